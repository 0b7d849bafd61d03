// rt_kernels.hip -- the wavefront path-tracing pipeline for gfx950.
//
//   camera_rgen      (raytracing.hlsl:128-163)  ->  folded into bounce 0 of k_traverse / k_shade (+ k_camera_resolve)
//   translucent_rgen (raytracing.hlsl:224-265)  ->  k_probe_raygen  / k_probe_resolve
//   trace_path_sample loop (raytracing.hlsl:92-126) -> per bounce: k_traverse -> k_shade [-> k_gather]
//   TraceRay (driver)                            ->  k_traverse: one ray per lane, two-level BVH2, per-lane stack in LDS
//   lambert/light/translucent_chit, miss         ->  k_shade
//   translucent_chit's loop over all sample points (raytracing.hlsl:323-331) -> k_gather
//
// Rays live in SoA queues in HBM; a ray is (origin, direction, path id) and the hit record is 20 B.  The queues are
// cut into segments of 1024 rays (one 32x32 pixel tile, or 1024 sample points); a segment is owned by one workgroup of
// the ray generator / shade kernels, which compacts the rays that survive a bounce to the front of the segment with a
// workgroup-local scan.  No global atomics are involved (a single append counter tops out near 90 appends/us on
// MI355X, which made a 2 M-ray launch cost 0.37 ms), slot assignment is deterministic, and the 16 waves that the
// traverse kernel runs per segment are dense except for the last one.
#include "rt_internal.h"
#include "rt_dev_pool.h"

namespace rti {

using namespace rtd;

static constexpr int TRAVERSE_WAVE   = 64;
static constexpr int LDS_STACK_DEPTH = 16;  // entries per lane kept in LDS (rows 0-15 of the 6.5 KB per wave: 24 waves/CU = 6 per SIMD at 80 VGPRs);
static constexpr int LDS_SIDE_ROW    = 17;  // row 16 is a dump row, rows 17-25 hold the top-level box-test constants while a ray is inside an instance
static constexpr int OVF_STACK_DEPTH = 48;  // global spill; 64 entries with the sentinel; the host checks tlas_depth + 1 + blas_depth <= 64
static constexpr int BT = 256;
static constexpr int SEG = 1024;            // rays per queue segment = threads per raygen / shade workgroup
// seg_count[..][s] holds the fill of segment s; in the fused camera pass bit 31 says that the radiance of every ray in the
// segment is +0 and was not stored (a path gathers radiance only where it ends, at a light -- unless a translucent
// surface turns it into a NaN -- so this is the rule): 24 bytes per ray and bounce that neither shade moves
static constexpr u32 SEG_RAD_ZERO = 0x80000000u, SEG_FILL_MASK = 0x7fffffffu;

// per-lane traversal stack: the first LDS_STACK_DEPTH entries in LDS (entries of one lane are 64 dwords apart:
// conflict-free), deeper entries in a global spill array indexed [entry][queue slot] (touched only by the rare deep
// path; a private array would be promoted to 32 VGPRs or drag the stack pointer into scratch).  top() and put() touch
// LDS unconditionally -- indices past the LDS part go to the dump row -- so that the common case has no branch.
// SPILL = false: the host has checked that the trees are shallow enough for the LDS rows alone (sentinel + top-level depth +
// marker + bottom-level depth + the row above the top <= LDS_STACK_DEPTH + 1, PassArgs::stack_in_lds): no index clamps, no
// overflow branches -- a fifth of the node step's instructions on the small scenes (everything but the instance field).
// SIDE_REGS: the nine top-level box-test constants wait in registers instead of LDS rows 17-25 while a ray is inside an
// instance -- the "deep" kernels (k_traverse at 5 waves per SIMD), whose 32 LDS rows all belong to the stack.
template <bool SPILL, bool SIDE_REGS = false>
struct LaneStack {
    float side[SIDE_REGS ? 9 : 1];
    RT_LDS int* lds;
    RT_GLOBAL int* spill;   // &stack_spill[queue slot]
    u32 spill_stride;
    int sp;                 // index of the top entry (>= 0 while the ray runs): top and the row above it share one address + offset
    RT_DEV void reset() { lds[0] = RT_TRAVERSAL_DONE; sp = 0; }
    RT_DEV int top() const { // the entry a pop would return
        const int i = sp;
        if (!SPILL) return lds[i * TRAVERSE_WAVE];
        int v = lds[(i < LDS_STACK_DEPTH ? i : LDS_STACK_DEPTH) * TRAVERSE_WAVE];
        if (i >= LDS_STACK_DEPTH) v = i < LDS_STACK_DEPTH + OVF_STACK_DEPTH ? spill[(size_t) (i - LDS_STACK_DEPTH) * spill_stride] : RT_TRAVERSAL_DONE;
        return v;
    }
    RT_DEV void put(int v, bool keep) { // stores above the top without moving sp; `keep` = the caller will count it in
        const int i = sp + 1;
        if (!SPILL) { lds[i * TRAVERSE_WAVE] = v; return; }
        lds[(i < LDS_STACK_DEPTH ? i : LDS_STACK_DEPTH) * TRAVERSE_WAVE] = v;
        if (i >= LDS_STACK_DEPTH && keep && i < LDS_STACK_DEPTH + OVF_STACK_DEPTH) spill[(size_t) (i - LDS_STACK_DEPTH) * spill_stride] = v;
    }
    RT_DEV void save_side(const float* v) {
        if (SIDE_REGS) { for (int k = 0; k < 9; k++) side[SIDE_REGS ? k : 0] = v[k]; return; }
        for (int k = 0; k < 9; k++) ((RT_LDS float*) lds)[(LDS_SIDE_ROW + k) * TRAVERSE_WAVE] = v[k];
    }
    RT_DEV void load_side(float* v) const {
        if (SIDE_REGS) { for (int k = 0; k < 9; k++) v[k] = side[SIDE_REGS ? k : 0]; return; }
        for (int k = 0; k < 9; k++) v[k] = ((const RT_LDS float*) lds)[(LDS_SIDE_ROW + k) * TRAVERSE_WAVE];
    }
    RT_DEV void push(int v) { put(v, true); sp++; }
    RT_DEV int pop() { int v = top(); sp--; return v; }
};

// rank of this thread among the threads of its 1024-thread workgroup with pred set, and their total
RT_DEV u32 block_rank(bool pred, u32* s_wave /* [16] */, u32& total) {
    unsigned long long m = __ballot(pred);
    u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads(); // s_wave may still be read from a previous call
    if (lane == 0) s_wave[wave] = (u32) __popcll(m);
    __syncthreads();
    u32 rank = (u32) __popcll(m & ((1ull << lane) - 1ull));
    total = 0;
    for (u32 w = 0; w < SEG / 64; w++) { u32 c = s_wave[w]; if (w < wave) rank += c; total += c; }
    return rank;
}

// the same with a BITS-bit key: threads with pred set are ranked by (key, thread index).  s_cnt: [(16 << BITS) + 1] words.
// One ballot per key bit gives the lane masks of all keys as scalar expressions: their sizes go to LDS through one store
// of the first 2^BITS lanes, and a lane's own mask is bp & ~((b0 ^ k0) | (b1 ^ k1) | ...), k_i = all ones where its key bit
// is set -- ~35 vector instructions per wave for 8 keys, where a ballot per key took 94; a wave without any such thread (the
// waves behind the fill of a compacted segment: half of all waves in the later bounces) only stores its zeros.
template <int BITS>
RT_DEV u32 block_rank_keyed(bool pred, u32 key, u32* s_cnt, u32& total) {
    constexpr u32 NK = 1u << BITS, PER = NK * 16u / 64u; // keys; (key, wave) counts per lane of the scan
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads(); // s_cnt may still be read from a previous call
    const unsigned long long bp = __ballot(pred);
    u32 below = 0; // threads of this wave with pred, the same key and a lower lane
    if (bp == 0ull) { // wave-uniform
        if (lane < NK) s_cnt[lane * 16u + wave] = 0u;
    } else {
        unsigned long long b[BITS];
#pragma unroll
        for (int i = 0; i < BITS; i++) b[i] = __ballot(((key >> i) & 1u) != 0u);
        int counts = 0;
#pragma unroll
        for (u32 k = 0; k < NK; k++) { // scalar work: mask and size of key k
            unsigned long long m = bp;
#pragma unroll
            for (int i = 0; i < BITS; i++) m &= ((k >> i) & 1u) ? b[i] : ~b[i];
            const int c = __popcll(m); (void) c;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(counts) : "s"(c), "n"(k)); // lane k of `counts` = c (a vector register written through a scalar operand: no store to memory)
#endif
        }
        if (lane < NK) s_cnt[lane * 16u + wave] = (u32) counts;
        u32 other_lo = 0u, other_hi = 0u; // lanes whose key differs from this lane's in some bit
#pragma unroll
        for (int i = 0; i < BITS; i++) {
            const u32 ki = 0u - ((key >> i) & 1u); // all ones where the key bit is set
            other_lo |= (u32) b[i] ^ ki; other_hi |= (u32) (b[i] >> 32) ^ ki;
        }
        below = __builtin_amdgcn_mbcnt_hi((u32) (bp >> 32) & ~other_hi, __builtin_amdgcn_mbcnt_lo((u32) bp & ~other_lo, 0u));
    }
    __syncthreads();
    if (wave == 0) { // exclusive scan of the (key, wave) counts in key-major order: PER per lane
        u32 c[PER], sum = 0;
#pragma unroll
        for (u32 k = 0; k < PER; k++) { c[k] = s_cnt[PER * lane + k]; sum += c[k]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 o = __shfl_up(inc, off); if ((int) lane >= off) inc += o; }
        u32 run = inc - sum;
#pragma unroll
        for (u32 k = 0; k < PER; k++) { s_cnt[PER * lane + k] = run; run += c[k]; }
        if (lane == 63u) s_cnt[NK * 16u] = inc;
    }
    __syncthreads();
    total = s_cnt[NK * 16u];
    return s_cnt[key * 16u + wave] + below;
}

// ranking by a key of up to 8 bits through an LDS histogram: one returning LDS atomic per thread (its rank among the threads
// of its key, in arrival order), a scan of the key counts by wave 0.  s_cnt: [(1 << BITS) + 1] words.  Cheaper than the
// ballots of block_rank_keyed from 16 keys on (C2 / C5 +0.6 % at equal keys).  The order inside a key varies from run to
// run; nothing depends on it: a path's arithmetic does not know its queue slot.
template <int BITS>
RT_DEV u32 block_rank_hist(bool pred, u32 key, u32* s_cnt, u32& total) {
    constexpr u32 NK = 1u << BITS, PER = (NK + 63u) / 64u;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads();
    if (threadIdx.x < NK) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    u32 r = 0;
    if (pred) r = atomicAdd(&s_cnt[key], 1u);
    __syncthreads();
    if (wave == 0) {
        u32 c[PER], sum = 0;
#pragma unroll
        for (u32 k = 0; k < PER; k++) { const u32 idx = PER * lane + k; c[k] = idx < NK ? s_cnt[idx] : 0u; sum += c[k]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 o = __shfl_up(inc, off); if ((int) lane >= off) inc += o; }
        u32 run = inc - sum;
#pragma unroll
        for (u32 k = 0; k < PER; k++) { const u32 idx = PER * lane + k; if (idx < NK) s_cnt[idx] = run; run += c[k]; }
        if (lane == 63u) s_cnt[NK] = inc;
    }
    __syncthreads();
    total = s_cnt[NK];
    return s_cnt[key] + r;
}

RT_DEV u32 wave_sum(u32 v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// thread -> pixel: 32x32 tiles (the multi-GPU split unit), 8x8 pixels per wave for coherent primary rays
RT_DEV bool thread_pixel(const PassArgs& a, u32 t, u32& px, u32& py) {
    if (t >= a.n_local_tiles * 1024u) return false;
    u32 tile = (t >> 10) * a.tile_world + a.tile_rank;
    u32 r = t & 1023u, sub = r >> 6, lane = r & 63u;
    // tile / tiles_x by multiplication with the host's rounded-up reciprocal (exact while tiles * tiles_x < 2^32, which
    // the host checks; 0 = divide): the compiler's 32-bit division is ~25 instructions, twice per thread
    const u32 ty = a.tiles_x_magic ? __umulhi(tile, a.tiles_x_magic) : tile / a.tiles_x;
    const u32 tx = tile - ty * a.tiles_x;
    px = tx * 32u + (sub & 3u) * 8u + (lane & 7u);
    py = ty * 32u + (sub >> 2) * 8u + (lane >> 3);
    return px < a.width && py < a.height;
}

// primary ray of a pixel for sample `sample_index` of this dispatch (raytracing.hlsl:130,141-149).  There is no camera
// ray-generation kernel: the bounce-0 traverse and shade kernels both derive the ray from the pixel (a hash, two draws
// and a 3x3 transform), which is cheaper than a 28-byte round trip through the ray queue.
RT_DEV void primary_ray(const PassArgs& a, u32 px, u32 py, u32 pixel, u32 sample_index, u32& rng, f3& o, f3& d) {
    rng = sample_index == 0 ? hash_u3(px, py, a.g.frame_rng * (u32) (a.g.accumulator_count != 0)) : a.ps.rng[pixel];
    camera_ray(a.g, a.camera_origin, a.inv_size, px, py, a.width, a.height, rng, o, d);
}

// the tail of camera_rgen (raytracing.hlsl:153-161): `tot` = sum of the dispatch's samples of one pixel
RT_DEV void resolve_pixel(const PassArgs& a, u32 pixel, float* tot /* [4] */) {
    float spp = (float) a.g.samples_per_pixel;
    float4 prev = make_float4(0, 0, 0, 0);
    if (a.g.accumulator_count != 0) prev = ((const float4*) a.accumulator)[pixel];
    float pv[4] = { prev.x, prev.y, prev.z, prev.w };
    float denom = (float) (a.g.accumulator_count + 1u);
    for (int k = 0; k < 4; k++) {
        if (spp != 1.0f) tot[k] = tot[k] / spp;                  // raytracing.hlsl:153 (x / 1 is x: the division is skipped, uniformly)
        if (a.g.accumulator_count != 0) tot[k] = tot[k] + pv[k]; // :155-158
    }
    ((float4*) a.accumulator)[pixel] = make_float4(tot[0], tot[1], tot[2], tot[3]);
    if (a.lazy_render_target) return; // pipelined frames: the 8-bit image is derived when it is read (k_resolve_only: four divisions and square roots per pixel that most frames of a sequence never show)
    unsigned char rt[4];
    for (int k = 0; k < 4; k++) rt[k] = to_unorm8(sqrtf(tot[k] / denom)); // :161
    ((uchar4*) a.render_target)[pixel] = make_uchar4(rt[0], rt[1], rt[2], rt[3]);
}

// separate resolve pass: only when the BSSRDF gather kernel may still add emission after a path's last shade
__global__ __launch_bounds__(BT) void k_camera_resolve(PassArgs a) {
    u32 t = blockIdx.x * BT + threadIdx.x;
    u32 px, py;
    if (!thread_pixel(a, t, px, py)) return;
    u32 pixel = py * a.width + px;
    float tot[4];
    tot[0] = a.ps.sum_x[pixel] + a.ps.rad_x[pixel];
    tot[1] = a.ps.sum_y[pixel] + a.ps.rad_y[pixel];
    tot[2] = a.ps.sum_z[pixel] + a.ps.rad_z[pixel];
    tot[3] = a.ps.sum_a[pixel] + a.ps.alpha[pixel];
    resolve_pixel(a, pixel, tot);
}

// Pipelined frames (a.defer_resolve) index their per-pixel state -- sample sums, RNG state between samples -- by the
// thread of the bounce-0 kernels (tile-major: thread t <-> thread_pixel(t)) instead of by pixel: k_frame_resolve's thread
// t reads record t, coalesced, and is the one that knows the pixel.
// With one sample per pixel and a frame already in the accumulator, most pixels need no record at all.  Two bits per
// thread, in the (otherwise unused) alpha array of the fused pass:
//   miss    the primary ray missed: the sample is (0, 0, 0, 0) and adds nothing (x + 0 is x; the sums are never -0)
//   record  the path ended with radiance: a float4 record holds the sample (set by the shade that ended it)
//   neither the path ended dark: the sample is (0, 0, 0, 1), known without a record
// The bounce-0 shade writes both words of its wave; later shades set `record` bits atomically.
RT_DEV bool frame_masks_in_use(const PassArgs& a) { return a.defer_resolve != 0 && a.g.samples_per_pixel == 1u && a.g.accumulator_count != 0u; }
RT_DEV unsigned long long* frame_miss_mask(const PassArgs& a) { return (unsigned long long*) a.ps.alpha; }
RT_DEV unsigned long long* frame_record_mask(const PassArgs& a) { return (unsigned long long*) a.ps.alpha + (a.n_paths >> 4); } // halfway into the array

// pipelined frames (a.defer_resolve): the fused shades leave each pixel's sample sum in ps.sum_* and this kernel, the only
// one of the frame that touches the accumulator, adds it in -- the frames' resolves run in frame order, everything else
// of a frame overlaps its neighbours.  Same additions in the same order as the resolve inside k_shade.
__global__ __launch_bounds__(BT) void k_frame_resolve(PassArgs a) {
    u32 t = blockIdx.x * BT + threadIdx.x;
    u32 px, py;
    if (!thread_pixel(a, t, px, py)) return;
    const bool masks = frame_masks_in_use(a);
    if (masks && ((frame_miss_mask(a)[t >> 6] >> (t & 63u)) & 1ull)) return;
    u32 pixel = py * a.width + px;
    float tot[4] = { 0.0f, 0.0f, 0.0f, 1.0f };
    if (masks && !((frame_record_mask(a)[t >> 6] >> (t & 63u)) & 1ull)) { /* ended dark: (0, 0, 0, 1) */ }
    else if (a.g.samples_per_pixel == 1u) { // one 16-byte record per thread (see k_shade)
        const float4 v = ((const float4*) a.ps.sum_x)[t];
        tot[0] = v.x; tot[1] = v.y; tot[2] = v.z; tot[3] = v.w;
    } else { tot[0] = a.ps.sum_x[t]; tot[1] = a.ps.sum_y[t]; tot[2] = a.ps.sum_z[t]; tot[3] = a.ps.sum_a[t]; }
    resolve_pixel(a, pixel, tot);
}

// re-derives the render target from an accumulator that already holds `accumulator_count` frames
__global__ __launch_bounds__(BT) void k_resolve_only(PassArgs a) {
    u32 pixel = blockIdx.x * BT + threadIdx.x;
    if (pixel >= a.width * a.height) return;
    float4 v = ((const float4*) a.accumulator)[pixel];
    float denom = (float) (a.g.accumulator_count == 0 ? 1u : a.g.accumulator_count);
    ((uchar4*) a.render_target)[pixel] = make_uchar4(to_unorm8(sqrtf(v.x / denom)), to_unorm8(sqrtf(v.y / denom)),
                                                      to_unorm8(sqrtf(v.z / denom)), to_unorm8(sqrtf(v.w / denom)));
}

// Tile split over ranks (rt_set_tile_split): the accumulator pixels of the tiles one rank owns, packed tile after tile
// (1,024 float4 per tile in thread_pixel order; slots outside the image are zero) -- what a rank sends instead of its
// whole, mostly untouched, frame -- and the way back into the accumulator of the rank that collects them.  `a` carries
// the rank / world whose tiles are meant.  Plain copies: the collected image equals the sum of the ranks' frames bit for bit.
__global__ __launch_bounds__(BT) void k_pack_tiles(PassArgs a, float4* __restrict__ packed) {
    const u32 t = blockIdx.x * BT + threadIdx.x;
    if (t >= a.n_local_tiles * 1024u) return;
    u32 px, py;
    packed[t] = thread_pixel(a, t, px, py) ? ((const float4*) a.accumulator)[py * a.width + px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ __launch_bounds__(BT) void k_unpack_tiles(PassArgs a, const float4* __restrict__ packed) {
    const u32 t = blockIdx.x * BT + threadIdx.x;
    u32 px, py;
    if (thread_pixel(a, t, px, py)) ((float4*) a.accumulator)[py * a.width + px] = packed[t];
}

RT_DEV void probe_finish_sample(const PassArgs& a, u32 p, f3 normal, f3& tsum) { // raytracing.hlsl:257-261
    f3 radiance = mk3(a.ps.rad_x[p], a.ps.rad_y[p], a.ps.rad_z[p]);
    f3 fd = mk3(a.ps.fd_x[p], a.ps.fd_y[p], a.ps.fd_z[p]);
    float cosine = -dot3(fd, normal);
    float fresnel = 1.0f - schlick(a.g.translucent_refractive_index, cosine);
    tsum = tsum + radiance * cosine * fresnel;
}

__global__ __launch_bounds__(SEG) void k_probe_raygen(PassArgs a, u32 sample_index) {
    __shared__ u32 s_wave[SEG / 64];
    u32 p = blockIdx.x * SEG + threadIdx.x;
    bool active = p < a.n_points && p >= a.probe_lo && p < a.probe_hi; // (rt_set_probe_split: this rank's block of the points)
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    if (active) {
        const TranslucentSlice& sl = a.shade.slices[a.sp_slice[p]];
        u32 x = p - sl.offset;
        f3 normal = mk3(a.sp_normal[3 * p], a.sp_normal[3 * p + 1], a.sp_normal[3 * p + 2]);
        u32 rng;
        if (sample_index == 0) {
            rng = hash_u3(x, sl.slot, a.g.frame_rng * (u32) (a.g.translucent_accumulator_count != 0)); // :246
            a.ps.sum_x[p] = 0.0f; a.ps.sum_y[p] = 0.0f; a.ps.sum_z[p] = 0.0f;
        } else {
            rng = a.ps.rng[p];
            f3 tsum = mk3(a.ps.sum_x[p], a.ps.sum_y[p], a.ps.sum_z[p]);
            probe_finish_sample(a, p, normal, tsum);
            a.ps.sum_x[p] = tsum.x; a.ps.sum_y[p] = tsum.y; a.ps.sum_z[p] = tsum.z;
        }
        const RtSamplePoint& sp = a.sp_write[p];
        o = mk3(sp.position[0], sp.position[1], sp.position[2]);
        d = random_on_hemisphere(rng, normal);
        a.ps.rng[p] = rng;
        a.ps.thr_x[p] = 1.0f; a.ps.thr_y[p] = 1.0f; a.ps.thr_z[p] = 1.0f;
        a.ps.rad_x[p] = 0.0f; a.ps.rad_y[p] = 0.0f; a.ps.rad_z[p] = 0.0f; a.ps.alpha[p] = 1.0f;
        a.ps.fd_x[p] = d.x; a.ps.fd_y[p] = d.y; a.ps.fd_z[p] = d.z;
    }
    u32 total;
    u32 slot = blockIdx.x * SEG + block_rank(active, s_wave, total);
    if (active) {
        const RayQueue& q = a.q[0];
        q.ox[slot] = o.x; q.oy[slot] = o.y; q.oz[slot] = o.z;
        q.dx[slot] = d.x; q.dy[slot] = d.y; q.dz[slot] = d.z;
        q.path[slot] = p;
    }
    if (threadIdx.x == 0) a.seg_count[0][blockIdx.x] = total;
}

__global__ __launch_bounds__(BT) void k_probe_resolve(PassArgs a) {
    u32 p = blockIdx.x * BT + threadIdx.x;
    if (p >= a.n_points || p < a.probe_lo || p >= a.probe_hi) return; // (the other ranks' blocks keep their payloads until the exchange)
    const TranslucentSlice& sl = a.shade.slices[a.sp_slice[p]];
    f3 normal = mk3(a.sp_normal[3 * p], a.sp_normal[3 * p + 1], a.sp_normal[3 * p + 2]);
    f3 tsum = mk3(a.ps.sum_x[p], a.ps.sum_y[p], a.ps.sum_z[p]);
    probe_finish_sample(a, p, normal, tsum);
    RtSamplePoint sp = a.sp_write[p];
    f3 payload = mk3(sp.payload[0], sp.payload[1], sp.payload[2]);
    if (a.g.translucent_accumulator_count == 0) payload = mk3(0.0f, 0.0f, 0.0f);     // raytracing.hlsl:243
    payload = payload + (tsum * sl.mean_area) / (RT_TAU_F / 2.0f * (float) a.g.samples_per_pixel); // :263
    sp.payload[0] = payload.x; sp.payload[1] = payload.y; sp.payload[2] = payload.z;
    a.sp_write[p] = sp;
}

// A camera path of the fused pass has ended at a bounce > 0 with radiance `rad`: its sample goes into the pixel exactly as
// k_shade<false, true> folds it (the same additions in the same order; pipelined frames: a record for k_frame_resolve).
RT_DEV void fused_sample_done(const PassArgs& a, u32 path, f3 rad, u32 sample_index, u32 rng) {
    float smp[4] = { rad.x, rad.y, rad.z, 1.0f }, tot[4];
    float* const sum[4] = { a.ps.sum_x, a.ps.sum_y, a.ps.sum_z, a.ps.sum_a };
    for (int k = 0; k < 4; k++) tot[k] = (sample_index == 0 ? 0.0f : sum[k][path]) + smp[k];
    if (sample_index + 1u < a.g.samples_per_pixel) {
        a.ps.rng[path] = rng; // the next sample of this pixel continues the stream (raytracing.hlsl:130,141)
        for (int k = 0; k < 4; k++) sum[k][path] = tot[k];
    } else if ((as_u32(rad.x) | as_u32(rad.y) | as_u32(rad.z)) == 0u && frame_masks_in_use(a)) { /* ended dark: no record */ }
    else if (a.defer_resolve) { // k_frame_resolve adds it in
        if (frame_masks_in_use(a)) atomicOr(frame_record_mask(a) + (path >> 6), 1ull << (path & 63u));
        if (a.g.samples_per_pixel == 1u) ((float4*) a.ps.sum_x)[path] = make_float4(tot[0], tot[1], tot[2], tot[3]);
        else for (int k = 0; k < 4; k++) sum[k][path] = tot[k];
    }
    else resolve_pixel(a, path, tot);
}

// ---- TraceRay ---------------------------------------------------------------------------------------------------
RT_DEV void traverse_counters_flush(const PassArgs& a, const TraverseCounters& cnt) {
    u32 sn = wave_sum(cnt.nodes), stt = wave_sum(cnt.tris), si = wave_sum(cnt.insts);
    u32 tn = wave_sum(cnt.node_trips), tt = wave_sum(cnt.tri_trips), ti = wave_sum(cnt.inst_trips);
    u32 mx = cnt.nodes + cnt.tris + cnt.insts;
    for (int off = 32; off >= 1; off >>= 1) { u32 o2 = __shfl_xor(mx, off); mx = o2 > mx ? o2 : mx; }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&a.stats->nodes, (unsigned long long) sn);
        atomicAdd(&a.stats->tris, (unsigned long long) stt);
        atomicAdd(&a.stats->insts, (unsigned long long) si);
        atomicAdd(&a.stats->node_trips, (unsigned long long) tn);
        atomicAdd(&a.stats->tri_trips, (unsigned long long) tt);
        atomicAdd(&a.stats->inst_trips, (unsigned long long) ti);
        atomicMax(&a.stats->max_ray_steps, (unsigned long long) mx);
    }
}

// PRIMARY: bounce 0 of the camera pass; lane t traces the primary ray of thread_pixel(t), the hit goes to slot t.
// PERSISTENT: the launch has one workgroup per wave slot of the GPU; a wave that has finished its 64 rays takes the next
// 64 from a global counter instead of ending.  The workgroup dispatcher keeps only ~80 % of the slots occupied when it
// has to place tens of thousands of one-wave workgroups whose lifetimes range from 4 us (the empty waves behind the
// compacted part of a segment) to ~90 us (timeline from per-wave timestamps); persistent waves
// keep 90 % occupied and balance the end of the launch by themselves.  The price is one device-scope atomic round trip
// per task (~2.5 us under load, exposed: memory operations return in order) and one wave per SIMD less (96 VGPRs).  It
// was worth +24 % on the 52,743-instance field (tasks of ~100 us) while the dispatched kernels ran 5 waves per SIMD in
// segment order; against today's dispatched kernels (6 waves, wave-major task order) it loses everywhere: -7 % on the
// field at 4K, -14 % at 1080p, -19 % on the bunny.  So it is off by default and kept as an option (RT_PERSISTENT).
// The tasks (waves of 64 queue slots) are dealt to TASK_PARTS interleaved partitions, each with its own counter in its
// own cache line: atomics on ONE address run at ~90 per microsecond on MI355X, which would make the 32,640 tasks of a
// 1080p launch cost 0.36 ms.  A wave works through its home partition, then moves to the one with the most tasks left.
static constexpr u32 TASK_PARTS = 64, TASK_STRIDE = 32; // counters 128 bytes apart; counter TASK_PARTS counts the waves that are done
// FLAT = K > 0: the scene has K <= RT_FLAT_MAX_INSTANCES instances and one tree over all their triangles (trace_flat,
// rt_dev_scene.h); the stack then has all 26 LDS rows to itself (no side store; the host checks the depth).
template <bool COUNT, int WAVES_PER_SIMD, bool PRIMARY, bool PERSISTENT, bool SPILL, int FLAT = 0>
__global__ __launch_bounds__(TRAVERSE_WAVE, WAVES_PER_SIMD) void k_traverse(PassArgs a, u32 inq, u32 sample_index, u32 n_tasks, u32 tail) {
    // tail (queued launches of the fused camera pass, last bounce): every path ends with this ray and only the emission of what it
    // hits still counts, so the wave shades its rays itself -- what k_shade<false, true> does at that bounce (no compaction there:
    // nothing the workgroup has to do together) -- instead of handing 16-byte hit records to one more kernel
    // DEEP: two-level trees too deep for the 17 stack rows of the 6-wave kernels but not for 33 (sentinel + top-level depth +
    // marker + bottom-level depth + the row above the top: the 52,743-instance field needs exactly 33): 5 waves per SIMD
    // by registers, 19 per CU by their 8.25 KB of LDS, all of it stack, the side store in registers (LaneStack SIDE_REGS)
    // -- the node step without the clamps and overflow branches of the spilling stack (41 instead of 50 vector instructions)
    constexpr bool DEEP = !PERSISTENT && !SPILL && FLAT == 0 && WAVES_PER_SIMD == 5;
    // FLAT = 4: two-level scene with a grid top level (trace_run_grid): the stack holds bottom-level entries only, RT_GRID_STACK_ROWS rows
    constexpr bool GRID = FLAT == 4;
    static_assert(!GRID || (!PERSISTENT && !SPILL), "the grid walk keeps its stack in LDS");
    __shared__ int smem[(GRID ? RT_GRID_STACK_ROWS : DEEP ? 33 : WAVES_PER_SIMD == 8 ? 20 : WAVES_PER_SIMD == 7 ? 22 : 26) * TRAVERSE_WAVE]; // 8 waves per SIMD = 32 per CU: 5 KB each of the 160 KB; 7: 28 per CU, 5.5 KB
    typedef LaneStack<SPILL, DEEP> Stack;
    TraverseCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.insts = 0; cnt.node_trips = 0; cnt.tri_trips = 0; cnt.inst_trips = 0;
    u32 part = blockIdx.x % TASK_PARTS;
    // next task of partition `p`, or n_tasks when it has none left (wave-uniform; costs one atomic round trip)
    auto take = [&](u32 p) -> u32 {
        u32 k = 0;
        if (threadIdx.x == 0) k = atomicAdd(a.task_counter + p * TASK_STRIDE, 1u);
        k = (u32) __builtin_amdgcn_readfirstlane((int) k);
        const u32 t = k * TASK_PARTS + p; // interleaved: the partitions advance through the queue together, so the rays in flight stay neighbours
        return t < n_tasks ? t : n_tasks;
    };
    // the home partition has run dry: move to the partition with the most tasks left (one 64-lane load of all counters,
    // one atomic), or end when none has any.  The counters only grow, so a partition seen dry stays dry.
    auto steal = [&]() -> u32 {
        for (int tries = 0; tries < 8; tries++) {
            const u32 p = threadIdx.x; // TASK_PARTS == the wave width
            const u32 taken = __hip_atomic_load(a.task_counter + p * TASK_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 size = n_tasks > p ? (n_tasks - p + TASK_PARTS - 1u) / TASK_PARTS : 0u;
            u32 left = size > taken ? size - taken : 0u;
            u32 best = (left << 6) | p; // largest remainder, ties to the higher partition index
            for (int off = 32; off >= 1; off >>= 1) { u32 o2 = __shfl_xor(best, off); best = o2 > best ? o2 : best; }
            if ((best >> 6) == 0u) return n_tasks;
            part = best & 63u;
            const u32 t = take(part);
            if (t < n_tasks) return t;
        }
        return n_tasks;
    };
    u32 task = blockIdx.x; // = the wave of 64 consecutive queue slots (or pixels of a tile) this wave works on
    // XCD tiles: one-wave workgroups are dealt to the 8 XCDs round-robin (b and b + 8 share an L2), so in launch order the 16
    // waves of a tile would run on all 8 XCDs and every L2 would hold what the tile touches.  Workgroup 128 g + 8 w + x
    // takes wave w of tile 8 g + x instead: a tile stays on one XCD, the tiles go round the XCDs (bands of the image per XCD
    // were measured too: -25 %, the bands cost different amounts).  Queued launches in wave-major order have this by
    // themselves when the number of segments is a multiple of 8.
    if (!PERSISTENT && PRIMARY && a.xcd_bands && (task >> 7) < (n_tasks >> 7)) task = (((task >> 7) << 3) + (task & 7u)) * 16u + ((task >> 3) & 15u);

    if (PERSISTENT) { task = take(part); if (task >= n_tasks) task = steal(); }
    while (task < n_tasks) {
        u32 next = 0;
        if (PERSISTENT && threadIdx.x == 0) next = atomicAdd(a.task_counter + part * TASK_STRIDE, 1u); // asked for early: back before the rays are
        u32 qi = task * TRAVERSE_WAVE + threadIdx.x;
        u32 seg_fill = 0;
        bool active;
        f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);
        if (PRIMARY) {
            u32 px, py;
            active = thread_pixel(a, qi, px, py);
            if (active) { u32 rng; primary_ray(a, px, py, a.defer_resolve ? qi : py * a.width + px, sample_index, rng, o, d); }
        } else {
            // queued launches take the waves of the segments "wave-major" (wave 0 of every segment, then wave 1, ...):
            // segments are compacted, so the empty waves behind their fill all land at the end of the launch instead of
            // taking dispatch slots between the working ones (C3, whose later bounces fill few waves per segment: +20 %)
            const u32 seg = a.wave_major ? task % a.n_segments : task / (SEG / TRAVERSE_WAVE);
            const u32 wis = a.wave_major ? task / a.n_segments : task % (SEG / TRAVERSE_WAVE);
            const u32 idx = wis * TRAVERSE_WAVE + threadIdx.x;
            qi = seg * SEG + idx;
            if (task == 0) { // ray statistics: one wave sums the segment counts of this launch
                u32 sum = 0;
                for (u32 k = threadIdx.x; k < a.n_segments; k += TRAVERSE_WAVE) sum += a.seg_count[inq][k] & SEG_FILL_MASK;
                sum = wave_sum(sum);
                if (threadIdx.x == 0) atomicAdd(a.probe ? &a.stats->rays_probe : &a.stats->rays_camera, (unsigned long long) sum);
            }
            seg_fill = a.seg_count[inq][seg];
            active = idx < (seg_fill & SEG_FILL_MASK);
            if (active) {
                const RayQueue& q = a.q[inq];
                o = mk3(q.ox[qi], q.oy[qi], q.oz[qi]);
                d = mk3(q.dx[qi], q.dy[qi], q.dz[qi]);
            }
        }
        if (PERSISTENT) next = (u32) __builtin_amdgcn_readfirstlane((int) next);
        if (active) {
            Stack st; st.lds = (RT_LDS int*) &smem[threadIdx.x]; st.spill = (RT_GLOBAL int*) (a.stack_spill + qi);
            st.spill_stride = a.n_paths; st.sp = 0;
            HitOut best;
            if (GRID) {
                TraceState S;
                trace_begin(a.scene, o, d, a.tmax, S, false);
                trace_run_grid<Stack, COUNT>(a.scene, o, d, a.tmin, a.tmax, st, S, cnt);
                best = S.best;
            } else if (FLAT > 0) {
                trace_flat<(FLAT == 2 ? 2 : 1), Stack, COUNT, !PRIMARY, PRIMARY, FLAT == 3>(a.scene, o, d, a.tmin, a.tmax, st, best, cnt); // FLAT = 3: one instance in the tree, a small one tested first (SceneView::flat_inl_inst)
            } else {
                TraceState S;
                trace_begin(a.scene, o, d, a.tmax, S);
                trace_run<Stack, COUNT>(a.scene, o, d, a.tmin, a.tmax, st, S, cnt);
                best = S.best;
            }
            // one 16-byte record per slot (t, u, v, id) in the space of the t, u, v and inst arrays, which are consecutive in
            // the arena (carve_path_buffers): one store here and one load in k_shade instead of five each.  Scenes with a flat
            // tree (whichever walk runs: the persistent kernels take the two levels) pack the instance into bit 30 of the id --
            // RT_FLAT_MAX_INSTANCES is 2 and their triangles number < 2^20; otherwise id = instance and the triangle goes to its
            // own array.  A miss is id = -1; the other words are then not read.
            static_assert(RT_FLAT_MAX_INSTANCES <= 2 && RT_FLAT_MAX_TRIS <= (1u << 30), "hit id packing");
            if (!PRIMARY && !COUNT && FLAT != 2 && tail) { // uniform (not compiled into the two-instance flat kernels: they sit at their 72-VGPR limit, and their scenes shade in k_shade)
                const u32 path = a.q[inq].path[qi];
                const SlotState& ss = a.st[inq];
                const f3 thr_prev = mk3(ss.thr_x[qi], ss.thr_y[qi], ss.thr_z[qi]);
                f3 rad = mk3(0.0f, 0.0f, 0.0f);
                if (!(seg_fill & SEG_RAD_ZERO)) rad = mk3(ss.rad_x[qi], ss.rad_y[qi], ss.rad_z[qi]);
                u32 rng = sample_index + 1u < a.g.samples_per_pixel ? ss.rng[qi] : 0u; // dead unless the pixel has another sample to come
                if (best.inst < 0) { best.t = inf_f32(); best.u = 0.0f; best.v = 0.0f; best.tri = -1; }
                ShadeOut r;
                shade_ray(a.shade, a.g, a.fudge, o, d, best, (float) a.g.bounces_per_sample, rng, r, true);
                rad = rad + r.emission * thr_prev;            // raytracing.hlsl:113
                fused_sample_done(a, path, rad, sample_index, rng);
            } else {
            const bool packed = (FLAT > 0 && !GRID) || a.scene.flat_instances != 0u; // uniform
            int id = best.inst;
            if (packed) id = best.inst < 0 ? -1 : (best.inst << 30) | best.tri;
            ((float4*) a.hits.t)[qi] = make_float4(best.t, best.u, best.v, __int_as_float(id));
            if (!packed && best.inst >= 0) a.hits.tri[qi] = best.tri;
            }
        }
        if (!PERSISTENT) break;
        task = next * TASK_PARTS + part < n_tasks ? next * TASK_PARTS + part : n_tasks;
        if (task >= n_tasks) task = steal();
    }
    if (COUNT) traverse_counters_flush(a, cnt);
    if (PERSISTENT && threadIdx.x == 0) { // the last wave out leaves the counters at zero for the next launch
        const u32 done = atomicAdd(a.task_counter + TASK_PARTS * TASK_STRIDE, 1u);
        if (done == gridDim.x - 1u) for (u32 k = 0; k <= TASK_PARTS; k++) a.task_counter[k * TASK_STRIDE] = 0u;
    }
}

// ---- pooled traversal of two-level scenes (rt_dev_pool.h) ---------------------------------------------------------------
// Queued launches only (bounce-0 rays are coherent and keep k_traverse).  One workgroup = 4 waves = a pool of 256 rays
// in LDS; grid = the workgroups that fit the GPU at once (2 per CU by their ~70 KB of LDS), each working through the
// queue segments g, g + G, g + 2 G, ... (G = grid size).  ROWS = stack rows per ray (17 / 33, as for k_traverse).
static constexpr int POOL_NODE_STEPS_MAX = 12; // a NODE batch goes back to the pool after this many steps at the latest
// RT_POOL_DEBUG (a separate build, tools/build_pool_debug.sh): every index and address the pooled kernel is about to use is
// checked against the scene's bounds first; a violation is recorded in DeviceStats::pool_err (code, value: the first one
// wins) and the ray is ended instead of the access being made -- a wild access takes the GPU down with it.
#if defined(RT_POOL_DEBUG)
#define RT_POOL_CHECK(cond, code, value, on_fail) if (!(cond)) { if (atomicCAS(&a.stats->pool_err[0], 0ull, (unsigned long long) (code)) == 0ull) a.stats->pool_err[1] = (unsigned long long) (value); on_fail; }
#else
#define RT_POOL_CHECK(cond, code, value, on_fail)
#endif
// Workgroup barrier of the pooled kernel: the LDS stores of this wave have landed before any other wave goes on.  The wait is
// explicit: at the barrier that heads the round loop hipcc (ROCm 7.2) emitted a bare s_barrier -- the wait that __syncthreads()'s
// release fence asks for was dropped on the loop's back edge -- and, once in ~10^8 rays, a wave read the class of a ray
// before the store of the round before had landed: the ray then sat in two lists and was walked by two waves at once.
RT_DEV void pool_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
}
template <bool COUNT, int ROWS>
__global__ __launch_bounds__(POOL_RAYS, 2) void k_traverse_pool(PassArgs a, u32 inq) {
    extern __shared__ int pool_lds[];
    RT_LDS int* const s_stack = (RT_LDS int*) pool_lds;                 // [ROWS][POOL_RAYS]
    RT_LDS int* const s_f = s_stack + ROWS * POOL_RAYS;                 // [PF_COUNT][POOL_RAYS]
    RT_LDS int* const s_list = s_f + PF_COUNT * POOL_RAYS;              // [4][POOL_RAYS]: slots by class
    RT_LDS int* const s_wc = s_list + 4 * POOL_RAYS;                    // [5][4]: rays per class and wave, pending results per wave
    RT_LDS int* const s_src = s_wc + 20;                                // the queue segment being fed into the pool: index, next ray, fill
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const SceneView& sc = a.scene;
    TraverseCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.insts = 0; cnt.node_trips = 0; cnt.tri_trips = 0; cnt.inst_trips = 0;
    auto fi = [&](int f, u32 slot) -> RT_LDS int& { return s_f[f * POOL_RAYS + slot]; };
    auto ff = [&](int f, u32 slot) -> RT_LDS float& { return ((RT_LDS float*) s_f)[f * POOL_RAYS + slot]; };
    const u32 seg_stride = gridDim.x;
    // the top level's box-test constants of a ray wait in global memory while it is inside an instance: three float4 per queue
    // slot in the (otherwise unused) overflow rows of the traversal stack
    float4* const side = (float4*) a.stack_spill;
    s_stack[tid] = RT_TRAVERSAL_DONE; // row 0: the sentinel under every stack
    fi(PF_CLS, tid) = PC_FREE; fi(PF_QI, tid) = -1; fi(PF_CUR, tid) = RT_TRAVERSAL_DONE; fi(PF_SP, tid) = 0;
#if defined(RT_POOL_DEBUG)
    s_src[4 + tid] = 0; // s_owner
#endif
    if (tid == 0) {
        s_src[0] = (int) blockIdx.x; s_src[1] = 0;
        s_src[2] = blockIdx.x < a.n_segments ? (int) (a.seg_count[inq][blockIdx.x] & SEG_FILL_MASK) : 0;
    }
    if (blockIdx.x == 0 && wave == 0) { // ray statistics: one wave sums the segment counts of this launch
        u32 sum = 0;
        for (u32 k = lane; k < a.n_segments; k += 64u) sum += a.seg_count[inq][k] & SEG_FILL_MASK;
        sum = wave_sum(sum);
        if (lane == 0) atomicAdd(a.probe ? &a.stats->rays_probe : &a.stats->rays_camera, (unsigned long long) sum);
    }
    // leaves the instance a ray has finished: the world-space box-test constants come back, the marker is popped
    auto leave_instance = [&](u32 slot, PoolStack& st, int& cur, u32& kzw) {
        u32 qi = (u32) fi(PF_QI, slot);
        RT_POOL_CHECK(qi < a.n_paths, 7, qi, qi = 0)
        const float4 s0 = side[qi], s1 = side[(size_t) a.n_paths + qi], s2 = side[2 * (size_t) a.n_paths + qi];
        ff(PF_IDX, slot) = s0.x; ff(PF_IDY, slot) = s0.y; ff(PF_IDZ, slot) = s0.z; ff(PF_CNX, slot) = s0.w;
        ff(PF_CNY, slot) = s1.x; ff(PF_CNZ, slot) = s1.y; ff(PF_CFX, slot) = s1.z; ff(PF_CFY, slot) = s1.w; ff(PF_CFZ, slot) = s2.x;
        const unsigned long long tn = (unsigned long long) sc.tlas_nodes;
        fi(PF_NLO, slot) = (int) (u32) tn; fi(PF_NHI, slot) = (int) (u32) (tn >> 32);
        kzw &= ~0x100u; fi(PF_KZ, slot) = (int) kzw;
        cur = st.pop();
    };
#if defined(RT_POOL_DEBUG)
    u32 rounds = 0;
    // where the time goes (lane 0 of every wave): cycles in scheduling (steps 1-3 with their barriers), in the wave's batch, and
    // waiting at the barrier that ends the round; batches and rays by class
    unsigned long long prof_sched = 0, prof_batch[4] = { 0, 0, 0, 0 }, prof_wait = 0, prof_nb[4] = { 0, 0, 0, 0 }, prof_nr[4] = { 0, 0, 0, 0 }, prof_idle = 0, prof_steps = 0;
    unsigned long long t_mark = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
#if defined(RT_POOL_DEBUG)
        if (++rounds > 2000000u) { if (tid == 0) { RT_POOL_CHECK(false, 9, blockIdx.x, ) } break; } // uniform
#endif
        // ---- 1. what every ray of the pool needs next --------------------------------------------------------------------
#if defined(RT_POOL_DEBUG)
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); (void) t; }
        const unsigned long long t_arrive = __builtin_amdgcn_s_memtime();
#endif
        pool_barrier(); // the phases of the round before have stored their rays
#if defined(RT_POOL_DEBUG)
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_wait += t - t_arrive; t_mark = t; }
#endif
        const int c = fi(PF_CLS, tid);
        const bool pend = c == PC_FREE && fi(PF_QI, tid) >= 0;
        unsigned long long m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = __ballot(c == k);
        const unsigned long long mp = __ballot(pend);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_wc[k * 4 + wave] = __popcll(m[k]);
            s_wc[16 + wave] = mp != 0ull ? 1 : 0;
        }
        if (tid == 0) { // the segment in hand is used up: on to this workgroup's next one that holds rays
            int seg = s_src[0], pos = s_src[1], fill = s_src[2];
            while (pos >= fill && (u32) seg < a.n_segments) {
                seg += (int) seg_stride; pos = 0;
                fill = (u32) seg < a.n_segments ? (int) (a.seg_count[inq][seg] & SEG_FILL_MASK) : 0;
            }
            s_src[0] = seg; s_src[1] = pos; s_src[2] = fill;
        }
        pool_barrier();
        // ---- 2. lists of slots by class (rank = rays of the class in the waves before + in the lanes before) ---------------
        int n[4], base[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            n[k] = 0; base[k] = 0;
#pragma unroll
            for (u32 w = 0; w < 4; w++) { const int v = s_wc[k * 4 + w]; if (w < wave) base[k] += v; n[k] += v; }
        }
        const bool any_pend = (s_wc[16] | s_wc[17] | s_wc[18] | s_wc[19]) != 0;
        const int seg = s_src[0], pos = s_src[1], fill = s_src[2];
        if (c < 4) {
            const unsigned long long mine = m[c];
            const int rank = base[c] + (int) __builtin_amdgcn_mbcnt_hi((u32) (mine >> 32), __builtin_amdgcn_mbcnt_lo((u32) mine, 0u));
            s_list[c * POOL_RAYS + rank] = (int) tid;
        }
        const int avail = fill - pos; // > 0 unless the queue is used up (step 1 moved on otherwise)
        if (!(avail > 0 || any_pend)) n[PC_FREE] = 0; // nothing to write, nothing to take: the free slots stay as they are
        if (n[0] + n[1] + n[2] + n[3] == 0) break; // uniform: every ray is done and written
        pool_barrier();
        // ---- 3. batches: the full ones of every class first, then the partial ones by size; wave w takes the w-th ----------
        int b_cls = -1, b_first = 0, b_cnt = 0, free_taken = 0;
        {
            int nb[4], rem[4], total_full = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { nb[k] = n[k] >> 6; rem[k] = n[k] & 63; total_full += nb[k]; }
            int key[4]; // partial batches, largest first: (size, class)
#pragma unroll
            for (int k = 0; k < 4; k++) key[k] = (rem[k] << 2) | k;
#define RT_POOL_CSWAP(i, j) { const int lo_ = key[i] < key[j] ? key[i] : key[j], hi_ = key[i] < key[j] ? key[j] : key[i]; key[i] = hi_; key[j] = lo_; }
            RT_POOL_CSWAP(0, 1) RT_POOL_CSWAP(2, 3) RT_POOL_CSWAP(0, 2) RT_POOL_CSWAP(1, 3) RT_POOL_CSWAP(1, 2)
#undef RT_POOL_CSWAP
#pragma unroll
            for (int w = 0; w < 4; w++) { // the batch of wave w (every wave works out all four: the FREE batches of the round decide how many rays the queue hands out)
                int cls = -1, first = 0, count = 0;
                if (w < total_full) {
                    int j = w;
#pragma unroll
                    for (int k = 0; k < 4; k++) { if (cls < 0 && j < nb[k]) { cls = k; first = j * 64; count = 64; } j -= nb[k]; }
                } else {
                    const int kk = key[(w - total_full) & 3];
                    if ((w - total_full) < 4 && (kk >> 2) > 0) { cls = kk & 3; first = nb[cls] * 64; count = kk >> 2; }
                }
                if (cls == PC_FREE) free_taken += count;
                if (w == (int) wave) { b_cls = cls; b_first = first; b_cnt = count; }
            }
        }
        b_cls = __builtin_amdgcn_readfirstlane(b_cls); b_first = __builtin_amdgcn_readfirstlane(b_first); b_cnt = __builtin_amdgcn_readfirstlane(b_cnt);
        const bool valid = (int) lane < b_cnt;
        const u32 slot = valid ? (u32) s_list[(b_cls < 0 ? 0 : b_cls) * POOL_RAYS + b_first + (int) lane] : 0u;
#if defined(RT_POOL_DEBUG)
        if (valid) { // every slot belongs to one batch of the round, and its class is the batch's
            RT_LDS int* const s_owner = s_src + 4;
            const int prev = atomicExch((int*) &s_owner[slot], (int) rounds);
            const int cls_now = fi(PF_CLS, slot);
            if (prev == (int) rounds || cls_now != b_cls || slot >= (u32) POOL_RAYS) {
                if (atomicCAS(&a.stats->pool_err[0], 0ull, prev == (int) rounds ? 11ull : 12ull) == 0ull) {
                    a.stats->pool_err[1] = slot; a.stats->pool_err[2] = (unsigned long long) b_cls; a.stats->pool_err[3] = (unsigned long long) cls_now;
                    a.stats->pool_err[4] = ((unsigned long long) wave << 32) | lane; a.stats->pool_err[5] = ((unsigned long long) b_first << 32) | (u32) b_cnt;
                    a.stats->pool_err[6] = rounds; a.stats->pool_err[7] = ((unsigned long long) n[0] << 48) | ((unsigned long long) n[1] << 32) | ((unsigned long long) n[2] << 16) | (unsigned long long) n[3];
                }
            }
        }
#endif
#if defined(RT_POOL_DEBUG)
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_sched += t - t_mark; t_mark = t; }
        if (b_cls >= 0) { prof_nb[b_cls] += 1; prof_nr[b_cls] += (unsigned long long) b_cnt; } else prof_idle += 1;
#endif
        // ---- 4. the wave's batch ---------------------------------------------------------------------------------------
        if (b_cls == PC_NODE) {
            int cur = RT_TRAVERSAL_DONE; u32 kzw = 0; float best_t = 0.0f;
            PoolStack st; st.lds = s_stack + slot; st.sp = 0;
            BoxRay br; br.idx = br.idy = br.idz = 1.0f; br.cnx = br.cny = br.cnz = 0.0f; br.cfx = br.cfy = br.cfz = 0.0f; br.px = br.py = br.pz = true;
            const RT_GLOBAL PackedNode* nodes = nullptr;
            if (valid) {
                cur = fi(PF_CUR, slot); st.sp = fi(PF_SP, slot); kzw = (u32) fi(PF_KZ, slot); best_t = ff(PF_T, slot);
                br.idx = ff(PF_IDX, slot); br.idy = ff(PF_IDY, slot); br.idz = ff(PF_IDZ, slot);
                br.cnx = ff(PF_CNX, slot); br.cny = ff(PF_CNY, slot); br.cnz = ff(PF_CNZ, slot);
                br.cfx = ff(PF_CFX, slot); br.cfy = ff(PF_CFY, slot); br.cfz = ff(PF_CFZ, slot);
                br.px = br.idx >= 0.0f; br.py = br.idy >= 0.0f; br.pz = br.idz >= 0.0f;
                nodes = (const RT_GLOBAL PackedNode*) (((unsigned long long) (u32) fi(PF_NHI, slot) << 32) | (u32) fi(PF_NLO, slot));
            }
            // steps down the tree while at least half of the batch still holds inner nodes: the rest wait with their leaf (or
            // their end) in hand, and go back to the pool with the others
            for (int steps = 0; steps < POOL_NODE_STEPS_MAX; steps++) {
                const bool inner = valid && cur >= 0;
                const int ni = __popcll(__ballot(inner));
                if (ni == 0 || (steps > 0 && 2 * ni < b_cnt)) break;
#if defined(RT_POOL_DEBUG)
                prof_steps += 1;
#endif
#if defined(RT_POOL_DEBUG)
                if (inner) {
                    const unsigned long long adr = (unsigned long long) nodes + ((unsigned long long) (u32) cur << 4);
                    const bool top_ok = (kzw & 0x100u) == 0u && nodes == (const RT_GLOBAL PackedNode*) sc.tlas_nodes && ((unsigned long long) (u32) cur << 4) + 32ull <= a.dbg_tlas_bytes;
                    const bool bot_ok = (kzw & 0x100u) != 0u && adr >= a.dbg_blas_lo && adr + 32ull <= a.dbg_blas_hi;
                    RT_POOL_CHECK(top_ok || bot_ok, 1, adr, cur = RT_TRAVERSAL_DONE)
                    RT_POOL_CHECK(st.sp >= 0 && st.sp + 1 < ROWS, 2, st.sp, cur = RT_TRAVERSAL_DONE)
                }
                if (valid && cur >= 0)
#else
                if (inner)
#endif
                    cur = node_step<PoolStack, COUNT>(nodes, br, cur, a.tmin, best_t, st, cnt);
            }
            if (valid) {
                if (cur == RT_STACK_MARKER) leave_instance(slot, st, cur, kzw);
                fi(PF_CUR, slot) = cur; fi(PF_SP, slot) = st.sp; fi(PF_CLS, slot) = pool_class_of(cur, (kzw & 0x100u) != 0u);
            }
        } else if (b_cls == PC_LEAF) {
            if (valid) {
                int cur = fi(PF_CUR, slot); u32 kzw = (u32) fi(PF_KZ, slot);
                PoolStack st; st.lds = s_stack + slot; st.sp = fi(PF_SP, slot);
                RayPrep rp;
                rp.o = mk3(ff(PF_OX, slot), ff(PF_OY, slot), ff(PF_OZ, slot)); rp.sx = ff(PF_SX, slot); rp.sy = ff(PF_SY, slot); rp.sz = ff(PF_SZ, slot);
                rp.kz = (int) (kzw & 3u); rp.flip = kzw & 0x80000000u;
                ClosestHit ch;
                ch.t = ff(PF_T, slot); ch.u_num = ff(PF_UN, slot); ch.v_num = ff(PF_VN, slot); ch.det = ff(PF_DET, slot);
                ch.inst = fi(PF_INST, slot); ch.tri = fi(PF_TRI, slot); ch.prim_global = (u32) fi(PF_PRIM, slot);
                const RT_GLOBAL TriRec* tris = (const RT_GLOBAL TriRec*) (((unsigned long long) (u32) fi(PF_THI, slot) << 32) | (u32) fi(PF_TLO, slot));
                const int cinst = fi(PF_CINST, slot);
                const float t_before = ch.t; const int tri_before = ch.tri, inst_before = ch.inst;
                const int v = ~cur, first = v >> 3;
                int count = (v & 7) + 1;
#if defined(RT_POOL_DEBUG)
                {
                    const unsigned long long adr = (unsigned long long) (tris + first);
                    if (!(adr >= a.dbg_blas_lo && adr + 48ull * (unsigned long long) count <= a.dbg_blas_hi && (kzw & 0x100u) != 0u)) {
                        if (atomicCAS(&a.stats->pool_err[0], 0ull, 3ull) == 0ull) {
                            a.stats->pool_err[1] = adr; a.stats->pool_err[2] = (unsigned long long) (u32) cur; a.stats->pool_err[3] = kzw;
                            a.stats->pool_err[4] = (unsigned long long) tris; a.stats->pool_err[5] = (unsigned long long) (u32) cinst;
                            a.stats->pool_err[6] = (unsigned long long) (u32) st.sp; a.stats->pool_err[7] = ((unsigned long long) slot << 32) | (u32) fi(PF_QI, slot);
                        }
                        count = 0;
                    }
                    RT_POOL_CHECK(st.sp >= 0 && st.sp < ROWS, 4, st.sp, st.sp = 0)
                }
#endif
                for (int k = 0; k < count; k++) {
                    const RT_GLOBAL TriRec* tr = tris + first + k;
                    const Quad qa = ldq(tr, 0), qb = ldq(tr, 1), qc = ldq(tr, 2);
                    if (COUNT) { cnt.tris++; if (first_active_lane()) cnt.tri_trips++; }
                    triangle_candidate(rp, mk3(qa.x, qa.y, qa.z), mk3(qa.w, qb.x, qb.y), mk3(qb.z, qb.w, qc.x), a.tmin, cinst, first + k, as_u32(qc.y), ch);
                }
                if (ch.t != t_before || ch.tri != tri_before || ch.inst != inst_before) {
                    ff(PF_T, slot) = ch.t; ff(PF_UN, slot) = ch.u_num; ff(PF_VN, slot) = ch.v_num; ff(PF_DET, slot) = ch.det;
                    fi(PF_INST, slot) = ch.inst; fi(PF_TRI, slot) = ch.tri; fi(PF_PRIM, slot) = (int) ch.prim_global;
                }
                cur = st.pop();
                if (cur == RT_STACK_MARKER) leave_instance(slot, st, cur, kzw);
                fi(PF_CUR, slot) = cur; fi(PF_SP, slot) = st.sp; fi(PF_CLS, slot) = pool_class_of(cur, (kzw & 0x100u) != 0u);
            }
        } else if (b_cls == PC_INST) {
            if (valid) { // a top-level leaf: into the instance (trace_bottom_level)
                const int p = ~fi(PF_CUR, slot);
                int cinst = p == sc.inl_inst ? (int) sc.n_insts - 1 : p; // the slot of the inlined instance holds the last one (SceneView)
                u32 qi = (u32) fi(PF_QI, slot);
                RT_POOL_CHECK((u32) cinst < sc.n_insts, 5, cinst, cinst = 0)
                RT_POOL_CHECK(qi < a.n_paths, 6, qi, qi = 0)
                PoolStack st; st.lds = s_stack + slot; st.sp = fi(PF_SP, slot);
                if (COUNT) { cnt.insts++; if (first_active_lane()) cnt.inst_trips++; }
                const RayQueue& q = a.q[inq];
                const f3 o = mk3(q.ox[qi], q.oy[qi], q.oz[qi]), d = mk3(q.dx[qi], q.dy[qi], q.dz[qi]);
                const RT_GLOBAL InstRec* in = (const RT_GLOBAL InstRec*) sc.insts + cinst;
                const Quad m0 = ldq(in, 0), m1 = ldq(in, 1), m2 = ldq(in, 2), m3 = ldq(in, 3), m4 = ldq(in, 4), m5 = ldq(in, 5);
                const float w2o[12] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w };
                const f3 oo = xform_point43(w2o, o), od = xform_dir43(w2o, d);
                const RayPrep rp = prepare_ray(oo, od);
                const unsigned long long pn = ((unsigned long long) as_u32(m3.y) << 32) | as_u32(m3.x);
                const unsigned long long pt = pn + as_u32(m3.z) + (unsigned long long) ((u32) rp.kz * as_u32(m5.z)); // the triangle copy of this ray's kz
                const BoxRay br = make_box_ray(oo, od, m3.w, mk3(m4.x, m4.y, m4.z), mk3(m4.w, m5.x, m5.y));
                ff(PF_IDX, slot) = br.idx; ff(PF_IDY, slot) = br.idy; ff(PF_IDZ, slot) = br.idz;
                ff(PF_CNX, slot) = br.cnx; ff(PF_CNY, slot) = br.cny; ff(PF_CNZ, slot) = br.cnz;
                ff(PF_CFX, slot) = br.cfx; ff(PF_CFY, slot) = br.cfy; ff(PF_CFZ, slot) = br.cfz;
                ff(PF_OX, slot) = rp.o.x; ff(PF_OY, slot) = rp.o.y; ff(PF_OZ, slot) = rp.o.z;
                ff(PF_SX, slot) = rp.sx; ff(PF_SY, slot) = rp.sy; ff(PF_SZ, slot) = rp.sz;
                fi(PF_KZ, slot) = (int) ((u32) rp.kz | rp.flip | 0x100u);
                fi(PF_NLO, slot) = (int) (u32) pn; fi(PF_NHI, slot) = (int) (u32) (pn >> 32);
                fi(PF_TLO, slot) = (int) (u32) pt; fi(PF_THI, slot) = (int) (u32) (pt >> 32);
                fi(PF_CINST, slot) = cinst;
                st.push(RT_STACK_MARKER);
                fi(PF_CUR, slot) = 0; fi(PF_SP, slot) = st.sp; fi(PF_CLS, slot) = PC_NODE;
            }
        } else if (b_cls == PC_FREE) {
            if (valid) {
                // the ray that held the slot is finished: its hit record (as k_traverse writes it for two-level scenes)
                int qold = fi(PF_QI, slot);
                RT_POOL_CHECK(qold < (int) a.n_paths, 8, qold, qold = -1)
                if (qold >= 0) {
                    const int inst = fi(PF_INST, slot);
                    float t = inf_f32(), u = 0.0f, v = 0.0f;
                    if (inst >= 0) { const float det = ff(PF_DET, slot); t = ff(PF_T, slot); u = ff(PF_UN, slot) / det; v = ff(PF_VN, slot) / det; } // S4: u = V / det, v = W / det
                    ((float4*) a.hits.t)[qold] = make_float4(t, u, v, __int_as_float(inst));
                    if (inst >= 0) a.hits.tri[qold] = fi(PF_TRI, slot);
                }
            }
            // the next rays of the queue segment, one per free slot in list order
            const int idx = pos + b_first + (int) lane;
            const bool take = valid && idx < fill;
            if (take) {
                u32 qi = (u32) seg * SEG + (u32) idx;
                RT_POOL_CHECK(qi < a.n_paths && (u32) seg < a.n_segments, 10, qi, qi = 0)
                const RayQueue& q = a.q[inq];
                const f3 o = mk3(q.ox[qi], q.oy[qi], q.oz[qi]), d = mk3(q.dx[qi], q.dy[qi], q.dz[qi]);
                const BoxRay br = make_box_ray(o, d, sc.tlas_bound, mk3(sc.tlas_ctr[0], sc.tlas_ctr[1], sc.tlas_ctr[2]), mk3(sc.tlas_inv_h[0], sc.tlas_inv_h[1], sc.tlas_inv_h[2]));
                side[qi] = make_float4(br.idx, br.idy, br.idz, br.cnx);
                side[(size_t) a.n_paths + qi] = make_float4(br.cny, br.cnz, br.cfx, br.cfy);
                side[2 * (size_t) a.n_paths + qi] = make_float4(br.cfz, 0.0f, 0.0f, 0.0f);
                ff(PF_IDX, slot) = br.idx; ff(PF_IDY, slot) = br.idy; ff(PF_IDZ, slot) = br.idz;
                ff(PF_CNX, slot) = br.cnx; ff(PF_CNY, slot) = br.cny; ff(PF_CNZ, slot) = br.cnz;
                ff(PF_CFX, slot) = br.cfx; ff(PF_CFY, slot) = br.cfy; ff(PF_CFZ, slot) = br.cfz;
                const unsigned long long tn = (unsigned long long) sc.tlas_nodes;
                fi(PF_NLO, slot) = (int) (u32) tn; fi(PF_NHI, slot) = (int) (u32) (tn >> 32);
                fi(PF_KZ, slot) = 0; fi(PF_QI, slot) = (int) qi;
                ClosestHit ch; ch.t = a.tmax; ch.u_num = 0.0f; ch.v_num = 0.0f; ch.det = 1.0f; ch.inst = -1; ch.tri = -1; ch.prim_global = 0;
                // NaN or infinite rays miss without a walk (trace_begin)
                const float big = 3.0e38f;
                const bool finite = fabsf(o.x) <= big && fabsf(o.y) <= big && fabsf(o.z) <= big && fabsf(d.x) <= big && fabsf(d.y) <= big && fabsf(d.z) <= big;
                int cur = finite ? 0 : RT_TRAVERSAL_DONE;
                if (sc.inl_inst >= 0 && finite) { // the triangles of the inlined instance first (SceneView::inl_inst)
                    inlined_triangles_first<COUNT>(sc, o, d, a.tmin, sc.inl_inst, ch, cnt);
                    if (sc.n_insts < 2u) cur = RT_TRAVERSAL_DONE;
                }
                ff(PF_T, slot) = ch.t; ff(PF_UN, slot) = ch.u_num; ff(PF_VN, slot) = ch.v_num; ff(PF_DET, slot) = ch.det;
                fi(PF_INST, slot) = ch.inst; fi(PF_TRI, slot) = ch.tri; fi(PF_PRIM, slot) = (int) ch.prim_global;
                fi(PF_CUR, slot) = cur; fi(PF_SP, slot) = 0; fi(PF_CLS, slot) = pool_class_of(cur, false);
            } else if (valid) {
                fi(PF_QI, slot) = -1; // written, and no ray to take in this round
                fi(PF_CLS, slot) = avail > 0 ? PC_FREE : PC_IDLE; // (a segment may end inside a batch: the slot asks again)
            }
        }
#if defined(RT_POOL_DEBUG)
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (b_cls >= 0) prof_batch[b_cls] += t - t_mark; t_mark = t; }
#endif
        if (tid == 0) s_src[1] = pos + (free_taken < avail ? free_taken : (avail > 0 ? avail : 0)); // rays handed out in this round
    }
#if defined(RT_POOL_DEBUG)
    if (lane == 0) {
        unsigned long long* p = a.stats->pool_prof;
        atomicAdd(p + 0, (unsigned long long) rounds); atomicAdd(p + 1, prof_sched); atomicAdd(p + 2, prof_wait); atomicAdd(p + 3, prof_idle); atomicAdd(p + 4, prof_steps);
        for (int k = 0; k < 4; k++) { atomicAdd(p + 5 + k, prof_batch[k]); atomicAdd(p + 9 + k, prof_nb[k]); atomicAdd(p + 13 + k, prof_nr[k]); }
        atomicAdd(p + 17, 1ull);
    }
#endif
    if (COUNT) traverse_counters_flush(a, cnt);
}

// ---- hit / miss shaders + the tail of the trace_path_sample loop body (raytracing.hlsl:113-119) ------------------
// PRIMARY: bounce 0 of the camera pass (thread t shades pixel thread_pixel(t); the path state starts here).
// FUSED: a path that ends here also finishes its pixel (sample sum, accumulate, tone-map) -- no resolve kernel, and no
// radiance / alpha / sum traffic for finished paths.  Not usable while the gather kernel may add emission afterwards.
// TRACE > 0 (with PRIMARY and FUSED; flat scenes, TRACE = the FLAT kind of k_traverse: 1 one instance, 3 one instance and a
// small one tested first): bounce 0 in ONE kernel -- the thread derives its primary ray once, walks it here (trace_flat, the
// stack in its wave's rows of dynamic LDS) and shades what it found: no second derivation of the ray (~110 vector instructions
// per lane of the separate bounce-0 shade), no hit record through memory, one launch less per sample.  The stack rows and the
// staging arrays of the compaction share the dynamic LDS (the walk is over, for all 16 waves, before the first staging store:
// the ranking in between starts with a barrier).
template <bool PRIMARY, bool FUSED, int TRACE = 0>
__global__ __launch_bounds__(SEG, 8) void k_shade(PassArgs a, u32 inq, u32 bounce, u32 sample_index) { // 8 waves per SIMD = two workgroups per CU: at most 64 VGPRs
    static_assert(TRACE == 0 || (FUSED && (TRACE == 1 || TRACE == 3)), "the kernels that walk and shade walk a flat tree of one instance");
    extern __shared__ int s_dyn[]; // TRACE: max(a.flat_rows, 14) x 1024 words
    __shared__ u32 s_wave[SEG / 64];
    const u32 qi = blockIdx.x * SEG + threadIdx.x;
    bool active, cont = false, gather = false, zero_sample = false, record = false;
    u32 path = 0, rng = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    f3 no = mk3(0, 0, 0), nd = mk3(0, 0, 1);
    ShadeOut r;
    int hit_inst = 0;
    f3 thr_prev = mk3(1.0f, 1.0f, 1.0f), rad = mk3(0.0f, 0.0f, 0.0f), thr = mk3(0.0f, 0.0f, 0.0f);
    // the last bounce in the fused pass: every path ends here and only the emission of what it hit still counts -- the
    // ray origin and (shade_ray) the scattering of Lambert surfaces are not needed, nor the RNG state unless the pixel
    // has another sample to come in this dispatch
    const bool tail = FUSED && !PRIMARY && !a.capture && bounce == a.g.bounces_per_sample;
    const bool rng_dead = tail && sample_index + 1u == a.g.samples_per_pixel;
    if (PRIMARY) {
        u32 px = 0, py = 0;
        active = thread_pixel(a, qi, px, py);
        path = a.defer_resolve ? qi : py * a.width + px; // see frame_masks_in_use
        if (active) {
            if (!FUSED) { // per-dispatch sample sum (raytracing.hlsl:139-151), resolved by k_camera_resolve
                if (sample_index == 0) {
                    a.ps.sum_x[path] = 0.0f; a.ps.sum_y[path] = 0.0f; a.ps.sum_z[path] = 0.0f; a.ps.sum_a[path] = 0.0f;
                } else {
                    a.ps.sum_x[path] += a.ps.rad_x[path]; a.ps.sum_y[path] += a.ps.rad_y[path];
                    a.ps.sum_z[path] += a.ps.rad_z[path]; a.ps.sum_a[path] += a.ps.alpha[path];
                }
            }
            primary_ray(a, px, py, path, sample_index, rng, o, d);
        }
    } else {
        if (TRACE > 0 && blockIdx.x == 0 && threadIdx.x < 64u) { // ray statistics, as k_traverse keeps them: one wave sums the segment counts of this launch
            u32 sum = 0;
            for (u32 k = threadIdx.x; k < a.n_segments; k += 64u) sum += a.seg_count[inq][k] & SEG_FILL_MASK;
            sum = wave_sum(sum);
            if (threadIdx.x == 0) atomicAdd(&a.stats->rays_camera, (unsigned long long) sum);
        }
        const u32 fill = a.seg_count[inq][blockIdx.x];
        active = threadIdx.x < (fill & SEG_FILL_MASK);
        if (active) {
            const RayQueue& q = a.q[inq];
            if (TRACE == 0) path = q.path[qi];
            if (TRACE > 0 || !tail || (a.gather_enabled != 0u && (int) bounce <= a.g.translucent_emission_bounces)) o = mk3(q.ox[qi], q.oy[qi], q.oz[qi]); // (a translucent hit of an emitting bounce queues its hit point)
            d = mk3(q.dx[qi], q.dy[qi], q.dz[qi]);
            if (FUSED && TRACE == 0) { // the state travels with the ray (TRACE: fetched after the walk, below -- eight registers that need not live through it)
                const SlotState& st = a.st[inq];
                if (!rng_dead) rng = st.rng[qi];
                thr_prev = mk3(st.thr_x[qi], st.thr_y[qi], st.thr_z[qi]);
                if (!(fill & SEG_RAD_ZERO)) rad = mk3(st.rad_x[qi], st.rad_y[qi], st.rad_z[qi]);
            } else {
                rng = a.ps.rng[path];
                thr_prev = mk3(a.ps.thr_x[path], a.ps.thr_y[path], a.ps.thr_z[path]);
                rad = mk3(a.ps.rad_x[path], a.ps.rad_y[path], a.ps.rad_z[path]);
            }
        }
    }
    if (active) {
        HitOut h;
        if (TRACE > 0) { // the walk of k_traverse<.., PRIMARY, .., FLAT = TRACE>, here
            LaneStack<false> st; st.lds = (RT_LDS int*) &s_dyn[(threadIdx.x >> 6) * a.flat_rows * TRAVERSE_WAVE + (threadIdx.x & 63u)];
            st.spill = nullptr; st.spill_stride = 0; st.sp = 0;
            TraverseCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.insts = 0; cnt.node_trips = 0; cnt.tri_trips = 0; cnt.inst_trips = 0;
            trace_flat<1, LaneStack<false>, false, !PRIMARY, PRIMARY, TRACE == 3>(a.scene, o, d, a.tmin, a.tmax, st, h, cnt); // (queued rays: the speculative walk, like k_traverse's)
            if (h.inst < 0) { h.t = inf_f32(); h.u = 0.0f; h.v = 0.0f; h.tri = -1; }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the last stack store has landed before the rows are reused (pool_barrier: hipcc can drop the barrier's own wait)
#endif
            if (!PRIMARY) { // the path's state, as the other kernels fetch it before the hit record
                const RayQueue& q = a.q[inq];
                const SlotState& sst = a.st[inq];
                const u32 fill = a.seg_count[inq][blockIdx.x];
                path = q.path[qi];
                if (!rng_dead) rng = sst.rng[qi];
                thr_prev = mk3(sst.thr_x[qi], sst.thr_y[qi], sst.thr_z[qi]);
                if (!(fill & SEG_RAD_ZERO)) rad = mk3(sst.rad_x[qi], sst.rad_y[qi], sst.rad_z[qi]);
            }
        } else {
        const float4 rec = ((const float4*) a.hits.t)[qi]; // (t, u, v, id) as k_traverse wrote it
        h.inst = __float_as_int(rec.w); h.t = inf_f32(); h.u = 0.0f; h.v = 0.0f; h.tri = -1;
        if (h.inst >= 0) {
            h.t = rec.x; h.u = rec.y; h.v = rec.z;
            if (a.scene.flat_instances) { h.tri = h.inst & 0x3fffffff; h.inst = h.inst >> 30; } // uniform
            else h.tri = a.hits.tri[qi];
        }
        }
        h.prim_global = 0;
        float payload_t = a.probe ? inf_f32() : (float) bounce; // raytracing.hlsl:102-107
        hit_inst = h.inst;
        shade_ray(a.shade, a.g, a.fudge, o, d, h, payload_t, rng, r, tail);
        rad = rad + r.emission * thr_prev;            // :113
        thr = thr_prev * r.reflectance;               // :114
        bool alive = any3(r.reflectance);             // :116
        cont = alive && bounce < a.g.bounces_per_sample;
        gather = r.gather;
        const bool missed_primary = bounce == 0 && h.inst < 0; // result.a = 0, :122
        if (!FUSED) {
            a.ps.rng[path] = rng;
            a.ps.rad_x[path] = rad.x; a.ps.rad_y[path] = rad.y; a.ps.rad_z[path] = rad.z;
            a.ps.thr_x[path] = thr.x; a.ps.thr_y[path] = thr.y; a.ps.thr_z[path] = thr.z;
            if (PRIMARY) a.ps.alpha[path] = missed_primary ? 0.0f : 1.0f;
            else if (missed_primary) a.ps.alpha[path] = 0.0f;
        } else if (!cont) {
            // the sample is complete: fold it into the pixel (same order of additions as the separate resolve pass)
            float smp[4] = { rad.x, rad.y, rad.z, missed_primary ? 0.0f : 1.0f }, tot[4];
            float* const sum[4] = { a.ps.sum_x, a.ps.sum_y, a.ps.sum_z, a.ps.sum_a };
            for (int k = 0; k < 4; k++) tot[k] = (sample_index == 0 ? 0.0f : sum[k][path]) + smp[k];
            if (sample_index + 1u < a.g.samples_per_pixel) {
                a.ps.rng[path] = rng; // the next sample of this pixel continues the stream (raytracing.hlsl:130,141)
                for (int k = 0; k < 4; k++) sum[k][path] = tot[k];
            } else if (PRIMARY && missed_primary && frame_masks_in_use(a)) zero_sample = true; // nothing to add: one bit for k_frame_resolve
            else if (!missed_primary && (as_u32(rad.x) | as_u32(rad.y) | as_u32(rad.z)) == 0u && frame_masks_in_use(a)) { /* ended dark: no record */ }
            else if (a.defer_resolve) { // k_frame_resolve adds it in
                if (frame_masks_in_use(a)) {
                    if (PRIMARY) record = true; // goes out with the wave's word below
                    else atomicOr(frame_record_mask(a) + (path >> 6), 1ull << (path & 63u));
                }
                // with one sample per pixel nothing reads the four sum arrays as such: they are consecutive in the arena
                // (rt_api.hip) and hold one float4 per pixel instead -- one 16-byte store, 128-byte runs per 8 pixels
                if (a.g.samples_per_pixel == 1u) ((float4*) a.ps.sum_x)[path] = make_float4(tot[0], tot[1], tot[2], tot[3]);
                else for (int k = 0; k < 4; k++) sum[k][path] = tot[k];
            }
            else resolve_pixel(a, path, tot);
        }
        if (alive) { no = o + r.t * d; nd = r.scatter; }       // :118-119
        if (a.probe && !cont) {
            // ray.Direction as trace_path_sample leaves it: the last traced direction after a break,
            // the untraced scatter direction when the bounce budget ran out
            f3 fd = alive ? r.scatter : d;
            a.ps.fd_x[path] = fd.x; a.ps.fd_y[path] = fd.y; a.ps.fd_z[path] = fd.z;
        }
        if (a.capture && sample_index == 0) {
            RtHit rec;
            rec.instance = h.inst; rec.geometry = r.geometry; rec.primitive = r.primitive;
            rec.t = h.inst < 0 ? inf_f32() : h.t; rec.u = h.u; rec.v = h.v; rec.rng_after = rng; rec.valid = 1u;
            a.capture[(size_t) bounce * a.width * a.height + path] = rec;
        }
    }
    if (PRIMARY && FUSED && frame_masks_in_use(a)) { // uniform
        const unsigned long long m = __ballot(zero_sample), rec = __ballot(record);
        if ((threadIdx.x & 63u) == 0) { frame_miss_mask(a)[qi >> 6] = m; frame_record_mask(a)[qi >> 6] = rec; }
    }
    u32 total;
    // gather items of the translucent hits of this launch: compacted to the front of the segment's part of the gather queue
    // (the host names a row of its own for every launch that can have any: batched gather, rt_api.hip run_pass)
    auto queue_gather_items = [&]() {
        u32 gtotal;
        const u32 gslot = blockIdx.x * SEG + block_rank(gather, s_wave, gtotal);
        if (threadIdx.x == 0) a.gseg_count[blockIdx.x] = gtotal;
        if (gather) {
            const GatherQueue& g = a.gq;
            g.path[gslot] = path;
            g.hx[gslot] = r.hit_point.x; g.hy[gslot] = r.hit_point.y; g.hz[gslot] = r.hit_point.z;
            g.tf[gslot] = r.transmitted_fresnel;
            g.kx[gslot] = thr_prev.x; g.ky[gslot] = thr_prev.y; g.kz[gslot] = thr_prev.z;
            g.ti[gslot] = r.ti;
        }
    };
    // the fused pass in a frame with the batched gather: only the bounces whose translucent hits emit have items (raytracing.hlsl:321)
    const bool fused_gather = FUSED && a.gather_enabled != 0u && (int) bounce <= a.g.translucent_emission_bounces; // uniform
    if (tail) { // no path goes on: nothing to compact
        if (threadIdx.x == 0) a.seg_count[inq ^ 1u][blockIdx.x] = 0u;
        if (fused_gather) queue_gather_items();
        return;
    }
    __shared__ u32 s_cnt[1025]; // up to 10 key bits (block_rank_hist)
    // rank of a surviving path in the next queue: by (key, thread) when a.ray_sort is set, by thread otherwise
    auto rank_sorted = [&]() -> u32 {
        if (!a.ray_sort || (!FUSED && !a.sort_box_on)) return block_rank(cont, s_wave, total); // (a direction sort alone does not pay in the unfused pass: cornell as shipped -2.7 %, scattered stores or not)
        // direction key: the octant and the dominant axis (24 of 32 values in use).  Rays of one tile that agree in both walk the
        // tree alike; the dominant axis on top of the octant: C3 +1.2 %, C2 +1.9 % (profiles/r02_ab_sort_box.txt)
        const float ax = fabsf(nd.x), ay = fabsf(nd.y), az = fabsf(nd.z);
        const u32 dom = az > fmaxf(ax, ay) ? 2u : (ay > ax ? 1u : 0u);
        const u32 key = ((nd.x < 0.0f ? 1u : 0u) | (nd.y < 0.0f ? 2u : 0u) | (nd.z < 0.0f ? 4u : 0u)) << 2 | dom;
        if (!a.sort_box_on) return block_rank_hist<5>(cont, key, s_cnt, total);
        // Flat scenes (a mesh in a room, a mesh on a floor, one mesh alone): how far the new ray goes through the world box of
        // the instance with the most triangles is a second sort key.  A wave takes as long as its longest ray, and the rays
        // that go through the mesh are the long ones: apart, they leave the other waves short.  (uniform branch)
        const float ix = fast_rcp(nd.x), iy = fast_rcp(nd.y), iz = fast_rcp(nd.z);
        const float x0 = (a.sort_box[0] - no.x) * ix, x1 = (a.sort_box[3] - no.x) * ix;
        const float y0 = (a.sort_box[1] - no.y) * iy, y1 = (a.sort_box[4] - no.y) * iy;
        const float z0 = (a.sort_box[2] - no.z) * iz, z1 = (a.sort_box[5] - no.z) * iz;
        const float en = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
        const float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        if (a.sort_box_on == 2u) {
            // Two-level scenes: the box is the top level's, and the key is the length of the ray's way through it, in eight classes
            // an octave apart around a.sort_chord (0.5 % of the box's diagonal: a ray that stays that long among the instances
            // meets several of them), longest first -- queued launches run the waves of a segment in order, and the launch ends
            // with its longest wave (52,743-instance field: +7.3 % at 1080p with two classes, +8.4 % with four, +1 % more with
            // eight, the same with sixteen; longest last: -4 %) -- then three bits of the instance the ray leaves (rays off the
            // same mesh start alike: +1.3 % with two bits, +0.6 % more with three, less again with four; the dominant axis
            // beside the octant: nothing; a bit in front of them for the rays that leave the inlined instance), then the octant.
            // The unfused pass: four classes two octaves apart, nothing else
            const float ch = ex - en;
            u32 c8 = 0;
            for (int k = -3; k <= 3; k++) c8 += ch >= ldexpf(a.sort_chord, k) ? 1u : 0u;
            if (!FUSED) return block_rank_keyed<3>(cont, (7u - c8) >> 1, s_cnt, total); // four classes where the stores scatter (below); eight: no difference (Cornell as shipped)
            const u32 inl = hit_inst == a.scene.inl_inst ? 1u : 0u; // off the inlined instance (the floor): no box around the origin to enter first (+1.3 %; as the first key: -2 %)
            return block_rank_hist<10>(cont, ((7u - c8) << 7) | (inl << 6) | (((u32) hit_inst & 7u) << 3) | (key >> 2), s_cnt, total);
        }
        // Flat scenes of two instances: eight classes, longest way through the mesh's box first (seven equal steps of the box's
        // diagonal), the rays that miss the box last.  Measured (profiles/r02_ab_sort_box.txt): with one bit (crosses or not)
        // the fused pass did best with the crossing rays last (C2 +2 %, first: +1.2 %) and the unfused pass, whose seven
        // stores per ray scatter with the key, with that bit alone, crossing first (C4 +8.7 %; with the octants: +3.6 %);
        // the eight classes on top of that: C2 +4.8 % (its queued launches 12.7 -> 15.0 Grays/s), C4 +1 % (traversal +4 %);
        // sixteen classes: the same; without the dominant axis in the direction key: -1 %; a bit for the instance the ray leaves: nothing.  One-instance scenes: the box is the
        // scene's, the way through it is how far the ray can go at all (C3 +2.2 %, its queued launches 12.7 -> 13.5 Grays/s)
        const float bx = a.sort_box[3] - a.sort_box[0], by = a.sort_box[4] - a.sort_box[1], bz = a.sort_box[5] - a.sort_box[2];
        const float per_diag = 7.0f * __builtin_amdgcn_rsqf(__builtin_fmaf(bx, bx, __builtin_fmaf(by, by, bz * bz))); // uniform
        const u32 cls = en <= ex ? 6u - (u32) fminf((ex - en) * per_diag, 6.0f) : 7u; // nd is a unit vector: ex - en is a length
        if (!FUSED) return block_rank_keyed<3>(cont, cls, s_cnt, total);
        return block_rank_hist<8>(cont, (cls << 5) | key, s_cnt, total);
    };
    if (FUSED) {
        // Survivors are compacted to the front of the segment, grouped by the octant of their new direction when
        // a.ray_sort is set: rays of one origin neighbourhood (the tile) and one octant walk the BVH alike, which cuts
        // the trips a wave makes through the traversal loop by ~20 % (C2: SIMD efficiency of node steps 0.35 -> 0.44).
        // Ray and path state go through LDS so that the queue is written in slot order, coalesced, whatever the ranking.
        __shared__ float s_stage_own[TRACE > 0 ? 1 : 14][SEG];
        float (*const s_stage)[SEG] = TRACE > 0 ? (float (*)[SEG]) s_dyn : s_stage_own; // TRACE: the stack rows, no longer in use
        const u32 ls = rank_sorted();
        const bool rad_zero = __syncthreads_or(cont && (as_u32(rad.x) | as_u32(rad.y) | as_u32(rad.z)) != 0u) == 0; // uniform
        if (threadIdx.x == 0) a.seg_count[inq ^ 1u][blockIdx.x] = total | (rad_zero ? SEG_RAD_ZERO : 0u);
        if (cont) {
            s_stage[0][ls] = no.x; s_stage[1][ls] = no.y; s_stage[2][ls] = no.z;
            s_stage[3][ls] = nd.x; s_stage[4][ls] = nd.y; s_stage[5][ls] = nd.z; s_stage[6][ls] = __uint_as_float(path);
            s_stage[7][ls] = __uint_as_float(rng);
            s_stage[8][ls] = thr.x; s_stage[9][ls] = thr.y; s_stage[10][ls] = thr.z;
            if (!rad_zero) { s_stage[11][ls] = rad.x; s_stage[12][ls] = rad.y; s_stage[13][ls] = rad.z; }
        }
        __syncthreads();
        if (threadIdx.x < total) {
            const RayQueue& q = a.q[inq ^ 1u];
            const SlotState& st = a.st[inq ^ 1u];
            const u32 t = threadIdx.x, o2 = blockIdx.x * SEG + t;
            q.ox[o2] = s_stage[0][t]; q.oy[o2] = s_stage[1][t]; q.oz[o2] = s_stage[2][t];
            q.dx[o2] = s_stage[3][t]; q.dy[o2] = s_stage[4][t]; q.dz[o2] = s_stage[5][t];
            q.path[o2] = __float_as_uint(s_stage[6][t]);
            st.rng[o2] = __float_as_uint(s_stage[7][t]);
            st.thr_x[o2] = s_stage[8][t]; st.thr_y[o2] = s_stage[9][t]; st.thr_z[o2] = s_stage[10][t];
            if (!rad_zero) { st.rad_x[o2] = s_stage[11][t]; st.rad_y[o2] = s_stage[12][t]; st.rad_z[o2] = s_stage[13][t]; }
        }
        if (fused_gather) queue_gather_items(); // (their terms are added to the pixels' sample sums at the end of the frame: k_gather_apply)
        return;
    }
    u32 slot = blockIdx.x * SEG + rank_sorted(); // (sorted: the seven stores scatter inside the segment's 4 KB of each array)
    if (threadIdx.x == 0) a.seg_count[inq ^ 1u][blockIdx.x] = total;
    if (cont) {
        const RayQueue& q = a.q[inq ^ 1u];
        q.ox[slot] = no.x; q.oy[slot] = no.y; q.oz[slot] = no.z;
        q.dx[slot] = nd.x; q.dy[slot] = nd.y; q.dz[slot] = nd.z;
        q.path[slot] = path;
    }
    if (!a.gather_enabled) return; // uniform: no translucent sample points in the scene
    queue_gather_items();
}

// ---- translucent_chit: sum of bssrdf(|p_i - hit|) * payload_i over all sample points (raytracing.hlsl:321-333) -------
// One hit per lane, every lane walks all points of its instance in one fixed order -- index order for the dipole, Morton order of
// the positions for the tabulated profile (below) -- (the order of the float sum is part of
// the result).  The points are the same for all lanes of a wave and arrive through the scalar cache (k_gather); the
// tabulated profile sits in LDS as one float4 per texel, so the two texel fetches of the linear filter are LDS gathers --
// reading both from global memory had the texture-address unit as the bottleneck (12 divergent loads per pair).  Hits
// of different translucent instances in one workgroup take turns.
static constexpr int GATHER_BLOCK = 256;
static constexpr int GATHER_TILE = 512;        // points per tile (dipole): 16 KB
static constexpr int GATHER_TABLE_MAX = 2048;  // texels kept in LDS: 32 KB (raytracing.cpp:113 pads the profile to 2048)

struct alignas(16) GatherPoint { float px, py, pz, pad0, wx, wy, wz, pad1; };

// S8 with the table in LDS as (r, g, b, 0) per texel, stored with its zero border: entries 0 and 1 are texels -2 and
// -1, entries 2..n+1 the texels, entries n+2 and n+3 zero again (tab_fill_lds).  Same arithmetic as
// sample_bssrdf_table for every finite coordinate; the border replaces the range checks (a clamp of the texel index
// instead of seven compares, seven selects and a branch): outside the table both texels are 0 and so is 0 + f * 0.
// (radiance only, 1e-4: the filter is one fused multiply-add per channel.  The texel coordinate keeps the oracle's two
// roundings: x is ~2000 where the profile ends, an ulp of it is 1.2e-4 of a texel, and the texels there are steep --
// a fused u n - 0.5 put one pixel of the C4 test 1.6e-4 off)
RT_DEV f3 sample_bssrdf_table_lds(const float4* tab, int n, float u) {
    const float x = u * (float) n - 0.5f;
    const float fl = floorf(x);
    const float f = x - fl;
    const int k0 = (int) fminf(fmaxf(fl, -2.0f), (float) n) + 2; // 0 .. n + 2
    const float4 t0 = tab[k0], t1 = tab[k0 + 1];
    return mk3(__builtin_fmaf(f, t1.x - t0.x, t0.x), __builtin_fmaf(f, t1.y - t0.y, t0.y), __builtin_fmaf(f, t1.z - t0.z, t0.z));
}
// The same value with the second fetch taken off the LDS pipe: the difference to the next texel comes from a table in global
// memory (k_gather_delta: the same scaled texels, subtracted once instead of per pair) through the vector L1.  The LDS pipe
// is shared by the four SIMDs of a CU and spends half its active cycles on the bank conflicts of these per-lane fetches
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.53, profiles/pmc_traffic.json c4), the texture path idles (TA busy 0.02).
RT_DEV f3 sample_bssrdf_table_lds_l1(const float4* tab, const float4* __restrict__ delta, int n, float u) {
    const float x = u * (float) n - 0.5f;
    const float fl = floorf(x);
    const float f = x - fl;
    const int k0 = (int) fminf(fmaxf(fl, -2.0f), (float) n) + 2; // 0 .. n + 2
    const float4 t0 = tab[k0], dt = delta[k0];
    return mk3(__builtin_fmaf(f, dt.x, t0.x), __builtin_fmaf(f, dt.y, t0.y), __builtin_fmaf(f, dt.z, t0.z));
}
// delta[t] = entry t + 1 - entry t of the table as tab_fill_lds lays it out (scaled texels, zero border), t = 0 .. n + 3
__global__ void k_gather_delta(const float* __restrict__ table, int n, float factor, float4* __restrict__ delta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n + 4) return;
    float e0[3], e1[3];
    for (int c = 0; c < 3; c++) {
        const int k0 = t - 2, k1 = t - 1;
        e0[c] = (k0 >= 0 && k0 < n) ? factor * table[3 * k0 + c] : 0.0f;
        e1[c] = (k1 >= 0 && k1 < n) ? factor * table[3 * k1 + c] : 0.0f;
    }
    delta[t] = make_float4(e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2], 0.0f);
}
int launch_gather_delta(hipStream_t s, const float* table, int n, float factor, float4* delta) {
    hipLaunchKernelGGL(k_gather_delta, dim3((n + 4 + 255) / 256), dim3(256), 0, s, table, n, factor, delta);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
// `factor` (fudge / scale^2, raytracing.hlsl:276) is multiplied into the texels here, once, instead of into every filtered value
template <int BLOCK>
RT_DEV void tab_fill_lds(float4* tab, const float* table, int n, float factor) {
    for (int t = threadIdx.x; t < n + 4; t += BLOCK) {
        const int k = t - 2;
        tab[t] = (k >= 0 && k < n) ? make_float4(factor * table[3 * k], factor * table[3 * k + 1], factor * table[3 * k + 2], 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// Gather items sit at the front of their 1024-slot segments; after the first bounce most segments hold only a few
// (secondary rays that happen to reach a translucent mesh), so the gather kernel numbers the items across segments:
// goff[s] = items in segments before s, goff[n_segments] = all of them.  One workgroup, wave scans with a carry.
__global__ __launch_bounds__(SEG) void k_gather_offsets(PassArgs a) {
    __shared__ u32 s_wave[SEG / 64];
    __shared__ u32 s_carry;
    // batched gather: the counts of all rows (shade launches) kept for this launch, row after row
    const u32 n = a.bg_rows ? a.bg_rows * a.bg_cols : a.n_segments;
    const u32* cnt = a.bg_rows ? a.bg_counts : a.gseg_count;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < n; b0 += SEG) {
        const u32 i = b0 + threadIdx.x;
        const u32 v = i < n ? cnt[i] : 0u;
        u32 inc = v;
        for (int off = 1; off < 64; off <<= 1) { u32 t = __shfl_up(inc, off); if ((int) lane >= off) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        u32 base = s_carry, total = 0;
        for (u32 w = 0; w < SEG / 64; w++) { u32 c = s_wave[w]; if (w < wave) base += c; total += c; }
        if (i < n) a.goff[i] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.goff[n] = s_carry; if (a.bg_rows) a.goff[n + 1u] = 0u; /* the block counter of the batched launch */ }
}

// MODE: 0 = dipole, 1 = tabulated profile in LDS, 2 = tabulated profile too long for LDS (read from global memory);
// a template parameter so that the unrolled loop body is straight-line code the scheduler can interleave
template <int MODE, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void k_gather(PassArgs a) {
    __shared__ GatherPoint s_pts[MODE == 0 ? GATHER_TILE : 1];
    __shared__ float4 s_tab[(MODE == 1 || MODE == 3) ? GATHER_TABLE_MAX + 4 : 1];
    __shared__ int s_next;
    __shared__ u32 s_blk;
    // Batched launch (a.bg_rows > 0: the items of several shade launches, launch_gather_batched): goff numbers the items of all
    // rows, the grid is sized to the machine and its workgroups take blocks of 256 items from a counter until none is left --
    // the table is filled once per workgroup, and a launch holds enough waves to keep the SIMDs busy (one shade launch of C4
    // leaves ~115,000 items: 1.8 waves per SIMD, each a chain of 759 dependent point evaluations).
    const u32 n_entries = a.bg_rows ? a.bg_rows * a.bg_cols : a.n_segments;
    const u32 n_items = a.goff[n_entries];
    if (!a.bg_rows && blockIdx.x * BLOCK >= n_items) return; // uniform
    const GatherQueue& g = a.gq;
    const float scale = a.g.translucent_bssrdf_scale;
    if (MODE == 1 || MODE == 3)
        tab_fill_lds<BLOCK>(s_tab, a.bssrdf_table, a.bssrdf_texels, a.fudge / (scale * scale)); // (the first barrier below covers it)
    u32 pairs = 0;
    u32 blk = blockIdx.x;
    for (;;) {
    if (a.bg_rows) { // uniform
        __syncthreads();
        if (threadIdx.x == 0) s_blk = atomicAdd(&a.goff[n_entries + 1u], 1u);
        __syncthreads();
        blk = s_blk;
        if (blk * BLOCK >= n_items) break; // uniform
    }
    const u32 j = blk * BLOCK + threadIdx.x;
    bool active = j < n_items;
    size_t i;
    {   // item j of the numbering -> its segment (last s with goff[s] <= j) -> its queue slot
        const u32 jj = active ? j : n_items - 1u;
        u32 lo = 0, hi = n_entries; // goff[lo] <= jj < goff[hi]
        while (hi - lo > 1u) { u32 mid = (lo + hi) >> 1; if (a.goff[mid] <= jj) lo = mid; else hi = mid; }
        const u32 row = a.bg_rows ? lo / a.bg_cols : 0u, col = lo - row * a.bg_cols;
        i = (size_t) row * a.bg_stride + (size_t) col * SEG + (jj - a.goff[lo]);
    }
    if ((MODE == 1 || MODE == 3) && a.sort_box_on == 1u) {
        // The workgroup's items regrouped by the cell of the mesh's box their hit lies in: the hits of a wave then sit
        // together, and more of the far point groups below are far for the whole wave (the skip needs every lane to agree)
        __shared__ u32 s_key[66];
        __shared__ size_t s_perm[BLOCK];
        if (threadIdx.x < 66) s_key[threadIdx.x] = 0u;
        __syncthreads();
        u32 key = 64u; // inactive threads last
        if (active) { // 4 x 4 x 4 cells of the mesh's box, numbered along the Morton curve
            const float h[3] = { g.hx[i], g.hy[i], g.hz[i] };
            u32 q[3];
            for (int k = 0; k < 3; k++) {
                const float t = (h[k] - a.sort_box[k]) * fast_rcp(a.sort_box[3 + k] - a.sort_box[k]) * 4.0f;
                q[k] = (u32) fminf(fmaxf(t, 0.0f), 3.0f);
            }
            key = 0u;
            for (int k = 0; k < 3; k++) key |= ((q[k] & 1u) << k) | ((q[k] & 2u) << (k + 2));
        }
        const u32 r = atomicAdd(&s_key[key], 1u);
        __syncthreads();
        if (threadIdx.x < 64) { // exclusive scan of the 64 cell counts by the first wave; [65] = all of them = the number of items
            const u32 c = s_key[threadIdx.x];
            u32 inc = c;
            for (int off = 1; off < 64; off <<= 1) { u32 o = __shfl_up(inc, off); if ((int) threadIdx.x >= off) inc += o; }
            s_key[threadIdx.x] = inc - c;
            if (threadIdx.x == 63) s_key[65] = inc;
        }
        __syncthreads();
        const u32 n_active = s_key[65];
        s_perm[(key < 64u ? s_key[key] : n_active) + r] = i;
        __syncthreads();
        i = s_perm[threadIdx.x];
        active = threadIdx.x < n_active;
    }
    const f3 hit = mk3(g.hx[i], g.hy[i], g.hz[i]);
    const int my_ti = active ? g.ti[i] : -1;
    // radiance-only arithmetic (covered by the 1e-4 tolerance): fudge / scale^2 is folded into one
    // factor, and the dipole uses the hardware sqrt / rcp / exp2.  The table coordinate radius / scale stays (all but)
    // correctly rounded -- near the end of the profile's support the texels are ~1e-7 and an ulp of the coordinate is a
    // visible relative error -- but by refinement steps (sqrt_refined, div_by_shared) instead of the library sequences.
    const float fudge_over_z = a.fudge / (scale * scale);
    const float inv_scale = 1.0f / scale;
    const float eta = a.g.translucent_refractive_index;
    DipoleConst cr = dipole_constants(a.g.translucent_scattering[0], a.g.translucent_absorption[0], eta);
    DipoleConst cg = dipole_constants(a.g.translucent_scattering[1], a.g.translucent_absorption[1], eta);
    DipoleConst cb = dipole_constants(a.g.translucent_scattering[2], a.g.translucent_absorption[2], eta);
    // the shipped defaults use one scattering / absorption pair for all channels (main.cpp:420-422): evaluate once
    const bool grey = a.g.translucent_scattering[0] == a.g.translucent_scattering[1] && a.g.translucent_scattering[0] == a.g.translucent_scattering[2] &&
                      a.g.translucent_absorption[0] == a.g.translucent_absorption[1] && a.g.translucent_absorption[0] == a.g.translucent_absorption[2];
    f3 E = mk3(0.0f, 0.0f, 0.0f);
    // translucent instances present in this workgroup, in increasing index order
    int cur = -1;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0x7fffffff;
        __syncthreads();
        if (my_ti > cur) atomicMin(&s_next, my_ti);
        __syncthreads();
        cur = s_next;
        if (cur == 0x7fffffff) break; // uniform
        if (MODE == 0) {
            // dipole: bound by its transcendental arithmetic, not by LDS -- the points are staged through LDS in tiles (every
            // lane reads the same point: a broadcast), which keeps them off the scalar cache's miss path
            const TranslucentSlice sl = a.shade.slices[cur];
            const RtSamplePoint* pts = a.sp_read + sl.offset;
            const bool mine = my_ti == cur;
            for (u32 k0 = 0; k0 < sl.count; k0 += GATHER_TILE) {
                const u32 nk = sl.count - k0 < (u32) GATHER_TILE ? sl.count - k0 : (u32) GATHER_TILE;
                __syncthreads(); // the previous tile is no longer read
                for (u32 t = threadIdx.x; t < nk; t += BLOCK) {
                    const RtSamplePoint sp = pts[k0 + t];
                    GatherPoint gp;
                    gp.px = sp.position[0]; gp.py = sp.position[1]; gp.pz = sp.position[2]; gp.pad0 = 0.0f;
                    gp.wx = sp.payload[0]; gp.wy = sp.payload[1]; gp.wz = sp.payload[2]; gp.pad1 = 0.0f;
                    s_pts[t] = gp;
                }
                __syncthreads();
                if (!mine) continue;
                // four points per trip: their sqrt / divide chains are independent and overlap; the sum stays in index order.
                // `GREY` is a compile-time flag of the loop (two copies): a branch on it between the four terms would keep
                // the scheduler from interleaving their chains.
                auto run = [&](auto grey_tag) {
                    constexpr bool GREY = decltype(grey_tag)::value;
                    struct Term { f3 bs, w; };
                    auto term = [&](u32 k) -> Term {
                        const GatherPoint gp = s_pts[k];
                        f3 dp = mk3(gp.px, gp.py, gp.pz) - hit;
                        float radius = MODE != 0 ? sqrt_refined(dot3(dp, dp))
                                                 : fast_sqrt(__builtin_fmaf(dp.z, dp.z, __builtin_fmaf(dp.y, dp.y, dp.x * dp.x)));
                        f3 bs;
                        if (MODE != 0) {
                            const float coord = div_by_shared(radius, scale, inv_scale);
                            bs = MODE == 1 ? sample_bssrdf_table_lds(s_tab, a.bssrdf_texels, coord)            // :272-276 (the factor is in the LDS texels)
                                           : fudge_over_z * sample_bssrdf_table(a.bssrdf_table, a.bssrdf_texels, coord);
                        } else if (GREY) { float v = dipole_eval_fast(cr, radius); bs = mk3(v, v, v); }               // :278-301
                        else bs = mk3(dipole_eval_fast(cr, radius), dipole_eval_fast(cg, radius), dipole_eval_fast(cb, radius));
                        return Term{bs, mk3(gp.wx, gp.wy, gp.wz)};
                    };
                    // the sum runs in index order, one fused multiply-add per channel and pair
                    auto add = [&](const Term& c) {
                        E.x = __builtin_fmaf(c.bs.x, c.w.x, E.x); E.y = __builtin_fmaf(c.bs.y, c.w.y, E.y); E.z = __builtin_fmaf(c.bs.z, c.w.z, E.z);
                    };
                    u32 k = 0;
                    for (; k + 4 <= nk; k += 4) {
                        Term c0 = term(k), c1 = term(k + 1), c2 = term(k + 2), c3 = term(k + 3);
                        add(c0); add(c1); add(c2); add(c3);
                    }
                    for (; k < nk; k++) add(term(k));
                };
                if (MODE == 0 && grey) run(std::true_type()); else run(std::false_type());
                pairs += nk;
            }
        } else {
            // The sample points are the same for every lane: they come through the scalar cache (uniform address in the
            // constant address space: s_load), straight into scalar registers that the vector instructions take as operands
            // -- no LDS staging, no barriers, and the LDS pipe, which bounded this kernel (two broadcast reads of the point
            // and two gathers of texels per pair), is left with the texels alone.
            const int cur_u = __builtin_amdgcn_readfirstlane(cur);
            const TranslucentSlice sl = a.shade.slices[cur_u];
            typedef const __attribute__((address_space(4))) float* const_floats;
            // the points are walked in Morton order of their positions (the hierarchical gather's `pos` and `order`, built with
            // every set of sample points: rt_api.hip hg_build_all): neighbours in the walk are neighbours in space, which is what
            // makes whole groups of them far from a wave's hits (the skip below).  The sum runs in that order -- a fixed order,
            // other than the reference's index order by rounding only (1e-7, under the radiance tolerance).
            const const_floats spos = (const_floats) (unsigned long long) (a.hg.pos + sl.offset);     // float4 per point, sorted
            const const_floats spay = (const_floats) (unsigned long long) (a.sp_pay + sl.offset);    // their payloads, sorted alike (k_sort_payload)
            const u32 nk = sl.count;
            if (my_ti != cur_u) continue;
            // four points per trip: their sqrt / divide chains are independent and overlap; the sum stays in the order of the walk.
            // `GREY` is a compile-time flag of the loop (two copies): a branch on it between the four terms would keep
            // the scheduler from interleaving their chains.
            auto run = [&](auto grey_tag) {
                constexpr bool GREY = decltype(grey_tag)::value;
                struct Term { f3 bs, w; };
                auto term = [&](u32 k) -> Term {
                    const const_floats sp = spos + 4u * k, q = spay + 4u * k;
                    f3 dp = mk3(sp[0], sp[1], sp[2]) - hit;
                    float radius = MODE != 0 ? sqrt_refined(dot3(dp, dp))
                                             : fast_sqrt(__builtin_fmaf(dp.z, dp.z, __builtin_fmaf(dp.y, dp.y, dp.x * dp.x)));
                    f3 bs;
                    if (MODE != 0) {
                        const float coord = div_by_shared(radius, scale, inv_scale);
                        bs = MODE == 1 ? sample_bssrdf_table_lds(s_tab, a.bssrdf_texels, coord)            // :272-276 (the factor is in the LDS texels)
                           : MODE == 3 ? sample_bssrdf_table_lds_l1(s_tab, a.gather_delta, a.bssrdf_texels, coord)
                                       : fudge_over_z * sample_bssrdf_table(a.bssrdf_table, a.bssrdf_texels, coord);
                    } else if (GREY) { float v = dipole_eval_fast(cr, radius); bs = mk3(v, v, v); }               // :278-301
                    else bs = mk3(dipole_eval_fast(cr, radius), dipole_eval_fast(cg, radius), dipole_eval_fast(cb, radius));
                    return Term{bs, mk3(q[0], q[1], q[2])};
                };
                // the sum runs in the order of the walk, one fused multiply-add per channel and pair
                auto add = [&](const Term& c) {
                    E.x = __builtin_fmaf(c.bs.x, c.w.x, E.x); E.y = __builtin_fmaf(c.bs.y, c.w.y, E.y); E.z = __builtin_fmaf(c.bs.z, c.w.z, E.z);
                };
                // Tabulated profile: beyond its support the filtered value is exactly zero (the table's zero border), and adding
                // 0 * payload leaves the sum as it is -- four points that lie that far from every hit of the wave are skipped
                // (the squared distances are the ones the terms start from; the cut is 0.1 % and two texels outside the support;
                // C4 +1.9 %; testing the points in pairs of two instead: +1 %, the tests cost more than the extra skips save)
                const float cut = (MODE == 1 || MODE == 3) ? scale * ((float) (a.bssrdf_texels + 2) / (float) a.bssrdf_texels) * 1.001f : 0.0f;
                const float cut2 = cut * cut;
                auto dist2 = [&](u32 k) -> float { const const_floats sp = spos + 4u * k; f3 dp = mk3(sp[0], sp[1], sp[2]) - hit; return dot3(dp, dp); };
                u32 k = 0;
                for (; k + 4 <= nk; k += 4) {
                    if ((MODE == 1 || MODE == 3) && all_lanes(fminf(fminf(dist2(k), dist2(k + 1)), fminf(dist2(k + 2), dist2(k + 3))) >= cut2)) continue;
                    Term c0 = term(k), c1 = term(k + 1), c2 = term(k + 2), c3 = term(k + 3);
                    add(c0); add(c1); add(c2); add(c3);
                }
                for (; k < nk; k++) add(term(k));
            };
            if (MODE == 0 && grey) run(std::true_type()); else run(std::false_type());
            pairs += nk;
        }
    }
    if (active) {
        E = E / (float) (a.g.translucent_accumulator_count + 1u);                                       // :332
        f3 emission = E * g.tf[i] / (RT_TAU_F / 2.0f);                                                 // :347
        f3 k3 = mk3(g.kx[i], g.ky[i], g.kz[i]);
        if (a.bg_rows) { g.kx[i] = emission.x * k3.x; g.ky[i] = emission.y * k3.y; g.kz[i] = emission.z * k3.z; } // the item's term: added to its path by k_gather_apply, row by row
        else {
            const u32 path = g.path[i];
            a.ps.rad_x[path] += emission.x * k3.x; a.ps.rad_y[path] += emission.y * k3.y; a.ps.rad_z[path] += emission.z * k3.z;
        }
    }
    if (!a.bg_rows) break;
    } // the next block of items
    pairs = wave_sum(pairs);
    if ((threadIdx.x & 63u) == 0 && pairs) atomicAdd(&a.stats->gather_pairs, (unsigned long long) pairs);
}

// Batched gather, last step: the terms k_gather left in the items (kx, ky, kz) are added to the radiance of their paths, row
// by row -- the rows are the shade launches in the order they ran, so a path that met translucent surfaces at several bounces
// (or, later in the frame, in several samples) gets its terms in that order, whichever workgroups evaluated them: the result
// does not depend on the scheduling of the gather launch.  One workgroup per queue segment (the items of a segment's paths
// sit in that segment of every row); a path has at most one item per row.
__global__ __launch_bounds__(SEG) void k_gather_apply(PassArgs a) {
    const GatherQueue& g = a.gq;
    for (u32 r = 0; r < a.bg_rows; r++) {
        const u32 cnt = a.bg_counts[r * a.bg_cols + blockIdx.x];
        if (threadIdx.x < cnt) {
            const size_t i = (size_t) r * a.bg_stride + (size_t) blockIdx.x * SEG + threadIdx.x;
            const u32 path = g.path[i];
            if (!a.bg_to_sums) { a.ps.rad_x[path] += g.kx[i]; a.ps.rad_y[path] += g.ky[i]; a.ps.rad_z[path] += g.kz[i]; }
            else if (a.g.samples_per_pixel != 1u) { a.ps.sum_x[path] += g.kx[i]; a.ps.sum_y[path] += g.ky[i]; a.ps.sum_z[path] += g.kz[i]; }
            else {
                // fused pass, one sample per pixel: the finished paths left one float4 per pixel (k_shade), or, with the frame's
                // masks in use, no record at all where the sample was (0, 0, 0, 1) -- the term then makes the record
                float4* const s4 = (float4*) a.ps.sum_x;
                bool have = true;
                unsigned long long* const word = frame_record_mask(a) + (path >> 6);
                if (frame_masks_in_use(a)) have = (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (path & 63u)) & 1ull; // (set by atomics of earlier rows: not through the L1)
                if (have) { float4 v = s4[path]; v.x += g.kx[i]; v.y += g.ky[i]; v.z += g.kz[i]; s4[path] = v; }
                else { s4[path] = make_float4(g.kx[i], g.ky[i], g.kz[i], 1.0f); atomicOr(word, 1ull << (path & 63u)); }
            }
        }
        __syncthreads(); // the next row may name the same paths
    }
}

// payloads of the sample points in the Morton order of their positions (HgView::order), for k_gather's walk: run whenever
// the read copy of the points is refreshed (rt_api.hip refresh_read_points)
__global__ __launch_bounds__(BT) void k_sort_payload(const RtSamplePoint* __restrict__ src, const u32* __restrict__ order, const u32* __restrict__ slice_of,
                                                    const TranslucentSlice* __restrict__ slices, u32 n, float4* __restrict__ pay) {
    const u32 j = blockIdx.x * BT + threadIdx.x;
    if (j >= n) return;
    const RtSamplePoint& p = src[slices[slice_of[j]].offset + order[j]];
    pay[j] = make_float4(p.payload[0], p.payload[1], p.payload[2], 0.0f);
}
int launch_sort_payload(hipStream_t s, const RtSamplePoint* src, const u32* order, const u32* slice_of, const TranslucentSlice* slices, u32 n, float4* pay) {
    if (!n) return RT_OK;
    hipLaunchKernelGGL(k_sort_payload, dim3((n + BT - 1) / BT), dim3(BT), 0, s, src, order, slice_of, slices, n, pay);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// ---- hierarchical gather (rt_set_gather_mode) ----------------------------------------
// one workgroup per translucent instance: payloads into sorted order, then node sums from the deepest level up
__global__ __launch_bounds__(SEG) void k_hg_sums(PassArgs a) {
    const TranslucentSlice sl = a.shade.slices[blockIdx.x];
    const HgSlice hs = a.hg.slices[blockIdx.x];
    const RtSamplePoint* pts = a.sp_read + sl.offset;
    float4* pay = a.hg.pay + sl.offset;
    const u32* order = a.hg.order + sl.offset;
    for (u32 j = threadIdx.x; j < sl.count; j += SEG) {
        const RtSamplePoint& p = pts[order[j]];
        pay[j] = make_float4(p.payload[0], p.payload[1], p.payload[2], 0.0f);
    }
    __syncthreads();
    const HgNode* nodes = a.hg.nodes + hs.node_offset;
    float4* sums = a.hg.sums + hs.node_offset;
    for (int depth = (int) hs.max_depth; depth >= 0; depth--) {
        for (u32 k = threadIdx.x; k < hs.node_count; k += SEG) {
            const HgNode n = nodes[k];
            if ((int) n.depth != depth) continue;
            f3 sum = mk3(0.0f, 0.0f, 0.0f);
            if (n.right == 0u) {
                for (u32 j = 0; j < n.count; j++) { float4 w = pay[n.first + j]; sum = sum + mk3(w.x, w.y, w.z); }
            } else {
                float4 l = sums[k + 1u], r = sums[n.right];
                sum = mk3(l.x, l.y, l.z) + mk3(r.x, r.y, r.z);
            }
            sums[k] = make_float4(sum.x, sum.y, sum.z, 0.0f);
        }
        __threadfence_block();
        __syncthreads(); // the next level reads these sums
    }
}

static constexpr int HG_STACK = 24; // the walk holds at most depth + 1 entries; depth <= 22 is checked on the host

// MODE as in k_gather.  One hit per lane, per-lane preorder walk with the stack in LDS.
template <int MODE>
__global__ __launch_bounds__(GATHER_BLOCK) void k_gather_hg(PassArgs a) {
    __shared__ float4 s_tab[GATHER_TABLE_MAX + 4];
    __shared__ u32 s_stack[HG_STACK * GATHER_BLOCK];
    const u32 n_items = a.goff[a.n_segments];
    if (blockIdx.x * GATHER_BLOCK >= n_items) return; // uniform
    const u32 j = blockIdx.x * GATHER_BLOCK + threadIdx.x;
    const bool active = j < n_items;
    u32 i;
    {
        const u32 jj = active ? j : n_items - 1u;
        u32 lo = 0, hi = a.n_segments;
        while (hi - lo > 1u) { u32 mid = (lo + hi) >> 1; if (a.goff[mid] <= jj) lo = mid; else hi = mid; }
        i = lo * SEG + (jj - a.goff[lo]);
    }
    if (MODE == 1) {
        tab_fill_lds<GATHER_BLOCK>(s_tab, a.bssrdf_table, a.bssrdf_texels, a.fudge / (a.g.translucent_bssrdf_scale * a.g.translucent_bssrdf_scale)); // as in k_gather
        __syncthreads();
    }
    if (!active) return;
    const GatherQueue& g = a.gq;
    const f3 hit = mk3(g.hx[i], g.hy[i], g.hz[i]);
    const int ti = g.ti[i];
    const TranslucentSlice sl = a.shade.slices[ti];
    const HgSlice hs = a.hg.slices[ti];
    const HgNode* nodes = a.hg.nodes + hs.node_offset;
    const float4* sums = a.hg.sums + hs.node_offset;
    const float4* pos = a.hg.pos + sl.offset;
    const float4* pay = a.hg.pay + sl.offset;
    const float scale = a.g.translucent_bssrdf_scale;
    const float fudge_over_z = a.fudge / (scale * scale); // as in k_gather
    const float inv_scale = 1.0f / scale;
    const float eta = a.g.translucent_refractive_index;
    DipoleConst cr = dipole_constants(a.g.translucent_scattering[0], a.g.translucent_absorption[0], eta);
    DipoleConst cg = dipole_constants(a.g.translucent_scattering[1], a.g.translucent_absorption[1], eta);
    DipoleConst cb = dipole_constants(a.g.translucent_scattering[2], a.g.translucent_absorption[2], eta);
    // the shipped defaults use one scattering / absorption pair for all channels (main.cpp:420-422): evaluate once
    const bool grey = a.g.translucent_scattering[0] == a.g.translucent_scattering[1] && a.g.translucent_scattering[0] == a.g.translucent_scattering[2] &&
                      a.g.translucent_absorption[0] == a.g.translucent_absorption[1] && a.g.translucent_absorption[0] == a.g.translucent_absorption[2];
    auto bssrdf = [&](float radius) -> f3 {
        if (MODE != 0) {
            const float coord = div_by_shared(radius, scale, inv_scale);
            return MODE == 1 ? sample_bssrdf_table_lds(s_tab, a.bssrdf_texels, coord)
                             : fudge_over_z * sample_bssrdf_table(a.bssrdf_table, a.bssrdf_texels, coord);
        }
        if (grey) { float v = dipole_eval_fast(cr, radius); return mk3(v, v, v); }
        return mk3(dipole_eval_fast(cr, radius), dipole_eval_fast(cg, radius), dipole_eval_fast(cb, radius));
    };
    f3 E = mk3(0.0f, 0.0f, 0.0f);
    u32 terms = 0;
    u32* stack = s_stack + threadIdx.x;
    int sp = 0;
    if (hs.node_count) stack[(sp++) * GATHER_BLOCK] = 0u;
    while (sp > 0) {
        const u32 id = stack[(--sp) * GATHER_BLOCK];
        const HgNode n = nodes[id];
        if (n.right == 0u) {
            for (u32 k = 0; k < n.count; k++) {
                const float4 p = pos[n.first + k], w = pay[n.first + k];
                f3 dp = mk3(p.x, p.y, p.z) - hit;
                E = E + bssrdf(MODE != 0 ? sqrt_refined(dot3(dp, dp)) : fast_sqrt(dot3(dp, dp))) * mk3(w.x, w.y, w.z);
            }
            terms += n.count;
            continue;
        }
        f3 d = mk3(n.cx, n.cy, n.cz) - hit;
        float d2 = dot3(d, d);
        if (d2 > n.radius2 && (float) n.count * sl.mean_area < a.hg.max_solid_angle * d2) {
            const float4 w = sums[id];
            E = E + bssrdf(MODE != 0 ? sqrt_refined(d2) : fast_sqrt(d2)) * mk3(w.x, w.y, w.z);
            terms += 1u;
        } else if (sp + 2 <= HG_STACK) {
            stack[(sp++) * GATHER_BLOCK] = n.right;
            stack[(sp++) * GATHER_BLOCK] = id + 1u;
        } else a.stats->hg_overflow = 1ull; // cannot happen while the host's depth check holds (hg_build_all); never silently
    }
    E = E / (float) (a.g.translucent_accumulator_count + 1u);                                           // :332
    f3 emission = E * g.tf[i] / (RT_TAU_F / 2.0f);                                                     // :347
    f3 k3 = mk3(g.kx[i], g.ky[i], g.kz[i]);
    const u32 path = g.path[i];
    a.ps.rad_x[path] += emission.x * k3.x; a.ps.rad_y[path] += emission.y * k3.y; a.ps.rad_z[path] += emission.z * k3.z;
    atomicAdd(&a.stats->gather_pairs, (unsigned long long) terms);
}

__global__ __launch_bounds__(BT) void k_fill_capture(RtHit* capture, size_t n) {
    size_t i = (size_t) blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    RtHit r; r.instance = -1; r.geometry = -1; r.primitive = -1; r.t = 0.0f; r.u = 0.0f; r.v = 0.0f; r.rng_after = 0u; r.valid = 0u;
    capture[i] = r;
}

// ---- launchers ---------------------------------------------------------------------------------------------------
static inline dim3 blocks(size_t n, int bt) { return dim3((unsigned) ((n + bt - 1) / bt)); }

int launch_probe_raygen(hipStream_t s, const PassArgs& a, u32 sample_index) {
    if (!a.n_points) return RT_OK;
    hipLaunchKernelGGL(k_probe_raygen, dim3(a.n_segments), dim3(SEG), 0, s, a, sample_index);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
static size_t pass_paths(const PassArgs& a) { return (size_t) a.n_segments * SEG; }
int launch_traverse(hipStream_t s, const PassArgs& a, u32 in_queue, bool count_traversal, bool primary, u32 sample_index, bool tail_shade) {
    size_t n = pass_paths(a);
    if (!n) return RT_OK;
    const u32 tail = (tail_shade && !primary && !count_traversal && a.scene.flat_instances != 2u) ? 1u : 0u;
    dim3 grid = blocks(n, TRAVERSE_WAVE), block(TRAVERSE_WAVE);
    const u32 n_tasks = grid.x;
    // 6 waves per SIMD = 80 VGPRs = what the LDS stacks allow as well (6.5 KB per wave); the persistent variant keeps 5 (it needs 96)
    const bool persistent = a.persistent_slots != 0 && !count_traversal && (primary ? (a.persistent & 2u) != 0 : (a.persistent & 1u) != 0) && n_tasks > a.persistent_slots;
    if (!primary && !persistent && a.pool_groups != 0u && a.scene.flat_instances == 0u && a.stack_in_lds != 0u) {
        // two-level scenes, queued rays: pooled traversal (rt_dev_pool.h), one persistent workgroup per 256-ray pool
        const int rows = a.stack_in_lds == 2u ? 33 : 17;
#if defined(RT_POOL_DEBUG)
        const size_t lds = ((size_t) (rows + PF_COUNT + 5) * POOL_RAYS + 32) * sizeof(int);
#else
        const size_t lds = ((size_t) (rows + PF_COUNT + 4) * POOL_RAYS + 24) * sizeof(int);
#endif
        const void* fn = count_traversal ? (rows == 33 ? (const void*) k_traverse_pool<true, 33> : (const void*) k_traverse_pool<true, 17>)
                                         : (rows == 33 ? (const void*) k_traverse_pool<false, 33> : (const void*) k_traverse_pool<false, 17>);
        RT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
#if defined(RT_POOL_DEBUG)
        RT_HIP(hipMemsetAsync(a.hits.t, 0xff, 16 * (size_t) a.n_paths, s)); // a hit record the kernel fails to write reads as a miss (id -1) in k_shade
#endif
        const u32 groups = a.pool_groups < a.n_segments ? a.pool_groups : a.n_segments;
        dim3 pgrid(groups), pblock(POOL_RAYS);
        if (count_traversal) { if (rows == 33) hipLaunchKernelGGL((k_traverse_pool<true, 33>), pgrid, pblock, lds, s, a, in_queue);
                               else hipLaunchKernelGGL((k_traverse_pool<true, 17>), pgrid, pblock, lds, s, a, in_queue); }
        else if (rows == 33) hipLaunchKernelGGL((k_traverse_pool<false, 33>), pgrid, pblock, lds, s, a, in_queue);
        else hipLaunchKernelGGL((k_traverse_pool<false, 17>), pgrid, pblock, lds, s, a, in_queue);
    } else if (persistent) {
        dim3 pgrid(a.persistent_slots);
        if (primary) hipLaunchKernelGGL((k_traverse<false, 5, true, true, true>), pgrid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else hipLaunchKernelGGL((k_traverse<false, 5, false, true, true>), pgrid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
    } else if (a.scene.flat_instances == 1u && a.scene.flat_inl_inst >= 0) {
        // one tree over the big instance of two, the small one tested first
        if (primary) {
            if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, true, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (a.flat_waves8) hipLaunchKernelGGL((k_traverse<false, 8, true, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 6, true, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        } else {
            if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, false, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (a.flat_waves8) hipLaunchKernelGGL((k_traverse<false, 8, false, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 6, false, false, false, 3>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        }
    } else if (a.scene.flat_instances == 1u || a.scene.flat_instances == 2u) {
        // one tree over the triangles of all (one or two) instances; count_traversal: the same walk with the counters on
        const bool two = a.scene.flat_instances == 2u;
        if (primary) {
            if (count_traversal) { if (two) hipLaunchKernelGGL((k_traverse<true, 1, true, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
                                   else hipLaunchKernelGGL((k_traverse<true, 1, true, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail); }
            else if (two && a.flat_waves7) hipLaunchKernelGGL((k_traverse<false, 7, true, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (two) hipLaunchKernelGGL((k_traverse<false, 6, true, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (a.flat_waves8) hipLaunchKernelGGL((k_traverse<false, 8, true, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 6, true, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        } else {
            if (count_traversal) { if (two) hipLaunchKernelGGL((k_traverse<true, 1, false, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
                                   else hipLaunchKernelGGL((k_traverse<true, 1, false, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail); }
            else if (two && a.flat_waves7) hipLaunchKernelGGL((k_traverse<false, 7, false, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (two) hipLaunchKernelGGL((k_traverse<false, 6, false, false, false, 2>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else if (a.flat_waves8) hipLaunchKernelGGL((k_traverse<false, 8, false, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 6, false, false, false, 1>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        }
    } else if (a.scene.grid_cells != 0u && a.scene.flat_instances == 0u) {
        // two-level scene with a grid top level (the host has checked that the bottom-level stacks fit RT_GRID_STACK_ROWS)
        if (primary) {
            if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, true, false, false, 4>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 5, true, false, false, 4>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        } else {
            if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, false, false, false, 4>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
            else hipLaunchKernelGGL((k_traverse<false, 5, false, false, false, 4>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        }
    } else if (primary) {
        if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, true, false, true>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else if (a.stack_in_lds == 2u) hipLaunchKernelGGL((k_traverse<false, 5, true, false, false>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else if (a.stack_in_lds) hipLaunchKernelGGL((k_traverse<false, 6, true, false, false>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else hipLaunchKernelGGL((k_traverse<false, 6, true, false, true>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
    } else {
        if (count_traversal) hipLaunchKernelGGL((k_traverse<true, 1, false, false, true>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else if (a.stack_in_lds == 2u) hipLaunchKernelGGL((k_traverse<false, 5, false, false, false>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else if (a.stack_in_lds) hipLaunchKernelGGL((k_traverse<false, 6, false, false, false>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
        else hipLaunchKernelGGL((k_traverse<false, 6, false, false, true>), grid, block, 0, s, a, in_queue, sample_index, n_tasks, tail);
    }
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_shade(hipStream_t s, const PassArgs& a, u32 in_queue, u32 bounce, u32 sample_index, bool primary, bool fused, bool walk) {
    size_t n = pass_paths(a);
    if (!n) return RT_OK;
    dim3 grid(a.n_segments), block(SEG);
    if (primary && fused && a.fuse_primary) { // bounce 0 in one kernel: walk and shade (no launch_traverse before it)
        const size_t lds = (size_t) (a.flat_rows > 14u ? a.flat_rows : 14u) * SEG * sizeof(int);
        if (a.scene.flat_inl_inst >= 0) {
            RT_HIP(hipFuncSetAttribute((const void*) k_shade<true, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            hipLaunchKernelGGL((k_shade<true, true, 3>), grid, block, lds, s, a, in_queue, bounce, sample_index);
        } else {
            RT_HIP(hipFuncSetAttribute((const void*) k_shade<true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            hipLaunchKernelGGL((k_shade<true, true, 1>), grid, block, lds, s, a, in_queue, bounce, sample_index);
        }
    }
    else if (!primary && fused && walk) { // a queued bounce in one kernel: walk and shade (no launch_traverse before it)
        const size_t lds = (size_t) (a.flat_rows > 14u ? a.flat_rows : 14u) * SEG * sizeof(int);
        if (a.scene.flat_inl_inst >= 0) {
            RT_HIP(hipFuncSetAttribute((const void*) k_shade<false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            hipLaunchKernelGGL((k_shade<false, true, 3>), grid, block, lds, s, a, in_queue, bounce, sample_index);
        } else {
            RT_HIP(hipFuncSetAttribute((const void*) k_shade<false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            hipLaunchKernelGGL((k_shade<false, true, 1>), grid, block, lds, s, a, in_queue, bounce, sample_index);
        }
    }
    else if (primary && fused) hipLaunchKernelGGL((k_shade<true, true>), grid, block, 0, s, a, in_queue, bounce, sample_index);
    else if (primary) hipLaunchKernelGGL((k_shade<true, false>), grid, block, 0, s, a, in_queue, bounce, sample_index);
    else if (fused) hipLaunchKernelGGL((k_shade<false, true>), grid, block, 0, s, a, in_queue, bounce, sample_index);
    else hipLaunchKernelGGL((k_shade<false, false>), grid, block, 0, s, a, in_queue, bounce, sample_index);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_gather(hipStream_t s, const PassArgs& a) {
    size_t n = pass_paths(a);
    if (!n) return RT_OK;
    hipLaunchKernelGGL(k_gather_offsets, dim3(1), dim3(SEG), 0, s, a);
    dim3 grid = blocks(n, GATHER_BLOCK), block(GATHER_BLOCK);
    if (a.hg.enabled) {
        if (a.g.translucent_bssrdf_scale == 0.0f) hipLaunchKernelGGL(k_gather_hg<0>, grid, block, 0, s, a);
        else if (a.bssrdf_texels <= GATHER_TABLE_MAX) hipLaunchKernelGGL(k_gather_hg<1>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(k_gather_hg<2>, grid, block, 0, s, a);
        RT_HIP(hipGetLastError());
        return RT_OK;
    }
    if (a.g.translucent_bssrdf_scale == 0.0f) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, s, a); // raytracing.hlsl:316-320
    else if (a.bssrdf_texels <= GATHER_TABLE_MAX && a.gather_delta) hipLaunchKernelGGL(k_gather<3>, grid, block, 0, s, a);
    else if (a.bssrdf_texels <= GATHER_TABLE_MAX) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_gather<2>, grid, block, 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
template <int BLOCK>
static void launch_gather_batched_block(hipStream_t s, const PassArgs& a) {
    dim3 grid(a.bg_groups), block(BLOCK);
    if (a.g.translucent_bssrdf_scale == 0.0f) hipLaunchKernelGGL((k_gather<0, BLOCK>), grid, block, 0, s, a); // raytracing.hlsl:316-320
    else if (a.bssrdf_texels <= GATHER_TABLE_MAX && a.gather_delta) hipLaunchKernelGGL((k_gather<3, BLOCK>), grid, block, 0, s, a);
    else if (a.bssrdf_texels <= GATHER_TABLE_MAX) hipLaunchKernelGGL((k_gather<1, BLOCK>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_gather<2, BLOCK>), grid, block, 0, s, a);
}
int launch_gather_batched(hipStream_t s, const PassArgs& a) {
    if (!a.bg_rows || !a.bg_cols || !a.bg_groups) return RT_OK;
    hipLaunchKernelGGL(k_gather_offsets, dim3(1), dim3(SEG), 0, s, a);
    if (a.bg_block == 1024u) launch_gather_batched_block<1024>(s, a);
    else if (a.bg_block == 512u) launch_gather_batched_block<512>(s, a);
    else launch_gather_batched_block<256>(s, a);
    hipLaunchKernelGGL(k_gather_apply, dim3(a.bg_cols), dim3(SEG), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_hg_sums(hipStream_t s, const PassArgs& a, u32 n_slices) {
    if (!n_slices || !a.hg.enabled) return RT_OK;
    hipLaunchKernelGGL(k_hg_sums, dim3(n_slices), dim3(SEG), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_camera_resolve(hipStream_t s, const PassArgs& a) {
    size_t threads = (size_t) a.n_local_tiles * 1024;
    if (!threads) return RT_OK;
    hipLaunchKernelGGL(k_camera_resolve, blocks(threads, BT), dim3(BT), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_frame_resolve(hipStream_t s, const PassArgs& a) {
    size_t threads = (size_t) a.n_local_tiles * 1024;
    if (!threads) return RT_OK;
    hipLaunchKernelGGL(k_frame_resolve, blocks(threads, BT), dim3(BT), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_probe_resolve(hipStream_t s, const PassArgs& a) {
    if (!a.n_points) return RT_OK;
    hipLaunchKernelGGL(k_probe_resolve, blocks(a.n_points, BT), dim3(BT), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_fill_capture(hipStream_t s, const PassArgs& a) {
    if (!a.capture) return RT_OK;
    size_t n = (size_t) (a.g.bounces_per_sample + 1) * a.width * a.height;
    hipLaunchKernelGGL(k_fill_capture, blocks(n, BT), dim3(BT), 0, s, a.capture, n);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_pack_tiles(hipStream_t s, const PassArgs& a, void* packed) {
    size_t threads = (size_t) a.n_local_tiles * 1024;
    if (!threads) return RT_OK;
    hipLaunchKernelGGL(k_pack_tiles, blocks(threads, BT), dim3(BT), 0, s, a, (float4*) packed);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_unpack_tiles(hipStream_t s, const PassArgs& a, const void* packed) {
    size_t threads = (size_t) a.n_local_tiles * 1024;
    if (!threads) return RT_OK;
    hipLaunchKernelGGL(k_unpack_tiles, blocks(threads, BT), dim3(BT), 0, s, a, (const float4*) packed);
    RT_HIP(hipGetLastError());
    return RT_OK;
}
int launch_resolve_only(hipStream_t s, const PassArgs& a) {
    size_t n = (size_t) a.width * a.height;
    if (!n) return RT_OK;
    hipLaunchKernelGGL(k_resolve_only, blocks(n, BT), dim3(BT), 0, s, a);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

} // namespace rti
