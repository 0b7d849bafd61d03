// rt_bluenoise.hip -- blue-noise (Poisson-disk) surface sample points for translucent meshes.
// Restates the four compute kernels of src/bluenoise.hlsl:60-245 and their host schedule src/bluenoise.cpp:263-351
// for gfx950.  Differences in mechanism, not in result definition:
//   * the bitonic network of single-thread dispatches (bluenoise.cpp:273-284) is a stable radix sort; the order
//     among points of one cell, unspecified in the reference, is (cell id, initial index);
//   * the hash table keeps, per bucket, the 5 smallest cell ids (the reference keeps whichever 5 win the atomics);
//   * the dart-throwing loop (24 phase-group dispatches per trial with a CPU read-back, bluenoise.cpp:327-351) is read
//     back every fourth trial, and accepted points are ordered by dispatch id instead of by the atomic append.
#include "rt_internal.h"

#include <algorithm>

namespace rti {

using namespace rtd;

struct BnEntry { // HashtableEntry, bluenoise.h:55-59
    u32 cell_id;
    u32 first_index;
    float sel[3];
};
static constexpr int BN_BUCKET = 5; // BLUENOISE_HASHTABLE_BUCKET_SIZE, bluenoise.h:61

RT_DEV u32 bn_cell_id_from_position(const BluenoiseParams& p, f3 pos) { // bluenoise.hlsl:48-58
    float cx = (pos.x - p.grid_origin[0]) / p.grid_cell_width;
    float cy = (pos.y - p.grid_origin[1]) / p.grid_cell_width;
    float cz = (pos.z - p.grid_origin[2]) / p.grid_cell_width;
    float fx = floorf(cx), fy = floorf(cy), fz = floorf(cz);
    u32 ix = fx > 0.0f ? (u32) fx : 0u, iy = fy > 0.0f ? (u32) fy : 0u, iz = fz > 0.0f ? (u32) fz : 0u;
    return ix + iy * p.grid_dims[0] + iz * p.grid_dims[0] * p.grid_dims[1];
}

RT_DEV f3 ld_pos(const RtVertex& v) { return mk3(v.position[0], v.position[1], v.position[2]); }

// generate_initial_sample_points, bluenoise.hlsl:60-103
__global__ void k_bn_initial(BluenoiseParams p, const RtVertex* __restrict__ verts, const uint16_t* __restrict__ indices,
                             const float* __restrict__ partial, float4* __restrict__ pos, u32* __restrict__ cell, u32* __restrict__ idx) {
    u32 index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= p.initial_count) return;
    u32 rng = hash_u2(index, p.rng_seed);
    float x = random01(rng) * p.total_surface_area;
    u32 min_index = 0, max_index = p.triangles_count - 1;
    while (min_index != max_index) {
        u32 i = (min_index + max_index) / 2;
        if (x > partial[i]) min_index = i + 1; else max_index = i;
    }
    const uint16_t* tri = indices + p.indices_offset + 3u * min_index;
    f3 p0 = ld_pos(verts[tri[0]]), p1 = ld_pos(verts[tri[1]]), p2 = ld_pos(verts[tri[2]]);
    float ux = random01(rng);
    float uy = random01(rng);
    ux = sqrtf(ux); uy = uy * ux; ux = 1.0f - ux;
    f3 q = p0 + ux * (p1 - p0) + uy * (p2 - p0);
    pos[index] = make_float4(q.x, q.y, q.z, as_f32(min_index));
    cell[index] = bn_cell_id_from_position(p, q);
    idx[index] = index;
}

__global__ void k_bn_gather(const float4* __restrict__ pos, const u32* __restrict__ perm, u32 n, float4* __restrict__ spos) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) spos[i] = pos[perm[i]];
}

RT_DEV bool bn_cell_exists(const u32* scell, u32 n, u32 c) { // sorted array membership
    u32 lo = 0, hi = n;
    while (lo < hi) { u32 mid = (lo + hi) / 2; if (scell[mid] < c) lo = mid + 1; else hi = mid; }
    return lo < n && scell[lo] == c;
}

// build_hashtable, bluenoise.hlsl:133-158, with a deterministic overflow rule
__global__ void k_bn_table(const u32* __restrict__ scell, u32 n, u32 buckets, BnEntry* __restrict__ entries,
                           u32* __restrict__ bucket_count, u32* __restrict__ out_counts /* [0] entries, [1] overflow */) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 c = scell[i];
    if (i > 0 && scell[i - 1] == c) return;
    u32 h = c % buckets;
    u32 rank = 0;
    for (u32 lower = h; lower < c && rank < BN_BUCKET; lower += buckets)
        if (bn_cell_exists(scell, n, lower)) rank++;
    if (rank >= BN_BUCKET) { atomicAdd(&out_counts[1], 1u); return; }
    BnEntry e; e.cell_id = c; e.first_index = i; e.sel[0] = e.sel[1] = e.sel[2] = inf_f32();
    entries[h * BN_BUCKET + rank] = e;
    atomicMax(&bucket_count[h], rank + 1u);
    atomicAdd(&out_counts[0], 1u);
}

RT_DEV int bn_search(const BnEntry* entries, const u32* bucket_count, u32 buckets, u32 cell_id) { // bluenoise.hlsl:160-170
    u32 h = cell_id % buckets;
    u32 cnt = bucket_count[h];
    for (u32 k = 0; k < cnt; k++) if (entries[h * BN_BUCKET + k].cell_id == cell_id) return (int) (h * BN_BUCKET + k);
    return -1;
}

// generate_sample_points (bluenoise.hlsl:180-245): one launch per phase group and trial, one thread per cell of the
// group, like the reference's Dispatch(ceil(gx/3), ceil(gy/3), ceil(gz/3)) (bluenoise.cpp:327-351).  Cells of one phase
// group are three apart on every axis and a candidate looks at most two cells away, so the candidates of a launch
// neither see nor disturb each other: the launch is a plain data-parallel map.  The reference appends accepted points
// with an atomic counter (order = whatever the hardware does); here the order is defined as dispatch order -- (trial,
// position of the phase group in the sequence, dispatch id) -- and applied at the end: an accepted candidate only
// leaves its order key in its hash-table entry, and after the last trial the entries are sorted by that key
// (k_bn_emit writes points and normals in sorted order).  Round 1 ran the whole loop in ONE workgroup to get the
// order for free: 42 ms for 759 points, 0.75 s for 4,911, 4.5 s for 20,058 (profiles/r02_bluenoise.txt).
struct BnControl { u32 count, prev, done, trials; };
static constexpr u32 BN_NOT_ACCEPTED = 0xffffffffu;

__global__ __launch_bounds__(256) void k_bn_try(BluenoiseParams p, u32 ph, u32 trial, const float4* __restrict__ spos, const u32* __restrict__ scell,
                                                BnEntry* entries, const u32* __restrict__ bucket_count, BnControl* ctl,
                                                u32* __restrict__ key_id, u32* __restrict__ key_round) {
    if (__hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return; // a trial added nothing: the loop is over (bluenoise.cpp:349)
    const u32 gx = (p.grid_dims[0] + 2) / 3, gy = (p.grid_dims[1] + 2) / 3, gz = (p.grid_dims[2] + 2) / 3;
    const u32 id = blockIdx.x * 256u + threadIdx.x;
    if (id >= gx * gy * gz) return;
    const u32 pg = p.phase_sequence[ph];
    const u32 cx = 3 * (id % gx) + pg % 3, cy = 3 * ((id / gx) % gy) + (pg / 3) % 3, cz = 3 * (id / (gx * gy)) + pg / 9;
    if (cx >= p.grid_dims[0] || cy >= p.grid_dims[1] || cz >= p.grid_dims[2]) return;
    const u32 cell_id = cx + cy * p.grid_dims[0] + cz * p.grid_dims[0] * p.grid_dims[1];
    const int entry_index = bn_search(entries, bucket_count, p.buckets, cell_id);
    if (entry_index < 0 || !isinf(entries[entry_index].sel[0])) return;
    const unsigned long long ti = (unsigned long long) entries[entry_index].first_index + trial;
    if (ti >= p.initial_count || scell[ti] != cell_id) return;
    const float4 tp = spos[ti];
    const float rr2 = p.rejection_radius * p.rejection_radius;
    for (int xo = -2; xo <= 2; xo++) for (int yo = -2; yo <= 2; yo++) for (int zo = -2; zo <= 2; zo++) {
        if ((xo == 0 && yo == 0 && zo == 0) || (abs(xo) == 2 && abs(yo) == 2 && abs(zo) == 2)) continue;
        const int nx = (int) cx + xo, ny = (int) cy + yo, nz = (int) cz + zo;
        if (nx < 0 || ny < 0 || nz < 0 || nx >= (int) p.grid_dims[0] || ny >= (int) p.grid_dims[1] || nz >= (int) p.grid_dims[2]) continue;
        const int ne = bn_search(entries, bucket_count, p.buckets, (u32) nx + (u32) ny * p.grid_dims[0] + (u32) nz * p.grid_dims[0] * p.grid_dims[1]);
        if (ne >= 0) {
            const f3 d = mk3(entries[ne].sel[0], entries[ne].sel[1], entries[ne].sel[2]) - mk3(tp.x, tp.y, tp.z);
            if (dot3(d, d) <= rr2) return;
        }
    }
    entries[entry_index].sel[0] = tp.x; entries[entry_index].sel[1] = tp.y; entries[entry_index].sel[2] = tp.z;
    entries[entry_index].first_index = (u32) ti; // the accepted candidate (k_bn_emit reads its triangle from spos[ti].w); the entry takes no further trials
    key_id[entry_index] = id;
    key_round[entry_index] = trial * p.phases + ph;
    atomicAdd(&ctl->count, 1u);
}

// end of a trial (all phase groups): the loop ends with the first trial that adds no point (bluenoise.cpp:347-350)
__global__ void k_bn_trial_end(BnControl* ctl) {
    if (ctl->done) return;
    if (ctl->count == ctl->prev) ctl->done = 1u;
    ctl->prev = ctl->count;
    ctl->trials += 1u;
}

__global__ void k_bn_order_init(u32* __restrict__ key_id, u32* __restrict__ key_round, u32* __restrict__ vals, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { key_id[i] = BN_NOT_ACCEPTED; key_round[i] = BN_NOT_ACCEPTED; vals[i] = i; }
}
// keys for the two sorting passes: unaccepted entries sort behind every accepted one
__global__ void k_bn_order_keys(const u32* __restrict__ src, const u32* __restrict__ vals, u32* __restrict__ keys, u32 n, u32 limit) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32 k = src[vals[i]]; keys[i] = k < limit ? k : limit; }
}

// accepted points in dispatch order: world-space position, zero payload, normalised world-space normal (bluenoise.hlsl:226-243)
__global__ void k_bn_emit(BluenoiseParams p, const u32* __restrict__ order, u32 count, const BnEntry* __restrict__ entries,
                          const float4* __restrict__ spos, const RtVertex* __restrict__ verts, const uint16_t* __restrict__ indices,
                          RtSamplePoint* __restrict__ out_points, float* __restrict__ out_normals) {
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= count) return;
    const BnEntry e = entries[order[slot]];
    const f3 q = mk3(e.sel[0], e.sel[1], e.sel[2]);
    RtSamplePoint sp;
    f3 wp = xform_point44(p.transform, q);
    sp.position[0] = wp.x; sp.position[1] = wp.y; sp.position[2] = wp.z;
    sp.payload[0] = 0.0f; sp.payload[1] = 0.0f; sp.payload[2] = 0.0f;
    out_points[slot] = sp;
    const uint16_t* tri = indices + p.indices_offset + 3u * as_u32(spos[e.first_index].w);
    const RtVertex& v0 = verts[tri[0]]; const RtVertex& v1 = verts[tri[1]]; const RtVertex& v2 = verts[tri[2]];
    // get_barycentrics, prelude.h:248-265
    f3 e0 = ld_pos(v1) - ld_pos(v0), e1 = ld_pos(v2) - ld_pos(v0), e2 = q - ld_pos(v0);
    float d00 = dot3(e0, e0), d01 = dot3(e0, e1), d11 = dot3(e1, e1), d20 = dot3(e2, e0), d21 = dot3(e2, e1);
    float denom = d00 * d11 - d01 * d01;
    float bx = (d11 * d20 - d01 * d21) / denom, by = (d00 * d21 - d01 * d20) / denom;
    f3 n0 = mk3(v0.normal[0], v0.normal[1], v0.normal[2]);
    f3 n1 = mk3(v1.normal[0], v1.normal[1], v1.normal[2]);
    f3 n2 = mk3(v2.normal[0], v2.normal[1], v2.normal[2]);
    f3 nrm = n0;
    nrm = nrm + bx * (n1 - n0);
    nrm = nrm + by * (n2 - n0);
    nrm = normalize3(xform_dir44(p.transform, nrm));
    out_normals[3 * slot] = nrm.x; out_normals[3 * slot + 1] = nrm.y; out_normals[3 * slot + 2] = nrm.z;
}

static int bits_for(unsigned long long values) { int b = 1; while (b < 32 && (1ull << b) < values) b++; return b; }

int bluenoise_generate(hipStream_t s, const BluenoiseParams& p, const RtVertex* verts, const uint16_t* indices,
                       const float* partial_areas, RtSamplePoint** points_out, float** normals_out, u32* count_out) {
    const u32 N0 = p.initial_count;
    const size_t cap = (size_t) p.buckets * BN_BUCKET;
    const u32 n_sort = (u32) std::max<size_t>(N0, cap);
    // one allocation for all the temporaries (a dozen hipMalloc / hipFree pairs cost more than the kernels of a small mesh)
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_pos = carve(sizeof(float4) * (size_t) N0), o_spos = carve(sizeof(float4) * (size_t) N0);
    const size_t o_cell = carve(4 * (size_t) n_sort), o_idx = carve(4 * (size_t) n_sort), o_cell_tmp = carve(4 * (size_t) n_sort), o_idx_tmp = carve(4 * (size_t) n_sort);
    const size_t o_scratch = carve(sort_scratch_bytes(n_sort));
    const size_t o_entries = carve(sizeof(BnEntry) * cap), o_bucket = carve(4 * (size_t) p.buckets), o_counts = carve(16), o_ctl = carve(sizeof(BnControl));
    const size_t o_key_id = carve(4 * cap), o_key_round = carve(4 * cap), o_okeys = carve(4 * cap), o_ovals = carve(4 * cap);
    char* arena = nullptr; RtSamplePoint* points = nullptr; float* normals = nullptr;
    int status = RT_OK;
    auto fail = [&](hipError_t e, const char* what) { set_error("bluenoise: %s failed: %s", what, hipGetErrorString(e)); status = RT_ERR_DEVICE; };
    hipError_t e;
#define BN_TRY(call) if (status == RT_OK && (e = (call)) != hipSuccess) fail(e, #call)
    BN_TRY(hipMalloc(&arena, off));
    BN_TRY(hipMalloc(&points, sizeof(RtSamplePoint) * cap));
    BN_TRY(hipMalloc(&normals, 12 * cap));
    float4 *pos = (float4*) (arena + o_pos), *spos = (float4*) (arena + o_spos);
    u32 *cell = (u32*) (arena + o_cell), *idx = (u32*) (arena + o_idx), *cell_tmp = (u32*) (arena + o_cell_tmp), *idx_tmp = (u32*) (arena + o_idx_tmp);
    void* scratch = arena + o_scratch;
    BnEntry* entries = (BnEntry*) (arena + o_entries);
    u32 *bucket_count = (u32*) (arena + o_bucket), *counts = (u32*) (arena + o_counts);
    BnControl* ctl = (BnControl*) (arena + o_ctl);
    u32 *key_id = (u32*) (arena + o_key_id), *key_round = (u32*) (arena + o_key_round), *okeys = (u32*) (arena + o_okeys), *ovals = (u32*) (arena + o_ovals);
    BN_TRY(hipMemsetAsync(bucket_count, 0, 4 * (size_t) p.buckets, s));
    BN_TRY(hipMemsetAsync(counts, 0, 16, s));
    BN_TRY(hipMemsetAsync(ctl, 0, sizeof(BnControl), s));
    if (status == RT_OK) {
        hipLaunchKernelGGL(k_bn_initial, dim3((N0 + 255) / 256), dim3(256), 0, s, p, verts, indices, partial_areas, pos, cell, idx);
        const unsigned long long cells = (unsigned long long) p.grid_dims[0] * p.grid_dims[1] * p.grid_dims[2];
        status = radix_sort_pairs(s, cell, idx, cell_tmp, idx_tmp, scratch, N0, bits_for(cells));
    }
    const u32 gx = (p.grid_dims[0] + 2) / 3, gy = (p.grid_dims[1] + 2) / 3, gz = (p.grid_dims[2] + 2) / 3;
    const u32 per_phase = gx * gy * gz;
    BnControl host_ctl = { 0, 0, 0, 0 };
    if (status == RT_OK) {
        hipLaunchKernelGGL(k_bn_gather, dim3((N0 + 255) / 256), dim3(256), 0, s, pos, idx, N0, spos);
        hipLaunchKernelGGL(k_bn_table, dim3((N0 + 255) / 256), dim3(256), 0, s, cell, N0, p.buckets, entries, bucket_count, counts);
        hipLaunchKernelGGL(k_bn_order_init, dim3((unsigned) ((cap + 255) / 256)), dim3(256), 0, s, key_id, key_round, ovals, (u32) cap);
        BN_TRY(hipGetLastError());
        // the trial loop of bluenoise.cpp:327-351, four trials per read-back (the reference reads the counter back after
        // every trial); launches behind the trial that found nothing return at once (ctl->done)
        const u32 max_trials = 0xffffffffu / std::max(p.phases, 1u) - 8u; // key_round = trial * phases + ph stays below BN_NOT_ACCEPTED
        for (u32 trial = 0; status == RT_OK && !host_ctl.done && trial < max_trials;) {
            for (u32 k = 0; k < 4; k++, trial++) {
                for (u32 ph = 0; ph < p.phases; ph++)
                    hipLaunchKernelGGL(k_bn_try, dim3((per_phase + 255) / 256), dim3(256), 0, s, p, ph, trial, spos, cell, entries, bucket_count, ctl, key_id, key_round);
                hipLaunchKernelGGL(k_bn_trial_end, dim3(1), dim3(1), 0, s, ctl);
            }
            BN_TRY(hipGetLastError());
            BN_TRY(hipMemcpyAsync(&host_ctl, ctl, sizeof host_ctl, hipMemcpyDeviceToHost, s));
            BN_TRY(hipStreamSynchronize(s));
        }
    }
    if (status == RT_OK && host_ctl.count > cap) { set_error("bluenoise: %u points accepted for %zu table entries", host_ctl.count, cap); status = RT_ERR_DEVICE; }
    if (status == RT_OK && host_ctl.count > 0) {
        // dispatch order = (trial, phase position, dispatch id): two stable passes, minor key first
        const u32 blocks = (u32) ((cap + 255) / 256);
        hipLaunchKernelGGL(k_bn_order_keys, dim3(blocks), dim3(256), 0, s, key_id, ovals, okeys, (u32) cap, per_phase);
        status = radix_sort_pairs(s, okeys, ovals, cell_tmp, idx_tmp, scratch, (u32) cap, bits_for((unsigned long long) per_phase + 1));
        if (status == RT_OK) {
            const u32 rounds = host_ctl.trials * p.phases;
            hipLaunchKernelGGL(k_bn_order_keys, dim3(blocks), dim3(256), 0, s, key_round, ovals, okeys, (u32) cap, rounds);
            status = radix_sort_pairs(s, okeys, ovals, cell_tmp, idx_tmp, scratch, (u32) cap, bits_for((unsigned long long) rounds + 1));
        }
        if (status == RT_OK) {
            hipLaunchKernelGGL(k_bn_emit, dim3((host_ctl.count + 255) / 256), dim3(256), 0, s, p, ovals, host_ctl.count, entries, spos, verts, indices, points, normals);
            BN_TRY(hipGetLastError());
            BN_TRY(hipStreamSynchronize(s));
        }
    }
#undef BN_TRY
    if (arena) (void) hipFree(arena);
    if (status != RT_OK) { if (points) (void) hipFree(points); if (normals) (void) hipFree(normals); return status; }
    *points_out = points; *normals_out = normals; *count_out = host_ctl.count;
    return RT_OK;
}

} // namespace rti
