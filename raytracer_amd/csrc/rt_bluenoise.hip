// rt_bluenoise.hip -- blue-noise (Poisson-disk) surface sample points for translucent meshes.
// Restates the four compute kernels of src/bluenoise.hlsl:60-245 and their host schedule src/bluenoise.cpp:263-351
// for gfx950.  Differences in mechanism, not in result definition:
//   * the bitonic network of single-thread dispatches (bluenoise.cpp:273-284) is a stable radix sort; the order
//     among points of one cell, unspecified in the reference, is (cell id, initial index);
//   * the hash table keeps, per bucket, the 5 smallest cell ids (the reference keeps whichever 5 win the atomics);
//   * the dart-throwing loop (24 phase-group dispatches per trial with a CPU read-back, bluenoise.cpp:327-351) runs
//     inside one persistent workgroup, and accepted points are appended in dispatch-id order instead of atomic order.
#include "rt_internal.h"

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

// generate_sample_points (bluenoise.hlsl:180-245) for every phase group and trial, one workgroup
__global__ __launch_bounds__(1024) void k_bn_throw(BluenoiseParams p, const float4* __restrict__ spos, const u32* __restrict__ scell,
                                                   BnEntry* entries, const u32* __restrict__ bucket_count,
                                                   const u32* __restrict__ table_counts, const RtVertex* __restrict__ verts,
                                                   const uint16_t* __restrict__ indices, RtSamplePoint* __restrict__ out_points,
                                                   float* __restrict__ out_normals, u32* __restrict__ out_count) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_count;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_count = 0;
    __syncthreads();
    const u32 capacity = table_counts[0];
    const u32 gx = (p.grid_dims[0] + 2) / 3, gy = (p.grid_dims[1] + 2) / 3, gz = (p.grid_dims[2] + 2) / 3;
    const u32 per_phase = gx * gy * gz;
    const float rr2 = p.rejection_radius * p.rejection_radius;
    for (u32 trial = 0;; trial++) {
        const u32 prev = s_count;
        for (u32 ph = 0; ph < p.phases; ph++) {
            const u32 pg = p.phase_sequence[ph];
            const u32 pgx = pg % 3, pgy = (pg / 3) % 3, pgz = pg / 9;
            for (u32 base = 0; base < per_phase; base += 1024) {
                const u32 id = base + tid;
                bool accept = false;
                int entry_index = -1;
                float4 tp = make_float4(0, 0, 0, 0);
                if (id < per_phase && s_count < capacity) {
                    u32 cx = 3 * (id % gx) + pgx, cy = 3 * ((id / gx) % gy) + pgy, cz = 3 * (id / (gx * gy)) + pgz;
                    if (cx < p.grid_dims[0] && cy < p.grid_dims[1] && cz < p.grid_dims[2]) {
                        u32 cell_id = cx + cy * p.grid_dims[0] + cz * p.grid_dims[0] * p.grid_dims[1];
                        entry_index = bn_search(entries, bucket_count, p.buckets, cell_id);
                        if (entry_index >= 0 && isinf(entries[entry_index].sel[0])) {
                            unsigned long long ti = (unsigned long long) entries[entry_index].first_index + trial;
                            if (ti < p.initial_count && scell[ti] == cell_id) {
                                tp = spos[ti];
                                accept = true;
                                for (int xo = -2; xo <= 2 && accept; xo++) for (int yo = -2; yo <= 2 && accept; yo++) for (int zo = -2; zo <= 2 && accept; zo++) {
                                    if ((xo == 0 && yo == 0 && zo == 0) || (abs(xo) == 2 && abs(yo) == 2 && abs(zo) == 2)) continue;
                                    int nx = (int) cx + xo, ny = (int) cy + yo, nz = (int) cz + zo;
                                    if (nx < 0 || ny < 0 || nz < 0 || nx >= (int) p.grid_dims[0] || ny >= (int) p.grid_dims[1] || nz >= (int) p.grid_dims[2]) continue;
                                    int ne = bn_search(entries, bucket_count, p.buckets,
                                                       (u32) nx + (u32) ny * p.grid_dims[0] + (u32) nz * p.grid_dims[0] * p.grid_dims[1]);
                                    if (ne >= 0) {
                                        f3 d = mk3(entries[ne].sel[0], entries[ne].sel[1], entries[ne].sel[2]) - mk3(tp.x, tp.y, tp.z);
                                        if (dot3(d, d) <= rr2) accept = false;
                                    }
                                }
                            }
                        }
                    }
                }
                // append in dispatch-id order
                unsigned long long m = __ballot(accept);
                if (lane == 0) s_wave[wave] = (u32) __popcll(m);
                __syncthreads();
                u32 offset = (u32) __popcll(m & ((1ull << lane) - 1ull)), total = 0;
                for (u32 w = 0; w < 16; w++) { u32 c = s_wave[w]; if (w < wave) offset += c; total += c; }
                if (accept) {
                    u32 slot = s_count + offset;
                    f3 q = mk3(tp.x, tp.y, tp.z);
                    entries[entry_index].sel[0] = q.x; entries[entry_index].sel[1] = q.y; entries[entry_index].sel[2] = q.z;
                    RtSamplePoint sp;
                    f3 wp = xform_point44(p.transform, q);
                    sp.position[0] = wp.x; sp.position[1] = wp.y; sp.position[2] = wp.z;
                    sp.payload[0] = 0.0f; sp.payload[1] = 0.0f; sp.payload[2] = 0.0f;
                    out_points[slot] = sp;
                    const uint16_t* tri = indices + p.indices_offset + 3u * as_u32(tp.w);
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
                __threadfence_block();
                __syncthreads();
                if (tid == 0) s_count += total;
                __syncthreads();
            }
        }
        if (s_count == prev) break;
    }
    if (tid == 0) *out_count = s_count;
}

int bluenoise_generate(hipStream_t s, const BluenoiseParams& p, const RtVertex* verts, const uint16_t* indices,
                       const float* partial_areas, RtSamplePoint** points_out, float** normals_out, u32* count_out) {
    const u32 N0 = p.initial_count;
    float4 *pos = nullptr, *spos = nullptr;
    u32 *cell = nullptr, *idx = nullptr, *cell_tmp = nullptr, *idx_tmp = nullptr, *bucket_count = nullptr, *counts = nullptr;
    BnEntry* entries = nullptr; void* scratch = nullptr;
    RtSamplePoint* points = nullptr; float* normals = nullptr;
    size_t cap = (size_t) p.buckets * BN_BUCKET;
    int status = RT_OK;
    auto fail = [&](hipError_t e, const char* what) { set_error("bluenoise: %s failed: %s", what, hipGetErrorString(e)); status = RT_ERR_DEVICE; };
    hipError_t e;
#define BN_TRY(call) if (status == RT_OK && (e = (call)) != hipSuccess) fail(e, #call)
    BN_TRY(hipMalloc(&pos, sizeof(float4) * (size_t) N0));
    BN_TRY(hipMalloc(&spos, sizeof(float4) * (size_t) N0));
    BN_TRY(hipMalloc(&cell, 4 * (size_t) N0));
    BN_TRY(hipMalloc(&idx, 4 * (size_t) N0));
    BN_TRY(hipMalloc(&cell_tmp, 4 * (size_t) N0));
    BN_TRY(hipMalloc(&idx_tmp, 4 * (size_t) N0));
    BN_TRY(hipMalloc(&scratch, sort_scratch_bytes(N0)));
    BN_TRY(hipMalloc(&entries, sizeof(BnEntry) * cap));
    BN_TRY(hipMalloc(&bucket_count, 4 * (size_t) p.buckets));
    BN_TRY(hipMalloc(&counts, 16));
    BN_TRY(hipMalloc(&points, sizeof(RtSamplePoint) * cap));
    BN_TRY(hipMalloc(&normals, 12 * cap));
    BN_TRY(hipMemsetAsync(bucket_count, 0, 4 * (size_t) p.buckets, s));
    BN_TRY(hipMemsetAsync(counts, 0, 16, s));
    if (status == RT_OK) {
        hipLaunchKernelGGL(k_bn_initial, dim3((N0 + 255) / 256), dim3(256), 0, s, p, verts, indices, partial_areas, pos, cell, idx);
        unsigned long long cells = (unsigned long long) p.grid_dims[0] * p.grid_dims[1] * p.grid_dims[2];
        int bits = 1; while (bits < 32 && (1ull << bits) < cells) bits++;
        status = radix_sort_pairs(s, cell, idx, cell_tmp, idx_tmp, scratch, N0, bits);
    }
    if (status == RT_OK) {
        hipLaunchKernelGGL(k_bn_gather, dim3((N0 + 255) / 256), dim3(256), 0, s, pos, idx, N0, spos);
        hipLaunchKernelGGL(k_bn_table, dim3((N0 + 255) / 256), dim3(256), 0, s, cell, N0, p.buckets, entries, bucket_count, counts);
        hipLaunchKernelGGL(k_bn_throw, dim3(1), dim3(1024), 0, s, p, spos, cell, entries, bucket_count, counts, verts, indices,
                           points, normals, counts + 2);
        BN_TRY(hipGetLastError());
    }
    u32 host_counts[4] = { 0, 0, 0, 0 };
    BN_TRY(hipMemcpyAsync(host_counts, counts, 16, hipMemcpyDeviceToHost, s));
    BN_TRY(hipStreamSynchronize(s));
#undef BN_TRY
    void* tmp[] = { pos, spos, cell, idx, cell_tmp, idx_tmp, scratch, entries, bucket_count, counts };
    for (void* q : tmp) if (q) (void) hipFree(q);
    if (status != RT_OK) { if (points) (void) hipFree(points); if (normals) (void) hipFree(normals); return status; }
    *points_out = points; *normals_out = normals; *count_out = host_counts[2];
    return RT_OK;
}

} // namespace rti
