// rt_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass, for gfx950 (wave64).
// Used by the LBVH build (Morton codes) and the blue-noise generator (grid cell ids), where the reference
// uses the DXR driver and a bitonic network of single-thread dispatches (bluenoise.cpp:273-284) respectively.
#include "rt_internal.h"

namespace rti {

static constexpr int SORT_THREADS = 256;
static constexpr int SORT_ROUNDS  = 8;
static constexpr int SORT_TILE    = SORT_THREADS * SORT_ROUNDS;

__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const u32* __restrict__ keys, u32 n, int shift,
                                                            u32* __restrict__ hist, u32 nblocks) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    u32 base = blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ROUNDS; r++) {
        u32 i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `total` counters in place, one workgroup
__global__ __launch_bounds__(1024) void k_sort_scan(u32* __restrict__ hist, u32 total) {
    __shared__ u32 part[1024];
    u32 per = (total + 1023u) / 1024u;
    u32 begin = threadIdx.x * per, end = begin + per < total ? begin + per : total;
    u32 sum = 0;
    for (u32 i = begin; i < end; i++) sum += hist[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = part[threadIdx.x] - sum;
    for (u32 i = begin; i < end; i++) { u32 c = hist[i]; hist[i] = run; run += c; }
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                               u32* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                               u32 n, int shift, const u32* __restrict__ hist, u32 nblocks) {
    __shared__ u32 run[256];
    __shared__ u32 wave_cnt[SORT_THREADS / 64][256];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    run[tid] = hist[tid * nblocks + blockIdx.x];
    for (int w = 0; w < SORT_THREADS / 64; w++) wave_cnt[w][tid] = 0;
    __syncthreads();
    u32 base = blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ROUNDS; r++) {
        u32 i = base + r * SORT_THREADS + tid;
        bool valid = i < n;
        u32 key = valid ? keys[i] : 0u;
        u32 val = valid ? vals[i] : 0u;
        u32 d = (key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        u32 rank = __popcll(peers & ((1ull << lane) - 1ull));
        u32 cnt = __popcll(peers);
        if (valid && rank == 0) wave_cnt[wave][d] = cnt;
        __syncthreads();
        if (valid) {
            u32 off = run[d];
            for (u32 w = 0; w < wave; w++) off += wave_cnt[w][d];
            keys_out[off + rank] = key;
            vals_out[off + rank] = val;
        }
        __syncthreads();
        u32 s = 0;
        for (int w = 0; w < SORT_THREADS / 64; w++) { s += wave_cnt[w][tid]; wave_cnt[w][tid] = 0; }
        run[tid] += s;
        __syncthreads();
    }
}

size_t sort_scratch_bytes(u32 n) {
    u32 nblocks = (n + SORT_TILE - 1) / SORT_TILE;
    if (nblocks == 0) nblocks = 1;
    return (size_t) 256 * nblocks * sizeof(u32);
}

int radix_sort_pairs(hipStream_t s, u32* keys, u32* vals, u32* keys_tmp, u32* vals_tmp, void* scratch, u32 n, int key_bits) {
    if (n == 0) return RT_OK;
    u32 nblocks = (n + SORT_TILE - 1) / SORT_TILE;
    u32* hist = (u32*) scratch;
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    u32 *ki = keys, *vi = vals, *ko = keys_tmp, *vo = vals_tmp;
    for (int p = 0; p < passes; p++) {
        int shift = 8 * p;
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(SORT_THREADS), 0, s, ki, n, shift, hist, nblocks);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, hist, 256u * nblocks);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(SORT_THREADS), 0, s, ki, vi, ko, vo, n, shift, hist, nblocks);
        u32* t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    if (ki != keys) {
        RT_HIP(hipMemcpyAsync(keys, ki, (size_t) n * 4, hipMemcpyDeviceToDevice, s));
        RT_HIP(hipMemcpyAsync(vals, vi, (size_t) n * 4, hipMemcpyDeviceToDevice, s));
    }
    RT_HIP(hipGetLastError());
    return RT_OK;
}

} // namespace rti
