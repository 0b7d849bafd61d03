// valu_exec_bench.hip -- does the issue cost of a wave64 vector instruction on gfx950 depend on WHICH lanes are active?
// k_traverse issues most of its instructions with a third of the lanes active (C5: 0.29).  If the SIMD skipped a pass
// over lanes whose exec bits are all zero, packing the live rays of a wave into one half (or quarter) would make every
// later instruction cheaper; if it does not, only fewer instructions help.  Same loop as valu_issue_bench.hip, run under
// an exec mask chosen by the host: the low N lanes, every other lane, the high half.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_exec_bench tools/valu_exec_bench.hip
//   tools/valu_exec_bench > profiles/r03_valu_exec_mask.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr int ITER = 4096, PER_TRIP = 32;

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define OP_FMA(r)     asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(r) : "v"(x), "s"(mask));
#define OP_FMAMIX(r)  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(y));
#define OP_MIXED(r)   asm volatile("v_cndmask_b32 %0, %1, %0, %4\n v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\n v_max_f32 %0, %1, %0\n v_fma_f32 %0, %1, %3, %0" : "+v"(r) : "v"(x), "v"(h), "v"(y), "s"(mask));

template <int KIND>
__global__ void k_exec(float* sink, unsigned long long* stamps, float x, float y, unsigned long long lanes) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned h = 0x3c003c00u;
    unsigned long long mask = __ballot(a0 < x * 40.0f);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((lanes >> (threadIdx.x & 63)) & 1) { // the loop runs with exec = lanes
        for (int i = 0; i < ITER; i++) {
            if (KIND == 0) { REP32(OP_FMA) }
            else if (KIND == 1) { REP32(OP_CNDMASK) }
            else if (KIND == 2) { REP32(OP_FMAMIX) }
            else { REP8(OP_MIXED) }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(size_t) blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) sink[0] = a0;
}

typedef void (*kern_t)(float*, unsigned long long*, float, float, unsigned long long);

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs; %d trips x %d instructions per wave; SIMD cycles per wave64 instruction under an exec mask\n", prop.name, cus, ITER, PER_TRIP);
    float* sink; unsigned long long* stamps;
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&stamps, 8 * (size_t) cus * 64));
    const char* names[] = { "v_fma_f32", "v_cndmask_b32", "v_fma_mix_f32", "mix(cndmask,fma_mix,max,fma)" };
    kern_t kerns[] = { k_exec<0>, k_exec<1>, k_exec<2>, k_exec<3> };
    struct { const char* name; unsigned long long lanes; } masks[] = {
        { "all 64", ~0ull }, { "low 32", 0xffffffffull }, { "high 32", 0xffffffff00000000ull }, { "low 16", 0xffffull },
        { "lanes 16-31", 0xffff0000ull }, { "low 1", 1ull }, { "even lanes", 0x5555555555555555ull }, { "low 16 + 32-47", 0x0000ffff0000ffffull },
    };
    const int ws[] = { 1, 4, 8 };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-30s %-16s %4s %10s %10s\n", "instruction", "exec", "w", "wave", "simd");
    for (int k = 0; k < 4; k++) for (auto& m : masks) for (int w : ws) {
        const int per_cu = w <= 4 ? 1 : 2, threads = 256 * w / per_cu, blocks = cus * per_cu;
        const size_t n_waves = (size_t) blocks * threads / 64;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(threads), 0, 0, sink, stamps, 1.0000001f, 1.0e-9f, m.lanes);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
        }
        std::vector<unsigned long long> st(n_waves);
        CHECK(hipMemcpy(st.data(), stamps, 8 * n_waves, hipMemcpyDeviceToHost));
        std::sort(st.begin(), st.end());
        const double wave = (double) st[n_waves / 2] / ((double) ITER * PER_TRIP);
        printf("%-30s %-16s %4d %10.3f %10.3f\n", names[k], m.name, w, wave, wave / w);
    }
    return 0;
}
