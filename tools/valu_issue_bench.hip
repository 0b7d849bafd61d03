// valu_issue_bench.hip -- how many cycles a SIMD of gfx950 needs to issue one wave64 vector instruction, as a function
// of the number of waves that share the SIMD.  Settles the "2 or 4 cycles" question behind bench.py's roofline
// (`bound: "valu_issue"`): peak = 1,024 SIMDs x clock / cycles per instruction.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue_bench tools/valu_issue_bench.hip
//   tools/valu_issue_bench > profiles/r02_valu_issue.txt
//
// Every wave runs ITER trips of 32 independent instructions (8 accumulators, so a dependent instruction is 8 issue slots
// behind its producer) and stamps s_memtime around the loop; w waves per SIMD = one workgroup of 256 w threads per CU
// (w <= 4) or two of 128 w (w = 6, 8), one or two workgroups per CU on all 256 CUs.  Reported per instruction kind and w:
//   wave   = a wave's own cycles per instruction (median over waves)
//   simd   = wave / w = issue cycles the SIMD spends per instruction when all its waves are at it
//   wall   = the same figure from the launch's wall time (HIP events) at the clock measured in-kernel
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr int ITER = 4096, PER_TRIP = 32;

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define OP_FMA(r)     asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_MUL(r)     asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_MAX3(r)    asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(r) : "v"(x), "s"(mask));
#define OP_FMAMIX(r)  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(y));
// vcc deliberately not declared as clobbered: the loop has no other user of vcc, and a declared clobber makes the compiler
// pad every statement with s_nop (the first version of this file measured those)
#define OP_CMP(r)     asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(x), "v"(r));
#define OP_RCP(r)     asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define OP_MIN(r)     asm volatile("v_min_f32 %0, %1, %0" : "+v"(r) : "v"(x));
// the node step of k_traverse in miniature: 2 selects, 1 fma_mix, 1 max, per accumulator
#define OP_MIXED(r)   asm volatile("v_cndmask_b32 %0, %1, %0, %4\n v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\n v_max_f32 %0, %1, %0\n v_fma_f32 %0, %1, %3, %0" : "+v"(r) : "v"(x), "v"(h), "v"(y), "s"(mask));

template <int KIND>
__global__ void k_issue(float* sink, unsigned long long* stamps, float x, float y) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned h = 0x3c003c00u; // two halves 1.0
    unsigned long long mask = __ballot(a0 < x * 40.0f), cm = 0; // a lane mask in an SGPR pair (vcc would make the compiler pad every asm statement with s_nop)
    __syncthreads(); // all waves of the workgroup start together
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) { REP32(OP_FMA) }
        else if (KIND == 1) { REP32(OP_MUL) }
        else if (KIND == 2) { REP32(OP_MAX3) }
        else if (KIND == 3) { REP32(OP_CNDMASK) }
        else if (KIND == 4) { REP32(OP_FMAMIX) }
        else if (KIND == 5) { REP32(OP_CMP) }
        else if (KIND == 6) { REP32(OP_RCP) }
        else if (KIND == 7) { REP32(OP_MIN) }
        else { REP8(OP_MIXED) }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t) blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f || cm == 0x123456789ull) sink[0] = a0; // keeps the accumulators alive
}

typedef void (*kern_t)(float*, unsigned long long*, float, float);

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    printf("# %d trips x %d instructions per wave; cycles per wave64 instruction\n", ITER, PER_TRIP);
    float* sink; unsigned long long* stamps;
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&stamps, 16 * (size_t) cus * 64));
    const char* names[] = { "v_fma_f32", "v_mul_f32", "v_max3_f32", "v_cndmask_b32", "v_fma_mix_f32", "v_cmp_le_f32", "v_rcp_f32", "v_min_f32", "mix(cndmask,fma_mix,max,fma)" };
    kern_t kerns[] = { k_issue<0>, k_issue<1>, k_issue<2>, k_issue<3>, k_issue<4>, k_issue<5>, k_issue<6>, k_issue<7>, k_issue<8> };
    const int ws[] = { 1, 2, 3, 4, 6, 8 };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-30s %4s %10s %10s %10s %10s\n", "instruction", "w", "wave", "simd", "wall", "clock_GHz");
    for (int k = 0; k < 9; k++) {
        for (int w : ws) {
            const int per_cu = w <= 4 ? 1 : 2, threads = 256 * w / per_cu, blocks = cus * per_cu;
            const size_t n_waves = (size_t) blocks * threads / 64;
            for (int rep = 0; rep < 3; rep++) { // the last repetition counts (clocks ramped up)
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kerns[k], dim3(blocks), dim3(threads), 0, 0, sink, stamps, 1.0000001f, 1.0e-9f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st(2 * n_waves);
            CHECK(hipMemcpy(st.data(), stamps, 16 * n_waves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(n_waves), clk(n_waves);
            for (size_t i = 0; i < n_waves; i++) { cyc[i] = (double) st[2 * i]; clk[i] = (double) st[2 * i] / ((double) st[2 * i + 1] / 100.0e6) / 1.0e9; }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double per_wave_instr = (double) ITER * PER_TRIP;
            const double wave = cyc[n_waves / 2] / per_wave_instr, clock = clk[n_waves / 2];
            const double wall = (ms * 1e-3) * clock * 1e9 / (per_wave_instr * w); // SIMD cycles per instruction over the whole launch
            printf("%-30s %4d %10.3f %10.3f %10.3f %10.3f\n", names[k], w, wave, wave / w, wall, clock);
        }
    }
    return 0;
}
