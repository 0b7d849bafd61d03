// valu_class_bench.hip -- which wave64 vector instructions of gfx950 issue at the fast rate (~2.25 SIMD cycles, measured for
// v_fma_f32 / v_mul_f32 in valu_issue_bench.hip) and which at the slow one (~4.2: v_cndmask, v_max3, v_fma_mix, compares)?
// k_traverse is bound by instruction issue, so the class of every instruction of its node and triangle steps is what a step
// costs.  Same method as valu_issue_bench.hip (8 independent accumulators, 4096 trips x 32 instructions, s_memtime), one line
// per instruction kind, at 4 and 8 waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_class_bench tools/valu_class_bench.hip
//   tools/valu_class_bench > profiles/r03_valu_classes.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr int ITER = 2048, PER_TRIP = 32;

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

// r: accumulator (float VGPR); x, y: float VGPRs; u: unsigned VGPR; s: float SGPR; mask: lane mask in an SGPR pair
#define KINDS(X, XS) \
    X(0,  "v_fma_f32",                 "v_fma_f32 %0, %1, %2, %0") \
    X(1,  "v_mul_f32",                 "v_mul_f32 %0, %1, %0") \
    X(2,  "v_add_f32",                 "v_add_f32 %0, %1, %0") \
    X(3,  "v_sub_f32",                 "v_sub_f32 %0, %1, %0") \
    X(4,  "v_fmac_f32",                "v_fmac_f32 %0, %1, %2") \
    X(5,  "v_fma_f32 |abs| -neg",      "v_fma_f32 %0, |%1|, -%2, %0") \
    X(6,  "v_fma_f32 sgpr operand",    "v_fma_f32 %0, %4, %2, %0") \
    X(7,  "v_mul_f32 literal",         "v_mul_f32 %0, 0x3f800001, %0") \
    X(8,  "v_max_f32",                 "v_max_f32 %0, %1, %0") \
    X(9,  "v_min_f32",                 "v_min_f32 %0, %1, %0") \
    X(10, "v_max3_f32",                "v_max3_f32 %0, %1, %2, %0") \
    X(11, "v_med3_f32",                "v_med3_f32 %0, %1, %2, %0") \
    X(12, "v_cndmask_b32 (sgpr mask)", "v_cndmask_b32 %0, %1, %0, %5") \
    X(13, "v_fma_mix_f32",             "v_fma_mix_f32 %0, %3, %2, %0 op_sel_hi:[1,0,0]") \
    X(14, "v_cvt_f32_f16",             "v_cvt_f32_f16 %0, %3") \
    X(15, "v_mov_b32",                 "v_mov_b32 %0, %1") \
    X(16, "v_and_b32",                 "v_and_b32 %0, %3, %0") \
    X(17, "v_or_b32",                  "v_or_b32 %0, %3, %0") \
    X(18, "v_xor_b32",                 "v_xor_b32 %0, %3, %0") \
    X(19, "v_lshlrev_b32",             "v_lshlrev_b32 %0, 1, %0") \
    X(20, "v_add_u32",                 "v_add_u32 %0, %3, %0") \
    X(21, "v_lshl_add_u32",            "v_lshl_add_u32 %0, %0, 1, %3") \
    X(22, "v_and_or_b32",              "v_and_or_b32 %0, %0, %3, %3") \
    X(23, "v_bfe_u32",                 "v_bfe_u32 %0, %0, 1, 30") \
    X(24, "v_mad_u32_u24",             "v_mad_u32_u24 %0, %3, %3, %0") \
    X(25, "v_mul_lo_u32",              "v_mul_lo_u32 %0, %3, %0") \
    X(29, "v_rcp_f32",                 "v_rcp_f32 %0, %0") \
    X(30, "v_rsq_f32",                 "v_rsq_f32 %0, %0") \
    X(31, "v_sqrt_f32",                "v_sqrt_f32 %0, %0") \
    X(32, "v_exp_f32",                 "v_exp_f32 %0, %0") \
    X(33, "v_cmp_le_f32 (vcc)",        "v_cmp_le_f32 vcc, %1, %0") \
    XS(34, "v_cmp_le_f32 (sgpr pair)",  "v_cmp_le_f32 s[20:21], %1, %0") \
    X(35, "v_cmp_class_f32",           "v_cmp_class_f32 vcc, %0, %3") \
    X(36, "v_bfi_b32",                 "v_bfi_b32 %0, %3, %1, %0") \
    X(37, "v_perm_b32",                "v_perm_b32 %0, %0, %1, %3") \
    X(38, "v_mov_b32 dpp row_shr:1",   "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    XS(39, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0") \
    X(40, "v_cvt_f32_u32",             "v_cvt_f32_u32 %0, %3") \
    X(41, "v_cvt_pkrtz_f16_f32",       "v_cvt_pkrtz_f16_f32 %0, %1, %0") \
    X(42, "v_sub_f32 sgpr",            "v_sub_f32 %0, %4, %0") \
    X(43, "v_mul_f32 dpp quad_perm",   "v_mul_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(44, "v_fma_f32 inline const",    "v_fma_f32 %0, %1, 2.0, %0") \
    X(45, "v_min3_f32",                "v_min3_f32 %0, %1, %2, %0") \
    X(46, "v_add_f32 sdwa-free e64 clamp", "v_add_f32_e64 %0, %1, %0 clamp") \
    X(47, "v_max_f32 e64 |abs|",       "v_max_f32_e64 %0, |%1|, %0") \
    X(48, "v_mul_f32 e64 mul:2",       "v_mul_f32_e64 %0, %1, %0 mul:2") \
    X(49, "v_ldexp_f32",               "v_ldexp_f32 %0, %0, %3")

template <int KIND>
__global__ void k_issue(float* sink, unsigned long long* stamps, float x, float y, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u = 0x3c003c00u;
    unsigned long long mask = __ballot(a0 < x * 40.0f);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
#define ASM8(text, ...) \
        asm volatile(text : "+v"(a0) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); asm volatile(text : "+v"(a1) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); \
        asm volatile(text : "+v"(a2) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); asm volatile(text : "+v"(a3) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); \
        asm volatile(text : "+v"(a4) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); asm volatile(text : "+v"(a5) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); \
        asm volatile(text : "+v"(a6) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__); asm volatile(text : "+v"(a7) : "v"(x), "v"(y), "v"(u), "s"(s), "s"(mask) __VA_ARGS__);
#define X(k, name, text) if (KIND == k) { _Pragma("unroll") for (int rep = 0; rep < 4; rep++) { ASM8(text) } }
// kinds that write an SGPR: declared, which makes the compiler pad them with s_nop (noted in the output)
#define XS(k, name, text) if (KIND == k) { _Pragma("unroll") for (int rep = 0; rep < 4; rep++) { ASM8(text, : "s20", "s21") } }
        KINDS(X, XS)
#undef XS
#undef X
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(size_t) blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) sink[0] = a0;
}

// packed fp32: two operations per lane and instruction, accumulators are VGPR pairs
template <int KIND>
__global__ void k_issue_pk(float* sink, unsigned long long* stamps, float x, float y, float s) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double xy = __hiloint2double(__float_as_int(y), __float_as_int(x));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
#define PK(r) if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(r) : "v"(xy)); else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r) : "v"(xy)); else asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r) : "v"(xy));
        REP32(PK)
#undef PK
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) stamps[(size_t) blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456) sink[0] = (float) a0;
}

typedef void (*kern_t)(float*, unsigned long long*, float, float, float);

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs; %d trips x %d instructions per wave; SIMD cycles per wave64 instruction (median wave / waves per SIMD)\n", prop.name, cus, ITER, PER_TRIP);
    float* sink; unsigned long long* stamps;
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&stamps, 8 * (size_t) cus * 64));
    struct { const char* name; kern_t k; } kinds[] = {
#define X(k, name, text) { name, k_issue<k> },
        KINDS(X, X)
#undef X
        { "v_pk_fma_f32 (2 per lane)", k_issue_pk<0> }, { "v_pk_mul_f32 (2 per lane)", k_issue_pk<1> }, { "v_pk_add_f32 (2 per lane)", k_issue_pk<2> },
    };
    const int ws[] = { 4, 8 };
    hipEvent_t e1; CHECK(hipEventCreate(&e1));
    printf("%-36s %10s %10s\n", "instruction", "w=4", "w=8");
    for (auto& kind : kinds) {
        double simd[2];
        for (int wi = 0; wi < 2; wi++) {
            const int w = ws[wi], per_cu = w <= 4 ? 1 : 2, threads = 256 * w / per_cu, blocks = cus * per_cu;
            const size_t n_waves = (size_t) blocks * threads / 64;
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(kind.k, dim3(blocks), dim3(threads), 0, 0, sink, stamps, 1.0000001f, 1.0e-9f, 0.5f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            std::vector<unsigned long long> st(n_waves);
            CHECK(hipMemcpy(st.data(), stamps, 8 * n_waves, hipMemcpyDeviceToHost));
            std::sort(st.begin(), st.end());
            simd[wi] = (double) st[n_waves / 2] / ((double) ITER * PER_TRIP) / w;
        }
        printf("%-36s %10.3f %10.3f\n", kind.name, simd[0], simd[1]);
        fflush(stdout);
    }
    return 0;
}
