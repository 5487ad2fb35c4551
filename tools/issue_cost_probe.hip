// Issue-cost probe for gfx950: how long one wavefront instruction of each kind holds a SIMD's issue port,
// relative to v_fma_f32.  Every kernel runs ITER trips of 32 INDEPENDENT instructions (one register each, so no
// instruction waits for the one before it) on 8 wavefronts per workgroup (2 per SIMD), 4 workgroups per CU in turn.
//     hipcc --offload-arch=gfx950 -O3 -o issue_cost_probe issue_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 2048;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#define PROBE1(NAME, ASM)                                                                 \
    __global__ __launch_bounds__(512) void NAME(float* out) {                             \
        float a[32];                                                                      \
        for (int i = 0; i < 32; ++i) a[i] = 0.001f * (float)((threadIdx.x & 15) + i);     \
        for (int it = 0; it < ITER; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 32; ++i) asm volatile(ASM : "+v"(a[i])); \
        }                                                                                 \
        float s = 0;                                                                      \
        for (int i = 0; i < 32; ++i) s += a[i];                                           \
        out[blockIdx.x * 512 + threadIdx.x] = s;                                          \
    }

#define PROBE2(NAME, ASM)                                                                 \
    __global__ __launch_bounds__(512) void NAME(float* out) {                             \
        f32x2 a[32];                                                                      \
        for (int i = 0; i < 32; ++i) a[i] = f32x2{0.001f * (float)((threadIdx.x & 15) + i), 0.5f}; \
        for (int it = 0; it < ITER; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 32; ++i) asm volatile(ASM : "+v"(a[i])); \
        }                                                                                 \
        float s = 0;                                                                      \
        for (int i = 0; i < 32; ++i) s += a[i][0] + a[i][1];                              \
        out[blockIdx.x * 512 + threadIdx.x] = s;                                          \
    }

PROBE1(k_fma, "v_fma_f32 %0, %0, %0, %0")
PROBE1(k_mul, "v_mul_f32 %0, %0, %0")
PROBE1(k_add, "v_add_f32 %0, %0, %0")
PROBE1(k_max3, "v_max3_f32 %0, %0, %0, %0")
PROBE1(k_exp, "v_exp_f32 %0, %0")
PROBE1(k_rcp, "v_rcp_f32 %0, %0")
PROBE1(k_rsq, "v_rsq_f32 %0, %0")
PROBE1(k_log, "v_log_f32 %0, %0")
PROBE1(k_exp16, "v_exp_f16 %0, %0")
PROBE1(k_rcp16, "v_rcp_f16 %0, %0")
PROBE1(k_pkmul16, "v_pk_mul_f16 %0, %0, %0")
PROBE1(k_pkfma16, "v_pk_fma_f16 %0, %0, %0, %0")
PROBE1(k_pkadd16, "v_pk_add_f16 %0, %0, %0")
PROBE1(k_cvtpkbf, "v_cvt_pk_bf16_f32 %0, %0, %0")
PROBE1(k_cvtpkh, "v_cvt_pk_f16_f32 %0, %0, %0")
PROBE1(k_cvth, "v_cvt_f16_f32 %0, %0")
PROBE1(k_cndmask, "v_cndmask_b32 %0, %0, %0, vcc")
PROBE1(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %0, s[2:3]")
PROBE1(k_cndmask0, "v_cndmask_b32_e64 %0, 0, %0, s[2:3]")
PROBE1(k_swap, "v_permlane32_swap_b32 %0, %0")
PROBE1(k_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
PROBE1(k_cmp, "v_cmp_gt_f32 vcc, %0, %0")
PROBE1(k_and, "v_and_b32 %0, %0, %0")
PROBE1(k_lshl, "v_lshlrev_b32 %0, 1, %0")
PROBE1(k_perm, "v_perm_b32 %0, %0, %0, %0")
PROBE1(k_fmac, "v_fmac_f32 %0, %0, %0")
PROBE1(k_sub, "v_sub_f32 %0, %0, %0")
PROBE1(k_max, "v_max_f32 %0, %0, %0")
PROBE1(k_mul64, "v_mul_f32_e64 %0, %0, %0")
PROBE1(k_cvtbfsingle, "v_cvt_pk_bf16_f32 %0, %0, 0")
PROBE1(k_mov, "v_mov_b32 %0, %0")
PROBE1(k_fmamix, "v_fma_mix_f32 %0, %0, %0, %0 op_sel_hi:[1,0,0]")
PROBE2(k_pkmul, "v_pk_mul_f32 %0, %0, %0")
PROBE2(k_pkadd, "v_pk_add_f32 %0, %0, %0")
PROBE2(k_pkfma, "v_pk_fma_f32 %0, %0, %0, %0")

// 8 independent accumulators of v_mfma_f32_32x32x16_bf16 (128 registers)
__global__ __launch_bounds__(512) void k_mfma(float* out) {
    f32x16 acc[8];
    b8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.01f * (threadIdx.x & 7));
        b[i] = (__bf16)0.5f;
    }
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 8 independent accumulators of v_mfma_f32_32x32x2_f32
__global__ __launch_bounds__(512) void k_mfma32(float* out) {
    f32x16 acc[8];
    const float a = 0.01f * (threadIdx.x & 7), b = 0.5f;
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
static float time_ms(K kernel, int blocks, float* out) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kernel<<<blocks, 512>>>(out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kernel<<<blocks, 512>>>(out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4;
    float* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    // instructions per SIMD: 4 workgroups in turn x 2 wavefronts x ITER x 32
    const double per_simd = 4.0 * 2.0 * ITER * 32.0;
    const float base = time_ms(k_fma, blocks, out);
    printf("%s, %d CUs; v_fma_f32: %.3f ms = %.2f clk per wave-instruction at 2.4 GHz\n", p.name, p.multiProcessorCount, base,
           base * 1e-3 * 2.4e9 / per_simd);
#define RUN(K, NAME) printf("%-22s %8.3f ms  %5.2f x v_fma_f32\n", NAME, time_ms(K, blocks, out), time_ms(K, blocks, out) / base);
    RUN(k_mul, "v_mul_f32")
    RUN(k_add, "v_add_f32")
    RUN(k_max3, "v_max3_f32")
    RUN(k_mov, "v_mov_b32")
    RUN(k_cndmask, "v_cndmask_b32")
    RUN(k_fmamix, "v_fma_mix_f32")
    RUN(k_cndmask64, "v_cndmask_b32_e64 s")
    RUN(k_cndmask0, "v_cndmask_b32_e64 0")
    RUN(k_swap, "v_permlane32_swap")
    RUN(k_dpp, "v_mov_b32_dpp")
    RUN(k_cmp, "v_cmp_gt_f32")
    RUN(k_and, "v_and_b32")
    RUN(k_lshl, "v_lshlrev_b32")
    RUN(k_perm, "v_perm_b32")
    RUN(k_fmac, "v_fmac_f32")
    RUN(k_sub, "v_sub_f32")
    RUN(k_max, "v_max_f32")
    RUN(k_mul64, "v_mul_f32_e64")
    RUN(k_pkmul, "v_pk_mul_f32")
    RUN(k_pkadd, "v_pk_add_f32")
    RUN(k_pkfma, "v_pk_fma_f32")
    RUN(k_exp, "v_exp_f32")
    RUN(k_rcp, "v_rcp_f32")
    RUN(k_rsq, "v_rsq_f32")
    RUN(k_log, "v_log_f32")
    RUN(k_exp16, "v_exp_f16")
    RUN(k_rcp16, "v_rcp_f16")
    RUN(k_pkmul16, "v_pk_mul_f16")
    RUN(k_pkadd16, "v_pk_add_f16")
    RUN(k_pkfma16, "v_pk_fma_f16")
    RUN(k_cvtpkbf, "v_cvt_pk_bf16_f32")
    RUN(k_cvtpkh, "v_cvt_pk_f16_f32")
    RUN(k_cvth, "v_cvt_f16_f32")
    {
        const float ms = time_ms(k_mfma, blocks, out);
        // 4 x 8 MFMAs per trip against 32 vector instructions per trip: same instruction count
        printf("%-22s %8.3f ms  %5.2f x v_fma_f32\n", "v_mfma_32x32x16_bf16", ms, ms / base);
        const float ms32 = time_ms(k_mfma32, blocks, out);
        printf("%-22s %8.3f ms  %5.2f x v_fma_f32\n", "v_mfma_32x32x2_f32", ms32, ms32 / base);
    }
    hipFree(out);
    return 0;
}
