// Does the vector ALU run while the matrix core is busy?  Timings on one workgroup per CU:
//   M: MFMA only   V: VALU only   I: both interleaved in every wave   S: waves split (waves 0-3 MFMA, 4-7 VALU)
//   M4 / V4: as M / V but only waves 0-3 work (one wave per SIMD)
// for the fp32 MFMA (32x32x2) and the f16 MFMA (32x32x16).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE, bool HALF>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {0}, a1 = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    h8 hx, hy;
    for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)(x + i); hy[i] = (_Float16)(y - i); }
    const bool do_m = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4) || (MODE == 5 && wave < 4);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (HALF) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, a1, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                }
                if (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < (HALF ? 1 : 2); ++r)
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                }
            }
        }
        if (do_v && MODE != 2) {
#pragma unroll
            for (int r = 0; r < (HALF ? 4 : 8); ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, bool HALF>
float run(float* d, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, HALF>), dim3(256), dim3(512), 0, 0, d, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, HALF>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <bool HALF>
void all(float* d, const char* name) {
    const int iters = 20000;
    printf("%s  M %.3f  V %.3f  I %.3f  S %.3f  M4 %.3f  V4 %.3f ms\n", name, run<0, HALF>(d, iters), run<1, HALF>(d, iters),
           run<2, HALF>(d, iters), run<3, HALF>(d, iters), run<4, HALF>(d, iters), run<5, HALF>(d, iters));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    all<false>(d, "fp32 32x32x2 :");
    all<true>(d, "f16 32x32x16 :");
    return 0;
}
