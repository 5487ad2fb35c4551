// Probe: v_mfma_f32_32x32x16_f16 operand lane map and fp16 subnormal inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {   // A [32][16], B [16][32], D [32][32]
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[r * 16 + 8 * h + j]; b[j] = (_Float16)B[(8 * h + j) * 32 + r]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
int main() {
    float hA[512], hB[512], hD[1024], ref[1024];
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (float)((i * 3 + kk * 7) % 11 - 5);
    for (int kk = 0; kk < 16; ++kk) for (int j = 0; j < 32; ++j) hB[kk * 32 + j] = (float)((kk * 5 + j * 2) % 13 - 6);
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {  // subnormal probe: A = 2^-20 everywhere, B = 2^10
            for (int i = 0; i < 512; ++i) { hA[i] = ldexpf(1.0f, -20); hB[i] = 1024.0f; }
        }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hA[i * 16 + kk] * hB[kk * 32 + j];
            ref[i * 32 + j] = (float)s; maxerr = fmax(maxerr, fabs(s - hD[i * 32 + j]));
        }
        printf("pass %d: max |D - ref| = %g  (D[0][0] = %g, ref = %g)\n", pass, maxerr, hD[0], ref[0]);
    }
    return 0;
}
