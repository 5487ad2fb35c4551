// Device helpers shared by the fused kernels (scoring: dptm_kernels.hip; training: dptm_train_fused.hip):
// the fp32 MFMA step, register-to-register dense layers in the chain layout of dptm_layout.h, swish and
// LayerNormalization over 64 features.
#pragma once
#include <hip/hip_runtime.h>

#include "dptm_layout.h"

namespace dptm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma(f32x16& acc, float a, float b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

// Keeps a lane-dependent index local to the code that follows: without it the compiler hoists `base + index` of every
// table it meets out of the enclosing loops as a 64-bit address per table and lane, two registers each that then live
// (or spill) across the whole kernel; with it the address is formed where it is used, scalar base + 32-bit lane offset.
#ifdef DPTM_KEEP_INDEX_LOCAL
#define DPTM_LOCAL_INDEX(v) asm volatile("" : "+v"(v))
#else
#define DPTM_LOCAL_INDEX(v) ((void)0)
#endif

// the same for an index used in an expression: returns v, opaque to the optimiser at this point
__device__ __forceinline__ unsigned local_index(unsigned v) {
    DPTM_LOCAL_INDEX(v);
    return v;
}

constexpr float kQScale = 0.25f * 1.44269504088896340736f;  // log2(e) / sqrt(head_dim)

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// keras 'swish': x * sigmoid(x)
__device__ __forceinline__ float swishf(float x) { return x * fast_rcp(1.0f + __expf(-x)); }

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// Exchange between the two half-waves: returns {a.lo|b.lo , a.hi|b.hi} (see
// v_permlane32_swap: lanes 32-63 of the first operand swap with lanes 0-31 of the second).
__device__ __forceinline__ void half_swap(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// acc tile t (16 regs) <-> feature groups u = 4t..4t+3 of the chain layout
template <int NT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[NT], const f32x4* __restrict__ vec,
                                              int h, int groups_per_half) {
    DPTM_LOCAL_INDEX(h);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 b = vec[h * groups_per_half + 4 * t + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
        }
}

// acc[t] += W^T-tile(t) * X^T with X held in registers (chain layout, K8 groups of 8 feats).
// Weight fragments are fetched DIST groups ahead of the MFMAs that consume them; the
// sched_barrier at the end of every group pins that order so the L2 latency of a fetch is
// covered by the 4*NT*DIST MFMAs issued after it.
template <int K8, int NT, int DIST = 2>
__device__ __forceinline__ void gemm_regs(f32x16 (&acc)[NT], const f32x4* __restrict__ wp,
                                          const f32x4 (&x)[K8], int lane) {
    DPTM_LOCAL_INDEX(lane);
    f32x4 a[K8][NT];
#pragma unroll
    for (int u = 0; u < DIST && u < K8; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) a[u][t] = wp[(t * K8 + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < K8; ++u) {
        if (u + DIST < K8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a[u + DIST][t] = wp[(t * K8 + u + DIST) * 64 + lane];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) mfma(acc[t], a[u][t][e], x[u][e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The first DIST groups of a product's weight fragments, requested while the PREVIOUS product still multiplies: a
// product that fetches its own first groups starts with one exposed trip to L2.
template <int NT, int DIST = 2>
struct WHead {
    f32x4 a[DIST][NT];
    // fragments (t*stride + u0 + u) of a packed matrix whose out tiles are `stride` groups apart
    __device__ __forceinline__ void load(const f32x4* __restrict__ wp, int stride, int u0, int lane) {
        DPTM_LOCAL_INDEX(lane);
#pragma unroll
        for (int u = 0; u < DIST; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) a[u][t] = wp[(t * stride + u0 + u) * 64 + lane];
    }
};

// gemm_regs with its first DIST groups already requested (hd.load(wp, K8, 0, lane) earlier)
template <int K8, int NT, int DIST = 2>
__device__ __forceinline__ void gemm_regs_head(f32x16 (&acc)[NT], const f32x4* __restrict__ wp, const f32x4 (&x)[K8], int lane,
                                               const WHead<NT, DIST>& hd) {
    DPTM_LOCAL_INDEX(lane);
    f32x4 a[K8][NT];
#pragma unroll
    for (int u = 0; u < DIST && u < K8; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) a[u][t] = hd.a[u][t];
#pragma unroll
    for (int u = 0; u < K8; ++u) {
        if (u + DIST < K8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a[u + DIST][t] = wp[(t * K8 + u + DIST) * 64 + lane];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) mfma(acc[t], a[u][t][e], x[u][e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
__device__ __forceinline__ void acc_to_x(const f32x16 (&acc)[NT], f32x4 (&x)[4 * NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * t + g][e] = acc[t][4 * g + e];
}

// keras LayerNormalization(epsilon=1e-6) over the 64 features of each row
__device__ __forceinline__ void layernorm64(f32x4 (&x)[8], const f32x4* __restrict__ g,
                                            const f32x4* __restrict__ b, int h) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (x[u][0] + x[u][1]) + (x[u][2] + x[u][3]);
    const float mean = xhalf_sum(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = x[u][e] - mean;
            x[u][e] = d;
            q = fmaf(d, d, q);
        }
    const float var = xhalf_sum(q) * (1.0f / 64.0f);
    const float inv = rsqrtf(var + 1e-6f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4 gg = g[h * 8 + u], bb = b[h * 8 + u];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[u][e] = fmaf(x[u][e] * inv, gg[e], bb[e]);
    }
}

}  // namespace dptm
