// One site in four wavefronts, two wavefronts per 32-row block ("pair" tiles): the pieces shared by the training kernels
// (dptm_train_pair.h) and the small-call scoring kernel (dptm_kernels.hip).  Wavefronts w and w + 2 hold the same rows and
// each computes half of every row product; the halves meet through an LDS exchange.
#pragma once
#include <hip/hip_runtime.h>

#include "dptm_layout.h"
#include "dptm_mfma.h"

namespace dptm {

constexpr int kPairImg = 64 * kLdsStride;          // floats of one image: rows [64][68] or transposed [64 features][68]
constexpr int kPairXf4 = 8 * 64;                   // float4 of one wavefront in one exchange buffer (32 floats per lane)
constexpr int kPairXFloats = 2 * 4 * kPairXf4 * 4; // two buffers x four wavefronts

// the exchange buffers, taken in turn: a buffer is written again only after the barrier of the exchange in between
struct PairX {
    f32x4* base;
    int turn;
    __device__ __forceinline__ f32x4* next() {
        f32x4* b = base + turn * (4 * kPairXf4);
        turn ^= 1;
        return b;
    }
};

// both wavefronts of a pair end up with the whole row vector; `mine` = feature groups 4 part .. 4 part + 3
__device__ __forceinline__ void pair_gather(PairX& px, int wave, int lane, int part, const f32x4 (&mine)[4], f32x4 (&x)[8]) {
    f32x4* xb = px.next();
    f32x4* d = xb + wave * kPairXf4 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) d[g * 64] = mine[g];
    __syncthreads();
    const f32x4* s = xb + (wave ^ 2) * kPairXf4 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 o = s[g * 64];
        x[g] = part ? o : mine[g];
        x[4 + g] = part ? mine[g] : o;
    }
}

// x += the partner's x (the same sum on both sides)
__device__ __forceinline__ void pair_sum(PairX& px, int wave, int lane, f32x4 (&x)[8]) {
    f32x4* xb = px.next();
    f32x4* d = xb + wave * kPairXf4 + lane;
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u * 64] = x[u];
    __syncthreads();
    const f32x4* s = xb + (wave ^ 2) * kPairXf4 + lane;
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] += s[u * 64];
}

// the wavefront's own four groups of a whole row vector
__device__ __forceinline__ void own_groups(const f32x4 (&x)[8], int part, f32x4 (&x4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) x4[g] = part ? x[4 + g] : x[g];
}

// four groups <-> the lane's ONE head: lane (h, j) of part p holds head 2p + h of row j, 16 features in natural order
__device__ __forceinline__ void x4_to_head(const f32x4 (&x4)[4], float (&hd)[16], float scale) {
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = x4[r >> 2][r & 3] * scale;
        b[r] = x4[2 + (r >> 2)][r & 3] * scale;
        half_swap(a[r], b[r]);
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int src = 4 * (d >> 3) + (d & 3);
        hd[d] = (d & 4) ? b[src] : a[src];
    }
}
__device__ __forceinline__ void head_to_x4(const float (&hd)[16], f32x4 (&x4)[4], float scale) {
    float a[8], b[8];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int dst = 4 * (d >> 3) + (d & 3);
        if (d & 4)
            b[dst] = hd[d] * scale;
        else
            a[dst] = hd[d] * scale;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        half_swap(a[r], b[r]);
        x4[r >> 2][r & 3] = a[r];
        x4[2 + (r >> 2)][r & 3] = b[r];
    }
}

// own groups of a row vector -> the lane's row of a row-major image (row_base: the row's first float)
__device__ __forceinline__ void chain_store4(float* row_base, int h, int part, const f32x4 (&x4)[4]) {
    row_base += 32 * part + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(row_base + 8 * g) = x4[g];
}

}  // namespace dptm
