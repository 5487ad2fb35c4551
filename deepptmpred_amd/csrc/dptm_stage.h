// Prologue copies of the fused branch kernels with every load of a thread in flight at once.
// Written as `for (i = tid; i < n; i += threads) lds[i] = mem[i]` such a copy compiles to a loop of
// load -> wait -> store, one trip to L2 (or further) per iteration and per dependent level, with all wavefronts
// of the workgroup in the same wait; here a thread first requests everything it will copy and stores afterwards.
#pragma once
#include <hip/hip_runtime.h>

#include "dptm_layout.h"

namespace dptm {

// a thread's share of the LDS copy of a branch's bias / scale / LayerNorm vectors (dptm_layout.h: vecs_source)
template <class V4, int NT>
struct VecsStage {
    static constexpr int R = (kVecsF4 + NT - 1) / NT;
    V4 v[R];
    __device__ __forceinline__ void load(const V4* __restrict__ P, int tid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * NT;
            v[r] = P[vecs_source(i < kVecsF4 ? i : kVecsF4 - 1)];
        }
    }
    __device__ __forceinline__ void store(V4* vecs, int tid) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // threads past the end repeat the last element (same value, same address): a guarded store would pull
            // its load into the guard, behind everything else
            const int i = tid + r * NT;
            vecs[i < kVecsF4 ? i : kVecsF4 - 1] = v[r];
        }
    }
};

// a thread's share of N consecutive float4 of the parameter buffer
template <class V4, int N, int NT>
struct CopyStage {
    static constexpr int R = (N + NT - 1) / NT;
    V4 v[R];
    __device__ __forceinline__ void load(const V4* __restrict__ src, int tid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * NT;
            v[r] = src[i < N ? i : N - 1];
        }
    }
    __device__ __forceinline__ void store(V4* dst, int tid) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * NT;
            dst[i < N ? i : N - 1] = v[r];      // as above
        }
    }
};

// Embedding + position embedding (model.py:157-158) of a tile of TR rows into an LDS image [TR][kLdsStride] floats:
// thread -> (row, float4 column) as idx = tid + r*NT, row = idx >> 4.  Rows from rows_valid on are zero.
// `tok` points at the tile's first token; tokens of one site are W consecutive bytes, sites follow each other.
template <class V4, int W, int TR, int NT>
struct EmbedStage {
    static_assert((TR * 16) % NT == 0, "whole iterations");
    static constexpr int R = TR * 16 / NT;
    int tk[R];
    V4 e[R], q[R];
    __device__ __forceinline__ void load_tokens(const uint8_t* __restrict__ tok, int rows_valid, int tid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = (tid + r * NT) >> 4;
            tk[r] = tok[row < rows_valid ? row : rows_valid - 1];     // rows_valid >= 1
        }
    }
    __device__ __forceinline__ void load_rows(const V4* __restrict__ P, int rows_valid, int tid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = tid + r * NT, row0 = idx >> 4, c4 = idx & 15;
            const int row = row0 < rows_valid ? row0 : rows_valid - 1;
            const int pos = row - (row / W) * W;
            const int t = tk[r] > kCodes - 1 ? kCodes - 1 : tk[r];
            e[r] = P[kOffEmb + t * 16 + c4];
            q[r] = P[kOffPos + pos * 16 + c4];
        }
    }
    __device__ __forceinline__ void store(float* img, int rows_valid, int tid) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int idx = tid + r * NT, row = idx >> 4, c4 = idx & 15;
            V4 v = e[r] + q[r];
            if (row >= rows_valid) v = V4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<V4*>(img + row * kLdsStride + c4 * 4) = v;
        }
    }
};

}  // namespace dptm
