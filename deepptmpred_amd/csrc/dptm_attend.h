// Softmax attention of one site's query rows against its keys on the vector unit, shared by the scoring
// kernel (dptm_kernels.hip) and the fused training kernels (dptm_train_branch.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "dptm_layout.h"
#include "dptm_mfma.h"

namespace dptm {

// value held by the neighbouring lane (lane ^ 1): a quad permutation [1,0,3,2] on the DPP path
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// Softmax attention of NR query rows of one site against that site's W keys for one head
// (model.py:62-71; q arrives scaled by log2(e)/sqrt(16), so exponentials are base 2).  kp / vp: first key / value row of
// the site in LDS at this head's 16 features.  Keys are walked in three chunks of W/3 with a
// running maximum (the usual rescaling form of softmax), which keeps only W/3 scores live per
// row; every K / V fragment read serves all NR rows.  m_out / rden_out: the row maximum (base-2 domain) and
// 1 / denominator, which a backward pass needs to rebuild the probabilities.
template <int W, int NR>
__device__ __forceinline__ void attend(const float* kp, const float* vp, const float (&q)[NR][16],
                                       float (&o)[NR][16], float (&m_out)[NR], float (&rden_out)[NR]) {
    constexpr int CH = W / 3;
    static_assert(CH * 3 == W, "window sizes are multiples of 3");
    float m[NR], den[NR];
    f32x4 acc[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        m[r] = -3.0e38f;
        den[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        float sc[NR][CH];
        float cm[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) cm[r] = -3.0e38f;
        // K / V fragments are requested one key ahead of the products that consume them (two register
        // sets); the sched_barrier keeps the compiler from sinking the request back to its use
        f32x4 kb[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kb[0][i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            if (t + 1 < CH) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    kb[(t + 1) & 1][i] = *reinterpret_cast<const f32x4*>(kp + (t + 1) * kLdsStride + 4 * i);
            }
            const f32x4 (&k)[4] = kb[t & 1];
            // two partial sums per row so that the 16 products go through 8 packed FMAs; the rows'
            // chains are written interleaved so that consecutive instructions are independent
            f32x2 a2[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) a2[r] = f32x2{0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; e += 2)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        a2[r] += f32x2{q[r][4 * i + e], q[r][4 * i + e + 1]} * f32x2{k[i][e], k[i][e + 1]};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float a = a2[r][0] + a2[r][1];
                sc[r][t] = a;
                cm[r] = fmaxf(cm[r], a);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 vb[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vb[0][i] = *reinterpret_cast<const f32x4*>(vp + 4 * i);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float mn = fmaxf(m[r], cm[r]);
            const float alpha = __builtin_amdgcn_exp2f(m[r] - mn);
            m[r] = mn;
            den[r] *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[r][i] *= alpha;
        }
        // all exponentials of the chunk first (independent of each other, so the quarter-rate unit
        // pipelines them), then the weighted sum
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                sc[r][t] = __builtin_amdgcn_exp2f(sc[r][t] - m[r]);
                den[r] += sc[r][t];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            if (t + 1 < CH) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    vb[(t + 1) & 1][i] = *reinterpret_cast<const f32x4*>(vp + (t + 1) * kLdsStride + 4 * i);
            }
            const f32x4 (&v)[4] = vb[t & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r][i] += sc[r][t] * v[i];
            __builtin_amdgcn_sched_barrier(0);
        }
        kp += CH * kLdsStride;
        vp += CH * kLdsStride;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float rden = 1.0f / den[r];
        m_out[r] = m[r];
        rden_out[r] = rden;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[r][4 * i + e] = acc[r][i][e] * rden;
    }
}

}  // namespace dptm
