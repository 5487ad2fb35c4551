// HIP kernels (gfx950 / CDNA4) for DeepPTMPred's site-scoring path.
//
//  branch_kernel<W>   one fused launch per window size: embedding + position add,
//                     residual conv block, two transformer blocks and attention pooling
//                     (reference graph: pred/train_PTM/model.py:157-182), fp32 throughout.
//  dense_kernel<..>   Y = epilogue(X*W + b) for the per-site dense layers
//                     (model.py:187-208): struct branch, ESM branch, trunk and head.
//  assemble_kernel    window assembly (pred/train_PTM/predict.py:487-501, 529-560).
//
// All matrix products run on v_mfma_f32_32x32x2_f32 in the transposed orientation
// Y^T = W^T * X^T, so that the data row sits on the lane and features sit on
// (register, half-wave); see dptm_layout.h for the register map ("chain layout").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dptm_layout.h"
#include "dptm_attend.h"
#include "dptm_mfma.h"
#include "dptm_pair.h"
#include "dptm_phase.h"
#include "dptm_stage.h"
#include "dptm_kernels.h"

namespace dptm {

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------
// Fused sequence branch.  One workgroup = 8 wavefronts = one tile of S = 256/W whole
// sites (S*W <= 256 rows, row = site_in_tile*W + position); wavefront w owns rows
// 32w..32w+31 for every row-wise layer.  LDS holds two [256][68] fp32 images:
//   conv phase:  bufA = embedded input, bufB = first conv output (neighbour rows are
//                read by the 3-tap convolutions);
//   transformer: bufA = K, bufB = V of the whole tile (attention reads other rows of
//                the same site); Q, the residual stream and the FFN hidden layer stay
//                in registers;
//   pooling:     bufA = final activations, plus one score per row.
// ------------------------------------------------------------------------------------
// TR: rows of the tile (256: eight wavefronts, the throughput form; 64: two wavefronts, one per SIMD, for small calls,
// whose duration is one wavefront's chain -- which is shorter when the wavefront has its SIMD to itself)
template <int W, int TR = kTileRows>
__device__ __forceinline__ void branch_body(const f32x4* __restrict__ P, const uint8_t* __restrict__ tok,
                                            int64_t n_sites, float* __restrict__ out, int ldo, int col_off) {
    constexpr int S = TR / W;  // sites per tile
    constexpr int NT = 2 * TR;  // threads of the workgroup
    static_assert(S >= 1, "a tile holds whole sites");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = lds + TR * kLdsStride;
    float* score = bufB + TR * kLdsStride;  // [TR]
    // bias / scale / LayerNorm vectors of the whole branch, copied once per workgroup: read from
    // memory they cost the texture path as many instructions as the weight fragments do
    f32x4* vecs = reinterpret_cast<f32x4*>(score + TR);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int64_t site0 = (int64_t)blockIdx.x * S;
    const int sites_here = (int)min((int64_t)S, n_sites - site0);
    const int rows_valid = sites_here * W;

    PH_DECL
    // ---- vectors -> LDS, embedding + position embedding -> bufA (model.py:157-158); the thread's loads of both
    // copies are requested before the first store (dptm_stage.h)
    {
        VecsStage<f32x4, NT> vst;
        EmbedStage<f32x4, W, TR, NT> est;
        vst.load(P, tid);
        est.load_tokens(tok + site0 * W, rows_valid, tid);
        est.load_rows(P, rows_valid, tid);
        vst.store(vecs, tid);
        est.store(bufA, rows_valid, tid);
    }
    __syncthreads();

    const int row = wave * 32 + j;
    const bool row_ok = row < rows_valid;
    const int site_l = row_ok ? row / W : 0;
    const int pos_l = row_ok ? row - site_l * W : 1;
    // first key row of the site whose keys this row (and its pair row, row ^ 1) attends to; padding rows at the
    // end of the tile borrow the last site so that they never split a pair
    const int att_start = min(row / W, S - 1) * W;
    const int att_start_pair = min((row ^ 1) / W, S - 1) * W;
    const bool tap0_ok = row_ok && pos_l > 0;
    const bool tap2_ok = row_ok && pos_l < W - 1;

    PH_MARK(9)    // set-up, embedding (incl. its barrier)
    f32x4 x1[8];  // residual stream of this lane's row, chain layout

    // ---- residual conv block (model.py:129-141)
    {
        const f32x4* cv = vecs;
        f32x16 acc_s[2], acc_a[2];
        acc_init_bias<2>(acc_s, cv + kVScB * kVec64, h, 8);
        acc_init_bias<2>(acc_a, cv + kVC3aB * kVec64, h, 8);
        const float* xc = bufA + row * kLdsStride + 4 * h;
        const float* xl = tap0_ok ? xc - kLdsStride : xc;
        const float* xr = tap2_ok ? xc + kLdsStride : xc;
        const f32x4* wsc = P + kOffScW;
        const f32x4* wa = P + kOffC3aW;
        f32x4 a_sc[2][2], a0[2][2], a1[2][2], a2[2][2];  // [parity of u][out tile]
        f32x4 bc[2], bl[2], br[2];
        auto fetch_a = [&](int u, int s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a_sc[s][t] = wsc[(t * 8 + u) * 64 + lane];
                a0[s][t] = wa[0 * kMat64 + (t * 8 + u) * 64 + lane];
                a1[s][t] = wa[1 * kMat64 + (t * 8 + u) * 64 + lane];
                a2[s][t] = wa[2 * kMat64 + (t * 8 + u) * 64 + lane];
            }
            bc[s] = *reinterpret_cast<const f32x4*>(xc + 8 * u);
            bl[s] = *reinterpret_cast<const f32x4*>(xl + 8 * u);
            br[s] = *reinterpret_cast<const f32x4*>(xr + 8 * u);
        };
        fetch_a(0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = u & 1;
            if (u + 1 < 8) fetch_a(u + 1, s ^ 1);
            if (!tap0_ok) bl[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!tap2_ok) br[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mfma(acc_s[t], a_sc[s][t][e], bc[s][e]);
                    mfma(acc_a[t], a0[s][t][e], bl[s][e]);
                    mfma(acc_a[t], a1[s][t][e], bc[s][e]);
                    mfma(acc_a[t], a2[s][t][e], br[s][e]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // y1 = BN1(swish(conv3a)) -> bufB (own rows)
        f32x4 y[8];
        acc_to_x<2>(acc_a, y);
        float* yb = bufB + row * kLdsStride + 4 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 sc = cv[kVBn1S * kVec64 + h * 8 + u], sh = cv[kVBn1T * kVec64 + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[u][e] = fmaf(swishf(y[u][e]), sc[e], sh[e]);
            *reinterpret_cast<f32x4*>(yb + 8 * u) = y[u];
        }
        __syncthreads();
        // conv3b on bufB
        f32x16 acc_b[2];
        acc_init_bias<2>(acc_b, cv + kVC3bB * kVec64, h, 8);
        const float* yc = bufB + row * kLdsStride + 4 * h;
        const float* yl = tap0_ok ? yc - kLdsStride : yc;
        const float* yr = tap2_ok ? yc + kLdsStride : yc;
        const f32x4* wb = P + kOffC3bW;
        f32x4 c0[2][2], c1[2][2], c2[2][2], dl[2], dr[2];
        auto fetch_b = [&](int u, int s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                c0[s][t] = wb[0 * kMat64 + (t * 8 + u) * 64 + lane];
                c1[s][t] = wb[1 * kMat64 + (t * 8 + u) * 64 + lane];
                c2[s][t] = wb[2 * kMat64 + (t * 8 + u) * 64 + lane];
            }
            dl[s] = *reinterpret_cast<const f32x4*>(yl + 8 * u);
            dr[s] = *reinterpret_cast<const f32x4*>(yr + 8 * u);
        };
        fetch_b(0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = u & 1;
            if (u + 1 < 8) fetch_b(u + 1, s ^ 1);
            if (!tap0_ok) dl[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!tap2_ok) dr[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mfma(acc_b[t], c0[s][t][e], dl[s][e]);
                    mfma(acc_b[t], c1[s][t][e], y[u][e]);
                    mfma(acc_b[t], c2[s][t][e], dr[s][e]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 s[8];
        acc_to_x<2>(acc_b, x1);
        acc_to_x<2>(acc_s, s);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 sc = cv[kVBn2S * kVec64 + h * 8 + u], sh = cv[kVBn2T * kVec64 + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) x1[u][e] = fmaf(swishf(x1[u][e]), sc[e], sh[e]) + s[u][e];
        }
    }

    PH_MARK(0)    // conv block
    // ---- two transformer blocks (model.py:93-118, 172-173)
    // Every product's first two groups of weight fragments are requested while the product before it multiplies
    // (WHead); the K projection's leave before the barrier in front of it.
    WHead<2> hk;
    hk.load(P + kOffTb + kTbWk, 8, 0, lane);
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        const f32x4* T = P + kOffTb + blk * kTbSize;
        const f32x4* tv = vecs + kVecsConv + blk * kVecsTb;
        // every wave is done reading bufA/bufB (conv taps, or the previous block's K/V)
        __syncthreads();
        PH_MARK(2)   // barrier wait
        WHead<2> hq, ho;
        {
            WHead<2> hv;
            hv.load(T + kTbWv, 8, 0, lane);
            f32x16 acc[2];
            acc_init_bias<2>(acc, tv + kVBk * kVec64, h, 8);
            gemm_regs_head<8, 2>(acc, T + kTbWk, x1, lane, hk);
            float* kb = bufA + row * kLdsStride + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(kb + 8 * (4 * t + g)) =
                        f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            hq.load(T + kTbWq, 8, 0, lane);
            acc_init_bias<2>(acc, tv + kVBv * kVec64, h, 8);
            gemm_regs_head<8, 2>(acc, T + kTbWv, x1, lane, hv);
            float* vb = bufB + row * kLdsStride + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(vb + 8 * (4 * t + g)) =
                        f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        }
        // q, scaled by log2(e)/sqrt(head_dim) (the softmax below works in base 2), regrouped so
        // that the lower half-wave holds heads 0,1 and the upper half-wave heads 2,3 of its row
        float qa[16], qb[16];
        {
            f32x16 acc[2];
            ho.load(T + kTbWo, 8, 0, lane);          // in flight across the attention
            acc_init_bias<2>(acc, tv + kVBq * kVec64, h, 8);
            gemm_regs_head<8, 2>(acc, T + kTbWq, x1, lane, hq);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                qa[r] = acc[0][r] * kQScale;
                qb[r] = acc[1][r] * kQScale;
                half_swap(qa[r], qb[r]);
            }
        }
        PH_MARK(3)   // K, V, Q projections
        __syncthreads();  // K and V of the whole tile are in LDS
        PH_MARK(4)   // barrier wait
        // attention: lanes j and j^1 of a half-wave pair up.  Each takes ONE of the half-wave's two heads
        // (head 2h + (j&1)) for BOTH rows of the pair, so every K / V fragment read from LDS serves two
        // query rows.  The lanes trade the 16 query values of the head they give away (one DPP move each).
        {
            const bool odd = (j & 1) != 0;
            float q[2][16], o[2][16];
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                // d-th feature of local head 0 / 1 of this lane's row
                const int src = 4 * (d >> 3) + (d & 3);
                const float h0 = (d & 4) ? qb[src] : qa[src];
                const float h1 = (d & 4) ? qb[8 + src] : qa[8 + src];
                q[0][d] = odd ? h1 : h0;
                q[1][d] = lane_xor1(odd ? h0 : h1);
            }
            const int head_off = 32 * h + (odd ? 16 : 0);
            const float* kp0 = bufA + att_start * kLdsStride + head_off;
            const float* kp1 = bufA + att_start_pair * kLdsStride + head_off;
            const int v_off = TR * kLdsStride;       // bufB = bufA + v_off
            float sm[2], sr[2];                             // softmax statistics: not needed for scoring
            if (__builtin_amdgcn_ballot_w64(att_start != att_start_pair) == 0) {
                attend<W, 2>(kp0, kp0 + v_off, q, o, sm, sr);
            } else {
                // a pair in this wavefront straddles two sites: the rows walk their own keys
                float q1[1][16], o1[1][16], sm1[1], sr1[1];
#pragma unroll
                for (int d = 0; d < 16; ++d) q1[0][d] = q[0][d];
                attend<W, 1>(kp0, kp0 + v_off, q1, o1, sm1, sr1);
#pragma unroll
                for (int d = 0; d < 16; ++d) { o[0][d] = o1[0][d]; q1[0][d] = q[1][d]; }
                attend<W, 1>(kp1, kp1 + v_off, q1, o1, sm1, sr1);
#pragma unroll
                for (int d = 0; d < 16; ++d) o[1][d] = o1[0][d];
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const float got = lane_xor1(o[1][d]);       // this lane's row, the other head
                const int dst = 4 * (d >> 3) + (d & 3);
                const float h0 = odd ? got : o[0][d];
                const float h1 = odd ? o[0][d] : got;
                if (d & 4) { qb[dst] = h0; qb[8 + dst] = h1; }
                else       { qa[dst] = h0; qa[8 + dst] = h1; }
            }
        }
        PH_MARK(5)   // attention
        // back to the chain layout, output projection, residual, LayerNorm 1
        f32x4 hres[8];
        WHead<2> hf;
        {
            f32x4 o[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                half_swap(qa[r], qb[r]);
                o[r >> 2][r & 3] = qa[r];
                o[4 + (r >> 2)][r & 3] = qb[r];
            }
            hf.load(T + kTbF1, 8, 0, lane);
            f32x16 acc[2];
            acc_init_bias<2>(acc, tv + kVBo * kVec64, h, 8);
            gemm_regs_head<8, 2>(acc, T + kTbWo, o, lane, ho);
            acc_to_x<2>(acc, hres);
#pragma unroll
            for (int u = 0; u < 8; ++u) hres[u] += x1[u];
            layernorm64(hres, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h);
        }
        PH_MARK(6)   // output projection, LayerNorm 1
        // FFN 64 -> 256 (swish) -> 64 in four hidden chunks of 64, residual, LayerNorm 2
        {
            f32x16 acc_o[2];
            acc_init_bias<2>(acc_o, tv + kVB2 * kVec64, h, 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x16 acc_h[2];
                // hidden tiles 2c, 2c+1 <-> bias groups 8c..8c+7 of the 256-vector
                WHead<2> h2;
                h2.load(T + kTbF2, 32, 8 * c, lane);
                acc_init_bias<2>(acc_h, tv + 9 * kVec64 + 8 * c, h, 32);
                gemm_regs_head<8, 2, 2>(acc_h, T + kTbF1 + (2 * c) * 8 * 64, hres, lane, hf);
                f32x4 hid[8];
                acc_to_x<2>(acc_h, hid);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hid[u][e] = swishf(hid[u][e]);
                // second layer: K = 256 -> 32 groups per out tile; this quarter is groups 8c..
                {
                    // the next product's first fragments: the next chunk's first layer, or the K projection of the
                    // next block (the same addresses again after the last block: harmless)
                    if (c + 1 < 4)
                        hf.load(T + kTbF1 + (2 * (c + 1)) * 8 * 64, 8, 0, lane);
                    else
                        hk.load(P + kOffTb + kTbSize + kTbWk, 8, 0, lane);
                    f32x4 a[8][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 2; ++t) a[u][t] = h2.a[u][t];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u + 2 < 8) {
#pragma unroll
                            for (int t = 0; t < 2; ++t)
                                a[u + 2][t] = T[kTbF2 + (t * 32 + 8 * c + u + 2) * 64 + lane];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int t = 0; t < 2; ++t) mfma(acc_o[t], a[u][t][e], hid[u][e]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            acc_to_x<2>(acc_o, x1);
#pragma unroll
            for (int u = 0; u < 8; ++u) x1[u] += hres[u];
            layernorm64(x1, tv + kVLn2G * kVec64, tv + kVLn2B * kVec64, h);
        }
        PH_MARK(7)   // FFN, LayerNorm 2
    }

    // ---- attention pooling (model.py:176-182)
    {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 wv = vecs[kVecsPool + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) part = fmaf(x1[u][e], wv[e], part);
        }
        // the score is a tanh, so it lies in [-1, 1]: softmax needs no running maximum, and every row computes
        // its own exponential once; the per-site sum and the weighted sum over positions follow from LDS
        const float ew = __expf(tanhf(xhalf_sum(part) + vecs[kVecsPool + kVec64][0]));
        __syncthreads();  // all attention reads of bufA are finished
        float* xb = bufA + row * kLdsStride + 4 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(xb + 8 * u) = x1[u];
        if (h == 0) score[row] = ew;
        __syncthreads();
        for (int oi = tid; oi < sites_here * 64; oi += NT) {
            const int s = oi >> 6, f = oi & 63;
            const float* sp = score + s * W;
            const float* xp = bufA + (s * W) * kLdsStride + f;
            float den = 0.f, acc = 0.f;
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const float e = sp[t];
                den += e;
                acc = fmaf(e, xp[t * kLdsStride], acc);
            }
            out[(site0 + s) * ldo + col_off + f] = acc / den;
        }
    }
    PH_MARK(8)   // pooling
    PH_FLUSH
}

// ------------------------------------------------------------------------------------
// The same branch for ONE site in FOUR wavefronts (dptm_pair.h): a call of a few dozen sites lasts as long as one
// wavefront's dependent chain, so two wavefronts share each 32-row block -- every row product is cut by output tile
// (32 of the 64 output features each; per output feature the same matrix instructions in the same order, so the
// probabilities are bit-identical to the other tilings'), a lane takes ONE attention head of its row, and the halves of a
// row vector meet through an LDS exchange wherever the next layer needs the whole row.
// ------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void branch_pair_body(const f32x4* __restrict__ P, const uint8_t* __restrict__ tok, int64_t site,
                                                 float* __restrict__ out, int ldo, int col_off) {
    static_assert(W <= 64, "a site fills at most two 32-row blocks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    PairX px{reinterpret_cast<f32x4*>(bufB + kPairImg), 0};
    float* score = bufB + kPairImg + kPairXFloats;  // [64]
    f32x4* vecs = reinterpret_cast<f32x4*>(score + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 1, part = wave >> 1;
    const int h = lane >> 5, j = lane & 31;
    const int row = pw * 32 + j;
    const bool row_ok = row < W;
    const bool tap0_ok = row_ok && row > 0, tap2_ok = row_ok && row < W - 1;
    const int head_off = 16 * (2 * part + h);

    {   // vectors and the site's embedded rows -> LDS, all of a thread's loads requested before its first store
        VecsStage<f32x4, 256> vst;
        EmbedStage<f32x4, W, 64, 256> est;
        vst.load(P, tid);
        est.load_tokens(tok + site * W, W, tid);
        est.load_rows(P, W, tid);
        vst.store(vecs, tid);
        est.store(bufA, W, tid);
    }
    __syncthreads();

    f32x4 x1[8];  // residual stream of this lane's row, chain layout (whole row in both wavefronts of a pair)
    // ---- residual conv block (model.py:129-141), this wavefront's output tile of every convolution
    {
        const f32x4* cv = vecs;
        const f32x4* cvh = vecs + 4 * part;
        f32x16 acc_s[1], acc_a[1];
        acc_init_bias<1>(acc_s, cvh + kVScB * kVec64, h, 8);
        acc_init_bias<1>(acc_a, cvh + kVC3aB * kVec64, h, 8);
        const float* xc = bufA + row * kLdsStride + 4 * h;
        const float* xl = tap0_ok ? xc - kLdsStride : xc;
        const float* xr = tap2_ok ? xc + kLdsStride : xc;
        const f32x4* wsc = P + kOffScW + part * 8 * 64;
        const f32x4* wa = P + kOffC3aW + part * 8 * 64;
        f32x4 a_sc[2], a0[2], a1[2], a2[2], bc[2], bl[2], br[2];  // [parity of u]
        auto fetch_a = [&](int u, int s) {
            a_sc[s] = wsc[u * 64 + lane];
            a0[s] = wa[0 * kMat64 + u * 64 + lane];
            a1[s] = wa[1 * kMat64 + u * 64 + lane];
            a2[s] = wa[2 * kMat64 + u * 64 + lane];
            bc[s] = *reinterpret_cast<const f32x4*>(xc + 8 * u);
            bl[s] = *reinterpret_cast<const f32x4*>(xl + 8 * u);
            br[s] = *reinterpret_cast<const f32x4*>(xr + 8 * u);
        };
        fetch_a(0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = u & 1;
            if (u + 1 < 8) fetch_a(u + 1, s ^ 1);
            if (!tap0_ok) bl[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!tap2_ok) br[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mfma(acc_s[0], a_sc[s][e], bc[s][e]);
                mfma(acc_a[0], a0[s][e], bl[s][e]);
                mfma(acc_a[0], a1[s][e], bc[s][e]);
                mfma(acc_a[0], a2[s][e], br[s][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // y1 = BN1(swish(conv3a)), own features -> bufB (own rows)
        f32x4 y[4];
        acc_to_x<1>(acc_a, y);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sc = cv[kVBn1S * kVec64 + h * 8 + 4 * part + g], sh = cv[kVBn1T * kVec64 + h * 8 + 4 * part + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[g][e] = fmaf(swishf(y[g][e]), sc[e], sh[e]);
        }
        chain_store4(bufB + row * kLdsStride, h, part, y);
        __syncthreads();
        // conv3b on bufB (all three taps from the image: the row's other half came from the partner)
        f32x16 acc_b[1];
        acc_init_bias<1>(acc_b, cvh + kVC3bB * kVec64, h, 8);
        const float* yc = bufB + row * kLdsStride + 4 * h;
        const float* yl = tap0_ok ? yc - kLdsStride : yc;
        const float* yr = tap2_ok ? yc + kLdsStride : yc;
        const f32x4* wb = P + kOffC3bW + part * 8 * 64;
        f32x4 c0[2], c1[2], c2[2], dl[2], dc[2], dr[2];
        auto fetch_b = [&](int u, int s) {
            c0[s] = wb[0 * kMat64 + u * 64 + lane];
            c1[s] = wb[1 * kMat64 + u * 64 + lane];
            c2[s] = wb[2 * kMat64 + u * 64 + lane];
            dl[s] = *reinterpret_cast<const f32x4*>(yl + 8 * u);
            dc[s] = *reinterpret_cast<const f32x4*>(yc + 8 * u);
            dr[s] = *reinterpret_cast<const f32x4*>(yr + 8 * u);
        };
        fetch_b(0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = u & 1;
            if (u + 1 < 8) fetch_b(u + 1, s ^ 1);
            if (!tap0_ok) dl[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!tap2_ok) dr[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mfma(acc_b[0], c0[s][e], dl[s][e]);
                mfma(acc_b[0], c1[s][e], dc[s][e]);
                mfma(acc_b[0], c2[s][e], dr[s][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 x4[4], s4[4];
        acc_to_x<1>(acc_b, x4);
        acc_to_x<1>(acc_s, s4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sc = cv[kVBn2S * kVec64 + h * 8 + 4 * part + g], sh = cv[kVBn2T * kVec64 + h * 8 + 4 * part + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) x4[g][e] = fmaf(swishf(x4[g][e]), sc[e], sh[e]) + s4[g][e];
        }
        pair_gather(px, wave, lane, part, x4, x1);
    }

    // ---- two transformer blocks (model.py:93-118, 172-173)
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        const f32x4* T = P + kOffTb + blk * kTbSize;
        const f32x4* Th = T + part * 8 * 64;           // output tile `part` of a packed 64 x 64 matrix
        const f32x4* tv = vecs + kVecsConv + blk * kVecsTb;
        const f32x4* tvh = tv + 4 * part;
        __syncthreads();   // every wavefront is done reading bufA / bufB (conv taps, or the previous block's K / V)
        {
            f32x16 acc[1];
            f32x4 t[4];
            acc_init_bias<1>(acc, tvh + kVBk * kVec64, h, 8);
            gemm_regs<8, 1, 8>(acc, Th + kTbWk, x1, lane);
            acc_to_x<1>(acc, t);
            chain_store4(bufA + row * kLdsStride, h, part, t);
            acc_init_bias<1>(acc, tvh + kVBv * kVec64, h, 8);
            gemm_regs<8, 1, 8>(acc, Th + kTbWv, x1, lane);
            acc_to_x<1>(acc, t);
            chain_store4(bufB + row * kLdsStride, h, part, t);
        }
        float q1[1][16];
        {
            f32x16 acc[1];
            f32x4 t[4];
            acc_init_bias<1>(acc, tvh + kVBq * kVec64, h, 8);
            gemm_regs<8, 1, 8>(acc, Th + kTbWq, x1, lane);
            acc_to_x<1>(acc, t);
            x4_to_head(t, q1[0], kQScale);
        }
        __syncthreads();   // K and V of the site are in LDS
        f32x4 o[8];
        {
            float o1[1][16], sm[1], sr[1];   // softmax statistics: not needed for scoring
            attend<W, 1>(bufA + head_off, bufB + head_off, q1, o1, sm, sr);
            f32x4 o4[4];
            head_to_x4(o1[0], o4, 1.0f);
            pair_gather(px, wave, lane, part, o4, o);
        }
        // output projection, residual, LayerNorm 1
        f32x4 hres[8];
        {
            f32x16 acc[1];
            acc_init_bias<1>(acc, tvh + kVBo * kVec64, h, 8);
            gemm_regs<8, 1, 8>(acc, Th + kTbWo, o, lane);
            f32x4 t4[4], r4[4];
            acc_to_x<1>(acc, t4);
            own_groups(x1, part, r4);
#pragma unroll
            for (int g = 0; g < 4; ++g) t4[g] += r4[g];
            pair_gather(px, wave, lane, part, t4, hres);
            layernorm64(hres, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h);
        }
        // FFN 64 -> 256 (swish) -> 64 in four hidden chunks of 64: first layer by output tile, the chunk's hidden units meet,
        // second layer by output tile again (so that an output feature sums the chunks in the order of the other tilings)
        {
            f32x16 acc_o[1];
            acc_init_bias<1>(acc_o, tvh + kVB2 * kVec64, h, 8);
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x16 acc_h[1];
                acc_init_bias<1>(acc_h, tv + 9 * kVec64 + 8 * c + 4 * part, h, 32);
                gemm_regs<8, 1, 8>(acc_h, T + kTbF1 + (2 * c + part) * 8 * 64, hres, lane);
                f32x4 h4[4], hid[8];
                acc_to_x<1>(acc_h, h4);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) h4[g][e] = swishf(h4[g][e]);
                pair_gather(px, wave, lane, part, h4, hid);
                {
                    const f32x4* w2 = T + kTbF2 + (part * 32 + 8 * c) * 64;
                    f32x4 a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = w2[u * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) mfma(acc_o[0], a[u][e], hid[u][e]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            f32x4 t4[4], r4[4];
            acc_to_x<1>(acc_o, t4);
            own_groups(hres, part, r4);
#pragma unroll
            for (int g = 0; g < 4; ++g) t4[g] += r4[g];
            pair_gather(px, wave, lane, part, t4, x1);
            layernorm64(x1, tv + kVLn2G * kVec64, tv + kVLn2B * kVec64, h);
        }
    }

    // ---- attention pooling (model.py:176-182)
    {
        float dotp = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 wv = vecs[kVecsPool + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) dotp = fmaf(x1[u][e], wv[e], dotp);
        }
        const float ew = __expf(tanhf(xhalf_sum(dotp) + vecs[kVecsPool + kVec64][0]));
        __syncthreads();  // all attention reads of bufA are finished
        f32x4 x4[4];
        own_groups(x1, part, x4);
        chain_store4(bufA + row * kLdsStride, h, part, x4);
        if (h == 0 && part == 0) score[row] = ew;
        __syncthreads();
        if (tid < 64) {
            const float* xp = bufA + tid;
            float den = 0.f, acc = 0.f;
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const float e = score[t];
                den += e;
                acc = fmaf(e, xp[t * kLdsStride], acc);
            }
            out[site * ldo + col_off + tid] = acc / den;
        }
    }
}

// ------------------------------------------------------------------------------------
// Per-site dense layer: out[r][col_off + n] = epi(sum_k X[src(r)][k] * Wt[k][n] + b[n]).
// One wavefront computes 64 rows x (NT*32) columns; a workgroup = 4 wavefronts = 256
// rows of one column slab (blockIdx.y), so its wavefronts share the weight stream.  Two workgroups
// per CU (launch bound): without the bound the compiler takes 154 + 128 accumulator registers, one
// wavefront per SIMD; with it 190 and none spilled, and the layers run 3-5 % faster.
// ------------------------------------------------------------------------------------
template <int NT, int EPI>
__global__ __launch_bounds__(256, 2) void dense_kernel(DenseArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (r0 >= p.m) return;
    const int slab = blockIdx.y;
    const int k8 = p.k >> 3;
    const f32x4* wp = p.w + (size_t)slab * NT * k8 * 64;

    const float* xrow[2];
    bool ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t r = r0 + 32 * i + j;
        if (r > p.m - 1) r = p.m - 1;
        int64_t src = r;
        ok[i] = true;
        if (p.row_index) {
            const int32_t s = p.row_index[r];
            ok[i] = s >= 0;
            src = s >= 0 ? s : 0;
        }
        xrow[i] = p.x + src * p.ldx + 4 * h;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = p.bias[h * (p.n_pad >> 3) + slab * NT * 4 + 4 * t + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][t][4 * g + e] = b[e];
                acc[1][t][4 * g + e] = b[e];
            }
        }

    f32x4 a_n[NT], b_n[2];
#pragma unroll
    for (int t = 0; t < NT; ++t) a_n[t] = wp[(t * k8) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) b_n[i] = *reinterpret_cast<const f32x4*>(xrow[i]);
    for (int u = 0; u < k8; ++u) {
        f32x4 a[NT], b[2];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = a_n[t];
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = ok[i] ? b_n[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (u + 1 < k8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a_n[t] = wp[(t * k8 + u + 1) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                b_n[i] = *reinterpret_cast<const f32x4*>(xrow[i] + 8 * (u + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mfma(acc[0][t], a[t][e], b[0][e]);
                mfma(acc[1][t], a[t][e], b[1][e]);
            }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + 32 * i + j;
        if (r >= p.m) continue;
        if (EPI == kEpiHead) {
            // logits 0,1 sit in registers 0,1 of the lower half-wave; softmax over 2
            if (h == 0) {
                const float l0 = acc[i][0][0], l1 = acc[i][0][1];
                const float m = fmaxf(l0, l1);
                const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
                const float rs = 1.0f / (e0 + e1);
                float2 o = {e0 * rs, e1 * rs};
                *reinterpret_cast<float2*>(p.out + r * p.ldo) = o;
            }
        } else {
            float* orow = p.out + r * p.ldo + p.col_off + slab * NT * 32 + 4 * h;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2],
                               acc[i][t][4 * g + 3]};
                    const int vi = h * (p.n_pad >> 3) + slab * NT * 4 + 4 * t + g;
                    if (EPI == kEpiSwish || EPI == kEpiSwishAffine) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = swishf(v[e]);
                    }
                    if (EPI == kEpiSwishAffine) {
                        const f32x4 sc = p.scale[vi], sh = p.shift[vi];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                    }
                    *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * g) = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------
// The same dense layer for a handful of rows (one protein's sites): a 64x128 tile per wavefront would
// leave the chip idle and walk the whole contraction in one wavefront.  Here a workgroup owns 32 rows x
// 32 columns and its 8 wavefronts split the contraction; partial tiles meet in LDS, where the bias and
// the epilogue are applied.  grid = (ceil(m/32), n_pad/32), block 512.
// ------------------------------------------------------------------------------------
// part: 32 KB of LDS, [8 wavefronts][16 registers][64 lanes]; (bx, by): row tile and column tile of this workgroup
template <int EPI>
__device__ __forceinline__ void dense_small_body(const DenseArgs& p, float (*part)[16][64], int bx, int by) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = (int64_t)bx * 32;
    const int tile = by;
    const int k8 = p.k >> 3;
    const int per = (k8 + 7) / 8;
    const int u0 = wave * per, u1 = min(k8, u0 + per);
    int64_t r = r0 + j;
    if (r > p.m - 1) r = p.m - 1;
    int64_t src = r;
    bool ok = true;
    if (p.row_index) {
        const int32_t s = p.row_index[r];
        ok = s >= 0;
        src = s >= 0 ? s : 0;
    }
    const float* xrow = p.x + src * p.ldx + 4 * h;
    const f32x4* wp = p.w + (size_t)tile * k8 * 64 + lane;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ub = u0; ub < u1; ub += 4) {      // four steps of operands in flight
        f32x4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = b[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ub + v < u1) {
                a[v] = wp[(size_t)(ub + v) * 64];
                if (ok) b[v] = *reinterpret_cast<const f32x4*>(xrow + 8 * (ub + v));
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) mfma(acc, a[v][e], b[v][e]);
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) part[wave][rg][lane] = acc[rg];
    __syncthreads();
    // 1024 outputs of the tile, two per thread: register rg of lane l is row r0 + (l & 31), feature
    // 8*(rg>>2) + 4*(l>>5) + (rg&3) of this 32-column tile
    for (int o = threadIdx.x; o < 1024; o += 512) {
        const int rg = o >> 6, l = o & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += part[w][rg][l];
        const int hh = l >> 5;
        const int g = rg >> 2, e = rg & 3;
        const int vi = hh * (p.n_pad >> 3) + tile * 4 + g;     // packed-vector group of this feature
        v += reinterpret_cast<const float*>(p.bias + vi)[e];
        const int64_t row = r0 + (l & 31);
        if (row >= p.m) continue;
        if (EPI == kEpiSwish || EPI == kEpiSwishAffine) v = swishf(v);
        if (EPI == kEpiSwishAffine)
            v = fmaf(v, reinterpret_cast<const float*>(p.scale + vi)[e], reinterpret_cast<const float*>(p.shift + vi)[e]);
        p.out[row * p.ldo + p.col_off + tile * 32 + 8 * g + 4 * hh + e] = v;
    }
}

// The 2-class head of a small call: softmax(x W + b) for 32 rows per workgroup, eight threads per row over the
// contraction (the throughput tiling walks it in one wavefront: 14 us for 53 rows against 4).  Reads the packed weight
// image: W[k][n] sits in element k & 3 of fragment (k >> 3) * 64 + n + 32 * ((k >> 2) & 1).
__global__ __launch_bounds__(256) void head_small_kernel(DenseArgs p) {
    const int r = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    const int64_t row = r < p.m ? r : p.m - 1;
    const float* x = p.x + row * p.ldx;
    float l0 = 0.f, l1 = 0.f;
    for (int k4 = part; k4 < (p.k >> 2); k4 += 8) {          // four consecutive inputs per step
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * k4);
        const int frag = (k4 >> 1) * 64 + 32 * (k4 & 1);
        const f32x4 w0 = p.w[frag], w1 = p.w[frag + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            l0 = fmaf(xv[e], w0[e], l0);
            l1 = fmaf(xv[e], w1[e], l1);
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        l0 += __shfl_xor(l0, o);
        l1 += __shfl_xor(l1, o);
    }
    if (part == 0 && r < p.m) {
        const float* b = reinterpret_cast<const float*>(p.bias);
        l0 += b[0];
        l1 += b[1];
        const float m = fmaxf(l0, l1);
        const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
        const float rs = 1.0f / (e0 + e1);
        float2 o = {e0 * rs, e1 * rs};
        *reinterpret_cast<float2*>(p.out + (int64_t)r * p.ldo) = o;
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void dense_small_kernel(DenseArgs p) {
    __shared__ float part[8][16][64];
    dense_small_body<EPI>(p, part, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------
// Window assembly (predict.py:529-560) and the ESM row index (predict.py:487-501) for sites of many
// proteins at once: column c of the W-window of 1-based, protein-local position `pos` is residue
// pos-1-(W-1)/2+c of that protein, or the padding code 20 outside 0..L-1 (also for codes > 20).
// A workgroup owns kAsmSites = 256 consecutive sites, whose three output ranges [s0*W, (s0+256)*W) are contiguous
// bytes; they are composed in LDS images and leave as 16-byte stores, 1 KiB per wavefront instruction.
// One thread per site: the W=33 and W=21 windows are the middle of the W=51 window, so the thread fetches the 51
// residue bytes once, as the (at most) 14 aligned 32-bit words that hold them, shifts them into 13 packed words,
// replaces codes above 20 and the bytes outside the protein by the padding code four bytes at a time, and writes
// three runs (51, 33 and 21 bytes) into the images.  Wavefront w takes the sites s = 4*lane + w: 51, 33 and 21 are
// odd, so the byte alignment of a run's start, (W*s) mod 4, is then the same for all lanes of a wavefront, and a
// run is a few single bytes at its ends and aligned 32-bit words in between (14 LDS stores for 51 bytes instead of
// 51; lanes 4 sites apart are an odd number of words apart: no bank conflicts).  The LDS stores and the per-byte
// work were what bounded the earlier forms of this kernel.
// seq_off == nullptr: one protein of length a.L (site_protein is then not read).
// ------------------------------------------------------------------------------------
constexpr int kAsmSites = 256;
constexpr int kAsmImg21 = 0, kAsmImg33 = kAsmSites * 21, kAsmImg51 = kAsmImg33 + kAsmSites * 33;   // byte offsets
constexpr int kAsmImgBytes = kAsmSites * (21 + 33 + 51);
static_assert(kAsmImg33 % 16 == 0 && kAsmImg51 % 16 == 0 && kAsmImgBytes % 16 == 0, "images start on 16-byte boundaries");

// bytes [0, bytes) of an LDS image to `out`; vec: `out` is 16-byte aligned
__device__ __forceinline__ void assemble_flush(const uint8_t* img, uint8_t* __restrict__ out, int bytes, bool vec) {
    const int whole = vec ? bytes >> 4 : 0;
    for (int d = threadIdx.x; d < whole; d += blockDim.x)
        reinterpret_cast<uint4*>(out)[d] = reinterpret_cast<const uint4*>(img)[d];
    for (int o = 16 * whole + threadIdx.x; o < bytes; o += blockDim.x) out[o] = img[o];
}

// four packed window bytes starting at byte b (compile-time) of r[0..12]
template <int B>
__device__ __forceinline__ uint32_t assemble_word(const uint32_t (&r)[14]) {
    constexpr int w = B >> 2, sft = B & 3;
    return sft == 0 ? r[w] : __builtin_amdgcn_alignbyte(r[w + 1], r[w], (uint32_t)sft);
}

// wp[k] = window bytes [B + 4k, B + 4k + 4) for k = K .. WORDS-1
template <int B, int K, int WORDS>
__device__ __forceinline__ void assemble_words(uint32_t* wp, const uint32_t (&r)[14]) {
    if constexpr (K < WORDS) {
        wp[K] = assemble_word<B + 4 * K>(r);
        assemble_words<B, K + 1, WORDS>(wp, r);
    }
}

// Window bytes [FROM, FROM + N) of r to LDS at dst, whose address is AL (mod 4): single bytes / a 16-bit piece up
// to the first word boundary, aligned words, then what is left.
template <int AL, int FROM, int N>
__device__ __forceinline__ void assemble_run(uint8_t* dst, const uint32_t (&r)[14]) {
    constexpr int head = (4 - AL) & 3;
    if (head != 0) {
        const uint32_t v = assemble_word<FROM>(r);
        if (head & 1) dst[0] = (uint8_t)v;                                                        // AL = 1 or 3
        if (head & 2) *reinterpret_cast<uint16_t*>(dst + (head & 1)) = (uint16_t)(v >> (8 * (head & 1)));   // AL = 1 or 2
    }
    constexpr int words = (N - head) >> 2;
    uint32_t* wp = reinterpret_cast<uint32_t*>(dst + head);
    assemble_words<FROM + head, 0, words>(wp, r);
    constexpr int done = head + 4 * words, tail = N - done;
    if (tail != 0) {
        const uint32_t v = assemble_word<FROM + done>(r);
        if (tail & 2) *reinterpret_cast<uint16_t*>(dst + done) = (uint16_t)v;
        if (tail & 1) dst[done + (tail & 2)] = (uint8_t)(v >> (8 * (tail & 2)));
    }
}

// the three runs of one site; S = site index (mod 4) of the wavefront: 51 = 3, 33 = 21 = 1 (mod 4)
template <int S>
__device__ __forceinline__ void assemble_runs(uint8_t* img, int sl, const uint32_t (&r)[14]) {
    assemble_run<(3 * S) & 3, 0, 51>(img + kAsmImg51 + sl * 51, r);
    assemble_run<S, 9, 33>(img + kAsmImg33 + sl * 33, r);        // columns 9..41
    assemble_run<S, 15, 21>(img + kAsmImg21 + sl * 21, r);       // columns 15..35
}

// WORDS: the residue array starts on a 4-byte boundary and is read in 32-bit words
template <bool WORDS>
__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t img[kAsmImgBytes];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = 4 * (tid & 63) + w;
    const int64_t tiles = (a.n + kAsmSites - 1) / kAsmSites;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t s0 = tile * kAsmSites;
        const int rows = (int)min((int64_t)kAsmSites, a.n - s0);
        if (sl < rows) {
            const int64_t site = s0 + sl;
            const int32_t pos = a.positions[site];
            int64_t base = 0, len = a.L, erow0 = 0, erows = a.esm_rows;
            if (a.seq_off) {
                const int32_t p = a.site_protein[site];
                base = a.seq_off[p];
                len = a.seq_off[p + 1] - base;
                erow0 = a.esm_off[p];
                erows = a.esm_off[p + 1] - erow0;
            }
            a.esm_row[site] = (pos - 1 < erows) ? (int32_t)(erow0 + pos - 1) : -1;
            const int i0 = pos - 1 - 25;                      // residue of window column 0
            const int last = (int)len - 1;
            // r[j] = window bytes 4j .. 4j+3 (r[12]: bytes 48..50 and one more), r[13] = 0
            uint32_t r[14];
            if (WORDS) {
                // every word is requested before the first is used; a word outside the protein's own words is replaced
                // by the nearest one of them -- every byte it stands for is padding -- so nothing beyond the residue
                // array's first and last word is touched
                const int64_t g = base + i0;                 // first byte, as an offset into codes (may be negative)
                const uint32_t sh = (uint32_t)(g & 3);
                const int64_t d0 = g >> 2, dlo = base >> 2, dhi = (base + last) >> 2;
                const uint32_t* cw = reinterpret_cast<const uint32_t*>(a.codes);
                uint32_t dw[14];
#pragma unroll
                for (int j = 0; j < 14; ++j) {
                    const int64_t d = d0 + j;
                    dw[j] = cw[d < dlo ? dlo : (d > dhi ? dhi : d)];
                }
#pragma unroll
                for (int j = 0; j < 13; ++j) r[j] = __builtin_amdgcn_alignbyte(dw[j + 1], dw[j], sh);
            } else {
                const uint8_t* src = a.codes + base;
#pragma unroll
                for (int j = 0; j < 13; ++j) {
                    uint32_t v = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = i0 + 4 * j + e;
                        v |= (uint32_t)src[i < 0 ? 0 : (i > last ? last : i)] << (8 * e);
                    }
                    r[j] = v;
                }
            }
            r[13] = 0u;
            // bytes [lo, hi) of the window lie inside the protein
            const int lo = min(max(-i0, 0), 52), hi = min(max(last + 1 - i0, 0), 52);
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                uint32_t x = r[j];
                // codes above 20 -> 20, four bytes at a time: bit 7 of (x & 0x7f) + 107 is set where the low seven bits
                // are 21 or more, bit 7 of x itself where the byte is 128 or more
                const uint32_t m7 = (((x & 0x7f7f7f7fu) + 0x6b6b6b6bu) | x) & 0x80808080u;
                const uint32_t big = m7 | (m7 - (m7 >> 7));                    // 0xff in every such byte
                x = (x & ~big) | (0x14141414u & big);
                // bytes of this word inside the protein: [a4, b4) of 0..4
                const int a4 = min(max(lo - 4 * j, 0), 4), b4 = min(max(hi - 4 * j, 0), 4);
                const uint32_t in = (uint32_t)((1ull << (8 * b4)) - 1ull) & ~(uint32_t)((1ull << (8 * a4)) - 1ull);
                r[j] = (x & in) | (0x14141414u & ~in);
            }
            switch (w) {          // the same for the whole wavefront
                case 0: assemble_runs<0>(img, sl, r); break;
                case 1: assemble_runs<1>(img, sl, r); break;
                case 2: assemble_runs<2>(img, sl, r); break;
                default: assemble_runs<3>(img, sl, r); break;
            }
        }
        __syncthreads();
        assemble_flush(img + kAsmImg21, a.t21 + s0 * 21, rows * 21, a.words);
        assemble_flush(img + kAsmImg33, a.t33 + s0 * 33, rows * 33, a.words);
        assemble_flush(img + kAsmImg51, a.t51 + s0 * 51, rows * 51, a.words);
        __syncthreads();
    }
}

template <int W>
__global__ __launch_bounds__(512) void branch_kernel(const f32x4* __restrict__ P,
                                                     const uint8_t* __restrict__ tok,
                                                     int64_t n_sites, float* __restrict__ out,
                                                     int ldo, int col_off) {
    branch_body<W>(P, tok, n_sites, out, ldo, col_off);
}

// The three window sizes of a SMALL call in one launch (grid.y = 0: W = 51, whose workgroups are handed out first;
// grid.x = the most tiles a branch has, a workgroup beyond its branch's tiles leaves at once): a call of a protein's
// worth of sites is one wavefront chain per branch, and as three launches on three streams it paid two hand-overs
// between streams -- and twice the chain whenever two of those streams shared a hardware queue.
struct Branch3Args {
    const f32x4* P[3];          // windows 21 | 33 | 51
    const uint8_t* tok[3];
    int64_t n_sites;
    float* out;
    int ldo;
};
__global__ __launch_bounds__(512) void branch3_kernel(Branch3Args p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x * (kTileRows / 21) < p.n_sites) branch_body<21>(p.P[0], p.tok[0], p.n_sites, p.out, p.ldo, 0);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x * (kTileRows / 33) < p.n_sites) branch_body<33>(p.P[1], p.tok[1], p.n_sites, p.out, p.ldo, kD);
    } else {
        if ((int64_t)blockIdx.x * (kTileRows / 51) < p.n_sites) branch_body<51>(p.P[2], p.tok[2], p.n_sites, p.out, p.ldo, 2 * kD);
    }
}

// ... and with the two layers that depend on the call's inputs only -- the structure layer and the first ESM layer
// (both swish + folded BatchNorm, small-call tiling) -- as grid.y = 3 and 4 of the same launch: everything of a small
// call that can start at once starts at once.
struct Front5Args {
    Branch3Args br;
    DenseArgs strct, esm1;
};
constexpr int kSmallTileRows = 64;      // two wavefronts: one site of 51 or 33 rows, three of 21
constexpr int64_t kSmallTileSites = 192; // calls up to this many sites take them: 2n + n/3 two-wavefront workgroups, at most two
                                         // per CU, so every wavefront still has a SIMD to itself (measured: 53 sites 120 us
                                         // against 190 with eight-wavefront tiles, 160 sites 131 against 197, 220 sites 203 against 197)
__global__ __launch_bounds__(2 * kSmallTileRows) void front5_kernel(Front5Args p) {
    extern __shared__ __attribute__((aligned(16))) float lds_front[];
    constexpr int TR = kSmallTileRows;
    if (blockIdx.y == 0) {
        if ((int64_t)blockIdx.x * (TR / 51) < p.br.n_sites) branch_body<51, TR>(p.br.P[2], p.br.tok[2], p.br.n_sites, p.br.out, p.br.ldo, 2 * kD);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x * (TR / 33) < p.br.n_sites) branch_body<33, TR>(p.br.P[1], p.br.tok[1], p.br.n_sites, p.br.out, p.br.ldo, kD);
    } else {
        if ((int64_t)blockIdx.x * (TR / 21) < p.br.n_sites) branch_body<21, TR>(p.br.P[0], p.br.tok[0], p.br.n_sites, p.br.out, p.br.ldo, 0);
    }
}
// The same for calls of at most kPairSites sites: ONE site in four wavefronts, two per 32-row block (branch_pair_body)
constexpr int64_t kPairSites = 85;      // 3 x 85 = 255 workgroups, one per CU
__global__ __launch_bounds__(256) void front_pair_kernel(Branch3Args p) {
    const int64_t site = blockIdx.x;
    if (blockIdx.y == 0)
        branch_pair_body<51>(p.P[2], p.tok[2], site, p.out, p.ldo, 2 * kD);
    else if (blockIdx.y == 1)
        branch_pair_body<33>(p.P[1], p.tok[1], site, p.out, p.ldo, kD);
    else
        branch_pair_body<21>(p.P[0], p.tok[0], site, p.out, p.ldo, 0);
}
// the structure layer (grid.y = 0) and the first ESM layer (grid.y = 1) of a small call in one launch
__global__ __launch_bounds__(512) void dense2_small_kernel(DenseArgs a0, DenseArgs a1) {
    __shared__ float part[8][16][64];
    const DenseArgs& d = blockIdx.y == 0 ? a0 : a1;
    const int ctiles = d.n_pad >> 5, rtiles = (int)((d.m + 31) >> 5);
    if ((int)blockIdx.x < ctiles * rtiles) dense_small_body<kEpiSwishAffine>(d, part, (int)blockIdx.x / ctiles, (int)blockIdx.x % ctiles);
}

// ------------------------------------------------------------------------------------ launchers

static constexpr size_t kBranchLds = (2 * kTileRows * kLdsStride + kTileRows) * sizeof(float) + kVecsF4 * 16;

// the LDS opt-in is per device: remember which devices of this process have it
template <int W>
static hipError_t branch_attr(int dev) {
    static bool attr_done_dev[64] = {};
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_done_dev[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&branch_kernel<W>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBranchLds);
    if (e == hipSuccess) attr_done_dev[dev] = true;
    return e;
}

template <int W>
static hipError_t launch_branch_t(const float* params, const uint8_t* tok, int64_t n, float* out,
                                  int ldo, int col_off, hipStream_t st) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = branch_attr<W>(dev)) != hipSuccess) return e;
    constexpr int S = kTileRows / W;
    const int64_t blocks = (n + S - 1) / S;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(branch_kernel<W>, dim3((unsigned)blocks), dim3(512), kBranchLds, st,
                       reinterpret_cast<const f32x4*>(params), tok, n, out, ldo, col_off);
    return hipGetLastError();
}

static hipError_t branch3_attr(int dev) {
    static bool attr_done_dev[64] = {};
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_done_dev[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&branch3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kBranchLds);
    if (e == hipSuccess) attr_done_dev[dev] = true;
    return e;
}

hipError_t launch_branch3(const float* const (&params)[3], const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo,
                          hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = branch3_attr(dev)) != hipSuccess) return e;
    Branch3Args p;
    for (int b = 0; b < 3; ++b) {
        p.P[b] = reinterpret_cast<const f32x4*>(params[b]);
        p.tok[b] = tok[b];
    }
    p.n_sites = n;
    p.out = out;
    p.ldo = ldo;
    constexpr int S51 = kTileRows / 51;
    const int64_t blocks = (n + S51 - 1) / S51;      // W = 51 has the fewest sites per tile, hence the most tiles
    hipLaunchKernelGGL(branch3_kernel, dim3((unsigned)blocks, 3), dim3(512), kBranchLds, st, p);
    return hipGetLastError();
}

static constexpr size_t kFrontPairLds = ((size_t)2 * kPairImg + kPairXFloats + 64) * sizeof(float) + kVecsF4 * 16;
static hipError_t front_pair_attr(int dev) {
    static bool attr_done_dev[64] = {};
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_done_dev[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&front_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kFrontPairLds);
    if (e == hipSuccess) attr_done_dev[dev] = true;
    return e;
}

static constexpr size_t kFrontLds = (2 * kSmallTileRows * kLdsStride + kSmallTileRows) * sizeof(float) + kVecsF4 * 16;
static_assert(kFrontLds <= 64 * 1024, "the small tile needs no LDS opt-in");

// The front of a small call: the three branches in two-wavefront tiles (one launch, grid.y = branch) and, next to it in
// a second launch on the same stream -- the first leaves most CUs idle -- the structure and first ESM layers.
hipError_t launch_front5(const float* const (&params)[3], const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo,
                         const DenseArgs& strct, const DenseArgs& esm1, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (strct.m != n || esm1.m != n || n > kDenseSmallRows || strct.n_pad % 32 != 0 || esm1.n_pad % 32 != 0) return hipErrorInvalidValue;
    Front5Args p;
    for (int b = 0; b < 3; ++b) {
        p.br.P[b] = reinterpret_cast<const f32x4*>(params[b]);
        p.br.tok[b] = tok[b];
    }
    p.br.n_sites = n;
    p.br.out = out;
    p.br.ldo = ldo;
    const int64_t rtiles = (n + 31) / 32;
    const int64_t gd = std::max<int64_t>(rtiles * (strct.n_pad / 32), rtiles * (esm1.n_pad / 32));
    hipLaunchKernelGGL(dense2_small_kernel, dim3((unsigned)gd, 2), dim3(512), 0, st, strct, esm1);
    static const int64_t small_tile_sites = [] {
        const char* e = std::getenv("DPTM_SMALL_TILE_SITES");
        return e ? (int64_t)std::atoll(e) : kSmallTileSites;
    }();
    static const int64_t pair_sites = [] {
        const char* e = std::getenv("DPTM_SCORE_PAIR_SITES");      // (measurements: 0 switches the four-wavefront sites off)
        return e ? (int64_t)std::atoll(e) : kPairSites;
    }();
    if (n <= pair_sites && n <= kPairSites) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if ((e = front_pair_attr(dev)) != hipSuccess) return e;
        hipLaunchKernelGGL(front_pair_kernel, dim3((unsigned)n, 3), dim3(256), kFrontPairLds, st, p.br);
        return hipGetLastError();
    }
    if (n <= small_tile_sites) {
        // one site per tile for W = 51 and 33 (the most tiles), three for W = 21
        hipLaunchKernelGGL(front5_kernel, dim3((unsigned)n, 3), dim3(2 * kSmallTileRows), kFrontLds, st, p);
        return hipGetLastError();
    }
    // more sites than two-wavefront tiles have SIMDs to themselves: the eight-wavefront tiles
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = branch3_attr(dev)) != hipSuccess) return e;
    constexpr int S51 = kTileRows / 51;
    hipLaunchKernelGGL(branch3_kernel, dim3((unsigned)((n + S51 - 1) / S51), 3), dim3(512), kBranchLds, st, p.br);
    return hipGetLastError();
}

hipError_t launch_branch(int w, const float* params, const uint8_t* tok, int64_t n, float* out,
                         int ldo, int col_off, hipStream_t st) {
    switch (w) {
        case 21: return launch_branch_t<21>(params, tok, n, out, ldo, col_off, st);
        case 33: return launch_branch_t<33>(params, tok, n, out, ldo, col_off, st);
        case 51: return launch_branch_t<51>(params, tok, n, out, ldo, col_off, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_dense(const DenseArgs& a, int epi, hipStream_t st) {
    if (a.m == 0) return hipSuccess;
    if (a.m <= kDenseSmallRows && epi != kEpiHead) {
        if (a.n_pad % 32 != 0) return hipErrorInvalidValue;
        const dim3 grid((unsigned)((a.m + 31) / 32), (unsigned)(a.n_pad / 32));
        if (epi == kEpiSwish)
            hipLaunchKernelGGL((dense_small_kernel<kEpiSwish>), grid, dim3(512), 0, st, a);
        else if (epi == kEpiSwishAffine)
            hipLaunchKernelGGL((dense_small_kernel<kEpiSwishAffine>), grid, dim3(512), 0, st, a);
        else
            return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (epi == kEpiHead && a.m <= kDenseSmallRows && a.k % 4 == 0 && !a.row_index) {
        hipLaunchKernelGGL(head_small_kernel, dim3((unsigned)((a.m + 31) / 32)), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    const unsigned gx = (unsigned)((a.m + 255) / 256);
    if (epi == kEpiHead) {
        hipLaunchKernelGGL((dense_kernel<1, kEpiHead>), dim3(gx, 1), dim3(256), 0, st, a);
    } else {
        if (a.n_pad % 128 != 0) return hipErrorInvalidValue;
        const unsigned gy = (unsigned)(a.n_pad / 128);
        if (epi == kEpiSwish)
            hipLaunchKernelGGL((dense_kernel<4, kEpiSwish>), dim3(gx, gy), dim3(256), 0, st, a);
        else if (epi == kEpiSwishAffine)
            hipLaunchKernelGGL((dense_kernel<4, kEpiSwishAffine>), dim3(gx, gy), dim3(256), 0, st, a);
        else
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_assemble(const AssembleArgs& a, hipStream_t st) {
    if (a.n == 0) return hipSuccess;
    int64_t blocks = (a.n + kAsmSites - 1) / kAsmSites;
    if (blocks > 256 * 32) blocks = 256 * 32;
    AssembleArgs k = a;
    // 16-byte stores when every output array starts on a 16-byte boundary (a tile's byte range starts at a
    // multiple of 64*W); single bytes otherwise
    k.words = ((reinterpret_cast<uintptr_t>(a.t21) | reinterpret_cast<uintptr_t>(a.t33) |
                reinterpret_cast<uintptr_t>(a.t51)) & 15) == 0;
    if ((reinterpret_cast<uintptr_t>(a.codes) & 3) == 0)
        hipLaunchKernelGGL(assemble_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, k);
    else
        hipLaunchKernelGGL(assemble_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, k);
    return hipGetLastError();
}

// Per-device kernel attributes (the LDS opt-in of the fp32 branch kernels) ahead of the first launch.
hipError_t prepare_kernels() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = branch_attr<21>(dev)) != hipSuccess) return e;
    if ((e = branch_attr<33>(dev)) != hipSuccess) return e;
    if ((e = branch_attr<51>(dev)) != hipSuccess) return e;
    if ((e = front_pair_attr(dev)) != hipSuccess) return e;
    return branch3_attr(dev);
}

}  // namespace dptm

PH_DEFINE_READER(dptm_debug_phase_cycles_f32)
