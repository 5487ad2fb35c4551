// "Pair" tiles of the fused training kernels (included by dptm_train_branch.hip behind its own helpers, and, for the
// shared pieces, by dptm_train_conv.hip): ONE site in FOUR wavefronts, for batches so small that a launch lasts as long as one wavefront's
// dependent chain (the reference trains with 64 sites per step, pred/train_PTM/config.py:56).  The site's rows fill two
// 32-row blocks; wavefronts w and w + 2 both hold block w & 1 and share its work: every row product is cut in two --
// by output tile (each wavefront 32 of the 64 output features), or by contraction half where the partial sums are
// needed anyway -- each wavefront takes two of the four attention heads (a head per half-wave) and half of every
// LDS image, and the halves meet through an LDS exchange: 16 or 32 floats per lane, double-buffered, one barrier.
// What the forward kernel keeps is laid out as in the other tiles (block w & 1 of the tile's four 32-row slots; the
// softmax statistics use all four slots, one head per lane), so the convolution kernels on either side do not change.
// (Included inside namespace dptm, behind dptm_pair.h: the pieces the scoring kernels share.)
#pragma once

// own groups -> a transposed image [feature][kLdsStride]; lane_base = image + 4 h kLdsStride + row of the lane
__device__ __forceinline__ void chain_store_t4(float* lane_base, int part, const f32x4 (&x4)[4]) {
    unsigned off = (unsigned)(32 * part * kLdsStride);
    DPTM_LOCAL_INDEX(off);
    lane_base += off;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) lane_base[(8 * g + e) * kLdsStride] = x4[g][e];
}

// Column sums of what THIS wavefront stored into a transposed image: its 32 features over its 32 rows (row0 = first
// row of its block), added to the workgroup's column sums `dst64` (64 floats of LDS) with 32 atomics.
__device__ __forceinline__ void colsum_own(const float* imgT, int row0, int part, int lane, float* dst64) {
    const int i = lane & 31, hh = lane >> 5;
    const f32x4* p = reinterpret_cast<const f32x4*>(imgT + (32 * part + i) * kLdsStride + row0 + 16 * hh);
    const f32x4 s = (p[0] + p[1]) + (p[2] + p[3]);
    const float v = xhalf_sum((s[0] + s[1]) + (s[2] + s[3]));
    if (hh == 0) atomicAdd(dst64 + 32 * part + i, v);
}

// inverted dropout of the own groups (the same element indices as dropout_chain)
__device__ __forceinline__ void dropout4(f32x4 (&x4)[4], int64_t r, int h, int part, unsigned seed, unsigned stream, unsigned thr,
                                         float scale) {
    if (thr == 0u) return;
    unsigned base = (unsigned)(r * 64) + 4 * h + 32 * part;
    DPTM_LOCAL_INDEX(base);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) x4[g][e] = hash_u32(seed, stream, base + 8 * g + e) >= thr ? x4[g][e] * scale : 0.f;
}

// acc[t] += contraction over the four feature groups g0 .. g0 + 3 of an image with K8 groups per output tile
template <int K8>
__device__ __forceinline__ void gemm_groups4(f32x16 (&acc)[2], const f32x4* __restrict__ wp, int g0, const f32x4 (&x4)[4], int lane) {
    DPTM_LOCAL_INDEX(lane);
    wp += g0 * 64;
    f32x4 a[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[u][t] = wp[(t * K8 + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t) mfma(acc[t], a[u][t][e], x4[u][e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// own groups of x1 = BN2(swish(a2)) + s for the lane's row (see x1_from_conv)
__device__ __forceinline__ void x1_from_conv4(const f32x4* a2w, const f32x4* sw, unsigned ln, const float* bnv, int h, int part,
                                              bool row_ok, f32x4 (&x4)[4]) {
    a2w += 4 * part * 64;
    sw += 4 * part * 64;
    bnv += 32 * part + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 a = (a2w + g * 64)[ln], sv = (sw + g * 64)[ln];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * g);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) x4[g][e] = row_ok ? fmaf(swishf(a[e]), sc[e], sh[e]) + sv[e] : 0.f;
    }
}

// LayerNormalization backward of the lane's row (see ln_backward); of the two images for the parameter gradients
// only the wavefront's own groups are stored
__device__ __forceinline__ void ln_backward_pair(const f32x4 (&dy)[8], const f32x4 (&xh)[8], float inv, const f32x4* __restrict__ g,
                                                 int h, int part, f32x4 (&dsum)[8], float* lane_a, float* lane_b) {
    DPTM_LOCAL_INDEX(h);
    float s1 = 0.f, s2 = 0.f;
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        t[u] = dy[u] * xh[u];
        dsum[u] = dy[u] * g[h * 8 + u];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += dsum[u][e];
            s2 = fmaf(dsum[u][e], xh[u][e], s2);
        }
    }
    f32x4 t4[4], d4[4];
    own_groups(t, part, t4);
    own_groups(dy, part, d4);
    chain_store_t4(lane_a, part, t4);
    chain_store_t4(lane_b, part, d4);
    s1 = xhalf_sum(s1) * (1.0f / 64.0f);
    s2 = xhalf_sum(s2) * (1.0f / 64.0f);
#pragma unroll
    for (int u = 0; u < 8; ++u) dsum[u] = inv * (dsum[u] - s1 - xh[u] * s2);
}

// One 64 x 64 weight-gradient unit of a pair tile: part_u (+)= X^T Y over the site's rows, X and Y transposed images
// [feature][kLdsStride]; wavefront w takes block w.  The caller has put a barrier between writing the images and this
// call; the call ends with one.  Rows at and beyond rows_valid hold zeros in at least one of the images.
__device__ __forceinline__ void dw_unit_pair(const float* imgXT, const float* imgYT, float* __restrict__ part_u, bool first, int lane,
                                             int wave, int rows_valid) {
    const int hh = lane >> 5, i = lane & 31;
    const int b = wave;
    const float* ap = imgXT + (32 * (b >> 1) + i) * kLdsStride + 4 * hh;
    const float* bp = imgYT + (32 * (b & 1) + i) * kLdsStride + 4 * hh;
    float* dst = part_u + b * 1024 + lane;
    f32x16 acc, prev;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = prev[rg] = 0.f;
    if (!first) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) prev[rg] = dst[rg * 64];
    }
    const int nq = (rows_valid + 7) >> 3;       // groups of 8 rows (at most 8)
    f32x4 ca = *reinterpret_cast<const f32x4*>(ap), cb = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll 2
    for (int g = 0; g < nq; ++g) {
        f32x4 na = ca, nb = cb;
        if (g + 1 < nq) {
            na = *reinterpret_cast<const f32x4*>(ap + 8 * (g + 1));
            nb = *reinterpret_cast<const f32x4*>(bp + 8 * (g + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) mfma(acc, ca[e], cb[e]);
        ca = na;
        cb = nb;
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) dst[rg * 64] = acc[rg] + prev[rg];
    __syncthreads();
}

// Two units behind ONE barrier (both pairs of images complete before the call): the two chains of matrix instructions
// interleave, and the second unit's LDS reads and partial-buffer traffic hide behind the first's products.
__device__ __forceinline__ void dw_unit_pair2(const float* x1, const float* y1, float* __restrict__ part1, const float* x2,
                                              const float* y2, float* __restrict__ part2, bool first, int lane, int wave,
                                              int rows_valid) {
    const int hh = lane >> 5, i = lane & 31;
    const int b = wave;
    const int ox = (32 * (b >> 1) + i) * kLdsStride + 4 * hh, oy = (32 * (b & 1) + i) * kLdsStride + 4 * hh;
    const float *ap1 = x1 + ox, *bp1 = y1 + oy, *ap2 = x2 + ox, *bp2 = y2 + oy;
    float* dst1 = part1 + b * 1024 + lane;
    float* dst2 = part2 + b * 1024 + lane;
    f32x16 acc1, acc2, prev1, prev2;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc1[rg] = acc2[rg] = prev1[rg] = prev2[rg] = 0.f;
    if (!first) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            prev1[rg] = dst1[rg * 64];
            prev2[rg] = dst2[rg * 64];
        }
    }
    const int nq = (rows_valid + 7) >> 3;
    f32x4 ca1 = *reinterpret_cast<const f32x4*>(ap1), cb1 = *reinterpret_cast<const f32x4*>(bp1);
    f32x4 ca2 = *reinterpret_cast<const f32x4*>(ap2), cb2 = *reinterpret_cast<const f32x4*>(bp2);
#pragma unroll 2
    for (int g = 0; g < nq; ++g) {
        f32x4 na1 = ca1, nb1 = cb1, na2 = ca2, nb2 = cb2;
        if (g + 1 < nq) {
            na1 = *reinterpret_cast<const f32x4*>(ap1 + 8 * (g + 1));
            nb1 = *reinterpret_cast<const f32x4*>(bp1 + 8 * (g + 1));
            na2 = *reinterpret_cast<const f32x4*>(ap2 + 8 * (g + 1));
            nb2 = *reinterpret_cast<const f32x4*>(bp2 + 8 * (g + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mfma(acc1, ca1[e], cb1[e]);
            mfma(acc2, ca2[e], cb2[e]);
        }
        ca1 = na1; cb1 = nb1; ca2 = na2; cb2 = nb2;
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        dst1[rg * 64] = acc1[rg] + prev1[rg];
        dst2[rg * 64] = acc2[rg] + prev2[rg];
    }
    __syncthreads();
}

// sum over the wavefront's 32 rows of its own 32 features of a ROW-major image (own_rows: first row of its block)
__device__ __forceinline__ float colsum_rows_own(const float* own_rows, int part, int lane) {
    const int i = lane & 31, hh = lane >> 5;
    const float* p = own_rows + 16 * hh * kLdsStride + 32 * part + i;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s0 += p[r * kLdsStride];
        s1 += p[(r + 1) * kLdsStride];
    }
    return xhalf_sum(s0 + s1);
}

#ifdef DPTM_PAIR_TB_BODIES      // the TransformerBlock bodies need the helpers of dptm_train_branch.hip

// What a small batch keeps for its backward pass fits the caches (64 sites: 15 MB over eight 4 MB L2s and the infinity
// cache behind them), and a workgroup meets its own rows again: plain stores and loads here, not the streaming ones.
#ifdef DPTM_EXP_PAIR_NT
__device__ __forceinline__ void pkeep_store(f32x4* p, const f32x4& v) { keep_store(p, v); }
__device__ __forceinline__ f32x4 pkeep_load(const f32x4* p) { return keep_load(p); }
#else
__device__ __forceinline__ void pkeep_store(f32x4* p, const f32x4& v) { *p = v; }
__device__ __forceinline__ f32x4 pkeep_load(const f32x4* p) { return *p; }
#endif

// ------------------------------------------------------------------------------------ forward

// LDS: K rows | V rows | exchange buffers | pooling scores, weights | BatchNorm 2 scale, shift
static size_t fwd_pair_lds() { return ((size_t)2 * kPairImg + kPairXFloats + 2 * 64 + 128) * sizeof(float); }

template <int W>
__device__ __forceinline__ void tb_fwd_pair_body(const TbFwdArgs& p) {
    static_assert(W <= 64, "a site fills at most two 32-row blocks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    PairX px{reinterpret_cast<f32x4*>(bufB + kPairImg), 0};
    float* score = bufB + kPairImg + kPairXFloats;   // [64] exp(tanh score)
    float* alpha_s = score + 64;                     // [64] pooling weights
    float* bnv = alpha_s + 64;                       // [scale | shift] of the second BatchNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 1, part = wave >> 1;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int row = pw * 32 + j;
    float* mine_a = bufA + row * kLdsStride;
    float* mine_b = bufB + row * kLdsStride;
    const bool row_ok = row < W;
    const int head_off = 16 * (2 * part + h);
    bn_prepare(p.bn2, bnv, tid, blockIdx.x == 0);
    __syncthreads();
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t grow = tile * W + row;
        const int64_t ct = ((tile * 4 + pw) * 8) * 64 + lane;
        const int64_t co = ct + 4 * part * 64;              // the wavefront's own groups
        const int64_t cl = (tile * 4 + pw) * 64 + lane;
        const int64_t clw = (tile * 4 + wave) * 64 + lane;
        f32x4 x1[8];
        x1_from_conv(p.cs, ct, bnv, h, row_ok, x1);
#pragma unroll 1
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4* T = p.P + kOffTb + blk * kTbSize;
            const f32x4* tv = T + kTbVec;
            const f32x4* Th = T + part * 8 * 64;            // output tile `part` of a packed 64 x 64 matrix
            const f32x4* tvh = tv + 4 * part;
            const TbSave& sv = p.sv[blk];
            __syncthreads();   // every wavefront is done with the previous contents of bufA / bufB
            {
                f32x16 acc[1];
                f32x4 t[4];
                acc_init_bias<1>(acc, tvh + kVBk * kVec64, h, 8);
                gemm_regs<8, 1, 8>(acc, Th + kTbWk, x1, lane);
                acc_to_x<1>(acc, t);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.k[co + g * 64], t[g]);
                chain_store4(mine_a, h, part, t);
                acc_init_bias<1>(acc, tvh + kVBv * kVec64, h, 8);
                gemm_regs<8, 1, 8>(acc, Th + kTbWv, x1, lane);
                acc_to_x<1>(acc, t);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.v[co + g * 64], t[g]);
                chain_store4(mine_b, h, part, t);
            }
            float q1[1][16];
            {
                f32x16 acc[1];
                f32x4 t[4];
                acc_init_bias<1>(acc, tvh + kVBq * kVec64, h, 8);
                gemm_regs<8, 1, 8>(acc, Th + kTbWq, x1, lane);
                acc_to_x<1>(acc, t);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.q[co + g * 64], t[g]);
                x4_to_head(t, q1[0], kQScale);
            }
            __syncthreads();   // K and V of the site are in LDS
            f32x4 o[8];
            {
                float o1[1][16], m1[1], r1[1];
                attend<W, 1>(bufA + head_off, bufB + head_off, q1, o1, m1, r1);
                sv.stat[clw] = f32x4{m1[0], r1[0], 0.f, 0.f};
                f32x4 o4[4];
                head_to_x4(o1[0], o4, 1.0f);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.o[co + g * 64], o4[g]);
                pair_gather(px, wave, lane, part, o4, o);
            }
            // output projection, dropout, residual, LayerNorm 1
            f32x4 hres[8];
            f32x2 iv;
            {
                f32x16 acc[1];
                acc_init_bias<1>(acc, tvh + kVBo * kVec64, h, 8);
                gemm_regs<8, 1, 8>(acc, Th + kTbWo, o, lane);
                f32x4 t4[4], r4[4];
                acc_to_x<1>(acc, t4);
                dropout4(t4, grow, h, part, seed, p.stream_base + 2 * blk, p.thr, p.scale);
                own_groups(x1, part, r4);
#pragma unroll
                for (int g = 0; g < 4; ++g) t4[g] += r4[g];
                f32x4 t[8];
                pair_gather(px, wave, lane, part, t4, t);
                iv[0] = ln_normalise(t);
                own_groups(t, part, t4);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.xh1[co + g * 64], t4[g]);
                ln_affine(t, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h, hres);
            }
            // FFN: this wavefront's two of the four hidden chunks; the partial sums meet
            {
                f32x16 acc_o[2];
                if (part == 0)
                    acc_init_bias<2>(acc_o, tv + kVB2 * kVec64, h, 8);
                else
                    acc_zero<2>(acc_o);
#pragma unroll 1
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * part + cc;
                    f32x16 acc_h[2];
                    acc_init_bias<2>(acc_h, tv + 9 * kVec64 + 8 * c, h, 32);
                    gemm_regs<8, 2, 2>(acc_h, T + kTbF1 + (2 * c) * 8 * 64, hres, lane);
                    f32x4 hid[8];
                    acc_to_x<2>(acc_h, hid);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) hid[u][e] = swishf(hid[u][e]);
                    gemm_k256_chunk(acc_o, T + kTbF2, c, hid, lane);
                }
                f32x4 t[8];
                acc_to_x<2>(acc_o, t);
                pair_sum(px, wave, lane, t);
                dropout_chain(t, grow, h, seed, p.stream_base + 2 * blk + 1, p.thr, p.scale);
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] += hres[u];
                iv[1] = ln_normalise(t);
                f32x4 t4[4];
                own_groups(t, part, t4);
#pragma unroll
                for (int g = 0; g < 4; ++g) pkeep_store(&sv.xh2[co + g * 64], t4[g]);
                ln_affine(t, tv + kVLn2G * kVec64, tv + kVLn2B * kVec64, h, x1);
            }
            if (part == 0) sv.inv[cl] = iv;
        }
        // ---- attention pooling (model.py:176-182): a = tanh(x.w + b), alpha = softmax over the site's positions
        {
            const f32x4* pv = p.P + kOffPoolV;
            float dotp = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wv = pv[h * 8 + u];
#pragma unroll
                for (int e = 0; e < 4; ++e) dotp = fmaf(x1[u][e], wv[e], dotp);
            }
            const float a = tanhf(xhalf_sum(dotp) + p.P[kOffPoolB][0]);
            const float ew = __expf(a);     // |a| <= 1: no running maximum needed
            __syncthreads();                // all attention reads of bufA are finished
            f32x4 x4[4];
            own_groups(x1, part, x4);
            chain_store4(mine_a, h, part, x4);
            if (h == 0 && part == 0) score[row] = ew;
            __syncthreads();
            float den = 0.f;
#pragma unroll 3
            for (int t = 0; t < W; ++t) den += score[t];
            const float al = ew / den;
            if (h == 0 && part == 0) alpha_s[row] = al;
            if (part == 0) p.pool[cl] = f32x2{a, al};
            __syncthreads();
            if (tid < 64) {
                const float* xp = bufA + tid;
                float acc = 0.f;
#pragma unroll 3
                for (int t = 0; t < W; ++t) acc = fmaf(alpha_s[t], xp[t * kLdsStride], acc);
                p.out[tile * p.ldo + p.col_off + tid] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward

// LDS: four images | exchange buffers | per-row arrays | column sums | BatchNorm 2 scale, shift | tile slot.  (Four images,
// so that what several weight-gradient units share -- d(f2) and h over the four FFN chunks, the block input over dWq / dWk /
// dWv -- is stored once, and two units at a time run behind one barrier.)
constexpr int kPairXreg = 1024;
static size_t bwd_pair_lds() { return ((size_t)4 * kPairImg + kPairXFloats + kPairXreg + kTbVecs * 64 + 128 + 64) * sizeof(float); }

template <int W>
__device__ __forceinline__ void tb_bwd_pair_body(const TbBwdArgs& p) {
    static_assert(W <= 64, "a site fills at most two 32-row blocks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    float* bufC = bufB + kPairImg;
    float* bufD = bufC + kPairImg;
    PairX px{reinterpret_cast<f32x4*>(bufD + kPairImg), 0};
    float* xreg = bufD + kPairImg + kPairXFloats;   // pooling scores [64]; softmax statistics 3 x [64][4]
    float* vacc = xreg + kPairXreg;                 // [kTbVecs][64] column sums of this workgroup
    float* bnv = vacc + kTbVecs * 64;               // [scale | shift] of the second BatchNorm
    int* tile_slot = reinterpret_cast<int*>(bnv + 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 1, part = wave >> 1;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int row = pw * 32 + j, row0 = pw * 32;
    float* mine_a = bufA + row * kLdsStride;
    float* mine_b = bufB + row * kLdsStride;
    float* lane_a = bufA + 4 * h * kLdsStride + row;       // the lane's place in a transposed image
    float* lane_b = bufB + 4 * h * kLdsStride + row;
    float* lane_c = bufC + 4 * h * kLdsStride + row;
    float* lane_d = bufD + 4 * h * kLdsStride + row;
    float* mine_c = bufC + row * kLdsStride;
    float* mine_d = bufD + row * kLdsStride;
    const bool row_ok = row < W;
    const int head_off = 16 * (2 * part + h);
    float* pbuf = p.part + (size_t)blockIdx.x * kTbPartStride;
    for (int i = tid; i < kTbVecs * 64; i += blockDim.x) vacc[i] = 0.f;
    bn_prepare(p.bn2, bnv, tid, false);
    __syncthreads();
    bool first = true;
    int tile_round = 0;
    PH_DECL
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue, tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue, tile_slot, tile_round)) {
        const int64_t grow = tile * W + row;
        const int64_t tb = (int64_t)__builtin_amdgcn_readfirstlane((int)tile) * 4;
        const int64_t tw = tb + pw;        // chain-tiled slab of this wavefront's row block
        const int64_t tww = tb + wave;     // per-wavefront slot (softmax statistics)
        unsigned ln = (unsigned)lane;
#define AT4(arr, u) ((arr) + tw * 512 + (u) * 64)[local_index(ln)]
#define AT4O(arr, g) ((arr) + tw * 512 + (4 * part + (g)) * 64)[local_index(ln)]
#define AT1(arr) ((arr) + tw * 64)[local_index(ln)]
        f32x4 dy[8];   // gradient w.r.t. the output of the block in hand
        __syncthreads();
        // ---- attention pooling backward: dx_t = alpha_t dout + dz_t w, dz_t = alpha_t (dalpha_t - sum alpha dalpha)(1 - a_t^2)
        {
            const f32x4* tv1 = p.P + kOffTb + kTbSize + kTbVec;
            const f32x4* pv = p.P + kOffPoolV;
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = pkeep_load(&AT4(p.sv[1].xh2, u));
            ln_affine(x, tv1 + kVLn2G * kVec64, tv1 + kVLn2B * kVec64, h, x);
            const f32x2 pa = AT1(p.pool);
            const float* dsite = p.dout + tile * p.ldd + p.col_off;
            if (p.up_da) {
                // the site's 64 pooled-row gradients from the layer above: four threads per feature, a quarter of the sum each
                float* dsl = xreg + kPairXreg - 64;
                const int f = tid >> 2, q = tid & 3, quarter = p.up_n >> 2;
                const f32x4* wr = reinterpret_cast<const f32x4*>(p.up_w + (int64_t)(p.col_off + f) * p.up_n + quarter * q);
                const f32x4* dr = reinterpret_cast<const f32x4*>(p.up_da + tile * p.up_n + quarter * q);
                f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
                for (int i = 0; i < (quarter >> 2); ++i) a4 += wr[i] * dr[i];
                float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (p.up_thr) {
                    const unsigned idx = (unsigned)(tile * p.up_cols) + (unsigned)(p.col_off + f);
                    acc = hash_u32(seed, p.up_stream, idx) >= p.up_thr ? acc * p.up_scale : 0.f;
                }
                if (q == 0) dsl[f] = acc;
                __syncthreads();
                dsite = dsl;
            }
            f32x4 ds[8];
            float dal = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ds[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row_ok) ds[u] = *reinterpret_cast<const f32x4*>(dsite + 8 * u + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) dal = fmaf(ds[u][e], x[u][e], dal);
            }
            dal = xhalf_sum(dal);
            if (h == 0 && part == 0) xreg[row] = row_ok ? pa[1] * dal : 0.f;
            __syncthreads();
            float dot = 0.f;
#pragma unroll 3
            for (int t = 0; t < W; ++t) dot += xreg[t];
            const float dz = row_ok ? pa[1] * (dal - dot) * (1.0f - pa[0] * pa[0]) : 0.f;
            const float al = row_ok ? pa[1] : 0.f;
            f32x4 t[8], t4[4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dy[u] = al * ds[u] + dz * pv[h * 8 + u];
                t[u] = dz * x[u];
            }
            own_groups(t, part, t4);
            chain_store_t4(lane_a, part, t4);
            wave_fence();
            colsum_own(bufA, row0, part, lane, vacc + kSPoolW * 64);
            if (part == 0) {
                float dzs = h == 0 ? dz : 0.f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dzs += __shfl_xor(dzs, o);
                if (lane == 0) atomicAdd(&vacc[kSPoolB * 64], dzs);
            }
            wave_fence();
            __syncthreads();   // the scores in xreg have been read
        }
        PH_MARK(0)   // pooling backward
#pragma unroll 1
        for (int blk = 1; blk >= 0; --blk) {
            const f32x4* T = p.P + kOffTb + blk * kTbSize;
            const f32x4* tv = T + kTbVec;
            const f32x4* TT = p.PT + blk * kBtSize;
            const TbSave& sv = p.sv[blk];
            float* pu = pbuf + (size_t)blk * 12 * 4096;
            float* va = vacc + blk * kSlotsPerBlock * 64;
            const f32x2 iv = AT1(sv.inv);
            // ---- LayerNorm 2 backward: dy -> dsum = d(h + f2)
            f32x4 dsum[8];
            {
                f32x4 xh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xh[u] = pkeep_load(&AT4(sv.xh2, u));
                ln_backward_pair(dy, xh, iv[1], tv + kVLn2G * kVec64, h, part, dsum, lane_a, lane_b);
                wave_fence();
                colsum_own(bufA, row0, part, lane, va + kSG2 * 64);
                colsum_own(bufB, row0, part, lane, va + kSB2 * 64);
                wave_fence();
            }
            PH_MARK(1)   // LayerNorm 2 backward
            f32x4 df2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) df2[u] = dsum[u];
            dropout_chain(df2, grow, h, seed, p.stream_base + 2 * blk + 1, p.thr, p.scale);
            f32x4 hx[8];
            {
                f32x4 xh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xh[u] = pkeep_load(&AT4(sv.xh1, u));
                ln_affine(xh, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h, hx);
            }
            // ---- feed-forward layers backward.  Every hidden chunk is cut by output tile: this wavefront rebuilds 32 of the
            // chunk's 64 hidden units, takes their gradient, and contracts them into its partial sum of d(h).
            f32x16 acc_dh[2];
            if (part == 0)
                x_to_acc(dsum, acc_dh);
            else
                acc_zero<2>(acc_dh);
            {
                // what the chunks' units share, stored once: d(f2) -> image C, h -> image D
                f32x4 df2o[4], hxo[4];
                own_groups(df2, part, df2o);
                own_groups(hx, part, hxo);
                chain_store_t4(lane_c, part, df2o);
                chain_store_t4(lane_d, part, hxo);
                wave_fence();
                colsum_own(bufC, row0, part, lane, va + kSFb2 * 64);
            }
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x4 d1[4];
                {
                    f32x4 g[4];
                    {
                        f32x16 acc_f[1];
                        acc_init_bias<1>(acc_f, tv + 9 * kVec64 + 8 * c + 4 * part, h, 32);
                        gemm_regs<8, 1, 8>(acc_f, T + kTbF1 + (2 * c + part) * 8 * 64, hx, lane);   // f1_c = h W1[:, my 32 units] + b1
                        f32x4 u_[4];
                        acc_to_x<1>(acc_f, u_);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = u_[u][e];
                                const float sg = fast_rcp(1.0f + __expf(-a));
                                u_[u][e] = a * sg;                                   // swish(f1_c)
                                g[u][e] = sg * (1.0f + a * (1.0f - sg));             // swish'(f1_c)
                            }
                        chain_store_t4(lane_a, part, u_);
                    }
                    f32x16 acc_u[1];
                    acc_zero<1>(acc_u);
                    gemm_regs<8, 1, 8>(acc_u, TT + kBtW2T + (2 * c + part) * 8 * 64, df2, lane);   // d(u_c) = d(f2) W2[my units, :]^T
                    acc_to_x<1>(acc_u, d1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) d1[u] *= g[u];                                     // d(f1_c)
                }
                chain_store_t4(lane_b, part, d1);
                wave_fence();
                colsum_own(bufB, row0, part, lane, va + (kSFb1 + c) * 64);
                gemm_groups4<32>(acc_dh, TT + kBtW1T, 8 * c + 4 * part, d1, lane);      // d(h) += d(f1_c) W1[:, my units]^T
                PH_MARK(2)   // feed-forward row products
                __syncthreads();
                // dW2[chunk, :] += u_c^T d(f2) and dW1[:, chunk] += h^T d(f1_c)
                dw_unit_pair2(bufA, bufC, pu + (4 + c) * 4096, bufD, bufB, pu + c * 4096, first, lane, wave, W);
                PH_MARK(3)   // feed-forward weight gradients
            }
            // ---- LayerNorm 1 backward: d(h) -> dsum = d(x + ao)
            {
                f32x4 dh[8], xh[8];
                acc_to_x<2>(acc_dh, dh);
#pragma unroll
                for (int u = 0; u < 8; ++u) xh[u] = pkeep_load(&AT4(sv.xh1, u));
                PH_MARK(2)
                pair_sum(px, wave, lane, dh);
                PH_MARK(9)   // exchanges between the two wavefronts of a pair
                ln_backward_pair(dh, xh, iv[0], tv + kVLn1G * kVec64, h, part, dsum, lane_a, lane_b);
                wave_fence();
                colsum_own(bufA, row0, part, lane, va + kSG1 * 64);
                colsum_own(bufB, row0, part, lane, va + kSB1 * 64);
                wave_fence();
            }
            PH_MARK(4)   // LayerNorm 1 backward
            // ---- output projection backward: this wavefront's output tile of d(o) holds exactly its two heads
            f32x4 dO[4];
            float D;           // D = dO . o of the lane's head (= sum_t p_t dO.v_t)
            {
                f32x4 dao[8], dao4[4], o4[4];
#pragma unroll
                for (int u = 0; u < 8; ++u) dao[u] = dsum[u];
#pragma unroll
                for (int g = 0; g < 4; ++g) o4[g] = pkeep_load(&AT4O(sv.o, g));
                dropout_chain(dao, grow, h, seed, p.stream_base + 2 * blk, p.thr, p.scale);
                own_groups(dao, part, dao4);
                // (o -> image C, d(ao) -> image D: their unit runs with dWq's, behind the query pass)
                chain_store_t4(lane_c, part, o4);
                chain_store_t4(lane_d, part, dao4);
                wave_fence();
                colsum_own(bufD, row0, part, lane, va + kSBo * 64);
                f32x16 acc[1];
                acc_zero<1>(acc);
                gemm_regs<8, 1, 8>(acc, TT + kBtWoT + part * 8 * 64, dao, lane);            // d(o) = d(ao) Wo^T, my tile
                acc_to_x<1>(acc, dO);
                float s01 = 0.f, s23 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s01 = fmaf(dO[0][e], o4[0][e], fmaf(dO[1][e], o4[1][e], s01));
                    s23 = fmaf(dO[2][e], o4[2][e], fmaf(dO[3][e], o4[3][e], s23));
                }
                s01 = xhalf_sum(s01);
                s23 = xhalf_sum(s23);
                D = h ? s23 : s01;
            }
            PH_MARK(5)   // output projection backward
            const f32x4 st = ((sv.stat) + tww * 64)[local_index(ln)];
            f32x16 acc_dx[2];   // d(x) = dsum + dq Wq^T + dk Wk^T + dv Wv^T: partial sums over this wavefront's groups
            if (part == 0)
                x_to_acc(dsum, acc_dx);
            else
                acc_zero<2>(acc_dx);
            // ---- attention backward, lane = query row: dq
            {
                f32x4 dq[4];
                {
                    f32x4 t[4], t2[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        t[g] = pkeep_load(&AT4O(sv.k, g));
                        t2[g] = pkeep_load(&AT4O(sv.v, g));
                    }
                    __syncthreads();   // the other wavefronts have summed their columns of images A / B (LayerNorm 1)
                    chain_store4(mine_a, h, part, t);
                    chain_store4(mine_b, h, part, t2);
                    float qs[16], doh[16], gh[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) t[g] = pkeep_load(&AT4O(sv.q, g));
                    x4_to_head(t, qs, kQScale);
                    x4_to_head(dO, doh, 1.0f);
                    __syncthreads();   // K and V of the site are in LDS
                    attn_bwd_q<W>(bufA + head_off, bufB + head_off, qs, doh, st[0], st[1], D, gh);
                    head_to_x4(gh, dq, row_ok ? 0.25f : 0.f);
                }
                PH_MARK(6)   // attention backward, query rows
                __syncthreads();   // K and V have been read
                PH_MARK(10)  // barrier after a vector-unit phase
                {
                    f32x4 xin[4];
                    if (blk == 0) {
                        x1_from_conv4(p.cs.a2 + tw * 512, p.cs.s + tw * 512, local_index(ln), bnv, h, part, row_ok, xin);
                    } else {
                        const f32x4* tv0 = p.P + kOffTb + kTbVec;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            xin[g] = pkeep_load(&AT4O(p.sv[0].xh2, g)) * tv0[kVLn2G * kVec64 + h * 8 + 4 * part + g] +
                                     tv0[kVLn2B * kVec64 + h * 8 + 4 * part + g];
                    }
                    chain_store_t4(lane_a, part, xin);
                }
                chain_store_t4(lane_b, part, dq);
                wave_fence();
                colsum_own(bufB, row0, part, lane, va + kSBq * 64);
                __syncthreads();
                // dWo += o^T d(ao) and dWq += x^T dq; the block input stays in image A for dWk and dWv
                dw_unit_pair2(bufC, bufD, pu + 8 * 4096, bufA, bufB, pu + 9 * 4096, first, lane, wave, W);
                gemm_groups4<8>(acc_dx, TT + kBtWqT, 4 * part, dq, lane);
            }
            PH_MARK(7)   // Q / K / V projections backward
            // ---- attention backward, lane = key row: dk, dv
            {
                f32x4 dk[4], dv[4];
                {
                    f32x4 t[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) t[g] = pkeep_load(&AT4O(sv.q, g)) * kQScale;
                    chain_store4(mine_c, h, part, t);
                    chain_store4(mine_d, h, part, dO);
                    xreg[64 + row * 4 + 2 * part + h] = st[0];
                    xreg[64 + 256 + row * 4 + 2 * part + h] = st[1];
                    xreg[64 + 512 + row * 4 + 2 * part + h] = D;
                    float kh[16], vh[16], gk[16], gv[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) t[g] = pkeep_load(&AT4O(sv.k, g));
                    x4_to_head(t, kh, 1.0f);
#pragma unroll
                    for (int g = 0; g < 4; ++g) t[g] = pkeep_load(&AT4O(sv.v, g));
                    x4_to_head(t, vh, 1.0f);
                    __syncthreads();   // scaled Q, dO and the statistics of the site are in LDS
                    attn_bwd_kv<W>(bufC + head_off, bufD + head_off, xreg + 64 + 2 * part + h, 256, kh, vh, gk, gv);
                    head_to_x4(gk, dk, row_ok ? kLn2 : 0.f);
                    head_to_x4(gv, dv, row_ok ? 1.0f : 0.f);
                }
                PH_MARK(8)   // attention backward, key rows
                __syncthreads();   // Q and dO have been read
                PH_MARK(10)
                chain_store_t4(lane_b, part, dk);
                chain_store_t4(lane_c, part, dv);
                wave_fence();
                colsum_own(bufB, row0, part, lane, va + kSBk * 64);
                colsum_own(bufC, row0, part, lane, va + kSBv * 64);
                __syncthreads();
                // dWk += x^T dk and dWv += x^T dv
                dw_unit_pair2(bufA, bufB, pu + 10 * 4096, bufA, bufC, pu + 11 * 4096, first, lane, wave, W);
                gemm_groups4<8>(acc_dx, TT + kBtWkT, 4 * part, dk, lane);
                gemm_groups4<8>(acc_dx, TT + kBtWvT, 4 * part, dv, lane);
            }
            acc_to_x<2>(acc_dx, dy);
            PH_MARK(7)
            pair_sum(px, wave, lane, dy);
            PH_MARK(9)
        }
        // d(x1) leaves chain-tiled; its column sums (with swish(a2)) feed the second BatchNorm's backward pass
        {
            f32x4 d4[4], t4[4];
            own_groups(dy, part, d4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                AT4O(p.dx1, g) = d4[g];
                const f32x4 a = AT4O(p.cs.a2, g);
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[g][e] = d4[g][e] * swishf(a[e]);
            }
            // the tile's two other 32-row slots are padding: the convolution kernels expect zeros there
            const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) (p.dx1 + (tb + 2 + pw) * 512 + (4 * part + g) * 64)[local_index(ln)] = zero4;
            chain_store_t4(lane_a, part, d4);
            chain_store_t4(lane_b, part, t4);
            wave_fence();
            colsum_own(bufA, row0, part, lane, vacc + kSBn2A * 64);
            colsum_own(bufB, row0, part, lane, vacc + kSBn2B * 64);
            wave_fence();
        }
        first = false;
#undef AT4
#undef AT4O
#undef AT1
    }
    __syncthreads();
    // a workgroup that was handed no tile still owns a partial buffer that the summing launch reads
    if (first) zero_floats(pbuf, kTbUnits * 4096);
    for (int i = tid; i < kTbVecs * 64; i += blockDim.x) {
        const int slot = i >> 6;
        if (slot == kSBn2A)
            atomicAdd(p.bn2_sums + (i & 63), vacc[i]);
        else if (slot == kSBn2B)
            atomicAdd(p.bn2_sums + 64 + (i & 63), vacc[i]);
        else
            pbuf[kTbUnits * 4096 + i] = vacc[i];
    }
    PH_FLUSH
}

#endif  // DPTM_PAIR_TB_BODIES
