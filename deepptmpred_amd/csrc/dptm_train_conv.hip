// Fused embedding + residual convolution block of the training step (gfx950; pred/train_PTM/model.py:129-141,
// 157-169).  BatchNormalization in training mode needs the statistics of the WHOLE batch between its input and its
// output, so the block is cut at its two BatchNorms: two forward kernels here (the third piece, BN2 + shortcut add,
// is the prologue of the TransformerBlock kernel) and two backward kernels behind the TransformerBlock backward
// kernel, each with the tiling, the chain layout and the weight-gradient units of dptm_train_tile.h.
//
//   conv_a_fwd   tokens -> x0 = spatial_dropout(E[tok] + P[pos]) (LDS only) -> s = conv1(x0), a1 = conv3a(x0);
//                column sums of swish(a1) for BN1
//   conv_b_fwd   a1 -> z1 = spatial_dropout(BN1(swish(a1))) (LDS only) -> a2 = conv3b(z1); column sums of swish(a2)
//   conv_b_bwd   d(x1) -> BN2, swish, conv3b backward -> d(z1) behind dropout, conv3b / BN2 gradients, sums for BN1
//   conv_a_bwd   d(z1), d(x1) -> BN1, swish, conv3a and shortcut backward, embedding-table gradients
//
// A 3-tap 'same' convolution is three products over rows shifted by -1 / 0 / +1 inside a site; the neighbour rows
// come from the LDS image of the tile, masked at the site edges (Keras pads with zeros per window).
#include <hip/hip_runtime.h>

#define DPTM_KEEP_INDEX_LOCAL 1   // see dptm_mfma.h: table addresses are formed where they are used

#include "dptm_layout.h"
#include "dptm_mfma.h"
#include "dptm_pair.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"
#include "dptm_train_tile.h"

namespace dptm {

#include "dptm_train_pair.h"

namespace {

// acc += w0^T x[r0] + w1^T x[r1] + w2^T x[r2] for three packed 64 x 64 matrices and three rows of an LDS image
// (pointers already at the lane's half: + 4h); rows r0 / r2 count only where ok0 / ok2 (pass any valid address otherwise)
// (NT output tiles from tile 0 of the matrices on: a caller that wants one tile passes the matrices at that tile)
template <int NT>
__device__ __forceinline__ void taps3(f32x16 (&acc)[NT], const f32x4* __restrict__ w0, const f32x4* __restrict__ w1,
                                      const f32x4* __restrict__ w2, const float* r0, const float* r1, const float* r2, bool ok0,
                                      bool ok2, int lane) {
    f32x4 a0[2][NT], a1[2][NT], a2[2][NT], b0[2], b1[2], b2[2];
    auto fetch = [&](int u, int s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a0[s][t] = w0[(t * 8 + u) * 64 + lane];
            a1[s][t] = w1[(t * 8 + u) * 64 + lane];
            a2[s][t] = w2[(t * 8 + u) * 64 + lane];
        }
        b0[s] = *reinterpret_cast<const f32x4*>(r0 + 8 * u);
        b1[s] = *reinterpret_cast<const f32x4*>(r1 + 8 * u);
        b2[s] = *reinterpret_cast<const f32x4*>(r2 + 8 * u);
    };
    fetch(0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int s = u & 1;
        if (u + 1 < 8) fetch(u + 1, s ^ 1);
        if (!ok0) b0[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!ok2) b2[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mfma(acc[t], a0[s][t][e], b0[s][e]);
                mfma(acc[t], a1[s][t][e], b1[s][e]);
                mfma(acc[t], a2[s][t][e], b2[s][e]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// x0 = spatial_dropout(E[tok] + P[pos]) for the rows of a tile into an LDS image; zero rows behind the last site
template <int W>
__device__ __forceinline__ void embed_tile(const ConvArgs& p, float* img, int rows_tile, int rows_valid, int64_t site0, unsigned seed) {
    for (int idx = threadIdx.x; idx < rows_tile * 16; idx += blockDim.x) {
        const int r = idx >> 4, c4 = idx & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows_valid) {
            const int s = r / W, pp = r - s * W;
            int t = p.tok[(site0 + s) * W + pp];
            t = t > kCodes - 1 ? kCodes - 1 : t;
            v = p.P[kOffEmb + t * 16 + c4] + p.P[kOffPos + pp * 16 + c4];
            if (p.thr_embed) {
                const unsigned base = (unsigned)((site0 + s) * 64) + 4 * c4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hash_u32(seed, p.stream_embed, base + e) >= p.thr_embed ? v[e] * p.scale_embed : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(img + r * kLdsStride + c4 * 4) = v;
    }
}

// The weight-gradient unit of dw_unit for one tap of a convolution: part_u (+)= shift(X)^T Y where row r of Y
// meets row r + xshift of X when that row lies in the same site (positions 0..W-1), zero otherwise.
template <int W, bool SHORT>
__device__ __forceinline__ void dw_unit_tap(const float* imgX, const float* imgY, int xshift, int NW, float* xchg,
                                            float* __restrict__ part_u, bool first, int lane, int wave, int rows_valid) {
    const int hh = lane >> 5, i = lane & 31;
    const int nblk = NW >= 4 ? 1 : 4 / NW;                 // blocks per wavefront
    const int sp = NW >= 4 ? wave >> 2 : 0;                // row half (NW = 8)
    // rows this wavefront contracts over; a four-wavefront tile that is mostly padding (one site, the shape small
    // batches take) stops at its last valid row: the rows behind it hold zeros in both images
    const int rows = SHORT && NW == 4 && rows_valid <= 96 ? rows_valid + (rows_valid & 1) : NW >= 4 ? 128 : NW * 32;
    for (int bi = 0; bi < nblk; ++bi) {
        const int b = NW >= 4 ? (wave & 3) : wave * nblk + bi;
        const int r0 = sp * 128 + hh;
        const float* ap = imgX + r0 * kLdsStride + 32 * (b >> 1) + i;
        const float* bp = imgY + r0 * kLdsStride + 32 * (b & 1) + i;
        float* dst = part_u + b * 1024 + lane;
        f32x16 acc;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
        if (sp == 0 && !first) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[rg] = dst[rg * 64];
        }
        int pos = r0 % W;                                   // position of row r0 + 2q inside its site
#pragma unroll 4
        for (int q = 0; q < rows / 2; ++q) {
            const bool ok = xshift < 0 ? pos > 0 : pos < W - 1;
            const float a = ok ? ap[(2 * q + xshift) * kLdsStride] : 0.f;
            mfma(acc, a, bp[2 * q * kLdsStride]);
            pos += 2;
            pos = pos >= W ? pos - W : pos;
        }
        if (NW == 8) {
            if (sp == 1) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) xchg[(b * 16 + rg) * 64 + lane] = acc[rg];
            }
            __syncthreads();
            if (sp == 0) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) acc[rg] += xchg[(b * 16 + rg) * 64 + lane];
            }
        }
        if (sp == 0) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) dst[rg * 64] = acc[rg];
        }
    }
    if (NW != 8) __syncthreads();
}

// BatchNormalization backward coefficients from the batch sums (threads 0..63): with M rows, mean, inv = 1/sigma,
//   dy = gamma*inv * (dz - sum_dz/M - xhat * sum_dz_xhat/M),  xhat = (y - mean)*inv,  sum_dz_xhat = inv*(sum_dz_y - mean*sum_dz)
// bb[0..63] = mean, [64..] = inv, [128..] = gamma*inv, [192..] = sum_dz/M, [256..] = sum_dz_xhat/M.
// The first workgroup also adds this replica's share to the gamma / beta gradients.
__device__ __forceinline__ void bn_bwd_prepare(const BnBwd& bn, float* bb, int tid, bool first_wg) {
    if (tid < 64) {
        const float mu = bn.stats[tid] * bn.inv_rows;
        float var = bn.stats[64 + tid] * bn.inv_rows - mu * mu;
        var = var < 0.f ? 0.f : var;
        const float inv = rsqrtf(var + 1e-3f);
        bb[tid] = mu;
        bb[64 + tid] = inv;
        bb[128 + tid] = bn.gamma[tid] * inv;
        bb[192 + tid] = bn.sums[tid] * bn.inv_rows;
        bb[256 + tid] = inv * (bn.sums[64 + tid] - mu * bn.sums[tid]) * bn.inv_rows;
        if (first_wg) {
            atomicAdd(bn.dgamma + tid, inv * (bn.own[64 + tid] - mu * bn.own[tid]));
            atomicAdd(bn.dbeta + tid, bn.own[tid]);
        }
    }
}

// for the lane's row: a (pre-activation), dz -> d(a) = BN backward, then swish backward; zero on padding rows
__device__ __forceinline__ void bn_swish_backward(const f32x4 (&a)[8], const f32x4 (&dz)[8], const float* bb, int h, bool row_ok,
                                                  f32x4 (&da)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = 8 * u + 4 * h;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(bb + c), inv = *reinterpret_cast<const f32x4*>(bb + 64 + c);
        const f32x4 gi = *reinterpret_cast<const f32x4*>(bb + 128 + c), s1 = *reinterpret_cast<const f32x4*>(bb + 192 + c);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(bb + 256 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = a[u][e];
            const float sg = fast_rcp(1.0f + __expf(-v));
            const float xh = (v * sg - mu[e]) * inv[e];
            const float dy = gi[e] * (dz[u][e] - s1[e] - xh * s2[e]);
            da[u][e] = row_ok ? dy * (sg * (1.0f + v * (1.0f - sg))) : 0.f;
        }
    }
}

constexpr int kConvLdsExtra = kXreg + 768;   // exchange buffer, coefficient vectors

}  // namespace

// ------------------------------------------------------------------------------------ forward

template <int W>
__device__ __forceinline__ void conv_a_fwd_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    float* bufA = lds;
    float* bufB = bufA + rows_tile * kLdsStride;
    float* sacc = bufB + rows_tile * kLdsStride;     // [sum | sum of squares] of swish(a1), this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* own_b = bufB + wave * 32 * kLdsStride;
    float* mine_b = own_b + j * kLdsStride;
    for (int i = tid; i < 128; i += blockDim.x) sacc[i] = 0.f;   // (a one-wavefront tile has 64 threads)
    const f32x4* cv = p.P + kOffConvV;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int pos = row_ok ? row % W : 1;
        const bool tap0_ok = row_ok && pos > 0, tap2_ok = row_ok && pos < W - 1;
        const int64_t ct = ((tile * NW + wave) * 8) * 64 + lane;
        __syncthreads();   // the previous tile's image has been read
        embed_tile<W>(p, bufA, rows_tile, rows_valid, site0, seed);
        __syncthreads();
        const float* xc = bufA + row * kLdsStride + 4 * h;
        f32x16 acc[2];
        f32x4 t[8];
        {
            f32x4 x0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x0[u] = *reinterpret_cast<const f32x4*>(xc + 8 * u);
            acc_init_bias<2>(acc, cv + kVScB * kVec64, h, 8);
            gemm_regs<8, 2>(acc, p.P + kOffScW, x0, lane);
            acc_to_x<2>(acc, t);
#pragma unroll
            for (int u = 0; u < 8; ++u) p.cs.s[ct + u * 64] = t[u];
        }
        acc_init_bias<2>(acc, cv + kVC3aB * kVec64, h, 8);
        taps3(acc, p.P + kOffC3aW, p.P + kOffC3aW + kMat64, p.P + kOffC3aW + 2 * kMat64, tap0_ok ? xc - kLdsStride : xc, xc,
              tap2_ok ? xc + kLdsStride : xc, tap0_ok, tap2_ok, lane);
        acc_to_x<2>(acc, t);
        f32x4 y2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            p.cs.a1[ct + u * 64] = t[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[u][e] = row_ok ? swishf(t[u][e]) : 0.f;
                y2[u][e] = t[u][e] * t[u][e];
            }
        }
        chain_store(mine_b, h, t);
        wave_fence();
        const float c1 = colsum32(own_b, lane);
        wave_fence();
        chain_store(mine_b, h, y2);
        wave_fence();
        const float c2 = colsum32(own_b, lane);
        wave_fence();
        atomicAdd(&sacc[lane], c1);
        atomicAdd(&sacc[64 + lane], c2);
    }
    __syncthreads();
    for (int i = tid; i < 128; i += blockDim.x) atomicAdd(p.stats1 + i, sacc[i]);
}

template <int W>
__device__ __forceinline__ void conv_b_fwd_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    float* bufA = lds;
    float* bufB = bufA + rows_tile * kLdsStride;
    float* sacc = bufB + rows_tile * kLdsStride;
    float* bnv = sacc + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* own_b = bufB + wave * 32 * kLdsStride;
    float* mine_b = own_b + j * kLdsStride;
    float* mine_a = bufA + row * kLdsStride;
    for (int i = tid; i < 128; i += blockDim.x) sacc[i] = 0.f;   // (a one-wavefront tile has 64 threads)
    bn_prepare(p.bn1, bnv, tid, blockIdx.x == 0);
    const f32x4* cv = p.P + kOffConvV;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int pos = row_ok ? row % W : 1;
        const bool tap0_ok = row_ok && pos > 0, tap2_ok = row_ok && pos < W - 1;
        const int64_t ct = ((tile * NW + wave) * 8) * 64 + lane;
        const int64_t gsite = site0 + row / W;
        __syncthreads();   // the previous tile's image has been read (and bnv is ready)
        {
            f32x4 z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 a = p.cs.a1[ct + u * 64];
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * u + 4 * h);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * u + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[u][e] = row_ok ? fmaf(swishf(a[e]), sc[e], sh[e]) : 0.f;
            }
            dropout_chain(z, gsite, h, seed, p.stream_conv, p.thr_conv, p.scale_conv);   // one mask value per (site, channel)
            chain_store(mine_a, h, z);
        }
        __syncthreads();
        const float* xc = mine_a + 4 * h;
        f32x16 acc[2];
        acc_init_bias<2>(acc, cv + kVC3bB * kVec64, h, 8);
        taps3(acc, p.P + kOffC3bW, p.P + kOffC3bW + kMat64, p.P + kOffC3bW + 2 * kMat64, tap0_ok ? xc - kLdsStride : xc, xc,
              tap2_ok ? xc + kLdsStride : xc, tap0_ok, tap2_ok, lane);
        f32x4 t[8], y2[8];
        acc_to_x<2>(acc, t);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            p.cs.a2[ct + u * 64] = t[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t[u][e] = row_ok ? swishf(t[u][e]) : 0.f;
                y2[u][e] = t[u][e] * t[u][e];
            }
        }
        chain_store(mine_b, h, t);
        wave_fence();
        const float c1 = colsum32(own_b, lane);
        wave_fence();
        chain_store(mine_b, h, y2);
        wave_fence();
        const float c2 = colsum32(own_b, lane);
        wave_fence();
        atomicAdd(&sacc[lane], c1);
        atomicAdd(&sacc[64 + lane], c2);
    }
    __syncthreads();
    for (int i = tid; i < 128; i += blockDim.x) atomicAdd(p.stats2 + i, sacc[i]);
}

// ------------------------------------------------------------------------------------ backward

// partial buffer of a workgroup (floats): conv_b_bwd: 3 units | conv3b bias gradient;
// conv_a_bwd: 4 units (conv3a taps, shortcut) | conv3a bias, shortcut bias | embedding rows [21][64] | position rows [W][64]
constexpr int kCbVec = 3 * 4096;
constexpr int kCaVec = 5 * 4096, kCaPos = kCaVec + 128;        // units 0..2 taps, 3 shortcut, 4 embedding table (rows 0..20)
static_assert(kCbVec + 64 <= kCbFloats && kCaPos + kMaxW * 64 <= kCaFloats && kMaxW <= kMaxWin, "regions of the partial buffer");

// SHORT: the instance for tiles of fewer than eight wavefronts (small batches), see dw_unit
template <int W, bool SHORT>
__device__ __forceinline__ void conv_b_bwd_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    float* bufA = lds;
    float* bufB = bufA + rows_tile * kLdsStride;
    float* xreg = bufB + rows_tile * kLdsStride;
    float* bb = xreg + kXreg;           // [5][64] BN2 backward coefficients
    float* bnv = bb + 320;              // [2][64] BN1 forward scale | shift
    float* vacc = bnv + 128;            // [3][64]: conv3b bias gradient, sum d(z1), sum d(z1)*swish(a1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* own_a = bufA + wave * 32 * kLdsStride;
    float* own_b = bufB + wave * 32 * kLdsStride;
    float* mine_a = own_a + j * kLdsStride;
    float* mine_b = own_b + j * kLdsStride;
    float* part = p.part + (size_t)blockIdx.x * kTbPartStride + kCbBase;
    for (int i = tid; i < 192; i += blockDim.x) vacc[i] = 0.f;
    bn_bwd_prepare(p.bb2, bb, tid, blockIdx.x == 0);
    {
        BnFwd f{p.bb1.stats, p.bb1.inv_rows, p.bb1.gamma, p.bb1.beta, nullptr, nullptr};
        bn_prepare(f, bnv, tid, false);
    }
    bool first = true;
    int tile_round = 0;
    __shared__ int tile_slot;
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue_b, &tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue_b, &tile_slot, tile_round)) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int pos = row_ok ? row % W : 1;
        const bool tap0_ok = row_ok && pos > 0, tap2_ok = row_ok && pos < W - 1;
        const int64_t ct = ((tile * NW + wave) * 8) * 64 + lane;
        const int64_t gsite = site0 + row / W;
        __syncthreads();   // vectors ready; the previous tile's images and exchange buffer have been read
        {
            // d(a2) -> image B
            f32x4 a[8], dz[8], da[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = p.cs.a2[ct + u * 64];
                dz[u] = p.dx1[ct + u * 64];
            }
            bn_swish_backward(a, dz, bb, h, row_ok, da);
            chain_store(mine_b, h, da);
            // z1 (the convolution's input) -> image A
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 a1 = p.cs.a1[ct + u * 64];
                const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * u + 4 * h);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * u + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[u][e] = row_ok ? fmaf(swishf(a1[e]), sc[e], sh[e]) : 0.f;
            }
            dropout_chain(a, gsite, h, seed, p.stream_conv, p.thr_conv, p.scale_conv);
            chain_store(mine_a, h, a);
            wave_fence();
            atomicAdd(&vacc[lane], colsum32(own_b, lane));
        }
        __syncthreads();
        // d(z1)[r] = sum_t d(a2)[r - (t-1)] K[t]^T: tap 0 meets row r+1, tap 2 row r-1
        f32x4 dz1[8];
        {
            const float* xc = mine_b + 4 * h;
            f32x16 acc[2];
            acc_zero<2>(acc);
            taps3(acc, p.PT + kBtC3bT, p.PT + kBtC3bT + 1024, p.PT + kBtC3bT + 2048, tap2_ok ? xc + kLdsStride : xc, xc,
                  tap0_ok ? xc - kLdsStride : xc, tap2_ok, tap0_ok, lane);
            acc_to_x<2>(acc, dz1);
            dropout_chain(dz1, gsite, h, seed, p.stream_conv, p.thr_conv, p.scale_conv);
#pragma unroll
            for (int u = 0; u < 8; ++u) p.dz1[ct + u * 64] = dz1[u];
        }
        // dK[t] += shift_{t-1}(z1)^T d(a2)
        dw_unit_tap<W, SHORT>(bufA, bufB, -1, NW, xreg, part + 0 * 4096, first, lane, wave, rows_valid);
        __syncthreads();   // the exchange buffer of the previous unit has been read
        dw_unit<SHORT>(bufA, bufB, NW, xreg, part + 1 * 4096, first, lane, wave, rows_valid);
        __syncthreads();
        dw_unit_tap<W, SHORT>(bufA, bufB, +1, NW, xreg, part + 2 * 4096, first, lane, wave, rows_valid);
        __syncthreads();
        // column sums for BN1's backward pass: d(z1) and d(z1) * swish(a1), through the wavefront's own image rows
        {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 a1 = p.cs.a1[ct + u * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[u][e] = dz1[u][e] * swishf(a1[e]);
            }
            chain_store(mine_a, h, dz1);
            chain_store(mine_b, h, t);
            wave_fence();
            atomicAdd(&vacc[64 + lane], colsum32(own_a, lane));
            atomicAdd(&vacc[128 + lane], colsum32(own_b, lane));
            wave_fence();
        }
        first = false;
    }
    __syncthreads();
    if (first) zero_floats(part, 3 * 4096);
    for (int i = tid; i < 192; i += blockDim.x) {
        if (i < 64)
            part[kCbVec + i] = vacc[i];
        else
            atomicAdd(p.bn1_sums + i - 64, vacc[i]);
    }
}

template <int W, bool SHORT>
__device__ __forceinline__ void conv_a_bwd_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    float* bufA = lds;
    float* bufB = bufA + rows_tile * kLdsStride;
    float* xreg = bufB + rows_tile * kLdsStride;
    float* bb = xreg + kXreg;           // [5][64] BN1 backward coefficients
    float* vacc = bb + 320;             // [2][64]: conv3a bias gradient, shortcut bias gradient
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* own_b = bufB + wave * 32 * kLdsStride;
    float* mine_b = own_b + j * kLdsStride;
    float* part = p.part + (size_t)blockIdx.x * kTbPartStride + kCaBase;
    for (int i = tid; i < 128; i += blockDim.x) vacc[i] = 0.f;
    bn_bwd_prepare(p.bb1, bb, tid, blockIdx.x == 0);
    bool first = true;
    int tile_round = 0;
    __shared__ int tile_slot;
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue_a, &tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue_a, &tile_slot, tile_round)) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int pos = row_ok ? row % W : 1;
        const bool tap0_ok = row_ok && pos > 0, tap2_ok = row_ok && pos < W - 1;
        const int64_t ct = ((tile * NW + wave) * 8) * 64 + lane;
        const int64_t gsite = site0 + row / W;
        __syncthreads();
        embed_tile<W>(p, bufA, rows_tile, rows_valid, site0, seed);   // x0 -> image A
        {
            f32x4 a[8], dz[8], da[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = p.cs.a1[ct + u * 64];
                dz[u] = p.dz1[ct + u * 64];
            }
            bn_swish_backward(a, dz, bb, h, row_ok, da);
            chain_store(mine_b, h, da);                               // d(a1) -> image B
            wave_fence();
            atomicAdd(&vacc[lane], colsum32(own_b, lane));
        }
        __syncthreads();
        // d(x0) = sum_t d(a1)[r - (t-1)] K3a[t]^T + d(x1) Ksc^T
        f32x16 acc[2];
        acc_zero<2>(acc);
        {
            const float* xc = mine_b + 4 * h;
            taps3(acc, p.PT + kBtC3aT, p.PT + kBtC3aT + 1024, p.PT + kBtC3aT + 2048, tap2_ok ? xc + kLdsStride : xc, xc,
                  tap0_ok ? xc - kLdsStride : xc, tap2_ok, tap0_ok, lane);
        }
        dw_unit_tap<W, SHORT>(bufA, bufB, -1, NW, xreg, part + 0 * 4096, first, lane, wave, rows_valid);
        __syncthreads();   // the exchange buffer of the previous unit has been read
        dw_unit<SHORT>(bufA, bufB, NW, xreg, part + 1 * 4096, first, lane, wave, rows_valid);
        __syncthreads();
        dw_unit_tap<W, SHORT>(bufA, bufB, +1, NW, xreg, part + 2 * 4096, first, lane, wave, rows_valid);
        __syncthreads();
        f32x4 dx0[8];
        {
            f32x4 d1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d1[u] = p.dx1[ct + u * 64];   // zero on padding rows
            chain_store(mine_b, h, d1);
            wave_fence();
            atomicAdd(&vacc[64 + lane], colsum32(own_b, lane));
            __syncthreads();
            dw_unit<SHORT>(bufA, bufB, NW, xreg, part + 3 * 4096, first, lane, wave, rows_valid);       // dKsc += x0^T d(x1)
            gemm_regs<8, 2>(acc, p.PT + kBtScT, d1, lane);
            acc_to_x<2>(acc, dx0);
        }
        dropout_chain(dx0, gsite, h, seed, p.stream_embed, p.thr_embed, p.scale_embed);
        // embedding tables: dP[pos] += rows of every site at that position, dE[code] += rows with that code.
        // The position rows are summed straight from the image.  The code rows are one more weight-gradient unit:
        // dE = onehot(code)^T d(x0), with the one-hot rows (32 columns wide, codes 0..20) as the X image.
        chain_store(mine_b, h, dx0);
        {
            int code = row_ok ? (int)p.tok[(site0 + row / W) * W + pos] : -1;
            code = code > kCodes - 1 ? kCodes - 1 : code;
            f32x4 oh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) oh[u][e] = (8 * u + 4 * h + e) == code ? 1.0f : 0.f;
            chain_store(bufA + row * kLdsStride, h, oh);      // the x0 image has been read by the shortcut's unit
        }
        __syncthreads();   // d(x0) and the one-hot rows of the tile are in the images; the exchange buffer has been read
        for (int o = tid; o < W * 64; o += blockDim.x) {
            const int pp = o >> 6, f = o & 63;
            float a = first ? 0.f : part[kCaPos + o];
            for (int s = 0; s < sites_here; ++s) a += bufB[(s * W + pp) * kLdsStride + f];
            part[kCaPos + o] = a;
        }
        dw_unit<SHORT>(bufA, bufB, NW, xreg, part + 4 * 4096, first, lane, wave, rows_valid);          // dE += onehot^T d(x0)
        first = false;
    }
    __syncthreads();
    if (first) {
        zero_floats(part, 5 * 4096);
        zero_floats(part + kCaPos, W * 64);
    }
    for (int i = tid; i < 128; i += blockDim.x) part[kCaVec + i] = vacc[i];
}


// ------------------------------------------------------------------------------------ one-site tiles (dptm_train_pair.h)

// The same four kernels for ONE site in FOUR wavefronts: wavefronts w and w + 2 share row block w & 1, each with one
// of the two output tiles of every product, its half of every image row and the column sums of its own 32 features.
// A convolution reads whole rows of the LDS image, so the halves need no exchange here.

namespace {
// own groups of bn_swish_backward
__device__ __forceinline__ void bn_swish_backward4(const f32x4 (&a)[4], const f32x4 (&dz)[4], const float* bb, int h, int part,
                                                   bool row_ok, f32x4 (&da)[4]) {
    bb += 32 * part + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 8 * g;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(bb + c), inv = *reinterpret_cast<const f32x4*>(bb + 64 + c);
        const f32x4 gi = *reinterpret_cast<const f32x4*>(bb + 128 + c), s1 = *reinterpret_cast<const f32x4*>(bb + 192 + c);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(bb + 256 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = a[g][e];
            const float sg = fast_rcp(1.0f + __expf(-v));
            const float xh = (v * sg - mu[e]) * inv[e];
            const float dy = gi[e] * (dz[g][e] - s1[e] - xh * s2[e]);
            da[g][e] = row_ok ? dy * (sg * (1.0f + v * (1.0f - sg))) : 0.f;
        }
    }
}
// z = BN(swish(a)) of the own groups; zero on padding rows
__device__ __forceinline__ void bn_swish4(const f32x4 (&a)[4], const float* bnv, int h, int part, bool row_ok, f32x4 (&z)[4]) {
    bnv += 32 * part + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * g);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[g][e] = row_ok ? fmaf(swishf(a[g][e]), sc[e], sh[e]) : 0.f;
    }
}
// swish of the own groups and its column sums [sum | sum of squares] over the wavefront's rows, added to sacc[128]
__device__ __forceinline__ void swish_sums4(f32x4 (&t)[4], bool row_ok, float* mine_b, float* own_b, int h, int part, int lane,
                                            float* sacc) {
    f32x4 y2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[g][e] = row_ok ? swishf(t[g][e]) : 0.f;
            y2[g][e] = t[g][e] * t[g][e];
        }
    chain_store4(mine_b, h, part, t);
    wave_fence();
    const float c1 = colsum_rows_own(own_b, part, lane);
    wave_fence();
    chain_store4(mine_b, h, part, y2);
    wave_fence();
    const float c2 = colsum_rows_own(own_b, part, lane);
    wave_fence();
    if (lane < 32) {
        atomicAdd(&sacc[32 * part + lane], c1);
        atomicAdd(&sacc[64 + 32 * part + lane], c2);
    }
}
}  // namespace

#define DPTM_PAIR_SETUP                                                                                      \
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);            \
    const int pw = wave & 1, part = wave >> 1;                                                               \
    const int h = lane >> 5, j = lane & 31;                                                                  \
    const unsigned seed = p.step[kStepSeed];                                                                 \
    const int row = pw * 32 + j;                                                                             \
    const bool row_ok = row < W;                                                                             \
    const bool tap0_ok = row_ok && row > 0, tap2_ok = row_ok && row < W - 1;                                 \
    float* own_a = bufA + pw * 32 * kLdsStride;                                                              \
    float* own_b = bufB + pw * 32 * kLdsStride;                                                              \
    float* mine_a = own_a + j * kLdsStride;                                                                  \
    float* mine_b = own_b + j * kLdsStride;                                                                  \
    const f32x4* cv = p.P + kOffConvV + 4 * part;       /* the own groups of a packed 64-feature vector */     \
    (void)seed; (void)own_a; (void)mine_a; (void)cv; (void)tap0_ok; (void)tap2_ok

template <int W>
__device__ __forceinline__ void conv_a_fwd_pair_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    float* sacc = bufB + kPairImg;     // [sum | sum of squares] of swish(a1), this workgroup
    DPTM_PAIR_SETUP;
    for (int i = tid; i < 128; i += blockDim.x) sacc[i] = 0.f;
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t co = ((tile * 4 + pw) * 8 + 4 * part) * 64 + lane;      // the wavefront's own groups
        __syncthreads();   // the previous tile's image has been read
        embed_tile<W>(p, bufA, 64, W, tile, seed);
        __syncthreads();
        const float* xc = bufA + row * kLdsStride + 4 * h;
        f32x16 acc[1];
        f32x4 t[4];
        {
            f32x4 x0[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x0[u] = *reinterpret_cast<const f32x4*>(xc + 8 * u);
            acc_init_bias<1>(acc, cv + kVScB * kVec64, h, 8);
            gemm_regs<8, 1, 8>(acc, p.P + kOffScW + part * 8 * 64, x0, lane);
            acc_to_x<1>(acc, t);
#pragma unroll
            for (int g = 0; g < 4; ++g) p.cs.s[co + g * 64] = t[g];
        }
        acc_init_bias<1>(acc, cv + kVC3aB * kVec64, h, 8);
        const f32x4* wt = p.P + kOffC3aW + part * 8 * 64;
        taps3<1>(acc, wt, wt + kMat64, wt + 2 * kMat64, tap0_ok ? xc - kLdsStride : xc, xc, tap2_ok ? xc + kLdsStride : xc, tap0_ok,
                 tap2_ok, lane);
        acc_to_x<1>(acc, t);
#pragma unroll
        for (int g = 0; g < 4; ++g) p.cs.a1[co + g * 64] = t[g];
        swish_sums4(t, row_ok, mine_b, own_b, h, part, lane, sacc);
    }
    __syncthreads();
    for (int i = tid; i < 128; i += blockDim.x) atomicAdd(p.stats1 + i, sacc[i]);
}

template <int W>
__device__ __forceinline__ void conv_b_fwd_pair_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    float* sacc = bufB + kPairImg;
    float* bnv = sacc + 128;
    DPTM_PAIR_SETUP;
    for (int i = tid; i < 128; i += blockDim.x) sacc[i] = 0.f;
    bn_prepare(p.bn1, bnv, tid, blockIdx.x == 0);
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t co = ((tile * 4 + pw) * 8 + 4 * part) * 64 + lane;
        __syncthreads();   // the previous tile's image has been read (and bnv is ready)
        {
            f32x4 a[4], z[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = p.cs.a1[co + g * 64];
            bn_swish4(a, bnv, h, part, row_ok, z);
            dropout4(z, tile, h, part, seed, p.stream_conv, p.thr_conv, p.scale_conv);   // one mask value per (site, channel)
            chain_store4(mine_a, h, part, z);
        }
        __syncthreads();
        const float* xc = mine_a + 4 * h;
        f32x16 acc[1];
        acc_init_bias<1>(acc, cv + kVC3bB * kVec64, h, 8);
        const f32x4* wt = p.P + kOffC3bW + part * 8 * 64;
        taps3<1>(acc, wt, wt + kMat64, wt + 2 * kMat64, tap0_ok ? xc - kLdsStride : xc, xc, tap2_ok ? xc + kLdsStride : xc, tap0_ok,
                 tap2_ok, lane);
        f32x4 t[4];
        acc_to_x<1>(acc, t);
#pragma unroll
        for (int g = 0; g < 4; ++g) p.cs.a2[co + g * 64] = t[g];
        swish_sums4(t, row_ok, mine_b, own_b, h, part, lane, sacc);
    }
    __syncthreads();
    for (int i = tid; i < 128; i += blockDim.x) atomicAdd(p.stats2 + i, sacc[i]);
}

template <int W>
__device__ __forceinline__ void conv_b_bwd_pair_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    float* bb = bufB + kPairImg;        // [5][64] BN2 backward coefficients
    float* bnv = bb + 320;              // [2][64] BN1 forward scale | shift
    float* vacc = bnv + 128;            // [3][64]: conv3b bias gradient, sum d(z1), sum d(z1)*swish(a1)
    int* tile_slot = reinterpret_cast<int*>(vacc + 192);
    DPTM_PAIR_SETUP;
    float* pbuf = p.part + (size_t)blockIdx.x * kTbPartStride + kCbBase;
    for (int i = tid; i < 192; i += blockDim.x) vacc[i] = 0.f;
    bn_bwd_prepare(p.bb2, bb, tid, blockIdx.x == 0);
    {
        BnFwd f{p.bb1.stats, p.bb1.inv_rows, p.bb1.gamma, p.bb1.beta, nullptr, nullptr};
        bn_prepare(f, bnv, tid, false);
    }
    bool first = true;
    int tile_round = 0;
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue_b, tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue_b, tile_slot, tile_round)) {
        const int64_t co = ((tile * 4 + pw) * 8 + 4 * part) * 64 + lane;
        __syncthreads();   // vectors ready; the previous tile's images have been read
        {
            // d(a2) -> image B, z1 (the convolution's input) -> image A
            f32x4 a[4], dz[4], da[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[g] = p.cs.a2[co + g * 64];
                dz[g] = p.dx1[co + g * 64];
            }
            bn_swish_backward4(a, dz, bb, h, part, row_ok, da);
            chain_store4(mine_b, h, part, da);
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = p.cs.a1[co + g * 64];
            bn_swish4(a, bnv, h, part, row_ok, dz);
            dropout4(dz, tile, h, part, seed, p.stream_conv, p.thr_conv, p.scale_conv);
            chain_store4(mine_a, h, part, dz);
            wave_fence();
            const float c = colsum_rows_own(own_b, part, lane);
            if (lane < 32) atomicAdd(&vacc[32 * part + lane], c);
        }
        __syncthreads();
        // d(z1)[r] = sum_t d(a2)[r - (t-1)] K[t]^T: tap 0 meets row r+1, tap 2 row r-1
        f32x4 dz1[4];
        {
            const float* xc = mine_b + 4 * h;
            f32x16 acc[1];
            acc_zero<1>(acc);
            const f32x4* wt = p.PT + kBtC3bT + part * 8 * 64;
            taps3<1>(acc, wt, wt + 1024, wt + 2048, tap2_ok ? xc + kLdsStride : xc, xc, tap0_ok ? xc - kLdsStride : xc, tap2_ok, tap0_ok,
                     lane);
            acc_to_x<1>(acc, dz1);
            dropout4(dz1, tile, h, part, seed, p.stream_conv, p.thr_conv, p.scale_conv);
#pragma unroll
            for (int g = 0; g < 4; ++g) p.dz1[co + g * 64] = dz1[g];
            // (the tile's two other 32-row slots are padding: the next kernel expects zeros there)
            const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) p.dz1[co + 2 * 512 + g * 64] = zero4;
        }
        // dK[t] += shift_{t-1}(z1)^T d(a2)
        dw_unit_tap<W, true>(bufA, bufB, -1, 4, nullptr, pbuf + 0 * 4096, first, lane, wave, W);
        dw_unit<true>(bufA, bufB, 4, nullptr, pbuf + 1 * 4096, first, lane, wave, W);
        dw_unit_tap<W, true>(bufA, bufB, +1, 4, nullptr, pbuf + 2 * 4096, first, lane, wave, W);
        // column sums for BN1's backward pass: d(z1) and d(z1) * swish(a1), through the wavefront's own part of the images
        {
            f32x4 t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a1 = p.cs.a1[co + g * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[g][e] = dz1[g][e] * swishf(a1[e]);
            }
            chain_store4(mine_a, h, part, dz1);
            chain_store4(mine_b, h, part, t);
            wave_fence();
            const float c1 = colsum_rows_own(own_a, part, lane), c2 = colsum_rows_own(own_b, part, lane);
            if (lane < 32) {
                atomicAdd(&vacc[64 + 32 * part + lane], c1);
                atomicAdd(&vacc[128 + 32 * part + lane], c2);
            }
            wave_fence();
        }
        first = false;
    }
    __syncthreads();
    if (first) zero_floats(pbuf, 3 * 4096);
    for (int i = tid; i < 192; i += blockDim.x) {
        if (i < 64)
            pbuf[kCbVec + i] = vacc[i];
        else
            atomicAdd(p.bn1_sums + i - 64, vacc[i]);
    }
}

template <int W>
__device__ __forceinline__ void conv_a_bwd_pair_body(const ConvArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kPairImg;
    float* bb = bufB + kPairImg;        // [5][64] BN1 backward coefficients
    float* vacc = bb + 320;             // [2][64]: conv3a bias gradient, shortcut bias gradient
    int* tile_slot = reinterpret_cast<int*>(vacc + 128);
    DPTM_PAIR_SETUP;
    float* pbuf = p.part + (size_t)blockIdx.x * kTbPartStride + kCaBase;
    for (int i = tid; i < 128; i += blockDim.x) vacc[i] = 0.f;
    bn_bwd_prepare(p.bb1, bb, tid, blockIdx.x == 0);
    bool first = true;
    int tile_round = 0;
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue_a, tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue_a, tile_slot, tile_round)) {
        const int64_t ct = ((tile * 4 + pw) * 8) * 64 + lane;
        const int64_t co = ct + 4 * part * 64;
        __syncthreads();
        embed_tile<W>(p, bufA, 64, W, tile, seed);   // x0 -> image A
        {
            f32x4 a[4], dz[4], da[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[g] = p.cs.a1[co + g * 64];
                dz[g] = p.dz1[co + g * 64];
            }
            bn_swish_backward4(a, dz, bb, h, part, row_ok, da);
            chain_store4(mine_b, h, part, da);                        // d(a1) -> image B
            wave_fence();
            const float c = colsum_rows_own(own_b, part, lane);
            if (lane < 32) atomicAdd(&vacc[32 * part + lane], c);
        }
        __syncthreads();
        // d(x0) = sum_t d(a1)[r - (t-1)] K3a[t]^T + d(x1) Ksc^T, this wavefront's output tile
        f32x16 acc[1];
        acc_zero<1>(acc);
        {
            const float* xc = mine_b + 4 * h;
            const f32x4* wt = p.PT + kBtC3aT + part * 8 * 64;
            taps3<1>(acc, wt, wt + 1024, wt + 2048, tap2_ok ? xc + kLdsStride : xc, xc, tap0_ok ? xc - kLdsStride : xc, tap2_ok, tap0_ok,
                     lane);
        }
        dw_unit_tap<W, true>(bufA, bufB, -1, 4, nullptr, pbuf + 0 * 4096, first, lane, wave, W);
        dw_unit<true>(bufA, bufB, 4, nullptr, pbuf + 1 * 4096, first, lane, wave, W);
        dw_unit_tap<W, true>(bufA, bufB, +1, 4, nullptr, pbuf + 2 * 4096, first, lane, wave, W);
        f32x4 dx0[4];
        {
            f32x4 d1[8], d4[4];
#pragma unroll
            for (int u = 0; u < 8; ++u) d1[u] = p.dx1[ct + u * 64];   // zero on padding rows
            own_groups(d1, part, d4);
            chain_store4(mine_b, h, part, d4);
            wave_fence();
            const float c = colsum_rows_own(own_b, part, lane);
            if (lane < 32) atomicAdd(&vacc[64 + 32 * part + lane], c);
            __syncthreads();
            dw_unit<true>(bufA, bufB, 4, nullptr, pbuf + 3 * 4096, first, lane, wave, W);       // dKsc += x0^T d(x1)
            gemm_regs<8, 1, 8>(acc, p.PT + kBtScT + part * 8 * 64, d1, lane);
            acc_to_x<1>(acc, dx0);
        }
        dropout4(dx0, tile, h, part, seed, p.stream_embed, p.thr_embed, p.scale_embed);
        // embedding tables: dP[pos] += the row at that position, dE[code] += rows with that code (one more
        // weight-gradient unit, dE = onehot(code)^T d(x0), with the one-hot rows as the X image)
        chain_store4(mine_b, h, part, dx0);
        {
            int code = row_ok ? (int)p.tok[tile * W + row] : -1;
            code = code > kCodes - 1 ? kCodes - 1 : code;
            f32x4 oh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) oh[g][e] = (32 * part + 8 * g + 4 * h + e) == code ? 1.0f : 0.f;
            chain_store4(mine_a, h, part, oh);      // the x0 image has been read by the shortcut's unit
        }
        __syncthreads();   // d(x0) and the one-hot rows of the site are in the images
        for (int o = tid; o < W * 64; o += blockDim.x) {
            const int pp = o >> 6, f = o & 63;
            pbuf[kCaPos + o] = (first ? 0.f : pbuf[kCaPos + o]) + bufB[pp * kLdsStride + f];
        }
        dw_unit<true>(bufA, bufB, 4, nullptr, pbuf + 4 * 4096, first, lane, wave, W);          // dE += onehot^T d(x0)
        first = false;
    }
    __syncthreads();
    if (first) {
        zero_floats(pbuf, 5 * 4096);
        zero_floats(pbuf + kCaPos, W * 64);
    }
    for (int i = tid; i < 128; i += blockDim.x) pbuf[kCaVec + i] = vacc[i];
}
#undef DPTM_PAIR_SETUP

// ------------------------------------------------------------------------------------ kernels

template <int W>
__global__ __launch_bounds__(512) void conv_a_fwd_kernel(ConvArgs p) { conv_a_fwd_body<W>(p); }
template <int W>
__global__ __launch_bounds__(512) void conv_b_fwd_kernel(ConvArgs p) { conv_b_fwd_body<W>(p); }
template <int W, bool SHORT>
__global__ __launch_bounds__(512) void conv_b_bwd_kernel(ConvArgs p) { conv_b_bwd_body<W, SHORT>(p); }
template <int W, bool SHORT>
__global__ __launch_bounds__(512) void conv_a_bwd_kernel(ConvArgs p) { conv_a_bwd_body<W, SHORT>(p); }

template <int W>
__global__ __launch_bounds__(256) void conv_a_fwd_pair_kernel(ConvArgs p) { conv_a_fwd_pair_body<W>(p); }
template <int W>
__global__ __launch_bounds__(256) void conv_b_fwd_pair_kernel(ConvArgs p) { conv_b_fwd_pair_body<W>(p); }
template <int W>
__global__ __launch_bounds__(256) void conv_b_bwd_pair_kernel(ConvArgs p) { conv_b_bwd_pair_body<W>(p); }
template <int W>
__global__ __launch_bounds__(256) void conv_a_bwd_pair_kernel(ConvArgs p) { conv_a_bwd_pair_body<W>(p); }

// The same launch of the three sequence branches as ONE launch, grid.y = branch (four-wavefront tiles, a tile per
// workgroup; grid.x = the most tiles a branch has, a workgroup beyond its branch's tiles leaves at once).  A small batch's
// step is then a chain of launches on one stream: no hand-over between streams sits on its critical path.
// (grid.y = 0 is the longest branch, W = 51: its workgroups are handed out first)
#define DPTM_BRANCH3(BODY21, BODY33, BODY51)                                         \
    if (blockIdx.y == 2) {                                                           \
        if ((int64_t)blockIdx.x < p.a[0].ntiles) BODY21(p.a[0]);                     \
    } else if (blockIdx.y == 1) {                                                    \
        if ((int64_t)blockIdx.x < p.a[1].ntiles) BODY33(p.a[1]);                     \
    } else {                                                                         \
        if ((int64_t)blockIdx.x < p.a[2].ntiles) BODY51(p.a[2]);                     \
    }
struct ConvArgs3 { ConvArgs a[3]; };
// MAXT: 256 for four-wavefront tiles, 512 for eight
template <int MAXT>
__global__ __launch_bounds__(MAXT) void conv_a_fwd3_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_a_fwd_body<21>, conv_a_fwd_body<33>, conv_a_fwd_body<51>) }
template <int MAXT>
__global__ __launch_bounds__(MAXT) void conv_b_fwd3_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_b_fwd_body<21>, conv_b_fwd_body<33>, conv_b_fwd_body<51>) }
#define DPTM_CBB(W) conv_b_bwd_body<W, MAXT == 256>
#define DPTM_CAB(W) conv_a_bwd_body<W, MAXT == 256>
template <int MAXT>
__global__ __launch_bounds__(MAXT) void conv_b_bwd3_kernel(ConvArgs3 p) { DPTM_BRANCH3(DPTM_CBB(21), DPTM_CBB(33), DPTM_CBB(51)) }
template <int MAXT>
__global__ __launch_bounds__(MAXT) void conv_a_bwd3_kernel(ConvArgs3 p) { DPTM_BRANCH3(DPTM_CAB(21), DPTM_CAB(33), DPTM_CAB(51)) }

__global__ __launch_bounds__(256) void conv_a_fwd3_pair_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_a_fwd_pair_body<21>, conv_a_fwd_pair_body<33>, conv_a_fwd_pair_body<51>) }
__global__ __launch_bounds__(256) void conv_b_fwd3_pair_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_b_fwd_pair_body<21>, conv_b_fwd_pair_body<33>, conv_b_fwd_pair_body<51>) }
__global__ __launch_bounds__(256) void conv_b_bwd3_pair_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_b_bwd_pair_body<21>, conv_b_bwd_pair_body<33>, conv_b_bwd_pair_body<51>) }
__global__ __launch_bounds__(256) void conv_a_bwd3_pair_kernel(ConvArgs3 p) { DPTM_BRANCH3(conv_a_bwd_pair_body<21>, conv_a_bwd_pair_body<33>, conv_a_bwd_pair_body<51>) }

// ------------------------------------------------------------------------------------ launchers

static size_t conv_fwd_pair_lds() { return ((size_t)2 * kPairImg + 256) * sizeof(float); }
static size_t conv_bwd_pair_lds() { return ((size_t)2 * kPairImg + 320 + 128 + 192 + 16) * sizeof(float); }

static size_t conv_fwd_lds(int NW) { return ((size_t)2 * NW * 32 * kLdsStride + 256) * sizeof(float); }
static size_t conv_bwd_lds(int NW) { return ((size_t)2 * NW * 32 * kLdsStride + kConvLdsExtra) * sizeof(float); }

template <typename K>
static hipError_t launch_conv(K kernel, bool (&done)[64], const ConvArgs& a, int NW, size_t lds, size_t lds_max, int64_t max_grid,
                              hipStream_t st) {
    hipError_t e = lds_opt_in(kernel, done, lds_max);
    if (e != hipSuccess) return e;
    const int64_t grid = a.ntiles < max_grid ? a.ntiles : max_grid;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * NW), lds, st, a);
    return hipGetLastError();
}

#define DPTM_CONV_DISPATCH(KERNEL, LDS, LDS_MAX, MAX_GRID)                                                        \
    static bool d21[64] = {}, d33[64] = {}, d51[64] = {};                                                         \
    switch (wlen) {                                                                                               \
        case 21: e = launch_conv(&KERNEL<21>, d21, a, NW, LDS, LDS_MAX, MAX_GRID, st); break;                     \
        case 33: e = launch_conv(&KERNEL<33>, d33, a, NW, LDS, LDS_MAX, MAX_GRID, st); break;                     \
        default: e = launch_conv(&KERNEL<51>, d51, a, NW, LDS, LDS_MAX, MAX_GRID, st); break;                     \
    }

// backward kernels: a second instance for tiles of fewer than eight wavefronts
#define DPTM_CONV_DISPATCH_BWD(KERNEL, LDS, LDS_MAX, MAX_GRID)                                                    \
    static bool d21[64] = {}, d33[64] = {}, d51[64] = {}, s21[64] = {}, s33[64] = {}, s51[64] = {};               \
    const bool small_tile = NW < 8;                                                                               \
    switch (wlen) {                                                                                               \
        case 21: e = small_tile ? launch_conv(&KERNEL<21, true>, s21, a, NW, LDS, LDS_MAX, MAX_GRID, st)          \
                                : launch_conv(&KERNEL<21, false>, d21, a, NW, LDS, LDS_MAX, MAX_GRID, st); break; \
        case 33: e = small_tile ? launch_conv(&KERNEL<33, true>, s33, a, NW, LDS, LDS_MAX, MAX_GRID, st)          \
                                : launch_conv(&KERNEL<33, false>, d33, a, NW, LDS, LDS_MAX, MAX_GRID, st); break; \
        default: e = small_tile ? launch_conv(&KERNEL<51, true>, s51, a, NW, LDS, LDS_MAX, MAX_GRID, st)          \
                                : launch_conv(&KERNEL<51, false>, d51, a, NW, LDS, LDS_MAX, MAX_GRID, st); break; \
    }

// one-site tiles: the pair kernels
#define DPTM_CONV_DISPATCH_PAIR(KERNEL, LDS, MAX_GRID)                                                            \
    static bool p21[64] = {}, p33[64] = {}, p51[64] = {};                                                         \
    switch (wlen) {                                                                                               \
        case 21: e = launch_conv(&KERNEL<21>, p21, a, 4, LDS, LDS, MAX_GRID, st); break;                          \
        case 33: e = launch_conv(&KERNEL<33>, p33, a, 4, LDS, LDS, MAX_GRID, st); break;                          \
        default: e = launch_conv(&KERNEL<51>, p51, a, 4, LDS, LDS, MAX_GRID, st); break;                          \
    }

hipError_t t_conv_a_fwd(const ConvArgs& a, int wlen, int NW, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    hipError_t e;
    if (t_pair_tile(a.S, NW)) {
        DPTM_CONV_DISPATCH_PAIR(conv_a_fwd_pair_kernel, conv_fwd_pair_lds(), 1024)
        return e;
    }
    DPTM_CONV_DISPATCH(conv_a_fwd_kernel, conv_fwd_lds(NW), conv_fwd_lds(8), 1024)
    return e;
}

hipError_t t_conv_b_fwd(const ConvArgs& a, int wlen, int NW, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    hipError_t e;
    if (t_pair_tile(a.S, NW)) {
        DPTM_CONV_DISPATCH_PAIR(conv_b_fwd_pair_kernel, conv_fwd_pair_lds(), 1024)
        return e;
    }
    DPTM_CONV_DISPATCH(conv_b_fwd_kernel, conv_fwd_lds(NW), conv_fwd_lds(8), 1024)
    return e;
}

// what the second / first convolution's backward launch leaves in a workgroup's partial buffer, appended to the plan
static hipError_t conv_b_bwd_plan(const ConvGrads& g, PartPlan* plan) {
    PartUnits& ut = plan->units;
    PartVecs& vt = plan->vecs;
    if (ut.n + 3 > kPartMaxUnits || vt.n + 1 > kPartMaxVecs) return hipErrorInvalidValue;
    for (int t = 0; t < 3; ++t) {
        ut.dst[ut.n] = g.c3bk + t * 4096;
        ut.off[ut.n] = kCbBase + t * 4096;
        ut.ld[ut.n] = 64;
        ut.rows[ut.n] = 64;
        ++ut.n;
    }
    vt.dst[vt.n] = g.c3bb; vt.off[vt.n] = kCbBase + kCbVec; vt.len[vt.n] = 64;
    ++vt.n;
    return hipSuccess;
}
static hipError_t conv_a_bwd_plan(const ConvGrads& g, int wlen, PartPlan* plan) {
    PartUnits& ut = plan->units;
    PartVecs& vt = plan->vecs;
    if (ut.n + 5 > kPartMaxUnits || vt.n + 3 > kPartMaxVecs) return hipErrorInvalidValue;
    float* const unit_dst[5] = {g.c3ak, g.c3ak + 4096, g.c3ak + 2 * 4096, g.sck, g.emb};
    for (int t = 0; t < 5; ++t) {
        ut.dst[ut.n] = unit_dst[t];
        ut.off[ut.n] = kCaBase + t * 4096;
        ut.ld[ut.n] = 64;
        ut.rows[ut.n] = t == 4 ? kCodes : 64;       // embedding table [kCodes][64]: only the unit's first kCodes rows exist
        ++ut.n;
    }
    vt.dst[vt.n] = g.c3ab; vt.off[vt.n] = kCaBase + kCaVec; vt.len[vt.n] = 64; ++vt.n;
    vt.dst[vt.n] = g.scb; vt.off[vt.n] = kCaBase + kCaVec + 64; vt.len[vt.n] = 64; ++vt.n;
    vt.dst[vt.n] = g.pos; vt.off[vt.n] = kCaBase + kCaPos; vt.len[vt.n] = wlen * 64; ++vt.n;
    return hipSuccess;
}

hipError_t t_conv_b_bwd(const ConvArgs& a, const ConvGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    hipError_t e;
    if (t_pair_tile(a.S, NW)) {
        DPTM_CONV_DISPATCH_PAIR(conv_b_bwd_pair_kernel, conv_bwd_pair_lds(), kTbMaxParts)
    } else {
        DPTM_CONV_DISPATCH_BWD(conv_b_bwd_kernel, conv_bwd_lds(NW), conv_bwd_lds(8), kTbMaxParts)
    }
    if (e != hipSuccess) return e;
    if ((e = conv_b_bwd_plan(g, plan)) != hipSuccess) return e;
    PartUnits& ut = plan->units;
    PartVecs& vt = plan->vecs;
    if (part_sums_deferred(a.ntiles)) return hipSuccess;
    const int grid = (int)(a.ntiles < kTbMaxParts ? a.ntiles : kTbMaxParts);
    e = t_reduce_parts(a.part, grid, ut, vt, st);
    ut.n = 0;
    vt.n = 0;
    return e;
}

hipError_t t_conv_a_bwd(const ConvArgs& a, const ConvGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    hipError_t e;
    if (t_pair_tile(a.S, NW)) {
        DPTM_CONV_DISPATCH_PAIR(conv_a_bwd_pair_kernel, conv_bwd_pair_lds(), kTbMaxParts)
    } else {
        DPTM_CONV_DISPATCH_BWD(conv_a_bwd_kernel, conv_bwd_lds(NW), conv_bwd_lds(8), kTbMaxParts)
    }
    if (e != hipSuccess) return e;
    const int grid = (int)(a.ntiles < kTbMaxParts ? a.ntiles : kTbMaxParts);
    if ((e = conv_a_bwd_plan(g, wlen, plan)) != hipSuccess) return e;
    // every backward launch of the branch has left its partial sums: one launch adds them to the gradients
    return t_reduce_parts(a.part, grid, plan->units, plan->vecs, st);
}

// ---- the three branches in one launch (see conv_a_fwd3_kernel)

static bool branch3_ok(const ConvArgs (&a)[3], int NW) {
    const int wl[3] = {21, 33, 51};
    if (NW != 4 && NW != 8) return false;
    for (int b = 0; b < 3; ++b)
        if (a[b].n_sites < 1 || a[b].ntiles > kTbMaxParts || !tile_ok(wl[b], a[b].S, NW, a[b].n_sites, a[b].ntiles)) return false;
    return true;
}

template <typename K>
static hipError_t launch_conv3(K kernel, bool (&done)[64], const ConvArgs (&a)[3], int NW, size_t lds, hipStream_t st) {
    if (!branch3_ok(a, NW)) return hipErrorInvalidValue;
    hipError_t e = lds_opt_in(kernel, done, lds);
    if (e != hipSuccess) return e;
    ConvArgs3 p;
    int64_t gx = 1;
    for (int b = 0; b < 3; ++b) {
        p.a[b] = a[b];
        gx = a[b].ntiles > gx ? a[b].ntiles : gx;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)gx, 3), dim3(64 * NW), lds, st, p);
    return hipGetLastError();
}

static bool pair3(const ConvArgs (&a)[3], int NW) {
    return t_pair_tile(a[0].S, NW) && t_pair_tile(a[1].S, NW) && t_pair_tile(a[2].S, NW);
}

hipError_t t_conv_a_fwd3(const ConvArgs (&a)[3], int NW, hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    if (pair3(a, NW)) return launch_conv3(&conv_a_fwd3_pair_kernel, dp, a, NW, conv_fwd_pair_lds(), st);
    return NW == 4 ? launch_conv3(&conv_a_fwd3_kernel<256>, d4, a, NW, conv_fwd_lds(4), st)
                   : launch_conv3(&conv_a_fwd3_kernel<512>, d8, a, NW, conv_fwd_lds(8), st);
}
hipError_t t_conv_b_fwd3(const ConvArgs (&a)[3], int NW, hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    if (pair3(a, NW)) return launch_conv3(&conv_b_fwd3_pair_kernel, dp, a, NW, conv_fwd_pair_lds(), st);
    return NW == 4 ? launch_conv3(&conv_b_fwd3_kernel<256>, d4, a, NW, conv_fwd_lds(4), st)
                   : launch_conv3(&conv_b_fwd3_kernel<512>, d8, a, NW, conv_fwd_lds(8), st);
}
hipError_t t_conv_b_bwd3(const ConvArgs (&a)[3], const ConvGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    hipError_t e = pair3(a, NW) ? launch_conv3(&conv_b_bwd3_pair_kernel, dp, a, NW, conv_bwd_pair_lds(), st)
                   : NW == 4    ? launch_conv3(&conv_b_bwd3_kernel<256>, d4, a, NW, conv_bwd_lds(4), st)
                                : launch_conv3(&conv_b_bwd3_kernel<512>, d8, a, NW, conv_bwd_lds(8), st);
    for (int b = 0; b < 3 && e == hipSuccess; ++b) e = conv_b_bwd_plan(g[b], &plan[b]);
    return e;
}
// ... and, behind the last backward launch, ONE launch that adds the three branches' partial sums to the gradients
hipError_t t_conv_a_bwd3(const ConvArgs (&a)[3], const ConvGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    hipError_t e = pair3(a, NW) ? launch_conv3(&conv_a_bwd3_pair_kernel, dp, a, NW, conv_bwd_pair_lds(), st)
                   : NW == 4    ? launch_conv3(&conv_a_bwd3_kernel<256>, d4, a, NW, conv_bwd_lds(4), st)
                                : launch_conv3(&conv_a_bwd3_kernel<512>, d8, a, NW, conv_bwd_lds(8), st);
    const int wl[3] = {21, 33, 51};
    for (int b = 0; b < 3 && e == hipSuccess; ++b) e = conv_a_bwd_plan(g[b], wl[b], &plan[b]);
    if (e != hipSuccess) return e;
    const float* parts[3] = {a[0].part, a[1].part, a[2].part};
    const int nparts[3] = {(int)a[0].ntiles, (int)a[1].ntiles, (int)a[2].ntiles};
    return t_reduce_parts3(parts, nparts, plan, st);
}

}  // namespace dptm
