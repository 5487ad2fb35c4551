// Device and launch helpers shared by the fused training kernels (dptm_train_branch.hip: TransformerBlocks and
// pooling; dptm_train_conv.hip: embedding and the residual convolution block): chain-layout vectors and LDS
// images, dropout on the stateless hash, LayerNormalization, and the weight-gradient unit on the matrix cores.
#pragma once
#include <hip/hip_runtime.h>

#include "dptm_layout.h"
#include "dptm_mfma.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"

namespace dptm {

constexpr int kXreg = 4096;   // floats of LDS next to the two images: exchange buffer of dw_unit, small per-row arrays

// Transposed images [feature][row] have ONE row stride for every tile size, so that a lane's 32 stores differ by
// compile-time offsets (a stride that depends on the tile size costs an address register per store).
constexpr int kTs = 260;      // 256 rows + 4: 16-byte reads by 16 lanes of consecutive features hit distinct banks
// floats of one LDS image of a tile of NW wavefronts: row-major [32 NW][68] or transposed [64][kTs]
__host__ __device__ constexpr int img_floats(int NW) {
    return 32 * NW * kLdsStride > 64 * kTs ? 32 * NW * kLdsStride : 64 * kTs;
}

// ------------------------------------------------------------------------------------ device helpers

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// inverse of acc_to_x: accumulator tiles that start from a chain-layout vector
__device__ __forceinline__ void x_to_acc(const f32x4 (&x)[8], f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = x[4 * t + g][e];
}

// chain-layout row vector <-> the lane's row of an LDS image [rows][kLdsStride]
__device__ __forceinline__ void chain_store(float* row_base, int h, const f32x4 (&x)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(row_base + 8 * u + 4 * h) = x[u];
}

// sum over 32 image rows of column `lane` (lane = feature)
__device__ __forceinline__ float colsum32(const float* img_rows, int lane) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        s0 += img_rows[r * kLdsStride + lane];
        s1 += img_rows[(r + 1) * kLdsStride + lane];
    }
    return s0 + s1;
}

// inverted dropout of a chain-layout row vector; the element index of (row r, feature c) is r*64 + c, as in the
// layer-wise dropout kernels and the NumPy restatement, so every implementation sees the same mask
__device__ __forceinline__ void dropout_chain(f32x4 (&x)[8], int64_t r, int h, unsigned seed, unsigned stream, unsigned thr,
                                              float scale) {
    if (thr == 0u) return;
    unsigned base = (unsigned)(r * 64) + 4 * h;
    // (kept local: the first mixing step of the hash depends on the element index and the seed only, and the compiler
    // would otherwise keep all 32 of them per lane alive across the kernel for the next dropout site to reuse)
    DPTM_LOCAL_INDEX(base);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[u][e] = hash_u32(seed, stream, base + 8 * u + e) >= thr ? x[u][e] * scale : 0.f;
}

// acc[t] += (rows 64c .. 64c+63 of a packed [256][64] matrix)^T-product with a 64-feature chain vector: the K = 256
// image has 32 feature groups per out tile, the chunk owns groups 8c .. 8c+7
__device__ __forceinline__ void gemm_k256_chunk(f32x16 (&acc)[2], const f32x4* __restrict__ wp, int c, const f32x4 (&x)[8],
                                                int lane) {
    DPTM_LOCAL_INDEX(lane);
    f32x4 a[8][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[u][t] = wp[(t * 32 + 8 * c + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u + 2 < 8) {
#pragma unroll
            for (int t = 0; t < 2; ++t) a[u + 2][t] = wp[(t * 32 + 8 * c + u + 2) * 64 + lane];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t) mfma(acc[t], a[u][t][e], x[u][e]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Chain layout -> the lane's two heads: lane (h, j) gets heads 2h and 2h+1 of row j, 16 features each in natural
// order (one exchange between the half-waves per register pair), times `scale`.
__device__ __forceinline__ void x_to_heads(const f32x4 (&x)[8], float (&hd)[2][16], float scale) {
    float qa[16], qb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        qa[r] = x[r >> 2][r & 3] * scale;
        qb[r] = x[4 + (r >> 2)][r & 3] * scale;
        half_swap(qa[r], qb[r]);
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int src = 4 * (d >> 3) + (d & 3);
        hd[0][d] = (d & 4) ? qb[src] : qa[src];
        hd[1][d] = (d & 4) ? qb[8 + src] : qa[8 + src];
    }
}
__device__ __forceinline__ void heads_to_x(const float (&hd)[2][16], f32x4 (&x)[8], float scale) {
    float qa[16], qb[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int dst = 4 * (d >> 3) + (d & 3);
        if (d & 4) {
            qb[dst] = hd[0][d] * scale;
            qb[8 + dst] = hd[1][d] * scale;
        } else {
            qa[dst] = hd[0][d] * scale;
            qa[8 + dst] = hd[1][d] * scale;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        half_swap(qa[r], qb[r]);
        x[r >> 2][r & 3] = qa[r];
        x[4 + (r >> 2)][r & 3] = qb[r];
    }
}

// keras LayerNormalization(epsilon=1e-6): y (the summed input) becomes xhat; returns 1/sigma
__device__ __forceinline__ float ln_normalise(f32x4 (&y)[8]) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (y[u][0] + y[u][1]) + (y[u][2] + y[u][3]);
    const float mean = xhalf_sum(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = y[u][e] - mean;
            y[u][e] = d;
            q = fmaf(d, d, q);
        }
    const float inv = rsqrtf(xhalf_sum(q) * (1.0f / 64.0f) + 1e-6f);
#pragma unroll
    for (int u = 0; u < 8; ++u) y[u] *= inv;
    return inv;
}

// out = xhat * gamma + beta with packed vectors (float4 index h*8 + u)
__device__ __forceinline__ void ln_affine(const f32x4 (&xh)[8], const f32x4* __restrict__ g, const f32x4* __restrict__ b, int h,
                                          f32x4 (&out)[8]) {
    DPTM_LOCAL_INDEX(h);
#pragma unroll
    for (int u = 0; u < 8; ++u) out[u] = xh[u] * g[h * 8 + u] + b[h * 8 + u];
}

// ---- transposed LDS images [feature][row] (row stride kTs): the layout the weight-gradient unit reads with
// 16-byte loads.  A lane stores its 32 features as single floats (consecutive lanes = consecutive rows =
// consecutive addresses).  lane_base = image + 4*h*kTs + row of the lane.
template <int TS = kTs>
__device__ __forceinline__ void chain_store_t(float* lane_base, const f32x4 (&x)[8]) {
    // the 32 addresses are formed from the base here, at the stores: hoisted out of a loop that calls this several
    // times they would be 15 more live registers each
    unsigned zero = 0;
    DPTM_LOCAL_INDEX(zero);
    lane_base += zero;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) lane_base[(8 * u + e) * TS] = x[u][e];
}

// sum over rows row0 .. row0+31 (row0 a multiple of 32) of feature `lane` of a transposed image
template <int TS = kTs>
__device__ __forceinline__ float colsum32_t(const float* imgT, int row0, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(imgT + lane * TS + row0);
    f32x4 s = p[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += p[i];
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// Backward of LayerNormalization for the lane's row: dy -> dsum = inv * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)).
// The parameter gradients need column sums over rows: dy * xhat goes to the lane's row of one transposed image
// (lane_a, see chain_store_t) and dy to a second one (the wavefront's own 32 rows of two idle images), from where
// colsum32_t takes them.
template <int TS = kTs>
__device__ __forceinline__ void ln_backward(const f32x4 (&dy)[8], const f32x4 (&xh)[8], float inv, const f32x4* __restrict__ g,
                                            int h, f32x4 (&dsum)[8], float* lane_a, float* lane_b) {
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
    chain_store_t<TS>(lane_a, t);
    chain_store_t<TS>(lane_b, dy);
    s1 = xhalf_sum(s1) * (1.0f / 64.0f);
    s2 = xhalf_sum(s2) * (1.0f / 64.0f);
#pragma unroll
    for (int u = 0; u < 8; ++u) dsum[u] = inv * (dsum[u] - s1 - xh[u] * s2);
}

// One 64 x 64 weight-gradient unit: part_u (+)= X^T Y over the rows of the tile, X and Y being LDS images.  The
// caller has put a barrier between writing the images and this call; the call ends with one barrier, after which
// the images (not xchg) may be overwritten.  xchg: 4096 floats of LDS that nothing else uses across the call.
// part_u: this workgroup's 4096 floats for the unit, accumulator order [block][register][lane].
// Rows at and beyond `rows_valid` hold zeros in both images (padding rows of a tile).  SHORT (the instances that
// small batches run): in a tile that is mostly padding -- one site in four wavefronts -- the contraction stops there.
template <bool SHORT>
__device__ __forceinline__ void dw_unit(const float* imgX, const float* imgY, int NW, float* xchg, float* __restrict__ part_u,
                                        bool first, int lane, int wave, int rows_valid) {
    const int hh = lane >> 5, i = lane & 31;
    if (NW >= 4) {
        // wavefront w: block w & 3 over row half w >> 2 (NW = 4: one "half" of all 128 rows)
        const int b = wave & 3, sp = wave >> 2;
        const float* ap = imgX + (sp * 128 + hh) * kLdsStride + 32 * (b >> 1) + i;
        const float* bp = imgY + (sp * 128 + hh) * kLdsStride + 32 * (b & 1) + i;
        float* dst = part_u + b * 1024 + lane;
        f32x16 acc;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
        if (sp == 0 && !first) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[rg] = dst[rg * 64];
        }
        if (!SHORT || rows_valid > sp * 128 + 96) {
#pragma unroll 8
            for (int q = 0; q < 64; ++q) mfma(acc, ap[2 * q * kLdsStride], bp[2 * q * kLdsStride]);
        } else {
            const int nsteps = max(0, (rows_valid - sp * 128 + 1) >> 1);
#pragma unroll 1
            for (int q = 0; q < nsteps; ++q) mfma(acc, ap[2 * q * kLdsStride], bp[2 * q * kLdsStride]);
        }
        if (sp == 1) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) xchg[(b * 16 + rg) * 64 + lane] = acc[rg];
        }
        __syncthreads();
        if (sp == 0) {
            if (NW == 8) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) acc[rg] += xchg[(b * 16 + rg) * 64 + lane];
            }
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) dst[rg * 64] = acc[rg];
        }
    } else {
        // one or two wavefronts: 4 / NW blocks each over all rows
        const int bpw = 4 / NW, steps = NW * 16;
        for (int bi = 0; bi < bpw; ++bi) {
            const int b = wave * bpw + bi;
            const float* ap = imgX + hh * kLdsStride + 32 * (b >> 1) + i;
            const float* bp = imgY + hh * kLdsStride + 32 * (b & 1) + i;
            float* dst = part_u + b * 1024 + lane;
            f32x16 acc;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[rg] = first ? 0.f : dst[rg * 64];
#pragma unroll 8
            for (int q = 0; q < steps; ++q) mfma(acc, ap[2 * q * kLdsStride], bp[2 * q * kLdsStride]);
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) dst[rg * 64] = acc[rg];
        }
        __syncthreads();
    }
}

#ifdef DPTM_EXP_NOSYNC
#define DW_SYNC() wave_fence()
#else
#define DW_SYNC() __syncthreads()
#endif
// dw_unit on transposed images: one 16-byte read per operand feeds four MFMA steps.  Step s of group Q contracts
// rows 8Q + s (lower half-wave) and 8Q + 4 + s (upper half-wave); every row of the tile is met once.
// Rows at and beyond `rows_valid` hold zeros in both images (padding rows of a tile).  SHORT (the instances that
// small batches run): in a four-wavefront tile that is mostly padding -- one site -- the contraction stops at the
// group of 8 rows that holds the last valid one.
template <bool SHORT>
__device__ __forceinline__ void dw_unit_t(const float* imgXT, const float* imgYT, int NW, float* xchg,
                                          float* __restrict__ part_u, bool first, int lane, int wave, int rows_valid) {
    const int hh = lane >> 5, i = lane & 31;
    const int nblk = NW >= 4 ? 1 : 4 / NW;                 // blocks per wavefront
    const int sp = NW >= 4 ? wave >> 2 : 0;                // row half (NW = 8)
    int nq = NW >= 4 ? 16 : NW * 4;                        // groups of 8 rows this wavefront contracts over
    if (SHORT && NW == 4 && rows_valid <= 96) nq = (rows_valid + 7) >> 3;
    for (int bi = 0; bi < nblk; ++bi) {
        const int b = NW >= 4 ? (wave & 3) : wave * nblk + bi;
        const float* ap = imgXT + (32 * (b >> 1) + i) * kTs + sp * 128 + 4 * hh;
        const float* bp = imgYT + (32 * (b & 1) + i) * kTs + sp * 128 + 4 * hh;
        float* dst = part_u + b * 1024 + lane;
        // the running sum of the partial buffer is requested now and added after the products, so that its
        // latency hides behind them
        f32x16 acc, prev;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = prev[rg] = 0.f;
#ifndef DPTM_EXP_NOPREV      // (experiment builds only: prices the read half of the partial-buffer read-modify-write)
        if (sp == 0 && !first) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) prev[rg] = dst[rg * 64];
        }
#endif
        // operands one group ahead of the four products that use them
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
        if (NW == 8) {
            if (sp == 1) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) xchg[(b * 16 + rg) * 64 + lane] = acc[rg];
            }
            DW_SYNC();
            if (sp == 0) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) acc[rg] += xchg[(b * 16 + rg) * 64 + lane];
            }
        }
#ifdef DPTM_EXP_NOWRITE      // (experiment builds only: no partial-buffer traffic at all but for the last tile's)
        if (sp == 0 && first) {
#else
        if (sp == 0) {
#endif
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) dst[rg * 64] = acc[rg] + prev[rg];
        }
    }
    if (NW != 8) DW_SYNC();
}

// The unit of the row-wise backward kernel: four wavefronts, 128 rows, transposed images of row stride kTsRow.  Wavefront
// w contracts block w over all 128 rows, so there is no exchange between row halves and no barrier inside; the caller
// puts one barrier between writing the images and this call and one before it overwrites them.
constexpr int kTsRow = 132;     // 128 rows + 4
__device__ __forceinline__ void dw_unit_row(const float* imgXT, const float* imgYT, float* __restrict__ part_u, bool first,
                                            int lane, int wave) {
    const int hh = lane >> 5, i = lane & 31;
    const int b = wave;
    const float* ap = imgXT + (32 * (b >> 1) + i) * kTsRow + 4 * hh;
    const float* bp = imgYT + (32 * (b & 1) + i) * kTsRow + 4 * hh;
    float* dst = part_u + b * 1024 + lane;
    f32x16 acc, prev;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) acc[rg] = prev[rg] = 0.f;
    if (!first) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) prev[rg] = dst[rg * 64];
    }
    f32x4 ca = *reinterpret_cast<const f32x4*>(ap), cb = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll 2
    for (int g = 0; g < 16; ++g) {
        f32x4 na = ca, nb = cb;
        if (g + 1 < 16) {
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
}

// scale | shift of a BatchNormalization in training mode from the column sums of its input (keras epsilon 1e-3,
// biased variance): bnv[c] = inv*gamma, bnv[64 + c] = beta - mean*inv*gamma.  Threads 0..63 of a workgroup; the
// caller puts a barrier behind it.  `publish`: also leave the batch mean / variance for the moving statistics.
__device__ __forceinline__ void bn_prepare(const BnFwd& bn, float* bnv, int tid, bool publish) {
    if (tid < 64) {
        const float mu = bn.stats[tid] * bn.inv_rows;
        float var = bn.stats[64 + tid] * bn.inv_rows - mu * mu;
        var = var < 0.f ? 0.f : var;
        const float sc = rsqrtf(var + 1e-3f) * bn.gamma[tid];
        bnv[tid] = sc;
        bnv[64 + tid] = bn.beta[tid] - mu * sc;
        if (publish) {
            bn.batch_mean[tid] = mu;
            bn.batch_var[tid] = var;
        }
    }
}

// input of the first TransformerBlock for the lane's row: x1 = BN2(swish(a2)) + s (model.py:138-141); zero on padding rows
__device__ __forceinline__ void x1_from_conv(const ConvSave& cs, int64_t ct, const float* bnv, int h, bool row_ok, f32x4 (&x1)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4 a = cs.a2[ct + u * 64], sv = cs.s[ct + u * 64];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * u + 4 * h);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * u + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) x1[u][e] = row_ok ? fmaf(swishf(a[e]), sc[e], sh[e]) + sv[e] : 0.f;
    }
}

// the same with the two arrays given as the wavefront's chain-tiled slab (scalar base) and the lane as a 32-bit offset
__device__ __forceinline__ void x1_from_conv(const f32x4* a2w, const f32x4* sw, unsigned ln, const float* bnv, int h, bool row_ok,
                                             f32x4 (&x1)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4 a = (a2w + u * 64)[ln], sv = (sw + u * 64)[ln];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 8 * u + 4 * h);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(bnv + 64 + 8 * u + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) x1[u][e] = row_ok ? fmaf(swishf(a[e]), sc[e], sh[e]) + sv[e] : 0.f;
    }
}

// Tiles of a backward launch with more tiles than workgroups are handed out through a counter instead of by workgroup
// index: the three branches' launches run side by side on the chip, a workgroup may start long after its launch began,
// and with a fixed share of the tiles per workgroup the late starters would decide when the launch ends.  Every thread
// of the workgroup calls this (two barriers inside; `slot` is one int of LDS); returns the tile, or a value >= ntiles
// when none is left.  The counter is zero when the launch starts (the step's first launch clears it).
// queue == nullptr (a launch with a workgroup per tile): call number `round` of the workgroup returns tile
// blockIdx.x + round * gridDim.x, without barriers.
__device__ __forceinline__ int64_t next_tile(unsigned* queue, int* slot, int& round) {
    if (!queue) return (int64_t)blockIdx.x + (int64_t)(round++) * gridDim.x;
    __syncthreads();          // the previous tile is finished and its slot value has been read
    if (threadIdx.x == 0) *slot = (int)atomicAdd(queue, 1u);
    __syncthreads();
    return *slot;
}

// a workgroup that was handed no tile at all still owns a partial buffer that the summing launch reads
__device__ __forceinline__ void zero_floats(float* __restrict__ dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = 0.f;
}

// ------------------------------------------------------------------------------------ launch helpers

template <typename K>
inline hipError_t lds_opt_in(K kernel, bool (&done)[64], size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

inline bool tile_ok(int wlen, int S, int NW, int64_t n, int64_t ntiles) {
    return (wlen == 21 || wlen == 33 || wlen == 51) && (NW == 1 || NW == 2 || NW == 4 || NW == 8) && S >= 1 && S * wlen <= NW * 32 &&
           ntiles == (n + S - 1) / S;
}

}  // namespace dptm
