// HIP kernels (gfx950) for one training step of DeepPTMPred's classifier: forward in
// training mode, backward and the AdamW update (reference: pred/train_PTM/trainer.py:
// 36-54, 304-339 on the graph of pred/train_PTM/model.py:143-210).
//
// Layer-wise design: activations live in HBM as row-major fp32 [rows][features]
// (rows = sample*W + position for the sequence branches, rows = samples for the
// per-site layers), weights and gradients in ONE flat fp32 buffer each, in the canonical
// tensor order (Keras shapes, unpacked).  Matrix products run on
// v_mfma_f32_32x32x2_f32 straight from global memory:
//   tgemm<NN>  Y  = shift(X) * W (+b)      forward dense / conv tap
//   tgemm<NT>  dX = shift(dY) * W^T        input gradients
//   tgemm_tn   dW += shift(X)^T * dY       weight gradients (+ bias gradient)
#include <hip/hip_runtime.h>

#include "dptm_train_device.h"
#include "dptm_train_kernels.h"

namespace dptm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma2(f32x16& acc, float a, float b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// row r of a [samples*wlen] stack shifted by s stays inside its own window?
__device__ __forceinline__ bool row_ok(int64_t r, int s, int64_t m, int wlen) {
    if (r >= m) return false;
    if (s == 0 || wlen == 0) return true;
    const int p = (int)(r % wlen) + s;
    return p >= 0 && p < wlen;
}

// ------------------------------------------------------------------------------------
// out[r][c0..c0+63] (+)= sum_j in[r+s][j] * Wm(j, c) (+ bias[c])
//   NN: Wm(j,c) = W[j*ldw + c]   (W is [K][N], contraction over its rows)
//   NT: Wm(j,c) = W[c*ldw + j]   (W is [C][J], contraction over its columns)
// One wavefront: 32 rows x 64 columns.  grid = (ceil(m/128), cols/64), block 256.
// ------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void tgemm_kernel(GemmArgs p) {
    if (p.batch > 1) {
        const int z = blockIdx.z;
        p.x += z * p.bx; p.w += z * p.bw; p.out += z * p.bo;
        if (p.bias) p.bias += z * p.bb;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (r0 >= p.m) return;
    const int c0 = blockIdx.y * 64;
    const int64_t r = r0 + j;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[t][4 * g + e] = p.bias ? p.bias[c0 + 32 * t + 8 * g + 4 * h + e] : 0.f;
    const int k8 = p.k >> 3;
    const int ntaps = p.taps > 1 ? p.taps : 1;
    for (int tp = 0; tp < ntaps; ++tp) {
        const int shift = p.taps > 1 ? p.tap_dir * (tp - p.taps / 2) : p.shift;
        const float* wm = p.w + tp * p.tap_stride;
        const bool ok = row_ok(r, shift, p.m, p.wlen);
        const float* xin = p.x + tp * p.x_tap_stride + (ok ? (r + shift) : 0) * p.ldx + 4 * h;
        // the contraction is walked in blocks of 64 inputs whose operands are all requested before the first
        // MFMA of the block: these layers are a few steps deep, so the loads' latency is the cost to hide
        for (int ub = 0; ub < k8; ub += 8) {
            f32x4 b[8], a[8][2];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int u = ub + v;
                b[v] = f32x4{0.f, 0.f, 0.f, 0.f};
                a[v][0] = a[v][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u < k8) {
                    if (ok) b[v] = *reinterpret_cast<const f32x4*>(xin + 8 * u);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int c = c0 + 32 * t + j;
                        if (NT) {
                            a[v][t] = *reinterpret_cast<const f32x4*>(wm + (int64_t)c * p.ldw + 8 * u + 4 * h);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[v][t][e] = wm[(int64_t)(8 * u + 4 * h + e) * p.ldw + c];
                        }
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < 2; ++t) mfma2(acc[t], a[v][t][e], b[v][e]);
        }
    }
    if (r >= p.m) return;
    float* orow = p.out + r * p.ldo + p.col_off + c0 + 4 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            f32x4* dst = reinterpret_cast<f32x4*>(orow + 32 * t + 8 * g);
            const int cc = c0 + 4 * h + 32 * t + 8 * g;
            if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + r * p.ldr + cc);
            if (p.accumulate) v += *dst;
            *dst = v;
            if (p.act_out) {
                f32x4 sw;
#pragma unroll
                for (int e = 0; e < 4; ++e) sw[e] = v[e] * sigmoidf_(v[e]);
                *reinterpret_cast<f32x4*>(p.act_out + r * p.lda + cc) = sw;
            }
        }
}

// ------------------------------------------------------------------------------------
// The same product for large row counts, built around what the first kernel spends its time on: the
// address unit.  There every wavefront fetches its operands row-per-lane (32 partial cache lines per
// load) and re-reads the weight tile for every 32 rows.  Here
//   * the weight tile of the block's 64 output columns (all taps, all of k) is packed ONCE per workgroup
//     into LDS in MFMA fragment order; a workgroup then walks many row tiles;
//   * an X tile (32 rows x 64 inputs) arrives with fully coalesced 16-byte loads, requested one segment
//     ahead, and is turned row-per-lane through a wavefront-private LDS stage;
//   * the output tile leaves the same way (coalesced, through the stage).
// A "segment" is one (tap, 64-input block) pair.  Needs k % 64 == 0 and taps * k <= 320 (LDS).
// grid = (workgroups, cols/64), block 512 = 8 wavefronts, dynamic LDS = taps*k*256 + 8*32*68*4 bytes.
// ------------------------------------------------------------------------------------
constexpr int kStageStride = 68;   // floats per staged row: row-per-lane ds_read_b128 is conflict-free

template <bool NT>
__global__ __launch_bounds__(512) void tgemm_lds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (p.batch > 1) {
        const int z = blockIdx.z;
        p.x += z * p.bx; p.w += z * p.bw; p.out += z * p.bo;
        if (p.bias) p.bias += z * p.bb;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int c0 = blockIdx.y * 64;
    const int ntaps = p.taps > 1 ? p.taps : 1;
    const int kblocks = p.k >> 6;
    const int nseg = ntaps * kblocks;
    f32x4* wl = reinterpret_cast<f32x4*>(lds);                          // [nseg][8][2][64] fragments
    float* stage = lds + (size_t)nseg * 8 * 2 * 64 * 4 + wave * 32 * kStageStride;

    // ---- weight fragments: element (seg, v, t, lane, e) = Wm_tap(64*kb + 8v + 4h + e, c0 + 32t + j)
    {
        float* wf = lds;
        const int per_seg = 64 * 64;
        for (int idx = tid; idx < nseg * per_seg; idx += 512) {
            const int seg = idx / per_seg, rem = idx - seg * per_seg;
            const int tp = seg / kblocks, kb = seg - tp * kblocks;
            int kk, cc;           // input index within the 64-block, output column within the 64-tile
            if (NT) { cc = rem >> 6; kk = rem & 63; } else { kk = rem >> 6; cc = rem & 63; }   // fastest index follows memory
            const float* wm = p.w + tp * p.tap_stride;
            const float val = NT ? wm[(int64_t)(c0 + cc) * p.ldw + 64 * kb + kk] : wm[(int64_t)(64 * kb + kk) * p.ldw + c0 + cc];
            const int v = kk >> 3, hh = (kk >> 2) & 1, e = kk & 3, t = cc >> 5, jj = cc & 31;
            wf[((((size_t)seg * 8 + v) * 2 + t) * 64 + 32 * hh + jj) * 4 + e] = val;
        }
    }
    __syncthreads();

    const int64_t ntile = (p.m + 31) / 32;
    // coalesced walk of a 32 x 64 tile: request i of a lane covers row (64 i + lane) / 16, floats 4*((64 i + lane) % 16)..
    const int crow = lane >> 4, ccol = (lane & 15) * 4;
    auto fetch = [&](f32x4 (&g)[8], int64_t r0, int seg) {
        const int tp = seg / kblocks, kb = seg - tp * kblocks;
        const int shift = p.taps > 1 ? p.tap_dir * (tp - p.taps / 2) : p.shift;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r0 + 4 * i + crow + shift;
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rr >= 0 && rr < p.m)
                g[i] = *reinterpret_cast<const f32x4*>(p.x + tp * p.x_tap_stride + rr * p.ldx + 64 * kb + ccol);
        }
    };
    const int64_t tstep = (int64_t)gridDim.x * 8;
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    f32x4 g[8], gn[8];
    if (tile < ntile) fetch(g, tile * 32, 0);
    for (; tile < ntile; tile += tstep) {
        const int64_t r0 = tile * 32;
        const int64_t r = r0 + j;
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[t][4 * g + e] = p.bias ? p.bias[c0 + 32 * t + 8 * g + 4 * h + e] : 0.f;
        for (int seg = 0; seg < nseg; ++seg) {
            // the next segment -- of this tile, or the first one of the wavefront's next tile -- is requested now
            const bool more = seg + 1 < nseg || tile + tstep < ntile;
            if (seg + 1 < nseg) fetch(gn, r0, seg + 1);
            else if (tile + tstep < ntile) fetch(gn, (tile + tstep) * 32, 0);
            const int tp = seg / kblocks;
            const int shift = p.taps > 1 ? p.tap_dir * (tp - p.taps / 2) : p.shift;
            const bool ok = row_ok(r, shift, p.m, p.wlen);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(stage + (4 * i + crow) * kStageStride + ccol) = g[i];
            wave_fence();
            f32x4 b[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                b[v] = *reinterpret_cast<const f32x4*>(stage + j * kStageStride + 8 * v + 4 * h);
                if (!ok) b[v] = f32x4{0.f, 0.f, 0.f, 0.f};     // a shifted row outside this row's window
            }
            wave_fence();
            const f32x4* ws = wl + (size_t)seg * 8 * 2 * 64 + lane;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const f32x4 a0 = ws[(v * 2 + 0) * 64], a1 = ws[(v * 2 + 1) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mfma2(acc[0], a0[e], b[v][e]);
                    mfma2(acc[1], a1[e], b[v][e]);
                }
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = gn[i];
            }
        }
        // ---- output tile through the stage: lane (row j, half h) holds columns 32t + 8g + 4h + e
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<f32x4*>(stage + j * kStageStride + 32 * t + 8 * gq + 4 * h) =
                    f32x4{acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]};
        wave_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r0 + 4 * i + crow;
            if (rr < p.m) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + (4 * i + crow) * kStageStride + ccol);
                f32x4* dst = reinterpret_cast<f32x4*>(p.out + rr * p.ldo + p.col_off + c0 + ccol);
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + rr * p.ldr + c0 + ccol);
                if (p.accumulate) v += *dst;
                *dst = v;
                if (p.act_out) {
                    f32x4 sw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) sw[e] = v[e] * sigmoidf_(v[e]);
                    *reinterpret_cast<f32x4*>(p.act_out + rr * p.lda + c0 + ccol) = sw;
                }
            }
        }
        wave_fence();
    }
}

// ------------------------------------------------------------------------------------
// dW[k][n] += sum_r X[r+s][k] * dY[r][n];  db[n] += sum_r dY[r][n] (blocks with k-tile 0).
// One wavefront: a 32(k) x 32(n) tile over a slab of rows; results added with float atomics.
// grid = (ceil(K/32), N/32, row slabs), block 64.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tgemm_tn_kernel(GemmTnArgs p) {
    const int zb = p.batch > 1 ? blockIdx.z / p.slabs : 0;
    const int zslab = blockIdx.z - zb * p.slabs;
    if (p.batch > 1) {
        p.x += zb * p.bx; p.dy += zb * p.bdy; p.dw += zb * p.bdw;
        if (p.db) p.db = p.bshift_dir ? (zb == 0 ? p.db : nullptr) : p.db + zb * p.bdb;   // taps share one bias
        if (p.bshift_dir) p.shift = p.bshift_dir * (zb - p.batch / 2);
    }
    const int lane = threadIdx.x, h = lane >> 5, i = lane & 31;
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int64_t rs = (int64_t)zslab * p.rows_per_slab;
    const int64_t re = min(p.m, rs + p.rows_per_slab);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool kin = k0 + i < p.k;
    // half-wave h takes row rr+h of every pair; four pairs per trip so that eight loads are in flight
    for (int64_t rr = rs; rr < re; rr += 8) {
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = rr + 2 * q + h;
            a[q] = 0.f;
            b[q] = 0.f;
            if (r < re) {
                b[q] = p.dy[r * p.ldy + p.dy_col_off + n0 + i];
                if (kin && row_ok(r, p.shift, p.m, p.wlen)) a[q] = p.x[(r + p.shift) * p.ldx + p.x_col_off + k0 + i];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mfma2(acc, a[q], b[q]);
            bsum += b[q];
        }
    }
    // D[row = k][col = n]: lane holds column n0+i, rows (reg&3) + 8*(reg>>2) + 4*h
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        const int k = k0 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
        if (k < p.k) atomicAdd(p.dw + (int64_t)k * p.ldw + n0 + i, acc[rg]);
    }
    if (p.db && blockIdx.x == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) atomicAdd(p.db + n0 + i, bsum);
    }
}

// ------------------------------------------------------------------------------------
// The weight gradient for large row stacks: one wavefront owns a whole 64(k) x 64(n) tile of dW (four MFMA
// tiles, so every operand value read feeds two products) and walks 32-row chunks of its slab.  X and dY
// chunks arrive with coalesced 16-byte loads, requested one chunk ahead, into wavefront-private LDS stages;
// rows that fall outside their window after the shift (or past the end) are zeroed as they are staged.
// The MFMA operands are then single-float LDS reads (lane = k or n index: conflict-free).
// grid = (k/64, n/64, slabs), block 256 = 4 wavefronts that interleave the slab's chunks.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tgemm_tn_lds_kernel(GemmTnArgs p) {
    const int zb = p.batch > 1 ? blockIdx.z / p.slabs : 0;
    const int zslab = blockIdx.z - zb * p.slabs;
    if (p.batch > 1) {
        p.x += zb * p.bx; p.dy += zb * p.bdy; p.dw += zb * p.bdw;
        if (p.db) p.db = p.bshift_dir ? (zb == 0 ? p.db : nullptr) : p.db + zb * p.bdb;   // taps share one bias
        if (p.bshift_dir) p.shift = p.bshift_dir * (zb - p.batch / 2);
    }
    __shared__ __attribute__((aligned(16))) float stage_all[4][2][32 * kStageStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int64_t rs = (int64_t)zslab * p.rows_per_slab;
    const int64_t re = min(p.m, rs + p.rows_per_slab);
    float* sx = stage_all[wave][0];
    float* sy = stage_all[wave][1];
    const int crow = lane >> 4, ccol = (lane & 15) * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum0 = 0.f, bsum1 = 0.f;
    auto fetch = [&](f32x4 (&gx)[8], f32x4 (&gy)[8], int64_t r0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t r = r0 + 4 * q + crow;
            gx[q] = gy[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < re) {
                gy[q] = *reinterpret_cast<const f32x4*>(p.dy + r * p.ldy + p.dy_col_off + n0 + ccol);
                if (row_ok(r, p.shift, p.m, p.wlen))
                    gx[q] = *reinterpret_cast<const f32x4*>(p.x + (r + p.shift) * p.ldx + p.x_col_off + k0 + ccol);
            }
        }
    };
    f32x4 gx[8], gy[8], nx[8], ny[8];
    int64_t r0 = rs + 32 * wave;
    if (r0 < re) fetch(gx, gy, r0);
    for (; r0 < re; r0 += 128) {
        const bool more = r0 + 128 < re;
        if (more) fetch(nx, ny, r0 + 128);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            *reinterpret_cast<f32x4*>(sx + (4 * q + crow) * kStageStride + ccol) = gx[q];
            *reinterpret_cast<f32x4*>(sy + (4 * q + crow) * kStageStride + ccol) = gy[q];
        }
        wave_fence();
        // half-wave h takes row 2q + h of every pair
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const float* xr = sx + (2 * q + h) * kStageStride + i;
            const float* yr = sy + (2 * q + h) * kStageStride + i;
            const float a0 = xr[0], a1 = xr[32], b0 = yr[0], b1 = yr[32];
            mfma2(acc[0][0], a0, b0);
            mfma2(acc[0][1], a0, b1);
            mfma2(acc[1][0], a1, b0);
            mfma2(acc[1][1], a1, b1);
            bsum0 += b0;
            bsum1 += b1;
        }
        wave_fence();
        if (more) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                gx[q] = nx[q];
                gy[q] = ny[q];
            }
        }
    }
    // The four wavefronts' partial tiles meet in LDS (the stages are free now) and one of them adds the block's sum
    // to dW: every float atomic lands on the same 64 x 64 values, so a quarter of them is a quarter of the queue.
    __syncthreads();
    float* red = &stage_all[0][0][0];                       // [3][64 regs][64 lanes]
    bsum0 += __shfl_xor(bsum0, 32);                         // both half-waves now hold the sum over their row pairs
    bsum1 += __shfl_xor(bsum1, 32);
    if (wave > 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) red[((wave - 1) * 64 + (a * 2 + b) * 16 + rg) * 64 + lane] = acc[a][b][rg];
        if (h == 0) {
            red[3 * 64 * 64 + ((wave - 1) * 2 + 0) * 32 + i] = bsum0;
            red[3 * 64 * 64 + ((wave - 1) * 2 + 1) * 32 + i] = bsum1;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // D[row = k][col = n]: lane holds column i, rows (reg&3) + 8*(reg>>2) + 4*h of each 32 x 32 tile
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int slot = (a * 2 + b) * 16 + rg;
                const float v = acc[a][b][rg] + red[(0 * 64 + slot) * 64 + lane] + red[(1 * 64 + slot) * 64 + lane] +
                                red[(2 * 64 + slot) * 64 + lane];
                const int k = k0 + 32 * a + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                atomicAdd(p.dw + (int64_t)k * p.ldw + n0 + 32 * b + i, v);
            }
    if (p.db && blockIdx.x == 0) {
        if (h == 0) {
            const float* rb = red + 3 * 64 * 64;
            atomicAdd(p.db + n0 + i, bsum0 + rb[0 * 32 + i] + rb[2 * 32 + i] + rb[4 * 32 + i]);
            atomicAdd(p.db + n0 + 32 + i, bsum1 + rb[1 * 32 + i] + rb[3 * 32 + i] + rb[5 * 32 + i]);
        }
    }
}

// ------------------------------------------------------------------------------------ elementwise

__global__ void swish_fwd_kernel(const float* __restrict__ a, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = a[i];
        y[i] = v * sigmoidf_(v);
    }
}

// da = dy * d/da[a*sigmoid(a)]   (in place on dy allowed)
__global__ void swish_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dy,
                                 float* __restrict__ da, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = a[i], s = sigmoidf_(v);
        da[i] = dy[i] * (s * (1.0f + v * (1.0f - s)));
    }
}
// n4 groups of four floats, 16-byte aligned
__global__ void swish_bwd4_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ dy, f32x4* __restrict__ da, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = a[i], g = dy[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s = sigmoidf_(v[e]);
            o[e] = g[e] * (s * (1.0f + v[e] * (1.0f - s)));
        }
        da[i] = o;
    }
}

__global__ void add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

// strided copy / add of a column block: dst[r][dc + c] (=|+=) src[r][sc + c]
__global__ void copy_cols_kernel(float* __restrict__ dst, int64_t ldd, int dc, const float* __restrict__ src,
                                 int64_t lds_, int sc, int64_t rows, int cols, int add) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = src[r * lds_ + sc + c];
        float* d = dst + r * ldd + dc + c;
        *d = add ? *d + v : v;
    }
}

// inverted dropout in place.  spatial: one mask value per (sample, channel) of a
// [samples][wlen][cols] stack; otherwise per element.  ld = row stride, cols = row width.
__global__ void dropout_kernel(float* __restrict__ x, const float* __restrict__ src, int64_t rows, int cols, int64_t ld,
                               int wlen_spatial, const unsigned* __restrict__ step_params, unsigned stream, unsigned thr,
                               float scale) {
    const unsigned seed = step_params[kStepSeed];
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const unsigned idx = wlen_spatial ? (unsigned)((r / wlen_spatial) * cols + c) : (unsigned)i;
        const float v = src[r * ld + c];
        x[r * ld + c] = hash_u32(seed, stream, idx) >= thr ? v * scale : 0.f;
    }
}

// The same for the common shape (cols a multiple of 4, 16-byte aligned rows, fewer than 2^31 elements): four
// columns per thread, 32-bit index arithmetic.  Element indices -- and with them the masks -- are unchanged.
__global__ void dropout4_kernel(float* __restrict__ x, const float* __restrict__ src, int rows, int cols, int ld,
                                int wlen_spatial, const unsigned* __restrict__ step_params, unsigned stream, unsigned thr,
                                float scale) {
    const unsigned seed = step_params[kStepSeed];
    const int c4n = cols >> 2;
    const int total = rows * c4n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / c4n;
        const int c = (i - r * c4n) << 2;
        const unsigned base = wlen_spatial ? (unsigned)((r / wlen_spatial) * cols + c) : (unsigned)(r * cols + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hash_u32(seed, stream, base + e) >= thr ? v[e] * scale : 0.f;
        *reinterpret_cast<f32x4*>(x + (int64_t)r * ld + c) = o;
    }
}

// ------------------------------------------------------------------------------------ embedding

// x[r][c] = E[tok[r]][c] + P[r % wlen][c]                       (model.py:157-158)
__global__ void embed_fwd_kernel(const uint8_t* __restrict__ tok, const float* __restrict__ E,
                                 const float* __restrict__ P, float* __restrict__ x, int64_t rows, int wlen) {
    const int64_t total = rows * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 6;
        const int c = (int)(i & 63);
        int t = tok[r];
        t = t > 20 ? 20 : t;
        x[i] = E[t * 64 + c] + P[(r % wlen) * 64 + c];
    }
}

// dE[tok[r]][c] += dx[r][c];  dP[r % wlen][c] += dx[r][c].  Each block first sums its rows
// per (code / position) in LDS, then issues one atomic per touched table entry.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const uint8_t* __restrict__ tok, const float* __restrict__ dx,
                                                        float* __restrict__ dE, float* __restrict__ dP,
                                                        int64_t rows, int wlen, int rows_per_block) {
    __shared__ float sE[21 * 64];
    __shared__ float sP[51 * 64];
    for (int i = threadIdx.x; i < 21 * 64; i += 256) sE[i] = 0.f;
    for (int i = threadIdx.x; i < 51 * 64; i += 256) sP[i] = 0.f;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    const int c = threadIdx.x & 63;
    for (int64_t r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
        int t = tok[r];
        t = t > 20 ? 20 : t;
        const float g = dx[r * 64 + c];
        atomicAdd(&sE[t * 64 + c], g);
        atomicAdd(&sP[(int)(r % wlen) * 64 + c], g);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 21 * 64; i += 256)
        if (sE[i] != 0.f) atomicAdd(dE + i, sE[i]);
    for (int i = threadIdx.x; i < wlen * 64; i += 256)
        if (sP[i] != 0.f) atomicAdd(dP + i, sP[i]);
}

// ------------------------------------------------------------------------------------ BatchNorm (training)

// stats[c] += sum_r x[r][c]; stats[C + c] += sum_r x[r][c]^2.  block (64, 4), C <= 512
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      int64_t rows, int cols, int64_t ld, float* __restrict__ stats,
                                                      int rows_per_block) {
    // with y == nullptr: sums of x and x*x; otherwise sums of x and x*y.  grid.x covers the columns (64 per block)
    __shared__ float red[2][4][64];
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    const int l = threadIdx.x & 63, q4 = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    float s1 = 0.f, s2 = 0.f;
    if (c < cols) {
        int64_t r = r0 + q4;
        for (; r + 12 < r1; r += 16) {          // four rows' loads in flight
            float v[4], w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = x[(r + 4 * q) * ld + c];
                w[q] = y ? y[(r + 4 * q) * ld + c] : v[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1 += v[q];
                s2 += v[q] * w[q];
            }
        }
        for (; r < r1; r += 4) {
            const float v = x[r * ld + c];
            s1 += v;
            s2 += v * (y ? y[r * ld + c] : v);
        }
    }
    // the block's four wavefronts meet in LDS: a quarter of the atomics on the same addresses (they serialise there)
    red[0][q4][l] = s1;
    red[1][q4][l] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 6;
        if (c < cols) atomicAdd(stats + k * cols + c, (red[k][0][l] + red[k][1][l]) + (red[k][2][l] + red[k][3][l]));
    }
}

// The 64-column case of the two column-sum kernels (the BatchNorms of the sequence branches, where the rows are
// many): 16 lanes per row with 16-byte loads, 16 rows per trip, partial sums met in LDS.
// MODE 0: stats = [sum x | sum x^2];  MODE 1: sums = [sum dz | sum dz * xhat] with xhat from y, mean, inv.
template <int MODE>
__global__ __launch_bounds__(256) void colsum64_kernel(const float* __restrict__ x, int64_t ldx, int x_col_off,
                                                       const float* __restrict__ y, const float* __restrict__ mean_inv,
                                                       int64_t rows, float* __restrict__ stats, int rows_per_block) {
    __shared__ f32x4 p1[16][16], p2[16][16];
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, inv = mu;
    if (MODE == 1) {
        mu = *reinterpret_cast<const f32x4*>(mean_inv + 4 * sub);
        inv = *reinterpret_cast<const f32x4*>(mean_inv + 64 + 4 * sub);
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    for (int64_t r = r0 + slot; r < r1; r += 16) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + x_col_off + 4 * sub);
        s1 += v;
        if (MODE == 0) {
            s2 += v * v;
        } else {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + r * 64 + 4 * sub);
            s2 += v * (yy - mu) * inv;
        }
    }
    p1[slot][sub] = s1;
    p2[slot][sub] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int f4 = threadIdx.x >> 2, e = threadIdx.x & 3;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            t1 += p1[q][f4][e];
            t2 += p2[q][f4][e];
        }
        atomicAdd(stats + threadIdx.x, t1);
        atomicAdd(stats + 64 + threadIdx.x, t2);
    }
}

// z = (y - mean) * inv * gamma + beta with mean / inv = rsqrt(var + eps) taken straight from the column sums
// (stats = [sum | sum of squares] over inv_rows^-1 rows).  The first workgroup also leaves mean | inv in
// mean_inv for the backward pass and the batch mean / (biased) variance for the moving-statistics update.
__global__ void bn_fwd_kernel(const float* __restrict__ y, float* __restrict__ z, int64_t rows, int cols,
                              const float* __restrict__ stats, float inv_rows, float* __restrict__ mean_inv,
                              float* __restrict__ batch_mean, float* __restrict__ batch_var,
                              const float* __restrict__ gamma, const float* __restrict__ beta, int64_t ldz, int z_col_off) {
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < cols; c += blockDim.x) {
            const float mu = stats[c] * inv_rows;
            float var = stats[cols + c] * inv_rows - mu * mu;
            var = var < 0.f ? 0.f : var;
            mean_inv[c] = mu;
            mean_inv[cols + c] = rsqrtf(var + 1e-3f);
            batch_mean[c] = mu;
            batch_var[c] = var;
        }
    }
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float mu = stats[c] * inv_rows;
        float var = stats[cols + c] * inv_rows - mu * mu;
        var = var < 0.f ? 0.f : var;
        z[r * ldz + z_col_off + c] = (y[i] - mu) * rsqrtf(var + 1e-3f) * gamma[c] + beta[c];
    }
}

// sums[c] += sum_r dz[r][c]; sums[C+c] += sum_r dz[r][c] * xhat[r][c]   (xhat from y, mean, inv)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, int64_t lddz, int dz_col_off,
                                                            const float* __restrict__ y, int64_t rows, int cols,
                                                            const float* __restrict__ mean_inv, float* __restrict__ sums,
                                                            int rows_per_block) {
    __shared__ float red[2][4][64];
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    const int l = threadIdx.x & 63, q4 = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    float s1 = 0.f, s2 = 0.f;
    if (c < cols) {
        const float mu = mean_inv[c], inv = mean_inv[cols + c];
        int64_t r = r0 + q4;
        for (; r + 12 < r1; r += 16) {          // four rows' loads in flight
            float g[4], v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g[q] = dz[(r + 4 * q) * lddz + dz_col_off + c];
                v[q] = y[(r + 4 * q) * cols + c];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1 += g[q];
                s2 += g[q] * (v[q] - mu) * inv;
            }
        }
        for (; r < r1; r += 4) {
            const float g = dz[r * lddz + dz_col_off + c];
            s1 += g;
            s2 += g * (y[r * cols + c] - mu) * inv;
        }
    }
    red[0][q4][l] = s1;
    red[1][q4][l] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 6;
        if (c < cols) atomicAdd(sums + k * cols + c, (red[k][0][l] + red[k][1][l]) + (red[k][2][l] + red[k][3][l]));
    }
}

// dy = gamma*inv * (dz - sum_dz/M - xhat * sum_dz_xhat/M) with the sums (and M) of the whole batch; also
// dgamma += sum_dz_xhat, dbeta += sum_dz from own_sums, this replica's share (the parameter gradients are
// added by the first block only).  On one device sums == own_sums.
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dz, int64_t lddz, int dz_col_off, const float* __restrict__ y,
                                    float* __restrict__ dy, int64_t rows, int cols, const float* __restrict__ mean_inv,
                                    const float* __restrict__ gamma, const float* __restrict__ sums, float inv_rows,
                                    const float* __restrict__ own_sums, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float inv = mean_inv[cols + c];
        const float xh = (y[i] - mean_inv[c]) * inv;
        dy[i] = gamma[c] * inv * (dz[r * lddz + dz_col_off + c] - sums[c] * inv_rows - xh * sums[cols + c] * inv_rows);
        if (i < cols) {
            atomicAdd(dgamma + c, own_sums[cols + c]);
            atomicAdd(dbeta + c, own_sums[c]);
        }
    }
}

// ------------------------------------------------------------------------------------ LayerNorm (64 features)

// One wavefront per row: out = LN(a + b) * g + beta; saves xhat and inv for the backward.
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ g, const float* __restrict__ beta,
                                                     float* __restrict__ out, float* __restrict__ xhat,
                                                     float* __restrict__ inv_out, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float v = a[r * 64 + lane] + b[r * 64 + lane];
    float s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / 64.0f);
    const float d = v - mean;
    float q = d * d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float inv = rsqrtf(q * (1.0f / 64.0f) + 1e-6f);
    const float xh = d * inv;
    xhat[r * 64 + lane] = xh;
    out[r * 64 + lane] = xh * g[lane] + beta[lane];
    if (lane == 0) inv_out[r] = inv;
}

// dsum = inv * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); dg += sum dy*xhat; dbeta += sum dy
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ inv_in, const float* __restrict__ g,
                                                     float* __restrict__ dsum, float* __restrict__ dg,
                                                     float* __restrict__ dbeta, int64_t rows, int rows_per_block) {
    // 16 lanes per row, four features each (16-byte accesses): a wavefront takes four rows per trip and the
    // two row sums cost four exchange steps instead of six
    __shared__ f32x4 sg[16][16], sb[16][16];
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;      // feature group, row slot 0..15
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    const f32x4 gl = *reinterpret_cast<const f32x4*>(g + 4 * sub);
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    for (int64_t rb = r0; rb < r1; rb += 16) {
        const int64_t r = rb + slot;
        const bool ok = r < r1;
        f32x4 d = {0.f, 0.f, 0.f, 0.f}, xh = d;
        if (ok) {
            d = *reinterpret_cast<const f32x4*>(dy + r * 64 + 4 * sub);
            xh = *reinterpret_cast<const f32x4*>(xhat + r * 64 + 4 * sub);
        }
        const f32x4 dgv = d * gl;
        float s1 = (dgv[0] + dgv[1]) + (dgv[2] + dgv[3]);
        float s2 = (dgv[0] * xh[0] + dgv[1] * xh[1]) + (dgv[2] * xh[2] + dgv[3] * xh[3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (ok) {
            const float inv = inv_in[r];
            *reinterpret_cast<f32x4*>(dsum + r * 64 + 4 * sub) = inv * (dgv - s1 * (1.0f / 64.0f) - xh * (s2 * (1.0f / 64.0f)));
        }
        ag += d * xh;
        ab += d;
    }
    sg[slot][sub] = ag;
    sb[slot][sub] = ab;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int f4 = threadIdx.x >> 2, e = threadIdx.x & 3;
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            tg += sg[q][f4][e];
            tb += sb[q][f4][e];
        }
        atomicAdd(dg + threadIdx.x, tg);
        atomicAdd(dbeta + threadIdx.x, tb);
    }
}

// ------------------------------------------------------------------------------------ attention (training)

// One wavefront per (sample, head).  q,k,v,o: [rows][64] with head hd in columns 16hd..;
// prob: [samples][4][wlen][wlen].   (model.py:67-82)
// Lane = query row.  The score row lives in LDS (row stride 53: conflict-free for row-per-lane access)
// until it is normalised; the probabilities then leave as one linear, coalesced copy.
__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ o,
                                                      float* __restrict__ prob, int wlen) {
    __shared__ float sk[51][17], sv[51][17], sp[51][53];
    const int b = blockIdx.x >> 2, hd = blockIdx.x & 3, lane = threadIdx.x;
    const int64_t base = (int64_t)b * wlen;
    for (int i = lane; i < wlen * 16; i += 64) {
        const int t = i >> 4, f = i & 15;
        sk[t][f] = k[(base + t) * 64 + 16 * hd + f];
        sv[t][f] = v[(base + t) * 64 + 16 * hd + f];
    }
    __syncthreads();
    if (lane < wlen) {
        float qr[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) qr[f] = q[(base + lane) * 64 + 16 * hd + f] * 0.25f;
        float* pr = sp[lane];
        float m = -3.0e38f;
        for (int t = 0; t < wlen; ++t) {
            float s = 0.f;
#pragma unroll
            for (int f = 0; f < 16; ++f) s = fmaf(qr[f], sk[t][f], s);
            pr[t] = s;
            m = fmaxf(m, s);
        }
        float den = 0.f;
        for (int t = 0; t < wlen; ++t) {
            const float e = __expf(pr[t] - m);
            pr[t] = e;
            den += e;
        }
        const float rd = 1.0f / den;
        float acc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = 0.f;
        for (int t = 0; t < wlen; ++t) {
            const float p_ = pr[t] * rd;
            pr[t] = p_;
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[f] = fmaf(p_, sv[t][f], acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) o[(base + lane) * 64 + 16 * hd + f] = acc[f];
    }
    __syncthreads();
    float* pb = prob + ((int64_t)b * 4 + hd) * wlen * wlen;
    for (int i = lane; i < wlen * wlen; i += 64) pb[i] = sp[i / wlen][i % wlen];
}

// dq, dk, dv from do (all [rows][64], head columns), with the saved probabilities.  One LDS image of the
// W x W matrix serves twice (first P, then dS written over it row by row), which keeps the workgroup at
// 25 KB of LDS; the price is that dP = dO V^T is formed twice per row.
__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const float* __restrict__ prob,
                                                      const float* __restrict__ d_o, float* __restrict__ dq,
                                                      float* __restrict__ dk, float* __restrict__ dv, int wlen) {
    __shared__ float sq[51][17], sk[51][17], sv[51][17], sdo[51][17];
    __shared__ float sp[51][53];   // odd row stride: row-per-lane and column-per-lane walks are both conflict-free
    const int b = blockIdx.x >> 2, hd = blockIdx.x & 3, lane = threadIdx.x;
    const int64_t base = (int64_t)b * wlen;
    for (int i = lane; i < wlen * 16; i += 64) {
        const int t = i >> 4, f = i & 15;
        const int64_t g = (base + t) * 64 + 16 * hd + f;
        sq[t][f] = q[g];
        sk[t][f] = k[g];
        sv[t][f] = v[g];
        sdo[t][f] = d_o[g];
    }
    const float* pb = prob + ((int64_t)b * 4 + hd) * wlen * wlen;
    for (int i = lane; i < wlen * wlen; i += 64) sp[i / wlen][i % wlen] = pb[i];
    __syncthreads();
    if (lane < wlen) {   // key row `lane`: dV = P^T dO
        float av[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) av[f] = 0.f;
        for (int i = 0; i < wlen; ++i) {
            const float p_ = sp[i][lane];
#pragma unroll
            for (int f = 0; f < 16; ++f) av[f] = fmaf(p_, sdo[i][f], av[f]);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) dv[(base + lane) * 64 + 16 * hd + f] = av[f];
    }
    __syncthreads();
    if (lane < wlen) {   // query row `lane`: dP, dS (over P) and dQ
        float dot = 0.f;
        for (int t = 0; t < wlen; ++t) {
            float dp = 0.f;
#pragma unroll
            for (int f = 0; f < 16; ++f) dp = fmaf(sdo[lane][f], sv[t][f], dp);
            dot = fmaf(dp, sp[lane][t], dot);
        }
        float acc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = 0.f;
        for (int t = 0; t < wlen; ++t) {
            float dp = 0.f;
#pragma unroll
            for (int f = 0; f < 16; ++f) dp = fmaf(sdo[lane][f], sv[t][f], dp);
            const float ds = sp[lane][t] * (dp - dot);
            sp[lane][t] = ds;
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[f] = fmaf(ds, sk[t][f], acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) dq[(base + lane) * 64 + 16 * hd + f] = acc[f] * 0.25f;
    }
    __syncthreads();
    if (lane < wlen) {   // key row `lane`: dK = dS^T Q
        float ak[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) ak[f] = 0.f;
        for (int i = 0; i < wlen; ++i) {
            const float ds = sp[i][lane];
#pragma unroll
            for (int f = 0; f < 16; ++f) ak[f] = fmaf(ds, sq[i][f], ak[f]);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) dk[(base + lane) * 64 + 16 * hd + f] = ak[f] * 0.25f;
    }
}

// ------------------------------------------------------------------------------------ attention pooling

// One wavefront per sample: a_t = tanh(x_t.w + b), alpha = softmax_t(a), out = sum alpha_t x_t.
// Saves a_t and alpha_t ([samples][wlen] each).                        (model.py:176-182)
__global__ __launch_bounds__(64) void pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int64_t ldo, int col_off, float* __restrict__ a_save,
                                                      float* __restrict__ alpha_save, int wlen) {
    __shared__ float sa[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t base = (int64_t)b * wlen;
    float a = -3.0e38f;
    if (lane < wlen) {
        float s = bias[0];
        for (int f = 0; f < 64; ++f) s = fmaf(x[(base + lane) * 64 + f], w[f], s);
        a = tanhf(s);
        a_save[base + lane] = a;
    }
    float m = a;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e = lane < wlen ? __expf(a - m) : 0.f;
    float den = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
    const float al = e / den;
    sa[lane] = al;
    if (lane < wlen) alpha_save[base + lane] = al;
    __syncthreads();
    float acc = 0.f;   // lane = feature
    for (int t = 0; t < wlen; ++t) acc = fmaf(sa[t], x[(base + t) * 64 + lane], acc);
    out[(int64_t)b * ldo + col_off + lane] = acc;
}

// dx_t = alpha_t*dout + dz_t*w with dz_t = alpha_t*(dalpha_t - sum alpha dalpha)*(1 - a_t^2),
// dalpha_t = dout.x_t; dw += sum_t dz_t x_t; db += sum_t dz_t.
__global__ __launch_bounds__(64) void pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ dout, int64_t ldd, int col_off,
                                                      const float* __restrict__ a_save, const float* __restrict__ alpha_save,
                                                      float* __restrict__ dx, float* __restrict__ dw,
                                                      float* __restrict__ db, int wlen) {
    __shared__ float sdz[64], sal[64], sdo[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t base = (int64_t)b * wlen;
    sdo[lane] = dout[(int64_t)b * ldd + col_off + lane];
    __syncthreads();
    float al = 0.f, dal = 0.f, a = 0.f;
    if (lane < wlen) {
        al = alpha_save[base + lane];
        a = a_save[base + lane];
        for (int f = 0; f < 64; ++f) dal = fmaf(sdo[f], x[(base + lane) * 64 + f], dal);
    }
    float dot = al * dal;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    const float dz = al * (dal - dot) * (1.0f - a * a);
    sdz[lane] = lane < wlen ? dz : 0.f;
    sal[lane] = al;
    float dzs = lane < wlen ? dz : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dzs += __shfl_xor(dzs, o);
    __syncthreads();
    const float wl = w[lane], dol = sdo[lane];   // lane = feature
    float gw = 0.f;
    for (int t = 0; t < wlen; ++t) {
        const float xv = x[(base + t) * 64 + lane];
        dx[(base + t) * 64 + lane] = sal[t] * dol + sdz[t] * wl;
        gw = fmaf(sdz[t], xv, gw);
    }
    atomicAdd(dw + lane, gw);
    if (lane == 0) atomicAdd(db, dzs);
}

// ------------------------------------------------------------------------------------ head + loss

// Per sample: logits = x.W + b (W [256][2]), softmax, class-balanced BCE (trainer.py:36-54), d(logits), d(x).
// A wavefront walks samples wave, wave + nwaves, ...; the gradients of W and b are summed in registers over the
// wavefront's samples, then over the workgroup in LDS, and leave through one set of atomics per workgroup.
__global__ __launch_bounds__(256) void head_loss_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const int32_t* __restrict__ labels,
                                                        int64_t n, const unsigned* __restrict__ step_params,
                                                        float* __restrict__ proba, float* __restrict__ loss,
                                                        float* __restrict__ dx, float* __restrict__ dW,
                                                        float* __restrict__ db) {
    __shared__ float red[4][8][64];
    __shared__ float red_s[4][4];
    const float n_global = __uint_as_float(step_params[kStepNGlobal]);
    const float n_pos_global = __uint_as_float(step_params[kStepNPosGlobal]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w0[i] = W[(lane + 64 * i) * 2];
        w1[i] = W[(lane + 64 * i) * 2 + 1];
    }
    const float b0 = bias[0], b1 = bias[1];
    float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f};
    float sl = 0.f, sd0 = 0.f, sd1 = 0.f;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < n; s += (int64_t)gridDim.x * 4) {
        float l0 = 0.f, l1 = 0.f;
        float xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xv[i] = x[s * 256 + lane + 64 * i];
            l0 = fmaf(xv[i], w0[i], l0);
            l1 = fmaf(xv[i], w1[i], l1);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            l0 += __shfl_xor(l0, o);
            l1 += __shfl_xor(l1, o);
        }
        l0 += b0;
        l1 += b1;
        const float m = fmaxf(l0, l1);
        const float e0 = __expf(l0 - m), e1 = __expf(l1 - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const float eps = 1e-7f;
        const int y = labels[s];
        const float wt = y ? n_global / (2.0f * n_pos_global + eps) : n_global / (2.0f * (n_global - n_pos_global) + eps);
        // keras binary_crossentropy over both softmax columns with the one-hot target.  For a
        // 2-class softmax 1-p0 = p1, so the four log terms collapse to -2*log(c_y + eps) with
        // c_y the clipped probability of the true class; this form is algebraically the same
        // as the literal one and avoids its 1-p cancellation in fp32.
        const float py = y ? p1 : p0;
        const float cy = fminf(fmaxf(py, eps), 1.0f - eps);
        const float bce = -2.0f * logf(cy + eps);
        const float scale = wt / (n_global * 2.0f);
        // d(scale*bce)/d(logit_i) = 2*scale * py/(cy+eps) * (p_i - [i == y]); zero where the clip is active
        const float gy = (py > eps && py < 1.0f - eps) ? 2.0f * scale * py / (cy + eps) : 0.f;
        const float d0 = gy * (p0 - (y ? 0.f : 1.f)), d1 = gy * (p1 - (y ? 1.f : 0.f));
        if (lane == 0) {
            proba[s * 2] = p0;
            proba[s * 2 + 1] = p1;
        }
        sl += scale * bce;
        sd0 += d0;
        sd1 += d1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dx[s * 256 + lane + 64 * i] = d0 * w0[i] + d1 * w1[i];
            g0[i] = fmaf(d0, xv[i], g0[i]);
            g1[i] = fmaf(d1, xv[i], g1[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[wave][i][lane] = g0[i];
        red[wave][4 + i][lane] = g1[i];
    }
    if (lane == 0) {
        red_s[wave][0] = sl;
        red_s[wave][1] = sd0;
        red_s[wave][2] = sd1;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 512; o += 256) {
        const int c = o >> 8, f = o & 255;          // class, feature
        const float v = red[0][4 * c + (f >> 6)][f & 63] + red[1][4 * c + (f >> 6)][f & 63] + red[2][4 * c + (f >> 6)][f & 63] +
                        red[3][4 * c + (f >> 6)][f & 63];
        if (v != 0.f) atomicAdd(dW + f * 2 + c, v);
    }
    if (threadIdx.x < 3) {
        float v = red_s[0][threadIdx.x] + red_s[1][threadIdx.x] + red_s[2][threadIdx.x] + red_s[3][threadIdx.x];
        if (threadIdx.x == 0 && blockIdx.x == 0) v += __uint_as_float(step_params[kStepL2 + (step_params[kStepL2Sel] & 1u)]);   // l2 regularisers (pack_kernel)
        atomicAdd(threadIdx.x == 0 ? loss : db + threadIdx.x - 1, v);
    }
}

// ------------------------------------------------------------------------------------ optimiser

// tensorflow-addons AdamW: w -= wd*w, then keras Adam with lr_t = lr*sqrt(1-b2^t)/(1-b1^t).
// trainable[i] = 0 marks the BatchNorm moving statistics: their slot of the gradient buffer
// holds the batch statistic (summed over replicas), and they move with momentum 0.99.
__global__ void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, const uint8_t* __restrict__ trainable, float stat_scale,
                             int64_t n, float lr_t, float wd, float b1, float b2, float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!trainable[i]) {
            w[i] = 0.99f * w[i] + 0.01f * (g[i] * stat_scale);
            continue;
        }
        const float gi = g[i] * gscale;
        float wi = w[i];
        wi -= wd * wi;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ------------------------------------------------------------------------------------ launchers

static inline unsigned grid_for(int64_t n, int block = 256, int64_t cap = 256 * 16) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

hipError_t t_gemm(const GemmArgs& a, bool nt, hipStream_t st) {
    if (a.m == 0) return hipSuccess;
    if (a.cols % 64 != 0 || a.k % 8 != 0) return hipErrorInvalidValue;
    const int ntaps = a.taps > 1 ? a.taps : 1;
    // large row stacks whose weight tile fits LDS take the staged kernel; operands must allow 16-byte accesses
    const bool aligned = a.ldx % 4 == 0 && a.ldo % 4 == 0 && a.col_off % 4 == 0 && a.x_tap_stride % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                         (!a.residual || (a.ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(a.residual) & 15) == 0)) &&
                         (!a.act_out || (a.lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a.act_out) & 15) == 0));
    if (a.m >= 4096 && a.k % 64 == 0 && ntaps * a.k <= 320 && aligned) {
        const size_t lds = (size_t)ntaps * a.k * 256 + (size_t)8 * 32 * kStageStride * sizeof(float);
        static bool attr_done[64][2] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!attr_done[dev][nt ? 1 : 0]) {
            const void* fn = nt ? reinterpret_cast<const void*>(&tgemm_lds_kernel<true>) : reinterpret_cast<const void*>(&tgemm_lds_kernel<false>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 320 * 256 + 8 * 32 * kStageStride * 4);
            if (e != hipSuccess) return e;
            attr_done[dev][nt ? 1 : 0] = true;
        }
        const int64_t tiles8 = (a.m + 255) / 256;
        const unsigned ycols = (unsigned)(a.cols / 64);
        int64_t gx = 256 / ycols;                     // one workgroup per CU over all column tiles (measured best of 128..2048)
        if (gx < 1) gx = 1;
        if (gx > tiles8) gx = tiles8;
        dim3 grid((unsigned)gx, ycols, (unsigned)(a.batch > 1 ? a.batch : 1));
        if (nt)
            hipLaunchKernelGGL(tgemm_lds_kernel<true>, grid, dim3(512), lds, st, a);
        else
            hipLaunchKernelGGL(tgemm_lds_kernel<false>, grid, dim3(512), lds, st, a);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((a.m + 127) / 128), (unsigned)(a.cols / 64), (unsigned)(a.batch > 1 ? a.batch : 1));
    if (nt)
        hipLaunchKernelGGL(tgemm_kernel<true>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(tgemm_kernel<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t t_gemm_tn(GemmTnArgs a, hipStream_t st) {
    if (a.m == 0) return hipSuccess;
    if (a.n % 32 != 0) return hipErrorInvalidValue;
    const bool aligned = a.ldx % 4 == 0 && a.ldy % 4 == 0 && a.x_col_off % 4 == 0 && a.dy_col_off % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0;
    if (a.m >= 4096 && a.k % 64 == 0 && a.n % 64 == 0 && aligned) {
        // ~2048 wavefronts over (tiles x slabs x 4), every wavefront with at least 128 rows of its own
        const int64_t tiles64 = (int64_t)(a.k / 64) * (a.n / 64);
        int64_t slabs = (512 + tiles64 - 1) / tiles64;
        if (slabs > (a.m + 511) / 512) slabs = (a.m + 511) / 512;
        if (slabs < 1) slabs = 1;
        a.rows_per_slab = (a.m + slabs - 1) / slabs;
        a.rows_per_slab = (a.rows_per_slab + 127) & ~(int64_t)127;   // whole 32-row chunks for all four wavefronts
        slabs = (a.m + a.rows_per_slab - 1) / a.rows_per_slab;
        a.slabs = (int)slabs;
        dim3 grid((unsigned)(a.k / 64), (unsigned)(a.n / 64), (unsigned)(slabs * (a.batch > 1 ? a.batch : 1)));
        hipLaunchKernelGGL(tgemm_tn_lds_kernel, grid, dim3(256), 0, st, a);
        return hipGetLastError();
    }
    // enough row slabs that tiles x slabs fills the chip (~2048 wavefronts), but at least 64 rows each
    const int64_t tiles = (int64_t)((a.k + 31) / 32) * (a.n / 32);
    int64_t slabs = (2048 + tiles - 1) / tiles;
    if (slabs > (a.m + 63) / 64) slabs = (a.m + 63) / 64;
    if (slabs < 1) slabs = 1;
    a.rows_per_slab = (a.m + slabs - 1) / slabs;
    a.rows_per_slab = (a.rows_per_slab + 7) & ~(int64_t)7;   // multiple of 8: four row pairs per loop trip
    slabs = (a.m + a.rows_per_slab - 1) / a.rows_per_slab;
    a.slabs = (int)slabs;
    dim3 grid((unsigned)((a.k + 31) / 32), (unsigned)(a.n / 32), (unsigned)(slabs * (a.batch > 1 ? a.batch : 1)));
    hipLaunchKernelGGL(tgemm_tn_kernel, grid, dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t t_swish_fwd(const float* a, float* y, int64_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(swish_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, y, n);
    return hipGetLastError();
}
hipError_t t_swish_bwd(const float* a, const float* dy, float* da, int64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(da)) & 15) == 0)
        hipLaunchKernelGGL(swish_bwd4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(a),
                           reinterpret_cast<const f32x4*>(dy), reinterpret_cast<f32x4*>(da), n / 4);
    else
        hipLaunchKernelGGL(swish_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, dy, da, n);
    return hipGetLastError();
}
hipError_t t_add(float* dst, const float* src, int64_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}
hipError_t t_copy_cols(float* dst, int64_t ldd, int dc, const float* src, int64_t lds_, int sc, int64_t rows,
                       int cols, bool add, hipStream_t st) {
    if (rows) hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, dst, ldd, dc, src, lds_, sc, rows, cols, add ? 1 : 0);
    return hipGetLastError();
}
hipError_t t_dropout(float* x, int64_t rows, int cols, int64_t ld, int wlen_spatial, const unsigned* step_params, unsigned stream,
                     float rate, hipStream_t st, const float* src) {
    if (rows == 0) return hipSuccess;
    if (rate <= 0.f) {
        if (!src || src == x) return hipSuccess;
        if (ld == cols) return hipMemcpyAsync(x, src, (size_t)rows * cols * sizeof(float), hipMemcpyDeviceToDevice, st);
        return hipMemcpy2DAsync(x, ld * sizeof(float), src, ld * sizeof(float), cols * sizeof(float), rows, hipMemcpyDeviceToDevice, st);
    }
    const unsigned thr = dropout_threshold(rate);
    const float* in = src ? src : x;
    if (cols % 4 == 0 && ld % 4 == 0 && rows * cols < (int64_t)1 << 31 && rows * ld < (int64_t)1 << 31 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(in)) & 15) == 0) {
        hipLaunchKernelGGL(dropout4_kernel, dim3(grid_for(rows * cols / 4)), dim3(256), 0, st, x, in, (int)rows, cols, (int)ld,
                           wlen_spatial, step_params, stream, thr, 1.0f / (1.0f - rate));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, x, in, rows, cols, ld, wlen_spatial,
                       step_params, stream, thr, 1.0f / (1.0f - rate));
    return hipGetLastError();
}
hipError_t t_embed_fwd(const uint8_t* tok, const float* E, const float* P, float* x, int64_t rows, int wlen, hipStream_t st) {
    if (rows) hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, st, tok, E, P, x, rows, wlen);
    return hipGetLastError();
}
hipError_t t_embed_bwd(const uint8_t* tok, const float* dx, float* dE, float* dP, int64_t rows, int wlen, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    int64_t rpb64 = rows / 1024;                     // ~1024 blocks for large batches, 64 rows each for small ones
    if (rpb64 < 64) rpb64 = 64;
    const int rpb = (int)rpb64;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, tok, dx, dE, dP, rows, wlen, rpb);
    return hipGetLastError();
}
static inline int rows_per_block_for(int64_t rows) {
    int64_t r = (rows + 63) / 64;
    if (r < 64) r = 64;
    return (int)r;
}
hipError_t t_colsum2(const float* x, const float* y, int64_t rows, int cols, float* stats, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    const int rpb = rows_per_block_for(rows);
    if (cols == 64 && !y && rows >= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        hipLaunchKernelGGL(colsum64_kernel<0>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, x, (int64_t)64, 0,
                           nullptr, nullptr, rows, stats, rpb);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(colsum2_kernel, grid, dim3(256), 0, st, x, y, rows, cols, (int64_t)cols, stats, rpb);
    return hipGetLastError();
}
hipError_t t_bn_fwd(const float* y, float* z, int64_t rows, int cols, const float* stats, int64_t rows_global, float* mean_inv,
                    float* batch_mean, float* batch_var, const float* gamma, const float* beta, int64_t ldz, int z_col_off,
                    hipStream_t st) {
    if (rows)
        hipLaunchKernelGGL(bn_fwd_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, y, z, rows, cols, stats,
                           1.0f / (float)rows_global, mean_inv, batch_mean, batch_var, gamma, beta, ldz, z_col_off);
    return hipGetLastError();
}
hipError_t t_bn_bwd_reduce(const float* dz, int64_t lddz, int dz_col_off, const float* y, int64_t rows, int cols,
                           const float* mean_inv, float* sums, hipStream_t st) {
    if (rows == 0) return hipSuccess;       // sums must arrive zeroed
    const int rpb = rows_per_block_for(rows);
    if (cols == 64 && rows >= 1024 && lddz % 4 == 0 && dz_col_off % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(mean_inv)) & 15) == 0) {
        hipLaunchKernelGGL(colsum64_kernel<1>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, dz, lddz, dz_col_off, y,
                           mean_inv, rows, sums, rpb);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, st, dz, lddz, dz_col_off, y, rows, cols, mean_inv, sums, rpb);
    return hipGetLastError();
}
hipError_t t_bn_bwd_apply(const float* dz, int64_t lddz, int dz_col_off, const float* y, float* dy, int64_t rows, int cols,
                          const float* mean_inv, const float* gamma, const float* sums, int64_t rows_global,
                          const float* own_sums, float* dgamma, float* dbeta, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, dz, lddz, dz_col_off, y, dy, rows,
                       cols, mean_inv, gamma, sums, 1.0f / (float)rows_global, own_sums, dgamma, dbeta);
    return hipGetLastError();
}
hipError_t t_ln_fwd(const float* a, const float* b, const float* g, const float* beta, float* out, float* xhat,
                    float* inv, int64_t rows, hipStream_t st) {
    if (rows) hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, b, g, beta, out, xhat, inv, rows);
    return hipGetLastError();
}
hipError_t t_ln_bwd(const float* dy, const float* xhat, const float* inv, const float* g, float* dsum, float* dg,
                    float* dbeta, int64_t rows, hipStream_t st) {
    if (rows == 0) return hipSuccess;
    const int rpb = rows_per_block_for(rows);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, dy, xhat, inv, g, dsum, dg, dbeta, rows, rpb);
    return hipGetLastError();
}
hipError_t t_attn_fwd(const float* q, const float* k, const float* v, float* o, float* prob, int64_t samples, int wlen, hipStream_t st) {
    if (samples) hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(samples * 4)), dim3(64), 0, st, q, k, v, o, prob, wlen);
    return hipGetLastError();
}
hipError_t t_attn_bwd(const float* q, const float* k, const float* v, const float* prob, const float* d_o, float* dq,
                      float* dk, float* dv, int64_t samples, int wlen, hipStream_t st) {
    if (samples) hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(samples * 4)), dim3(64), 0, st, q, k, v, prob, d_o, dq, dk, dv, wlen);
    return hipGetLastError();
}
hipError_t t_pool_fwd(const float* x, const float* w, const float* bias, float* out, int64_t ldo, int col_off, float* a_save,
                      float* alpha_save, int64_t samples, int wlen, hipStream_t st) {
    if (samples) hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)samples), dim3(64), 0, st, x, w, bias, out, ldo, col_off, a_save, alpha_save, wlen);
    return hipGetLastError();
}
hipError_t t_pool_bwd(const float* x, const float* w, const float* dout, int64_t ldd, int col_off, const float* a_save,
                      const float* alpha_save, float* dx, float* dw, float* db, int64_t samples, int wlen, hipStream_t st) {
    if (samples) hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)samples), dim3(64), 0, st, x, w, dout, ldd, col_off, a_save, alpha_save, dx, dw, db, wlen);
    return hipGetLastError();
}
hipError_t t_head_loss(const float* x, const float* W, const float* bias, const int32_t* labels, int64_t n,
                       const unsigned* step_params, float* proba, float* loss, float* dx, float* dW, float* db, hipStream_t st) {
    if (n) {
        const int64_t blocks = (n + 3) / 4 < 256 ? (n + 3) / 4 : 256;
        hipLaunchKernelGGL(head_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, W, bias, labels, n, step_params, proba, loss, dx, dW, db);
    }
    return hipGetLastError();
}
// Rows rows[0..n) of a device-resident data set -> the trainer's staging buffers (one workgroup per row: the 1280-float
// ESM row as 16-byte pieces, the 8 structure floats, the 21 + 33 + 51 window codes, the label).  A row index outside
// the data set is clamped into it (the kernel never reads beyond the arrays whatever the index buffer holds).
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherRowsArgs a) {
    const int64_t i = blockIdx.x;
    int64_t r = a.rows[i];
    r = r < 0 ? 0 : (r >= a.n_data ? a.n_data - 1 : r);
    const int tid = threadIdx.x;
    const f32x4* es = reinterpret_cast<const f32x4*>(a.esm + r * 1280);
    f32x4* ed = reinterpret_cast<f32x4*>(a.esm_out + i * 1280);
    for (int k = tid; k < 320; k += 256) ed[k] = es[k];
    if (tid < 8) a.strct_out[i * 8 + tid] = a.strct[r * 8 + tid];
    if (tid >= 64 && tid < 64 + 21) a.tok21_out[i * 21 + (tid - 64)] = a.tok21[r * 21 + (tid - 64)];
    if (tid >= 96 && tid < 96 + 33) a.tok33_out[i * 33 + (tid - 96)] = a.tok33[r * 33 + (tid - 96)];
    if (tid >= 160 && tid < 160 + 51) a.tok51_out[i * 51 + (tid - 160)] = a.tok51[r * 51 + (tid - 160)];
    if (tid == 255) a.labels_out[i] = a.labels[r];
}
hipError_t t_gather_rows(const GatherRowsArgs& a, int64_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, a);
    return hipGetLastError();
}
// the step's loss and class probabilities to addresses of the caller's, one launch
__global__ void step_outputs_kernel(const float* loss, float* loss_out, const float* proba, float* proba_out, int64_t n2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (proba_out && i < n2) proba_out[i] = proba[i];
    if (loss_out && i == 0) loss_out[0] = loss[0];
}
hipError_t t_step_outputs(const float* loss, float* loss_out, const float* proba, float* proba_out, int64_t n, hipStream_t st) {
    if (!loss_out && !proba_out) return hipSuccess;
    const int64_t n2 = proba_out ? 2 * n : 1;
    hipLaunchKernelGGL(step_outputs_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, loss, loss_out, proba, proba_out, n2);
    return hipGetLastError();
}
__global__ void set_step_kernel(unsigned* step_params, unsigned seed, float n_global, float n_pos_global) {
    if (threadIdx.x == 0) {
        step_params[kStepSeed] = seed;
        step_params[kStepNGlobal] = __float_as_uint(n_global);
        step_params[kStepNPosGlobal] = __float_as_uint(n_pos_global);
        step_params[kStepL2] = 0u;       // the step's first launch adds the l2 regularisers' value here
        step_params[kStepL2Sel] = 0u;
    }
}
hipError_t t_set_step(unsigned* step_params, unsigned seed, float n_global, float n_pos_global, hipStream_t st) {
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(64), 0, st, step_params, seed, n_global, n_pos_global);
    return hipGetLastError();
}
hipError_t t_adamw(float* w, const float* g, float* m, float* v, const uint8_t* trainable, float stat_scale, int64_t n,
                   float lr_t, float wd, float b1, float b2, float eps, float gscale, hipStream_t st) {
    if (n) hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, st, w, g, m, v, trainable, stat_scale, n, lr_t, wd, b1, b2, eps, gscale);
    return hipGetLastError();
}

}  // namespace dptm
