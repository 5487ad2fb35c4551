// Fused row-wise kernels of the training step (gfx950): the feed-forward half of a TransformerBlock
// (pred/train_PTM/model.py:102-105, 116-118) forward and backward in one launch each, so that the
// 256-wide hidden layer, its swish and the dropout / residual / LayerNormalization around it never
// travel through HBM.
//
//   ffn_fwd_kernel  h -> out = LN2(h + dropout(W2 swish(W1 h + b1) + b2)); keeps xhat2 and 1/sigma2
//   ffn_bwd_kernel  d(out) -> d(h) and the gradients of W1, b1, W2, b2, gamma2, beta2.  The hidden
//                   layer is recomputed from h in 64-wide chunks; weight gradients are contracted
//                   over the rows on the matrix cores from LDS images of the tile and stay in
//                   accumulator registers for the whole launch.
//   reduce_parts_kernel  sums the workgroups' partial parameter gradients into the gradient buffer.
//
// All matrix products run on v_mfma_f32_32x32x2_f32 in the chain layout of dptm_layout.h (a row per
// lane); weights are read in MFMA fragment order from images that pack_kernel rebuilds every step.
#include <hip/hip_runtime.h>

#include "dptm_layout.h"
#include "dptm_mfma.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"

namespace dptm {

// ------------------------------------------------------------------------------------ weight images

// dst[((t*(K/8) + u)*64 + lane)*4 + e] = M[8u + 4h + e][32t + j] for the logical matrix M [K][N]:
// M[k][n] = src[k*N + n], or src[n*K + k] when the job asks for the transpose of a raw [N][K] array.
__global__ __launch_bounds__(256) void pack_kernel(const PackJob* __restrict__ jobs) {
    const PackJob jb = jobs[blockIdx.y];
    const int total = jb.K * jb.N;
    const int k8 = jb.K >> 3;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int e = o & 3, lane = (o >> 2) & 63, tu = o >> 8;
        const int t = tu / k8, u = tu - t * k8;
        const int k = 8 * u + 4 * (lane >> 5) + e, n = 32 * t + (lane & 31);
        jb.dst[o] = jb.transpose ? jb.src[(int64_t)n * jb.K + k] : jb.src[(int64_t)k * jb.N + n];
    }
}

hipError_t t_pack(const PackJob* jobs_dev, int n_jobs, hipStream_t st) {
    if (n_jobs) hipLaunchKernelGGL(pack_kernel, dim3(16, (unsigned)n_jobs), dim3(256), 0, st, jobs_dev);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ helpers

// accumulator tiles from a raw (unpacked) feature vector: acc[t][4g + e] = v[8(4t + g) + 4h + e]
template <int NT>
__device__ __forceinline__ void acc_init_raw(f32x16 (&acc)[NT], const float* __restrict__ v, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(v + 8 * (4 * t + g) + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
        }
}

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// chain-layout row vector <-> the lane's row of an LDS image [rows][kLdsStride]
__device__ __forceinline__ void chain_store(float* row_base, int h, const f32x4 (&x)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(row_base + 8 * u + 4 * h) = x[u];
}
__device__ __forceinline__ void chain_load(const float* row_base, int h, f32x4 (&x)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(row_base + 8 * u + 4 * h);
}

// 32 rows x 64 floats between global memory (row-major, 256-byte rows) and 32 image rows, every access a
// coalesced 16-byte vector: request i of a lane covers row 4i + lane/16, floats 4*(lane%16)..; rows >= R read as 0
__device__ __forceinline__ void rows_in(float* img_rows, const float* __restrict__ g, int64_t r0, int64_t R, int lane) {
    const int crow = lane >> 4, ccol = (lane & 15) * 4;
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t rr = r0 + 4 * i + crow;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rr < R) v[i] = *reinterpret_cast<const f32x4*>(g + rr * 64 + ccol);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(img_rows + (4 * i + crow) * kLdsStride + ccol) = v[i];
}
__device__ __forceinline__ void rows_out(const float* img_rows, float* __restrict__ g, int64_t r0, int64_t R, int lane) {
    const int crow = lane >> 4, ccol = (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t rr = r0 + 4 * i + crow;
        if (rr < R)
            *reinterpret_cast<f32x4*>(g + rr * 64 + ccol) = *reinterpret_cast<const f32x4*>(img_rows + (4 * i + crow) * kLdsStride + ccol);
    }
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

// inverted dropout of a chain-layout row vector; the element index of (row r, feature c) is r*64 + c, as in
// dropout4_kernel, so forward and backward (and the layer-wise kernels) see the same mask
__device__ __forceinline__ void dropout_chain(f32x4 (&x)[8], int64_t r, int h, unsigned seed, unsigned stream, unsigned thr,
                                              float scale) {
    if (thr == 0u) return;
    const unsigned base = (unsigned)(r * 64) + 4 * h;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[u][e] = hash_u32(seed, stream, base + 8 * u + e) >= thr ? x[u][e] * scale : 0.f;
}

// acc[t] += (chunk c of a packed [256][64] matrix)^T-product with a 64-feature chain vector: the K = 256 image has 32
// feature groups per out tile, the chunk owns groups 8c .. 8c+7
__device__ __forceinline__ void gemm_k256_chunk(f32x16 (&acc)[2], const f32x4* __restrict__ wp, int c, const f32x4 (&x)[8],
                                                int lane) {
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

// ------------------------------------------------------------------------------------ forward

// One wavefront per 32-row tile, wavefronts independent.  grid-stride over tiles, block 256.
__global__ __launch_bounds__(256, 2) void ffn_fwd_kernel(FfnFwdArgs p) {
    __shared__ __attribute__((aligned(16))) float stage_all[4][32 * kLdsStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    float* stage = stage_all[wave];
    const unsigned seed = p.step[kStepSeed];
    const int64_t ntile = (p.R + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        const int64_t r0 = tile * 32;
        rows_in(stage, p.h, r0, p.R, lane);
        wave_fence();
        f32x4 hx[8];
        chain_load(stage + j * kLdsStride, h, hx);
        wave_fence();
        f32x16 acc_o[2];
        acc_init_raw<2>(acc_o, p.b2, h);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            f32x16 acc_h[2];
            acc_init_raw<2>(acc_h, p.b1 + 64 * c, h);
            gemm_regs<8, 2, 2>(acc_h, p.w1p + (2 * c) * 8 * 64, hx, lane);
            f32x4 hid[8];
            acc_to_x<2>(acc_h, hid);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) hid[u][e] = swishf(hid[u][e]);
            gemm_k256_chunk(acc_o, p.w2p, c, hid, lane);
        }
        f32x4 y[8];
        acc_to_x<2>(acc_o, y);
        dropout_chain(y, r0 + j, h, seed, p.stream, p.thr, p.scale);
        // LayerNormalization(epsilon 1e-6) of h + dropout(f2); xhat and 1/sigma are kept for the backward pass
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            y[u] += hx[u];
            s += (y[u][0] + y[u][1]) + (y[u][2] + y[u][3]);
        }
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
        for (int u = 0; u < 8; ++u) {
            y[u] *= inv;
            p.xhat[(tile * 8 + u) * 64 + lane] = y[u];          // tile-chain layout: a coalesced 1 KiB store
            const f32x4 gg = *reinterpret_cast<const f32x4*>(p.gamma + 8 * u + 4 * h);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.beta + 8 * u + 4 * h);
            y[u] = y[u] * gg + bb;
        }
        if (h == 0 && r0 + j < p.R) p.inv[r0 + j] = inv;
        chain_store(stage + j * kLdsStride, h, y);
        wave_fence();
        rows_out(stage, p.out, r0, p.R, lane);
        wave_fence();
    }
}

hipError_t t_ffn_fwd(const FfnFwdArgs& a, hipStream_t st) {
    if (a.R == 0) return hipSuccess;
    const int64_t tiles = (a.R + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    hipLaunchKernelGGL(ffn_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ backward

constexpr int kRwRows = 128;                          // rows per workgroup tile: 4 wavefronts x 32
constexpr int kRwImg = kRwRows * kLdsStride;          // floats per LDS image
constexpr size_t kFfnBwdLds = 4 * (size_t)kRwImg * sizeof(float);

// dW tile of wavefront-owned accumulators: acc += A[:, a_col + i]^T B[:, b_col + i] over the tile's 128 rows
// (row pair 2q + half-wave per MFMA step; single-float LDS reads with lane = column index: conflict-free)
__device__ __forceinline__ void dw_tile(f32x16& acc, const float* imgA, int a_col, const float* imgB, int b_col, int lane) {
    const int hh = lane >> 5, i = lane & 31;
    const float* ap = imgA + hh * kLdsStride + a_col + i;
    const float* bp = imgB + hh * kLdsStride + b_col + i;
#pragma unroll 8
    for (int q = 0; q < kRwRows / 2; ++q) mfma(acc, ap[2 * q * kLdsStride], bp[2 * q * kLdsStride]);
}

// One workgroup (4 wavefronts) per 128-row tile, grid-stride; LDS images: [0] h, [1] d(out) then d(f2),
// [2] scratch / swish(f1) chunk / d(h), [3] d(f1) chunk.  Wavefront w owns, for every hidden chunk c, the 32 x 32
// tile (w>>1, w&1) of dW1[:, 64c..] and of dW2[64c.., :].
__global__ __launch_bounds__(256) void ffn_bwd_kernel(FfnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* img_h = lds;
    float* img_b = lds + kRwImg;
    float* img_c = lds + 2 * kRwImg;
    float* img_d = lds + 3 * kRwImg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int own = wave * 32 * kLdsStride;               // this wavefront's rows inside an image
    const int mine = own + j * kLdsStride;                // this lane's row
    const unsigned seed = p.step[kStepSeed];
    const int64_t ntile = (p.R + kRwRows - 1) / kRwRows;
    const int64_t ntile32 = (p.R + 31) / 32;
    const int kt = wave >> 1, nt = wave & 1;

    f32x16 dw1[4], dw2[4];
    acc_zero<4>(dw1);
    acc_zero<4>(dw2);
    float db1[4] = {0.f, 0.f, 0.f, 0.f}, db2 = 0.f, dgam = 0.f, dbet = 0.f;     // lane = feature

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t r0 = tile * kRwRows + wave * 32;    // first row of this wavefront
        const int64_t t32 = r0 >> 5;
        rows_in(img_h + own, p.h, r0, p.R, lane);
        rows_in(img_b + own, p.dout, r0, p.R, lane);
        f32x4 xh[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xh[u] = t32 < ntile32 ? p.xhat[(t32 * 8 + u) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
        const float inv = r0 + j < p.R ? p.inv[r0 + j] : 0.f;
        wave_fence();
        f32x4 hx[8], dy[8];
        chain_load(img_h + mine, h, hx);
        chain_load(img_b + mine, h, dy);
        // ---- LayerNormalization 2 backward: dsum = inv * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat))
        f32x4 dsum[8];
        float s1 = 0.f, s2 = 0.f;
        {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t[u] = dy[u] * xh[u];
                const f32x4 gg = *reinterpret_cast<const f32x4*>(p.gamma + 8 * u + 4 * h);
                dsum[u] = dy[u] * gg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += dsum[u][e];
                    s2 = fmaf(dsum[u][e], xh[u][e], s2);
                }
            }
            chain_store(img_c + mine, h, t);
        }
        s1 = xhalf_sum(s1) * (1.0f / 64.0f);
        s2 = xhalf_sum(s2) * (1.0f / 64.0f);
#pragma unroll
        for (int u = 0; u < 8; ++u) dsum[u] = inv * (dsum[u] - s1 - xh[u] * s2);
        wave_fence();
        dgam += colsum32(img_c + own, lane);
        dbet += colsum32(img_b + own, lane);
        wave_fence();
        // ---- dropout backward: d(f2) with the forward mask
        f32x4 df2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) df2[u] = dsum[u];
        dropout_chain(df2, r0 + j, h, seed, p.stream, p.thr, p.scale);
        chain_store(img_b + mine, h, df2);
        wave_fence();
        db2 += colsum32(img_b + own, lane);
        // ---- d(h) = dsum + sum_c d(f1_c) W1_c^T, accumulated on the matrix cores
        f32x16 acc_dh[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc_dh[t][4 * g + e] = dsum[4 * t + g][e];
        // (a rolled loop: unrolled, the four chunks' weight fetches are hoisted over one another and spill)
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            f32x4 f1[8], du[8];
            {
                f32x16 acc_f[2];
                acc_init_raw<2>(acc_f, p.b1 + 64 * c, h);
                gemm_regs<8, 2, 2>(acc_f, p.w1p + (2 * c) * 8 * 64, hx, lane);      // f1_c = h W1[:, chunk] + b1
                acc_to_x<2>(acc_f, f1);
                f32x16 acc_u[2];
                acc_zero<2>(acc_u);
                gemm_regs<8, 2, 2>(acc_u, p.w2tp + (2 * c) * 8 * 64, df2, lane);    // d(u_c) = d(f2) W2[chunk, :]^T
                acc_to_x<2>(acc_u, du);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = f1[u][e];
                    const float sg = fast_rcp(1.0f + __expf(-a));
                    f1[u][e] = a * sg;                                               // u_c = swish(f1_c)
                    du[u][e] = du[u][e] * (sg * (1.0f + a * (1.0f - sg)));           // d(f1_c)
                }
            chain_store(img_c + mine, h, f1);
            chain_store(img_d + mine, h, du);
            gemm_k256_chunk(acc_dh, p.w1tp, c, du, lane);                            // += d(f1_c) W1[:, chunk]^T
            wave_fence();
            const float bs = colsum32(img_d + own, lane);
            __syncthreads();
            // weight gradients over the whole tile: dW1[:, chunk] += h^T d(f1_c);  dW2[chunk, :] += u_c^T d(f2).
            // The accumulators are named statically in every case so that they stay in registers.
#define DPTM_DW_CASE(C)                                              \
    case C:                                                          \
        db1[C] += bs;                                                \
        dw_tile(dw1[C], img_h, 32 * kt, img_d, 32 * nt, lane);       \
        dw_tile(dw2[C], img_c, 32 * kt, img_b, 32 * nt, lane);       \
        break;
            switch (c) {
                DPTM_DW_CASE(0)
                DPTM_DW_CASE(1)
                DPTM_DW_CASE(2)
                default:
                DPTM_DW_CASE(3)
            }
#undef DPTM_DW_CASE
            __syncthreads();
        }
        f32x4 dh[8];
        acc_to_x<2>(acc_dh, dh);
        chain_store(img_c + mine, h, dh);
        wave_fence();
        rows_out(img_c + own, p.dh, r0, p.R, lane);
        wave_fence();
    }

    // ---- this workgroup's partial gradients: [dW1 64x256 | dW2 256x64 | db1 256 | db2 64 | dgamma 64 | dbeta 64]
    float* part = p.part + (size_t)blockIdx.x * kFfnPartStride;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int row = (rg & 3) + 8 * (rg >> 2) + 4 * h;        // row of the 32 x 32 tile held by this register
            part[(32 * kt + row) * 256 + 64 * c + 32 * nt + j] = dw1[c][rg];
            part[16384 + (64 * c + 32 * kt + row) * 64 + 32 * nt + j] = dw2[c][rg];
        }
    __syncthreads();
    float* red = lds;                                        // [7][4 wavefronts][64]
    const float vals[7] = {db1[0], db1[1], db1[2], db1[3], db2, dgam, dbet};
#pragma unroll
    for (int v = 0; v < 7; ++v) red[(v * 4 + wave) * 64 + lane] = vals[v];
    __syncthreads();
    for (int o = threadIdx.x; o < 7 * 64; o += 256) {
        const int v = o >> 6, f = o & 63;
        part[32768 + o] = red[(v * 4 + 0) * 64 + f] + red[(v * 4 + 1) * 64 + f] + red[(v * 4 + 2) * 64 + f] + red[(v * 4 + 3) * 64 + f];
    }
}

// dst_s[e] += sum_p part[p*stride + src_s + e] for every segment s of the table (the kernel is the only writer of
// those gradient tensors in the step)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int nparts, int64_t stride, PartSegs segs) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < segs.total; o += gridDim.x * blockDim.x) {
        int s = 0;
        while (s + 1 < segs.n && o >= segs.src[s + 1]) ++s;
        float a0 = 0.f, a1 = 0.f;
        int pidx = 0;
        for (; pidx + 1 < nparts; pidx += 2) {
            a0 += part[(size_t)pidx * stride + o];
            a1 += part[(size_t)(pidx + 1) * stride + o];
        }
        if (pidx < nparts) a0 += part[(size_t)pidx * stride + o];
        segs.dst[s][o - segs.src[s]] += a0 + a1;
    }
}

static int ffn_bwd_grid(int64_t R) {
    int64_t tiles = (R + kRwRows - 1) / kRwRows;
    return (int)(tiles < 256 ? tiles : 256);
}

hipError_t t_ffn_bwd(const FfnBwdArgs& a, const FfnGrads& g, hipStream_t st) {
    if (a.R == 0) return hipSuccess;
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kFfnBwdLds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    const int blocks = ffn_bwd_grid(a.R);
    hipLaunchKernelGGL(ffn_bwd_kernel, dim3((unsigned)blocks), dim3(256), kFfnBwdLds, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    PartSegs segs;
    segs.n = 6;
    const int src[7] = {0, 16384, 32768, 32768 + 256, 32768 + 320, 32768 + 384, 32768 + 448};
    float* dst[6] = {g.dw1, g.dw2, g.db1, g.db2, g.dgamma, g.dbeta};
    for (int s = 0; s < 6; ++s) {
        segs.src[s] = src[s];
        segs.dst[s] = dst[s];
    }
    segs.src[6] = src[6];
    segs.total = src[6];
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(130), dim3(256), 0, st, a.part, blocks, (int64_t)kFfnPartStride, segs);
    return hipGetLastError();
}

int t_ffn_bwd_max_parts() { return 256; }

}  // namespace dptm
