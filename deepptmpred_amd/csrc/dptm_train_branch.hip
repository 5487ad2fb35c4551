// Fused sequence-branch kernels of the training step (gfx950): both TransformerBlocks and the attention
// pooling of one branch (pred/train_PTM/model.py:44-118, 172-182) in ONE forward and ONE backward launch per
// window size, so that the activations of a site stay in registers / LDS across the layers.
//
//   tb_fwd_kernel<W>   x1 -> pooled rows.  Saves per block what the backward pass cannot rebuild cheaply:
//                      q, k, v, the attention output, the two normalised rows (xhat) with 1/sigma, and the
//                      softmax statistics (row maximum, 1/denominator) instead of the W x W probabilities.
//   tb_bwd_kernel<W>   d(pooled rows) -> d(x1) and the gradients of every parameter of the two blocks and the
//                      pooling.  The FFN hidden layer and the attention probabilities are recomputed.
//   reduce_*_kernel    sum the workgroups' partial parameter gradients into the gradient buffer.
//   pack_kernel        rebuilds the weight images (MFMA fragment order) from the raw weights every step.
//
// Tiling: a workgroup owns S whole sites = S*W rows in NW wavefronts of 32 rows (dptm_layout.h "chain layout":
// a row per lane), NW*32 >= S*W; the host picks (S, NW) from the batch size so that small batches still spread
// over the chip.  Row-wise products run register to register on v_mfma_f32_32x32x2_f32.  A weight gradient
// dW = X^T dY contracts over the rows, which sit on the lanes, so both operands pass through LDS images
// [row][68]: the eight wavefronts take the four 32 x 32 blocks of a 64 x 64 unit over two row halves, and the
// workgroup adds the result to its private partial buffer in global memory (plain read-modify-write: a
// workgroup is the only writer of its buffer).  Attention backward runs on the vector unit in two passes,
// lane = query row (dq) and lane = key row (dk, dv), each rebuilding p = exp2(s - max) / den.
#include <hip/hip_runtime.h>

#include <cstdlib>

#define DPTM_KEEP_INDEX_LOCAL 1
#include "dptm_attend.h"
#include "dptm_layout.h"
#include "dptm_mfma.h"
#include "dptm_pair.h"
#include "dptm_phase.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"
#include "dptm_train_tile.h"

namespace dptm {

// ------------------------------------------------------------------------------------ weight images

// kind 0: dst[((t*(K/8) + u)*64 + lane)*4 + e] = M[8u + 4h + e][32t + j], M[k][n] = src[k*N + n] or src[n*K + k]
// kind 1: dst[(h*(K/8) + u)*4 + e] = src[8u + 4h + e];  kind 2: dst[o] = src[o];  kind 3: dst[o] = 0 (o < K)
// kind 4: l2 kernel regulariser (model.py:130-138,200,204) of K weights: dst[o] = 2*lambda*src[o] -- the gradient buffer
//         starts from it, every later writer adds -- and lambda * sum src^2 joins step_params[kStepL2]
// The launch is the step's first: block (0, 0) also writes the step's scalars (set = 1), names the slot of the l2 value and
// clears the other one for the next step.
__global__ __launch_bounds__(256) void pack_kernel(const PackJob* __restrict__ jobs, float lambda, unsigned* __restrict__ step_params,
                                                   int l2_slot, int set, StepVals sv) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (set) {
            step_params[kStepSeed] = sv.seed;
            step_params[kStepNGlobal] = __float_as_uint(sv.n_global);
            step_params[kStepNPosGlobal] = __float_as_uint(sv.n_pos_global);
            step_params[kStepL2Sel] = (unsigned)l2_slot;
            step_params[kStepL2 + (l2_slot ^ 1)] = 0u;
        }
    }
    const PackJob jb = jobs[blockIdx.y];
    if (jb.kind == 0) {
        const int total = jb.K * jb.N;
        const int k8 = jb.K >> 3;
        for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
            const int e = o & 3, lane = (o >> 2) & 63, tu = o >> 8;
            const int t = tu / k8, u = tu - t * k8;
            const int k = 8 * u + 4 * (lane >> 5) + e, n = 32 * t + (lane & 31);
            jb.dst[o] = jb.transpose ? jb.src[(int64_t)n * jb.K + k] : jb.src[(int64_t)k * jb.N + n];
        }
    } else if (jb.kind == 1) {
        const int g8 = jb.K >> 3;
        for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < jb.K; o += gridDim.x * blockDim.x) {
            const int e = o & 3, g = o >> 2;
            const int h = g / g8, u = g - h * g8;
            jb.dst[o] = jb.src[8 * u + 4 * h + e];
        }
    } else if (jb.kind == 2) {
        for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < jb.K; o += gridDim.x * blockDim.x) jb.dst[o] = jb.src[o];
    } else if (jb.kind == 3) {
        // (spans start and end on multiples of four floats wherever the tables allow: vector stores, scalar tail)
        const bool vec = ((reinterpret_cast<uintptr_t>(jb.dst) & 15) == 0);
        const int k4 = vec ? jb.K >> 2 : 0;
        f32x4* d4 = reinterpret_cast<f32x4*>(jb.dst);
        for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < k4; o += gridDim.x * blockDim.x) d4[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int o = 4 * k4 + blockIdx.x * blockDim.x + threadIdx.x; o < jb.K; o += gridDim.x * blockDim.x) jb.dst[o] = 0.f;
    } else {
        __shared__ float red[4];
        float s = 0.f;
        const bool vec = ((reinterpret_cast<uintptr_t>(jb.dst) | reinterpret_cast<uintptr_t>(jb.src)) & 15) == 0;
        const int k4 = vec ? jb.K >> 2 : 0;
        const f32x4* s4 = reinterpret_cast<const f32x4*>(jb.src);
        f32x4* d4 = reinterpret_cast<f32x4*>(jb.dst);
        const int stride4 = gridDim.x * blockDim.x;
        int o = blockIdx.x * blockDim.x + threadIdx.x;
        for (; o + stride4 < k4; o += 2 * stride4) {       // two loads in flight
            const f32x4 v = s4[o], w = s4[o + stride4];
            d4[o] = (2.0f * lambda) * v;
            d4[o + stride4] = (2.0f * lambda) * w;
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], v[e], fmaf(w[e], w[e], s));
        }
        if (o < k4) {
            const f32x4 v = s4[o];
            d4[o] = (2.0f * lambda) * v;
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], v[e], s);
        }
        for (int o1 = 4 * k4 + blockIdx.x * blockDim.x + threadIdx.x; o1 < jb.K; o1 += gridDim.x * blockDim.x) {
            const float v = jb.src[o1];
            jb.dst[o1] = 2.0f * lambda * v;
            s = fmaf(v, v, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (red[0] + red[1]) + (red[2] + red[3]);
            if (t != 0.f) atomicAdd(reinterpret_cast<float*>(step_params) + kStepL2 + l2_slot, lambda * t);
        }
    }
}

hipError_t t_pack(const PackJob* jobs_dev, int n_jobs, float lambda, unsigned* step_params, int l2_slot, const StepVals* sv,
                  hipStream_t st) {
    if (n_jobs < 1 || (l2_slot != 0 && l2_slot != 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_kernel, dim3(8, (unsigned)n_jobs), dim3(256), 0, st, jobs_dev, lambda, step_params, l2_slot, sv ? 1 : 0,
                       sv ? *sv : StepVals{0u, 0.f, 0.f});
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ tiling

namespace {
struct Cand { int S, NW; };
// Tiles per window size: whole sites in eight wavefronts, in four, and one site spread over FOUR wavefronts.  In the
// last one the wavefronts without rows of their own take a share of every weight-gradient unit (whose contraction
// stops at the site's last row), so the dependent chain of the wavefronts that do hold rows -- which is what such a
// launch lasts -- gets shorter.
const Cand kCand21[] = {{12, 8}, {6, 4}, {1, 4}};
const Cand kCand33[] = {{7, 8}, {3, 4}, {1, 4}};
const Cand kCand51[] = {{5, 8}, {2, 4}, {1, 4}};
constexpr int kNumCand = 3;
constexpr int kCUs = 256;

int64_t tiles_of(const Cand& c, int64_t n) { return (n + c.S - 1) / c.S; }
// workgroups of the three branches together (they run side by side on three streams, one workgroup per CU: a
// four-wavefront tile holds 91 KB of LDS)
int64_t workgroups_all(int i, int64_t n) { return tiles_of(kCand21[i], n) + tiles_of(kCand33[i], n) + tiles_of(kCand51[i], n); }
// one round: every workgroup of the three branches finds a CU at once.  The four-wavefront tiles stop a little
// short of the chip (measured: at 255 sites = 256 workgroups they are already 5 % behind the eight-wavefront tiles).
// One-site tiles go a little beyond one round (112 sites = 336 workgroups): with two wavefronts per row block their chain
// is half that of the four-wavefront tiles, and the shortest branch's workgroups fill in behind the longest's (measured
// per step: 86 sites 0.58 against 0.64 ms, 100 sites 0.62 against 0.64, 128 sites 0.67 against 0.65).
bool one_round(int i, int64_t n) { return workgroups_all(i, n) <= (i == 2 ? 336 : kCUs - 24); }

// DPTM_TRAIN_TILE=i pins candidate i (measurements: tools/tile_sweep.sh; tests of a tile shape on a small batch)
int forced_tile() {
    static const int v = [] {
        const char* e = std::getenv("DPTM_TRAIN_TILE");
        const int x = e ? std::atoi(e) : -1;
        return x >= kNumCand ? kNumCand - 1 : x;
    }();
    return v;
}
}  // namespace

// A launch of these kernels lasts about as long as ONE tile's dependent chain times the rounds of tiles a CU sees,
// and the chain of an eight-wavefront tile is only ~1.65x that of a four-wavefront tile that holds less than half
// the sites.  So: the smallest tile whose workgroups -- of all three branches -- are on the chip at once (one round),
// else the largest.  Measured per step (MI355X): 128 sites 0.81 ms with four-wavefront tiles against 1.02 with eight;
// 256 sites 1.10 ms with eight against 1.17 with four; 512 sites 1.36 against 1.56.
TileCfg t_tile_cfg(int64_t n, int wlen) {
    const Cand* c = wlen == 21 ? kCand21 : wlen == 33 ? kCand33 : kCand51;
    int i = forced_tile();
    if (i < 0) i = one_round(2, n) ? 2 : one_round(1, n) ? 1 : 0;
    TileCfg r{c[i].S, c[i].NW, 0};
    r.ntiles = tiles_of(c[i], n);
    return r;
}

int64_t t_tile_rows_max(int64_t max_sites, int wlen) {
    // the padded row count grows with n inside one tile regime: its maximum sits at a regime's upper end
    int64_t best = 0;
    for (int i = 0; i < kNumCand; ++i) {
        int64_t lo = 1, hi = max_sites;       // largest n <= max_sites whose three branches fit the chip with candidate i
        if (i > 0) {
            if (!one_round(i, 1)) continue;
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) / 2;
                if (one_round(i, mid)) lo = mid; else hi = mid - 1;
            }
        }
        const TileCfg t = t_tile_cfg(i > 0 ? lo : max_sites, wlen);
        const int64_t rows = t.ntiles * t.NW * 32;
        best = rows > best ? rows : best;
    }
    return best;
}

// What the forward kernel keeps for the backward pass is written once and read once, milliseconds apart: both
// sides carry the non-temporal hint, so that the arrays do not displace the weight images from the caches.
#ifdef DPTM_EXP_PLAIN_KEEP
__device__ __forceinline__ void keep_store(f32x4* p, const f32x4& v) { *p = v; }
__device__ __forceinline__ f32x4 keep_load(const f32x4* p) { return *p; }
#else
__device__ __forceinline__ void keep_store(f32x4* p, const f32x4& v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ f32x4 keep_load(const f32x4* p) { return __builtin_nontemporal_load(p); }
#endif

// ------------------------------------------------------------------------------------ forward

template <int W>
__device__ __forceinline__ void tb_fwd_body(const TbFwdArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    float* bufA = lds;
    float* bufB = bufA + rows_tile * kLdsStride;
    float* score = bufB + rows_tile * kLdsStride;   // [rows_tile] exp(tanh score)
    float* alpha_s = score + rows_tile;             // [rows_tile] pooling weights
    float* bnv = alpha_s + rows_tile;               // [scale | shift] of the second BatchNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* mine_a = bufA + row * kLdsStride;
    float* mine_b = bufB + row * kLdsStride;
    const int att_start = min(row / W, S - 1) * W;
    bn_prepare(p.bn2, bnv, tid, blockIdx.x == 0);
    __syncthreads();
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int64_t grow = site0 * W + row;
        const int64_t ct = ((tile * NW + wave) * 8) * 64 + lane;
        const int64_t cl = (tile * NW + wave) * 64 + lane;
        f32x4 x1[8];
        x1_from_conv(p.cs, ct, bnv, h, row_ok, x1);
#pragma unroll 1
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4* T = p.P + kOffTb + blk * kTbSize;
            const f32x4* tv = T + kTbVec;
            const TbSave& sv = p.sv[blk];
            __syncthreads();   // every wavefront is done with the previous contents of bufA / bufB
            {
                f32x16 acc[2];
                f32x4 t[8];
                acc_init_bias<2>(acc, tv + kVBk * kVec64, h, 8);
                gemm_regs<8, 2>(acc, T + kTbWk, x1, lane);
                acc_to_x<2>(acc, t);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.k[ct + u * 64], t[u]);
                chain_store(mine_a, h, t);
                acc_init_bias<2>(acc, tv + kVBv * kVec64, h, 8);
                gemm_regs<8, 2>(acc, T + kTbWv, x1, lane);
                acc_to_x<2>(acc, t);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.v[ct + u * 64], t[u]);
                chain_store(mine_b, h, t);
            }
            float qh[2][16];
            {
                f32x16 acc[2];
                f32x4 t[8];
                acc_init_bias<2>(acc, tv + kVBq * kVec64, h, 8);
                gemm_regs<8, 2>(acc, T + kTbWq, x1, lane);
                acc_to_x<2>(acc, t);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.q[ct + u * 64], t[u]);
                x_to_heads(t, qh, kQScale);
            }
            __syncthreads();   // K and V of the whole tile are in LDS
            f32x4 o[8];
            {
                float oh[2][16];
                f32x4 st;
#pragma unroll
                for (int hd = 0; hd < 2; ++hd) {
                    float q1[1][16], o1[1][16], m1[1], r1[1];
#pragma unroll
                    for (int d = 0; d < 16; ++d) q1[0][d] = qh[hd][d];
                    const int off = att_start * kLdsStride + 32 * h + 16 * hd;
                    attend<W, 1>(bufA + off, bufB + off, q1, o1, m1, r1);
#pragma unroll
                    for (int d = 0; d < 16; ++d) oh[hd][d] = o1[0][d];
                    st[2 * hd] = m1[0];
                    st[2 * hd + 1] = r1[0];
                }
                sv.stat[cl] = st;
                heads_to_x(oh, o, 1.0f);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.o[ct + u * 64], o[u]);
            }
            // output projection, dropout, residual, LayerNorm 1
            f32x4 hres[8];
            f32x2 iv;
            {
                f32x16 acc[2];
                acc_init_bias<2>(acc, tv + kVBo * kVec64, h, 8);
                gemm_regs<8, 2>(acc, T + kTbWo, o, lane);
                f32x4 t[8];
                acc_to_x<2>(acc, t);
                dropout_chain(t, grow, h, seed, p.stream_base + 2 * blk, p.thr, p.scale);
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] += x1[u];
                iv[0] = ln_normalise(t);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.xh1[ct + u * 64], t[u]);
                ln_affine(t, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h, hres);
            }
            // FFN 64 -> 256 (swish) -> 64 in four hidden chunks, dropout, residual, LayerNorm 2
            {
                f32x16 acc_o[2];
                acc_init_bias<2>(acc_o, tv + kVB2 * kVec64, h, 8);
#pragma unroll 1
                for (int c = 0; c < 4; ++c) {
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
                dropout_chain(t, grow, h, seed, p.stream_base + 2 * blk + 1, p.thr, p.scale);
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] += hres[u];
                iv[1] = ln_normalise(t);
#pragma unroll
                for (int u = 0; u < 8; ++u) keep_store(&sv.xh2[ct + u * 64], t[u]);
                ln_affine(t, tv + kVLn2G * kVec64, tv + kVLn2B * kVec64, h, x1);
            }
            sv.inv[cl] = iv;
        }
        // ---- attention pooling (model.py:176-182): a = tanh(x.w + b), alpha = softmax over the site's positions
        {
            const f32x4* pv = p.P + kOffPoolV;
            float part = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wv = pv[h * 8 + u];
#pragma unroll
                for (int e = 0; e < 4; ++e) part = fmaf(x1[u][e], wv[e], part);
            }
            const float a = tanhf(xhalf_sum(part) + p.P[kOffPoolB][0]);
            const float ew = __expf(a);     // |a| <= 1: no running maximum needed
            __syncthreads();                // all attention reads of bufA are finished
            chain_store(mine_a, h, x1);
            if (h == 0) score[row] = ew;
            __syncthreads();
            float den = 0.f;
#pragma unroll 3
            for (int t = 0; t < W; ++t) den += score[att_start + t];
            const float al = ew / den;
            if (h == 0) alpha_s[row] = al;
            p.pool[cl] = f32x2{a, al};
            __syncthreads();
            for (int oi = tid; oi < sites_here * 64; oi += blockDim.x) {
                const int s = oi >> 6, f = oi & 63;
                const float* ap = alpha_s + s * W;
                const float* xp = bufA + (s * W) * kLdsStride + f;
                float acc = 0.f;
#pragma unroll 3
                for (int t = 0; t < W; ++t) acc = fmaf(ap[t], xp[t * kLdsStride], acc);
                p.out[(site0 + s) * p.ldo + p.col_off + f] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward

// attention backward, lane = query row, one head: dq = sum_t dz_t k_t with dz_t = p_t (dO.v_t - D),
// p_t = exp2(q.k_t - m) * rden (q scaled by log2(e)/4 as in the forward pass)
template <int W>
__device__ __forceinline__ void attn_bwd_q(const float* kp, const float* vp, const float (&q)[16], const float (&dO)[16], float m,
                                           float rden, float D, float (&dq)[16]) {
    f32x4 q4[4], d4[4], acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q4[i] = f32x4{q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]};
        d4[i] = f32x4{dO[4 * i], dO[4 * i + 1], dO[4 * i + 2], dO[4 * i + 3]};
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 3
    for (int t = 0; t < W; ++t) {
        f32x4 k[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            k[i] = *reinterpret_cast<const f32x4*>(kp + t * kLdsStride + 4 * i);
            v[i] = *reinterpret_cast<const f32x4*>(vp + t * kLdsStride + 4 * i);
        }
        f32x4 s4 = q4[0] * k[0], p4 = d4[0] * v[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            s4 += q4[i] * k[i];
            p4 += d4[i] * v[i];
        }
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const float dp = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        const float pr = __builtin_amdgcn_exp2f(s - m) * rden;
        const float dz = pr * (dp - D);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += dz * k[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[4 * i + e] = acc[i][e];
}

// attention backward, lane = key row, one head: dk = sum_i dz_i q_i, dv = sum_i p_i dO_i over the site's query rows;
// qp / dop: scaled q and dO rows of the site in LDS, st: {max, 1/den, D} per (row, head) in three arrays `sstride` apart
template <int W>
__device__ __forceinline__ void attn_bwd_kv(const float* qp, const float* dop, const float* st, int sstride, const float (&k)[16],
                                            const float (&v)[16], float (&dk)[16], float (&dv)[16]) {
    f32x4 k4[4], v4[4], ak[4], av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k4[i] = f32x4{k[4 * i], k[4 * i + 1], k[4 * i + 2], k[4 * i + 3]};
        v4[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        ak[i] = av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 3
    for (int t = 0; t < W; ++t) {
        f32x4 q[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[i] = *reinterpret_cast<const f32x4*>(qp + t * kLdsStride + 4 * i);
            d[i] = *reinterpret_cast<const f32x4*>(dop + t * kLdsStride + 4 * i);
        }
        const float m = st[t * 4], rden = st[sstride + t * 4], D = st[2 * sstride + t * 4];
        f32x4 s4 = q[0] * k4[0], p4 = d[0] * v4[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            s4 += q[i] * k4[i];
            p4 += d[i] * v4[i];
        }
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const float dp = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        const float pr = __builtin_amdgcn_exp2f(s - m) * rden;
        const float dz = pr * (dp - D);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ak[i] += dz * q[i];
            av[i] += pr * d[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dk[4 * i + e] = ak[i][e];
            dv[4 * i + e] = av[i][e];
        }
}

constexpr float kLn2 = 0.69314718055994530942f;

// index of a block's vectors in the partial buffer
enum TbVecSlot { kSG2 = 0, kSB2, kSFb2, kSFb1, kSG1 = kSFb1 + 4, kSB1, kSBo, kSBq, kSBk, kSBv, kSlotsPerBlock = 14 };
constexpr int kSPoolW = 2 * kSlotsPerBlock, kSPoolB = kSPoolW + 1;
constexpr int kSBn2A = 13, kSBn2B = 27;     // the two spare slots: sum d(x1), sum d(x1) * swish(a2)
static_assert(kSPoolB < kTbVecs, "vector slots");

// (experiment builds only: DPTM_EXP_NOSYNC drops the barriers of the tile loop to price them; results are then wrong)
#ifdef DPTM_EXP_HOTLOADS
#define CTL(ct) ((int64_t)lane)     // experiment: every tile reads the same cache-resident rows
#else
#define CTL(ct) (ct)
#endif
#ifdef DPTM_EXP_NOSYNC
#define TB_SYNC() wave_fence()
#else
#define TB_SYNC() __syncthreads()
#endif

// MAXT: the largest workgroup the instance is launched with.  Tiles of up to four wavefronts get an instance of their
// own whose register budget is twice that of the eight-wavefront tile (one wavefront per SIMD instead of two).
// MODE 0: the whole backward pass of both blocks and the pooling in one launch.  Large batches split it (see TbBwdArgs):
// MODE 1 = the pooling alone, MODE 2 = the attention half of block p.blk (output projection, attention, Q / K / V); the
// row-wise half of a block runs in tb_row_bwd_kernel between them.
template <int W, int MAXT, int MODE>
__device__ __forceinline__ void tb_bwd_body(const TbBwdArgs& p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NW = blockDim.x >> 6, rows_tile = NW * 32;
    const int img = img_floats(NW);                  // room for a row-major [rows][68] or a transposed [64][rows + 4] image
    float* bufA = lds;
    float* bufB = bufA + img;
    float* xreg = bufB + img;
    float* vacc = xreg + kXreg;                     // [kTbVecs][64] column sums of this workgroup
    float* bnv = vacc + kTbVecs * 64;               // [scale | shift] of the second BatchNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int S = p.S;
    const int row = wave * 32 + j;
    float* own_a = bufA + wave * 32 * kLdsStride;
    float* own_b = bufB + wave * 32 * kLdsStride;
    float* mine_a = own_a + j * kLdsStride;
    float* mine_b = own_b + j * kLdsStride;
    const int att_start = min(row / W, S - 1) * W;
    const int row0 = wave * 32;
    float* lane_a = bufA + 4 * h * kTs + row;       // the lane's place in a transposed image (chain_store_t)
    float* lane_b = bufB + 4 * h * kTs + row;
    float* part = p.part + (size_t)blockIdx.x * kTbPartStride;
    for (int i = tid; i < kTbVecs * 64; i += blockDim.x) vacc[i] = 0.f;
    bn_prepare(p.bn2, bnv, tid, false);
    __syncthreads();
    bool first = true;
    // (the eight-wavefront tile uses every byte of LDS a workgroup can have: the tile slot lives in the unused part of
    // the pooling-bias slot of the column sums, of which only the first float is ever summed)
    int* tile_slot = reinterpret_cast<int*>(vacc + kSPoolB * 64 + 32);
    int tile_round = 0;
    PH_DECL
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue, tile_slot, tile_round); tile < p.ntiles; tile = next_tile(p.queue, tile_slot, tile_round)) {
        const int64_t site0 = tile * S;
        const int sites_here = (int)min((int64_t)S, p.n_sites - site0);
        const int rows_valid = sites_here * W;
        const bool row_ok = row < rows_valid;
        const int64_t grow = site0 * W + row;
        // Addresses of the kept arrays: the tile and the wavefront are the same for every lane, so (tile, wavefront) goes
        // into a scalar base per array and the lane adds one 32-bit offset that all arrays share -- a 64-bit address per
        // array and lane would cost two registers each, and there are two dozen arrays.
        const int64_t tw = (int64_t)__builtin_amdgcn_readfirstlane((int)tile) * NW + wave;     // chain-tiled slab of this wavefront
        unsigned ln = (unsigned)lane;
#define AT4(arr, u) ((arr) + tw * 512 + (u) * 64)[local_index(ln)]
#define AT1(arr) ((arr) + tw * 64)[local_index(ln)]
        f32x4 dy[8];   // gradient w.r.t. the output of the block in hand
        TB_SYNC();   // the previous tile's last exchange through xreg has been read
        // ---- attention pooling backward: dx_t = alpha_t dout + dz_t w, dz_t = alpha_t (dalpha_t - sum alpha dalpha)(1 - a_t^2)
        if (MODE != 2) {
            const f32x4* tv1 = p.P + kOffTb + kTbSize + kTbVec;
            const f32x4* pv = p.P + kOffPoolV;
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = keep_load(&AT4(p.sv[1].xh2, u));
            ln_affine(x, tv1 + kVLn2G * kVec64, tv1 + kVLn2B * kVec64, h, x);
            const f32x2 pa = AT1(p.pool);
            const float* dsite = p.dout + (site0 + row / W) * p.ldd + p.col_off;
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
            if (h == 0) xreg[row] = row_ok ? pa[1] * dal : 0.f;
            TB_SYNC();
            float dot = 0.f;
#pragma unroll 3
            for (int t = 0; t < W; ++t) dot += xreg[att_start + t];
            const float dz = row_ok ? pa[1] * (dal - dot) * (1.0f - pa[0] * pa[0]) : 0.f;
            const float al = row_ok ? pa[1] : 0.f;
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dy[u] = al * ds[u] + dz * pv[h * 8 + u];
                t[u] = dz * x[u];
            }
            chain_store_t(lane_a, t);
            wave_fence();
            atomicAdd(&vacc[kSPoolW * 64 + lane], colsum32_t(bufA, row0, lane));
            float dzs = h == 0 ? dz : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dzs += __shfl_xor(dzs, o);
            if (lane == 0) atomicAdd(&vacc[kSPoolB * 64], dzs);
            wave_fence();
            TB_SYNC();   // the scores in xreg have been read
        }
        PH_MARK(0)   // pooling backward
        if (MODE == 1) {       // the pooling alone: its input gradient goes to the row-wise kernel of block 1
#pragma unroll
            for (int u = 0; u < 8; ++u) AT4(p.gout, u) = dy[u];
            first = false;
            continue;
        }
        const int blk_hi = MODE == 2 ? p.blk : 1, blk_lo = MODE == 2 ? p.blk : 0;
#pragma unroll 1
        for (int blk = blk_hi; blk >= blk_lo; --blk) {
            const f32x4* T = p.P + kOffTb + blk * kTbSize;
            const f32x4* tv = T + kTbVec;
            const f32x4* TT = p.PT + blk * kBtSize;
            const TbSave& sv = p.sv[blk];
            float* pu = part + (size_t)blk * 12 * 4096;
            float* va = vacc + blk * kSlotsPerBlock * 64 + lane;
            const f32x2 iv = AT1(sv.inv);
            // ---- LayerNorm 2 backward: dy -> dsum = d(h + f2)
            f32x4 dsum[8];
            if (MODE == 2) {   // the row-wise half ran in tb_row_bwd_kernel: d(x + attention output) arrives in p.din
#pragma unroll
                for (int u = 0; u < 8; ++u) dsum[u] = AT4(p.gin, u);
            }
            if (MODE == 0) {
            {
                f32x4 xh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xh[u] = keep_load(&AT4(sv.xh2, u));
                ln_backward(dy, xh, iv[1], tv + kVLn2G * kVec64, h, dsum, lane_a, lane_b);
                wave_fence();
                atomicAdd(va + kSG2 * 64, colsum32_t(bufA, row0, lane));
                atomicAdd(va + kSB2 * 64, colsum32_t(bufB, row0, lane));
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
                for (int u = 0; u < 8; ++u) xh[u] = keep_load(&AT4(sv.xh1, u));
                ln_affine(xh, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h, hx);
            }
            // ---- feed-forward layers backward; the hidden layer is rebuilt in four 64-wide chunks
            f32x16 acc_dh[2];
            x_to_acc(dsum, acc_dh);
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                // all row products of the chunk first, so that little stays in registers across the weight-gradient units
                f32x4 d1[8];
                {
                    f32x4 g[8];
                    {
                        f32x16 acc_f[2];
                        acc_init_bias<2>(acc_f, tv + 9 * kVec64 + 8 * c, h, 32);
                        gemm_regs<8, 2, 2>(acc_f, T + kTbF1 + (2 * c) * 8 * 64, hx, lane);     // f1_c = h W1[:, chunk] + b1
                        f32x4 u_[8];
                        acc_to_x<2>(acc_f, u_);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = u_[u][e];
                                const float sg = fast_rcp(1.0f + __expf(-a));
                                u_[u][e] = a * sg;                                   // swish(f1_c)
                                g[u][e] = sg * (1.0f + a * (1.0f - sg));             // swish'(f1_c)
                            }
                        chain_store_t(lane_a, u_);
                    }
                    f32x16 acc_u[2];
                    acc_zero<2>(acc_u);
                    gemm_regs<8, 2, 2>(acc_u, TT + kBtW2T + (2 * c) * 8 * 64, df2, lane);   // d(u_c) = d(f2) W2[chunk, :]^T
                    acc_to_x<2>(acc_u, d1);
#pragma unroll
                    for (int u = 0; u < 8; ++u) d1[u] *= g[u];                             // d(f1_c)
                }
                chain_store_t(lane_b, df2);
                if (c == 0) {
                    wave_fence();
                    atomicAdd(va + kSFb2 * 64, colsum32_t(bufB, row0, lane));
                }
                PH_MARK(2)   // feed-forward row products
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + (4 + c) * 4096, first, lane, wave, rows_valid);     // dW2[chunk, :] += u_c^T d(f2)
                PH_MARK(3)   // feed-forward weight gradients
                chain_store_t(lane_a, hx);
                chain_store_t(lane_b, d1);
                wave_fence();
                atomicAdd(va + (kSFb1 + c) * 64, colsum32_t(bufB, row0, lane));
                gemm_k256_chunk(acc_dh, TT + kBtW1T, c, d1, lane);                         // d(h) += d(f1_c) W1[:, chunk]^T
                PH_MARK(2)
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + c * 4096, first, lane, wave, rows_valid);           // dW1[:, chunk] += h^T d(f1_c)
                PH_MARK(3)
            }
            PH_MARK(2)
            // ---- LayerNorm 1 backward: d(h) -> dsum = d(x + ao)
            {
                f32x4 dh[8], xh[8];
                acc_to_x<2>(acc_dh, dh);
#pragma unroll
                for (int u = 0; u < 8; ++u) xh[u] = keep_load(&AT4(sv.xh1, u));
                ln_backward(dh, xh, iv[0], tv + kVLn1G * kVec64, h, dsum, lane_a, lane_b);
                wave_fence();
                atomicAdd(va + kSG1 * 64, colsum32_t(bufA, row0, lane));
                atomicAdd(va + kSB1 * 64, colsum32_t(bufB, row0, lane));
                wave_fence();
            }
            PH_MARK(4)   // LayerNorm 1 backward
            }   // MODE == 0: row-wise half
            // ---- output projection backward
            f32x4 dO[8];
            float D[2];        // D = dO . o per head (= sum_t p_t dO.v_t), heads 2h and 2h+1 of the lane's row
            {
                f32x4 dao[8], o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dao[u] = dsum[u];
                    o[u] = keep_load(&AT4(sv.o, u));
                }
                dropout_chain(dao, grow, h, seed, p.stream_base + 2 * blk, p.thr, p.scale);
                chain_store_t(lane_a, o);
                chain_store_t(lane_b, dao);
                wave_fence();
                atomicAdd(va + kSBo * 64, colsum32_t(bufB, row0, lane));
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + 8 * 4096, first, lane, wave, rows_valid);           // dWo += o^T d(ao)
                f32x16 acc[2];
                acc_zero<2>(acc);
                gemm_regs<8, 2>(acc, TT + kBtWoT, dao, lane);                              // d(o) = d(ao) Wo^T
                acc_to_x<2>(acc, dO);
                // head hd owns feature groups 2hd, 2hd+1; each half-wave holds half of every group
                float d4[4];
#pragma unroll
                for (int hd = 0; hd < 4; ++hd) {
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(dO[2 * hd][e], o[2 * hd][e], fmaf(dO[2 * hd + 1][e], o[2 * hd + 1][e], s));
                    d4[hd] = xhalf_sum(s);
                }
                D[0] = h ? d4[2] : d4[0];
                D[1] = h ? d4[3] : d4[1];
            }
            PH_MARK(5)   // output projection backward
            const f32x4 st = AT1(sv.stat);
            f32x16 acc_dx[2];   // d(x) = dsum + dq Wq^T + dk Wk^T + dv Wv^T, summed as the terms appear
            // ---- attention backward, lane = query row: dq
            {
                f32x4 dq[8];
                {
                    f32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.k, u));
                    chain_store(mine_a, h, t);
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.v, u));
                    chain_store(mine_b, h, t);
                    float qs[2][16], doh[2][16], gh[2][16];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.q, u));
                    x_to_heads(t, qs, kQScale);
                    x_to_heads(dO, doh, 1.0f);
                    TB_SYNC();   // K and V of the tile are in LDS
#pragma unroll
                    for (int hd = 0; hd < 2; ++hd) {
                        const int off = att_start * kLdsStride + 32 * h + 16 * hd;
                        attn_bwd_q<W>(bufA + off, bufB + off, qs[hd], doh[hd], st[2 * hd], st[2 * hd + 1], D[hd], gh[hd]);
                    }
                    heads_to_x(gh, dq, row_ok ? 0.25f : 0.f);
                }
                PH_MARK(6)   // attention backward, query rows
                TB_SYNC();   // K and V have been read
                PH_MARK(10)  // barrier after a vector-unit phase
                {
                    f32x4 xin[8];
                    if (blk == 0) {
                        x1_from_conv(p.cs.a2 + tw * 512, p.cs.s + tw * 512, local_index(ln), bnv, h, row_ok, xin);
                    } else {
                        const f32x4* tv0 = p.P + kOffTb + kTbVec;
#pragma unroll
                        for (int u = 0; u < 8; ++u) xin[u] = keep_load(&AT4(p.sv[0].xh2, u));
                        ln_affine(xin, tv0 + kVLn2G * kVec64, tv0 + kVLn2B * kVec64, h, xin);
                    }
                    chain_store_t(lane_a, xin);
                }
                chain_store_t(lane_b, dq);
                wave_fence();
                atomicAdd(va + kSBq * 64, colsum32_t(bufB, row0, lane));
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + 9 * 4096, first, lane, wave, rows_valid);           // dWq += x^T dq
                x_to_acc(dsum, acc_dx);
                gemm_regs<8, 2>(acc_dx, TT + kBtWqT, dq, lane);
            }
            PH_MARK(7)   // Q / K / V projections backward
            // ---- attention backward, lane = key row: dk, dv
            {
                f32x4 dk[8], dv[8];
                {
                    f32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.q, u)) * kQScale;
                    chain_store(mine_a, h, t);
                    chain_store(mine_b, h, dO);
#pragma unroll
                    for (int hd = 0; hd < 2; ++hd) {
                        xreg[row * 4 + 2 * h + hd] = st[2 * hd];
                        xreg[rows_tile * 4 + row * 4 + 2 * h + hd] = st[2 * hd + 1];
                        xreg[2 * rows_tile * 4 + row * 4 + 2 * h + hd] = D[hd];
                    }
                    float kh[2][16], vh[2][16], gk[2][16], gv[2][16];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.k, u));
                    x_to_heads(t, kh, 1.0f);
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = keep_load(&AT4(sv.v, u));
                    x_to_heads(t, vh, 1.0f);
                    TB_SYNC();   // scaled Q, dO and the statistics of the tile are in LDS
#pragma unroll
                    for (int hd = 0; hd < 2; ++hd) {
                        const int off = att_start * kLdsStride + 32 * h + 16 * hd;
                        attn_bwd_kv<W>(bufA + off, bufB + off, xreg + att_start * 4 + 2 * h + hd, rows_tile * 4, kh[hd], vh[hd], gk[hd],
                                       gv[hd]);
                    }
                    heads_to_x(gk, dk, row_ok ? kLn2 : 0.f);
                    heads_to_x(gv, dv, row_ok ? 1.0f : 0.f);
                }
                PH_MARK(8)   // attention backward, key rows
                TB_SYNC();   // Q and dO have been read
                PH_MARK(10)
                {
                    f32x4 xin[8];
                    if (blk == 0) {
                        x1_from_conv(p.cs.a2 + tw * 512, p.cs.s + tw * 512, local_index(ln), bnv, h, row_ok, xin);
                    } else {
                        const f32x4* tv0 = p.P + kOffTb + kTbVec;
#pragma unroll
                        for (int u = 0; u < 8; ++u) xin[u] = keep_load(&AT4(p.sv[0].xh2, u));
                        ln_affine(xin, tv0 + kVLn2G * kVec64, tv0 + kVLn2B * kVec64, h, xin);
                    }
                    chain_store_t(lane_a, xin);
                }
                chain_store_t(lane_b, dk);
                wave_fence();
                atomicAdd(va + kSBk * 64, colsum32_t(bufB, row0, lane));
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + 10 * 4096, first, lane, wave, rows_valid);          // dWk += x^T dk
                gemm_regs<8, 2>(acc_dx, TT + kBtWkT, dk, lane);
                chain_store_t(lane_b, dv);
                wave_fence();
                atomicAdd(va + kSBv * 64, colsum32_t(bufB, row0, lane));
                TB_SYNC();
                dw_unit_t<MAXT == 256>(bufA, bufB, NW, xreg, pu + 11 * 4096, first, lane, wave, rows_valid);          // dWv += x^T dv
                gemm_regs<8, 2>(acc_dx, TT + kBtWvT, dv, lane);
            }
            acc_to_x<2>(acc_dx, dy);
            PH_MARK(7)
        }
        if (MODE == 2 && p.blk == 1) {   // block 1 done: the gradient goes to the row-wise kernel of block 0
#pragma unroll
            for (int u = 0; u < 8; ++u) AT4(p.gout, u) = dy[u];
            first = false;
            continue;
        }
        // d(x1) leaves chain-tiled; its column sums (with swish(a2)) feed the second BatchNorm's backward pass
        {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                AT4(p.dx1, u) = dy[u];
                const f32x4 a = AT4(p.cs.a2, u);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[u][e] = dy[u][e] * swishf(a[e]);
            }
            chain_store_t(lane_a, dy);
            chain_store_t(lane_b, t);
            wave_fence();
            atomicAdd(&vacc[kSBn2A * 64 + lane], colsum32_t(bufA, row0, lane));
            atomicAdd(&vacc[kSBn2B * 64 + lane], colsum32_t(bufB, row0, lane));
            wave_fence();
        }
        first = false;
    }
    __syncthreads();
    // a workgroup that was handed no tile still owns a partial buffer that the summing launch reads: the units (and, below,
    // the vector slots) this launch is responsible for
    if (first && MODE == 0) zero_floats(part, kTbUnits * 4096);
    if (first && MODE == 2) zero_floats(part + (p.blk * 12 + 8) * 4096, 4 * 4096);
    for (int i = tid; i < kTbVecs * 64; i += blockDim.x) {
        const int slot = i >> 6;
        bool mine = true;      // vector slots of this launch (the split form's launches share a workgroup's buffer)
        if (MODE == 1) mine = slot == kSPoolW || slot == kSPoolB;
        if (MODE == 2) {
            const int s_ = slot - p.blk * kSlotsPerBlock;
            mine = s_ == kSBo || s_ == kSBq || s_ == kSBk || s_ == kSBv;
        }
        const bool ends_here = MODE == 0 || (MODE == 2 && p.blk == 0);     // the launch that forms d(x1)
        if (slot == kSBn2A) {
            if (ends_here) atomicAdd(p.bn2_sums + (i & 63), vacc[i]);
        } else if (slot == kSBn2B) {
            if (ends_here) atomicAdd(p.bn2_sums + 64 + (i & 63), vacc[i]);
        } else if (mine) {
            part[kTbUnits * 4096 + i] = vacc[i];
        }
    }
    PH_FLUSH
#undef AT4
#undef AT1
}

#define DPTM_PAIR_TB_BODIES 1
#include "dptm_train_pair.h"

// ------------------------------------------------------------------------------------ kernels

template <int W>
__global__ __launch_bounds__(512) void tb_fwd_kernel(TbFwdArgs p) { tb_fwd_body<W>(p); }
template <int W, int MAXT, int MODE>
__global__ __launch_bounds__(MAXT) void tb_bwd_kernel(TbBwdArgs p) { tb_bwd_body<W, MAXT, MODE>(p); }
// one site in four wavefronts (dptm_train_pair.h)
template <int W>
__global__ __launch_bounds__(256) void tb_fwd_pair_kernel(TbFwdArgs p) { tb_fwd_pair_body<W>(p); }
template <int W>
__global__ __launch_bounds__(256) void tb_bwd_pair_kernel(TbBwdArgs p) { tb_bwd_pair_body<W>(p); }

// The three branches' launch as ONE (grid.y = branch; four-wavefront tiles, a tile per workgroup): see conv_a_fwd3_kernel
struct TbFwdArgs3 { TbFwdArgs a[3]; };
struct TbBwdArgs3 { TbBwdArgs a[3]; };
// (grid.y = 0 is the longest branch, W = 51: its workgroups are handed out first; MAXT: 256 for four-wavefront tiles, 512 for eight)
template <int MAXT>
__global__ __launch_bounds__(MAXT) void tb_fwd3_kernel(TbFwdArgs3 p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x < p.a[0].ntiles) tb_fwd_body<21>(p.a[0]);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x < p.a[1].ntiles) tb_fwd_body<33>(p.a[1]);
    } else {
        if ((int64_t)blockIdx.x < p.a[2].ntiles) tb_fwd_body<51>(p.a[2]);
    }
}
template <int MAXT>
__global__ __launch_bounds__(MAXT) void tb_bwd3_kernel(TbBwdArgs3 p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x < p.a[0].ntiles) tb_bwd_body<21, MAXT, 0>(p.a[0]);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x < p.a[1].ntiles) tb_bwd_body<33, MAXT, 0>(p.a[1]);
    } else {
        if ((int64_t)blockIdx.x < p.a[2].ntiles) tb_bwd_body<51, MAXT, 0>(p.a[2]);
    }
}

__global__ __launch_bounds__(256) void tb_fwd3_pair_kernel(TbFwdArgs3 p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x < p.a[0].ntiles) tb_fwd_pair_body<21>(p.a[0]);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x < p.a[1].ntiles) tb_fwd_pair_body<33>(p.a[1]);
    } else {
        if ((int64_t)blockIdx.x < p.a[2].ntiles) tb_fwd_pair_body<51>(p.a[2]);
    }
}
__global__ __launch_bounds__(256) void tb_bwd3_pair_kernel(TbBwdArgs3 p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x < p.a[0].ntiles) tb_bwd_pair_body<21>(p.a[0]);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x < p.a[1].ntiles) tb_bwd_pair_body<33>(p.a[1]);
    } else {
        if ((int64_t)blockIdx.x < p.a[2].ntiles) tb_bwd_pair_body<51>(p.a[2]);
    }
}

// ------------------------------------------------------------------------------------ row-wise half, large batches

// index of the vectors in a workgroup's partial buffer (behind its kRowUnits units)
enum RowVecSlot { kRG2 = 0, kRB2, kRFb2, kRFb1, kRG1 = kRFb1 + 4, kRB1 };
static_assert(kRB1 < kRowVecs, "vector slots of the row-wise kernel");

// LayerNorm 2, the feed-forward layers and LayerNorm 1 of ONE TransformerBlock, backward: nothing here looks beyond a
// row, so a tile is any 128 consecutive rows of the chain-tiled arrays (four 32-row blocks, padding rows of the forward
// tiling included: their gradients are zero), four wavefronts, two 34 KB transposed images -- and TWO workgroups fit a
// CU, whose barrier phases then interleave on every SIMD.  Wavefront w owns block w of every 64 x 64 weight-gradient
// unit over all 128 rows: no exchange between row halves.
__global__ __launch_bounds__(256, 2) void tb_row_bwd_kernel(TbRowBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;                               // transposed images [64][kTsRow]
    float* bufB = bufA + 64 * kTsRow;
    float* vacc = bufB + 64 * kTsRow;                // [kRowVecs][64] column sums of this workgroup
    int* tile_slot = reinterpret_cast<int*>(vacc + kRowVecs * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const unsigned seed = p.step[kStepSeed];
    const int row = wave * 32 + j, row0 = wave * 32;
    float* lane_a = bufA + 4 * h * kTsRow + row;
    float* lane_b = bufB + 4 * h * kTsRow + row;
    float* part = p.part + (size_t)blockIdx.x * kRowPartStride;
    float* va = vacc + lane;
    for (int i = tid; i < kRowVecs * 64; i += blockDim.x) vacc[i] = 0.f;
    __syncthreads();
    const f32x4* T = p.P;
    const f32x4* tv = T + kTbVec;
    const f32x4* TT = p.PT;
    const int64_t ntiles = (p.nblocks + 3) >> 2;
    bool first = true;
    int tile_round = 0;
#pragma unroll 1
    for (int64_t tile = next_tile(p.queue, tile_slot, tile_round); tile < ntiles; tile = next_tile(p.queue, tile_slot, tile_round)) {
        const int64_t rb = (int64_t)__builtin_amdgcn_readfirstlane((int)tile) * 4 + wave;     // this wavefront's 32-row block
        const bool blk_ok = rb < p.nblocks;                    // the last tile may be short of blocks: those wavefronts carry zeros
        const int64_t tw = blk_ok ? rb : p.nblocks - 1;
        // the row's index in the batch (it seeds the dropout mask): forward tile rb / NWf holds S sites from site tile*S on
        const int64_t ft = rb / p.NWf;
        const int64_t grow = ft * p.S * p.wlen + (rb - ft * p.NWf) * 32 + j;
        unsigned ln = (unsigned)lane;
#define AT4(arr, u) ((arr) + tw * 512 + (u) * 64)[local_index(ln)]
#define AT1(arr) ((arr) + tw * 64)[local_index(ln)]
        const f32x2 iv = AT1(p.inv);
        // ---- LayerNorm 2 backward: dy -> dsum = d(h + f2)
        f32x4 dsum[8];
        {
            f32x4 dy[8], xh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dy[u] = AT4(p.din, u);
                if (!blk_ok) dy[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                xh[u] = keep_load(&AT4(p.xh2, u));
            }
            ln_backward<kTsRow>(dy, xh, iv[1], tv + kVLn2G * kVec64, h, dsum, lane_a, lane_b);
            wave_fence();
            atomicAdd(va + kRG2 * 64, colsum32_t<kTsRow>(bufA, row0, lane));
            atomicAdd(va + kRB2 * 64, colsum32_t<kTsRow>(bufB, row0, lane));
            wave_fence();
        }
        f32x4 df2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) df2[u] = dsum[u];
        dropout_chain(df2, grow, h, seed, p.stream, p.thr, p.scale);
        f32x4 hx[8];
        {
            f32x4 xh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xh[u] = keep_load(&AT4(p.xh1, u));
            ln_affine(xh, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h, hx);
        }
        // ---- feed-forward layers backward; the hidden layer is rebuilt in four 64-wide chunks
        f32x16 acc_dh[2];
        x_to_acc(dsum, acc_dh);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            f32x4 d1[8];
            {
                f32x4 g[8];
                {
                    f32x16 acc_f[2];
                    acc_init_bias<2>(acc_f, tv + 9 * kVec64 + 8 * c, h, 32);
                    gemm_regs<8, 2, 2>(acc_f, T + kTbF1 + (2 * c) * 8 * 64, hx, lane);     // f1_c = h W1[:, chunk] + b1
                    f32x4 u_[8];
                    acc_to_x<2>(acc_f, u_);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = u_[u][e];
                            const float sg = fast_rcp(1.0f + __expf(-a));
                            u_[u][e] = a * sg;                                   // swish(f1_c)
                            g[u][e] = sg * (1.0f + a * (1.0f - sg));             // swish'(f1_c)
                        }
                    chain_store_t<kTsRow>(lane_a, u_);
                }
                f32x16 acc_u[2];
                acc_zero<2>(acc_u);
                gemm_regs<8, 2, 2>(acc_u, TT + kBtW2T + (2 * c) * 8 * 64, df2, lane);   // d(u_c) = d(f2) W2[chunk, :]^T
                acc_to_x<2>(acc_u, d1);
#pragma unroll
                for (int u = 0; u < 8; ++u) d1[u] *= g[u];                             // d(f1_c)
            }
            chain_store_t<kTsRow>(lane_b, df2);
            if (c == 0) {
                wave_fence();
                atomicAdd(va + kRFb2 * 64, colsum32_t<kTsRow>(bufB, row0, lane));
            }
            __syncthreads();
            dw_unit_row(bufA, bufB, part + (4 + c) * 4096, first, lane, wave);         // dW2[chunk, :] += u_c^T d(f2)
            __syncthreads();   // the images have been read
            chain_store_t<kTsRow>(lane_a, hx);
            chain_store_t<kTsRow>(lane_b, d1);
            wave_fence();
            atomicAdd(va + (kRFb1 + c) * 64, colsum32_t<kTsRow>(bufB, row0, lane));
            gemm_k256_chunk(acc_dh, TT + kBtW1T, c, d1, lane);                         // d(h) += d(f1_c) W1[:, chunk]^T
            __syncthreads();
            dw_unit_row(bufA, bufB, part + c * 4096, first, lane, wave);               // dW1[:, chunk] += h^T d(f1_c)
            __syncthreads();
        }
        // ---- LayerNorm 1 backward: d(h) -> d(x + attention output), which goes to the attention half
        {
            f32x4 dh[8], xh[8];
            acc_to_x<2>(acc_dh, dh);
#pragma unroll
            for (int u = 0; u < 8; ++u) xh[u] = keep_load(&AT4(p.xh1, u));
            ln_backward<kTsRow>(dh, xh, iv[0], tv + kVLn1G * kVec64, h, dsum, lane_a, lane_b);
            wave_fence();
            atomicAdd(va + kRG1 * 64, colsum32_t<kTsRow>(bufA, row0, lane));
            atomicAdd(va + kRB1 * 64, colsum32_t<kTsRow>(bufB, row0, lane));
            wave_fence();
        }
        if (blk_ok) {
#pragma unroll
            for (int u = 0; u < 8; ++u) AT4(p.dout, u) = dsum[u];
        }
        first = false;
#undef AT4
#undef AT1
    }
    __syncthreads();
    if (first) zero_floats(part, kRowUnits * 4096);
    for (int i = tid; i < kRowVecs * 64; i += blockDim.x) part[kRowUnits * 4096 + i] = vacc[i];
}

static size_t row_lds() { return ((size_t)2 * 64 * kTsRow + kRowVecs * 64 + 64) * sizeof(float); }

// ------------------------------------------------------------------------------------ partial sums -> gradients

// One launch sums every unit and vector of the partial buffers.  grid.x = work item: 16 per unit (256 accumulator
// elements each: thread o of a unit is element [block o>>10][register (o>>6)&15][lane o&63]), then the vectors in
// chunks of 256 floats; grid.y = part group: group y adds parts y, y + gridDim.y, ... and its sum joins the
// destination through one atomic per element (the destinations start from the step's zeroed gradient buffer).
__device__ __forceinline__ void reduce_parts_body(const float* __restrict__ part, int nparts, const PartUnits& ut, const PartVecs& vt,
                                                  int stride) {
    const int item = blockIdx.x;
    const float* src;
    float* dst;
    if (item < ut.n * 16) {
        const int u = item >> 4, o = (item & 15) * 256 + threadIdx.x;
        const int b = o >> 10, rg = (o >> 6) & 15, lane = o & 63;
        const int k = 32 * (b >> 1) + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5), n = 32 * (b & 1) + (lane & 31);
        if (k >= ut.rows[u]) return;
        src = part + ut.off[u] + o;
        dst = ut.dst[u] + (size_t)k * ut.ld[u] + n;
    } else {
        int c = item - ut.n * 16, v = 0;
        while (v < vt.n && c >= (vt.len[v] + 255) / 256) {
            c -= (vt.len[v] + 255) / 256;
            ++v;
        }
        const int f = c * 256 + threadIdx.x;
        if (v >= vt.n || f >= vt.len[v]) return;
        src = part + vt.off[v] + f;
        dst = vt.dst[v] + f;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four loads in flight: the launch is bound by their latency
    int pi = blockIdx.y;
    const int stepp = gridDim.y;
    for (; pi + 3 * stepp < nparts; pi += 4 * stepp) {
        a0 += src[(size_t)pi * stride];
        a1 += src[(size_t)(pi + stepp) * stride];
        a2 += src[(size_t)(pi + 2 * stepp) * stride];
        a3 += src[(size_t)(pi + 3 * stepp) * stride];
    }
    for (; pi < nparts; pi += stepp) a0 += src[(size_t)pi * stride];
    atomicAdd(dst, (a0 + a1) + (a2 + a3));
}

__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int nparts, PartUnits ut, PartVecs vt,
                                                           int stride) {
    reduce_parts_body(part, nparts, ut, vt, stride);
}
// the partial buffers of the three branches in one launch: grid.z = branch
struct ReduceParts3 {
    const float* part[3];
    int nparts[3];
    PartUnits ut[3];
    PartVecs vt[3];
};
__global__ __launch_bounds__(256) void reduce_parts3_kernel(ReduceParts3 r) {
    if (blockIdx.z == 0)
        reduce_parts_body(r.part[0], r.nparts[0], r.ut[0], r.vt[0], kTbPartStride);
    else if (blockIdx.z == 1)
        reduce_parts_body(r.part[1], r.nparts[1], r.ut[1], r.vt[1], kTbPartStride);
    else
        reduce_parts_body(r.part[2], r.nparts[2], r.ut[2], r.vt[2], kTbPartStride);
}

hipError_t t_reduce_parts(const float* part, int nparts, const PartUnits& u, const PartVecs& v, hipStream_t st, int part_stride) {
    int items = u.n * 16;
    for (int i = 0; i < v.n; ++i) items += (v.len[i] + 255) / 256;
    if (items == 0 || nparts == 0) return hipSuccess;
    int groups = (nparts + 15) / 16;
    groups = groups > 8 ? 8 : groups;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)items, (unsigned)groups), dim3(256), 0, st, part, nparts, u, v, part_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ launchers

static size_t fwd_lds(int NW) { return ((size_t)2 * NW * 32 * kLdsStride + 2 * NW * 32 + 128) * sizeof(float); }
static size_t bwd_lds(int NW) { return ((size_t)2 * img_floats(NW) + kXreg + kTbVecs * 64 + 128) * sizeof(float); }

// A one-site tile of four wavefronts runs the pair kernels (DPTM_TRAIN_PAIR=0: the plain ones, for A/B measurements)
bool t_pair_tile(int S, int NW) {
    static const bool on = [] {
        const char* e = std::getenv("DPTM_TRAIN_PAIR");
        return !(e && e[0] == '0');
    }();
    return on && S == 1 && NW == 4;
}

template <int W>
static hipError_t launch_fwd(const TbFwdArgs& a, int NW, hipStream_t st) {
    static bool done[64] = {}, done_pair[64] = {};
    const int64_t grid = a.ntiles < 1024 ? a.ntiles : 1024;
    if (t_pair_tile(a.S, NW)) {
        hipError_t e = lds_opt_in(&tb_fwd_pair_kernel<W>, done_pair, fwd_pair_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(tb_fwd_pair_kernel<W>, dim3((unsigned)grid), dim3(256), fwd_pair_lds(), st, a);
        return hipGetLastError();
    }
    hipError_t e = lds_opt_in(&tb_fwd_kernel<W>, done, fwd_lds(8));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tb_fwd_kernel<W>, dim3((unsigned)grid), dim3(64 * NW), fwd_lds(NW), st, a);
    return hipGetLastError();
}

hipError_t t_tb_fwd(const TbFwdArgs& a, int wlen, int NW, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    switch (wlen) {
        case 21: return launch_fwd<21>(a, NW, st);
        case 33: return launch_fwd<33>(a, NW, st);
        default: return launch_fwd<51>(a, NW, st);
    }
}

template <int W>
static hipError_t launch_bwd(const TbBwdArgs& a, int NW, int grid, hipStream_t st) {
    static bool done[64] = {}, done_small[64] = {}, done_pair[64] = {};
    if (t_pair_tile(a.S, NW)) {
        hipError_t e = lds_opt_in(&tb_bwd_pair_kernel<W>, done_pair, bwd_pair_lds());
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(tb_bwd_pair_kernel<W>, dim3((unsigned)grid), dim3(256), bwd_pair_lds(), st, a);
        return hipGetLastError();
    }
    if (NW <= 4) {
        hipError_t e = lds_opt_in(&tb_bwd_kernel<W, 256, 0>, done_small, bwd_lds(4));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((tb_bwd_kernel<W, 256, 0>), dim3((unsigned)grid), dim3(64 * NW), bwd_lds(NW), st, a);
        return hipGetLastError();
    }
    hipError_t e = lds_opt_in(&tb_bwd_kernel<W, 512, 0>, done, bwd_lds(8));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((tb_bwd_kernel<W, 512, 0>), dim3((unsigned)grid), dim3(64 * NW), bwd_lds(NW), st, a);
    return hipGetLastError();
}

// DPTM_TRAIN_SPLIT=0 keeps large batches on the one-launch form (A/B measurements, tests of that form at large tiles)
static bool split_enabled() {
    static const bool v = [] {
        const char* e = std::getenv("DPTM_TRAIN_SPLIT");
        return !(e && e[0] == '0');
    }();
    return v;
}

// The split form of a large batch: pooling | row-wise half of block 1 | attention half of block 1 | row-wise half of
// block 0 | attention half of block 0, the gradient travelling through a.dx1 and a.dtmp in turn (the last launch leaves
// d(x1) in a.dx1), then the sums of the two sets of partial buffers.
template <int W>
static hipError_t launch_bwd_split(const TbBwdArgs& a, const TbGrads& g, int grid, hipStream_t st) {
    static bool done1[64] = {}, done2[64] = {}, done_row[64] = {};
    hipError_t e = lds_opt_in(&tb_bwd_kernel<W, 512, 1>, done1, bwd_lds(8));
    if (e == hipSuccess) e = lds_opt_in(&tb_bwd_kernel<W, 512, 2>, done2, bwd_lds(8));
    if (e == hipSuccess) e = lds_opt_in(&tb_row_bwd_kernel, done_row, row_lds());
    if (e != hipSuccess) return e;
    const int NWf = 8;
    const int64_t nblocks = a.ntiles * NWf, row_tiles = (nblocks + 3) / 4;
    const int row_grid = (int)(row_tiles < a.row_parts ? row_tiles : a.row_parts);
    f32x4_tr* const Da = a.dx1;
    f32x4_tr* const Db = a.dtmp;
    auto queue = [&](int k, int64_t tiles, int launch_grid) { return tiles > launch_grid ? a.queues + k : nullptr; };
    {
        TbBwdArgs p1 = a;
        p1.gout = Da;
        p1.queue = queue(0, a.ntiles, grid);
        hipLaunchKernelGGL((tb_bwd_kernel<W, 512, 1>), dim3((unsigned)grid), dim3(512), bwd_lds(8), st, p1);
    }
    for (int blk = 1; blk >= 0; --blk) {
        TbRowBwdArgs r;
        r.P = a.P + kOffTb + blk * kTbSize;
        r.PT = a.PT + blk * kBtSize;
        r.xh1 = a.sv[blk].xh1;
        r.xh2 = a.sv[blk].xh2;
        r.inv = a.sv[blk].inv;
        r.din = Da;
        r.dout = Db;
        r.part = a.row_part + (size_t)blk * a.row_parts * kRowPartStride;
        r.queue = queue(blk == 1 ? 1 : 3, row_tiles, row_grid);
        r.n_sites = a.n_sites;
        r.S = a.S;
        r.wlen = W;
        r.NWf = NWf;
        r.nblocks = nblocks;
        r.step = a.step;
        r.stream = a.stream_base + 2 * blk + 1;
        r.thr = a.thr;
        r.scale = a.scale;
        hipLaunchKernelGGL(tb_row_bwd_kernel, dim3((unsigned)row_grid), dim3(256), row_lds(), st, r);
        TbBwdArgs p2 = a;
        p2.blk = blk;
        p2.gin = Db;
        p2.gout = Da;          // (block 0 writes a.dx1 itself, which is the same array)
        p2.queue = queue(blk == 1 ? 2 : 4, a.ntiles, grid);
        hipLaunchKernelGGL((tb_bwd_kernel<W, 512, 2>), dim3((unsigned)grid), dim3(512), bwd_lds(8), st, p2);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // sums: the site-aware launches' buffers (4 units per block, their bias vectors, the pooling) ...
    {
        PartUnits ut;
        PartVecs vt;
        for (int b = 0; b < 2; ++b) {
            float* sq[4] = {g.wo[b], g.wq[b], g.wk[b], g.wv[b]};
            float* vb[4] = {g.bo[b], g.bq[b], g.bk[b], g.bv[b]};
            const int vs[4] = {kSBo, kSBq, kSBk, kSBv};
            for (int i = 0; i < 4; ++i) {
                ut.dst[ut.n] = sq[i]; ut.off[ut.n] = (b * 12 + 8 + i) * 4096; ut.ld[ut.n] = 64; ut.rows[ut.n] = 64; ++ut.n;
                vt.dst[vt.n] = vb[i]; vt.off[vt.n] = kTbUnits * 4096 + (b * kSlotsPerBlock + vs[i]) * 64; vt.len[vt.n] = 64; ++vt.n;
            }
        }
        vt.dst[vt.n] = g.poolw; vt.off[vt.n] = kTbUnits * 4096 + kSPoolW * 64; vt.len[vt.n] = 64; ++vt.n;
        vt.dst[vt.n] = g.poolb; vt.off[vt.n] = kTbUnits * 4096 + kSPoolB * 64; vt.len[vt.n] = 1; ++vt.n;
        e = t_reduce_parts(a.part, grid, ut, vt, st, kTbPartStride);
        if (e != hipSuccess) return e;
    }
    // ... and the row-wise launches' (8 units and 9 vectors per block)
    for (int b = 0; b < 2; ++b) {
        PartUnits ut;
        PartVecs vt;
        for (int c = 0; c < 4; ++c) {
            ut.dst[ut.n] = g.w1[b] + 64 * c; ut.off[ut.n] = c * 4096; ut.ld[ut.n] = 256; ut.rows[ut.n] = 64; ++ut.n;
        }
        for (int c = 0; c < 4; ++c) {
            ut.dst[ut.n] = g.w2[b] + 64 * c * 64; ut.off[ut.n] = (4 + c) * 4096; ut.ld[ut.n] = 64; ut.rows[ut.n] = 64; ++ut.n;
        }
        auto vec = [&](float* dst, int slot) {
            vt.dst[vt.n] = dst; vt.off[vt.n] = kRowUnits * 4096 + slot * 64; vt.len[vt.n] = 64; ++vt.n;
        };
        vec(g.ln2g[b], kRG2); vec(g.ln2b[b], kRB2); vec(g.b2[b], kRFb2);
        for (int c = 0; c < 4; ++c) vec(g.b1[b] + 64 * c, kRFb1 + c);
        vec(g.ln1g[b], kRG1); vec(g.ln1b[b], kRB1);
        e = t_reduce_parts(a.row_part + (size_t)b * a.row_parts * kRowPartStride, row_grid, ut, vt, st, kRowPartStride);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// units 0 .. 23 and vectors 0 .. 29 of a branch's plan: what the TransformerBlock / pooling backward launch leaves in a
// workgroup's partial buffer (the convolution kernels append theirs)
static void tb_bwd_plan(const TbGrads& g, PartPlan* plan) {
    PartUnits& ut = plan->units;
    PartVecs& vt = plan->vecs;
    ut = PartUnits();
    vt = PartVecs();
    ut.n = kTbUnits;
    vt.n = kTbVecs;
    for (int i = 0; i < kTbVecs; ++i) {
        vt.off[i] = kTbUnits * 4096 + i * 64;
        vt.len[i] = 64;
        vt.dst[i] = nullptr;
    }
    for (int b = 0; b < 2; ++b) {
        for (int c = 0; c < 4; ++c) {
            ut.dst[b * 12 + c] = g.w1[b] + 64 * c;              // dW1[:, 64c ..]  (raw [64][256])
            ut.ld[b * 12 + c] = 256;
            ut.dst[b * 12 + 4 + c] = g.w2[b] + 64 * c * 64;     // dW2[64c .., :]  (raw [256][64])
            ut.ld[b * 12 + 4 + c] = 64;
        }
        float* sq[4] = {g.wo[b], g.wq[b], g.wk[b], g.wv[b]};
        for (int i = 0; i < 4; ++i) {
            ut.dst[b * 12 + 8 + i] = sq[i];
            ut.ld[b * 12 + 8 + i] = 64;
        }
        float** vd = vt.dst + b * kSlotsPerBlock;
        vd[kSG2] = g.ln2g[b]; vd[kSB2] = g.ln2b[b]; vd[kSFb2] = g.b2[b];
        for (int c = 0; c < 4; ++c) vd[kSFb1 + c] = g.b1[b] + 64 * c;
        vd[kSG1] = g.ln1g[b]; vd[kSB1] = g.ln1b[b]; vd[kSBo] = g.bo[b];
        vd[kSBq] = g.bq[b]; vd[kSBk] = g.bk[b]; vd[kSBv] = g.bv[b];
        vt.len[b * kSlotsPerBlock + 13] = 0;      // spare slot: those column sums leave through atomics
    }
    vt.dst[kSPoolW] = g.poolw;
    vt.dst[kSPoolB] = g.poolb; vt.len[kSPoolB] = 1;
}

hipError_t t_tb_bwd(const TbBwdArgs& a, const TbGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st) {
    if (a.n_sites == 0) return hipSuccess;
    if (!tile_ok(wlen, a.S, NW, a.n_sites, a.ntiles)) return hipErrorInvalidValue;
    const int grid = (int)(a.ntiles < kTbMaxParts ? a.ntiles : kTbMaxParts);
    hipError_t e;
    if (NW == 8 && a.row_parts > 0 && !part_sums_deferred(a.ntiles) && a.row_part && a.dtmp && a.queues && split_enabled()) {
        switch (wlen) {
            case 21: e = launch_bwd_split<21>(a, g, grid, st); break;
            case 33: e = launch_bwd_split<33>(a, g, grid, st); break;
            default: e = launch_bwd_split<51>(a, g, grid, st); break;
        }
        plan->units.n = 0;
        plan->vecs.n = 0;
        return e;
    }
    switch (wlen) {
        case 21: e = launch_bwd<21>(a, NW, grid, st); break;
        case 33: e = launch_bwd<33>(a, NW, grid, st); break;
        default: e = launch_bwd<51>(a, NW, grid, st); break;
    }
    if (e != hipSuccess) return e;
    tb_bwd_plan(g, plan);
    PartUnits& ut = plan->units;
    PartVecs& vt = plan->vecs;
    if (part_sums_deferred(a.ntiles)) return hipSuccess;
    e = t_reduce_parts(a.part, grid, ut, vt, st);
    ut.n = 0;
    vt.n = 0;
    return e;
}

// ---- the three branches in one launch (tb_fwd3_kernel, tb_bwd3_kernel, reduce_parts3_kernel)

static_assert(sizeof(ReduceParts3) <= 4000 && sizeof(TbBwdArgs3) <= 4000 && sizeof(TbFwdArgs3) <= 4000, "kernel argument blocks");

template <typename A>
static bool tb3_ok(const A (&a)[3], int NW) {
    const int wl[3] = {21, 33, 51};
    if (NW != 4 && NW != 8) return false;
    for (int b = 0; b < 3; ++b)
        if (a[b].n_sites < 1 || a[b].ntiles > kTbMaxParts || !tile_ok(wl[b], a[b].S, NW, a[b].n_sites, a[b].ntiles)) return false;
    return true;
}

hipError_t t_tb_fwd3(const TbFwdArgs (&a)[3], int NW, hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    if (!tb3_ok(a, NW)) return hipErrorInvalidValue;
    const bool pair = t_pair_tile(a[0].S, NW) && t_pair_tile(a[1].S, NW) && t_pair_tile(a[2].S, NW);
    hipError_t e = pair      ? lds_opt_in(&tb_fwd3_pair_kernel, dp, fwd_pair_lds())
                   : NW == 4 ? lds_opt_in(&tb_fwd3_kernel<256>, d4, fwd_lds(4))
                             : lds_opt_in(&tb_fwd3_kernel<512>, d8, fwd_lds(8));
    if (e != hipSuccess) return e;
    TbFwdArgs3 p;
    int64_t gx = 1;
    for (int b = 0; b < 3; ++b) {
        p.a[b] = a[b];
        gx = a[b].ntiles > gx ? a[b].ntiles : gx;
    }
    if (pair)
        hipLaunchKernelGGL(tb_fwd3_pair_kernel, dim3((unsigned)gx, 3), dim3(256), fwd_pair_lds(), st, p);
    else if (NW == 4)
        hipLaunchKernelGGL(tb_fwd3_kernel<256>, dim3((unsigned)gx, 3), dim3(256), fwd_lds(4), st, p);
    else
        hipLaunchKernelGGL(tb_fwd3_kernel<512>, dim3((unsigned)gx, 3), dim3(512), fwd_lds(8), st, p);
    return hipGetLastError();
}

hipError_t t_tb_bwd3(const TbBwdArgs (&a)[3], const TbGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st) {
    static bool d4[64] = {}, d8[64] = {}, dp[64] = {};
    if (!tb3_ok(a, NW)) return hipErrorInvalidValue;
    const bool pair = t_pair_tile(a[0].S, NW) && t_pair_tile(a[1].S, NW) && t_pair_tile(a[2].S, NW);
    hipError_t e = pair      ? lds_opt_in(&tb_bwd3_pair_kernel, dp, bwd_pair_lds())
                   : NW == 4 ? lds_opt_in(&tb_bwd3_kernel<256>, d4, bwd_lds(4))
                             : lds_opt_in(&tb_bwd3_kernel<512>, d8, bwd_lds(8));
    if (e != hipSuccess) return e;
    TbBwdArgs3 p;
    int64_t gx = 1;
    for (int b = 0; b < 3; ++b) {
        p.a[b] = a[b];
        p.a[b].queue = nullptr;        // a tile per workgroup
        gx = a[b].ntiles > gx ? a[b].ntiles : gx;
        tb_bwd_plan(g[b], &plan[b]);
    }
    if (pair)
        hipLaunchKernelGGL(tb_bwd3_pair_kernel, dim3((unsigned)gx, 3), dim3(256), bwd_pair_lds(), st, p);
    else if (NW == 4)
        hipLaunchKernelGGL(tb_bwd3_kernel<256>, dim3((unsigned)gx, 3), dim3(256), bwd_lds(4), st, p);
    else
        hipLaunchKernelGGL(tb_bwd3_kernel<512>, dim3((unsigned)gx, 3), dim3(512), bwd_lds(8), st, p);
    return hipGetLastError();
}

hipError_t t_reduce_parts3(const float* const (&part)[3], const int (&nparts)[3], const PartPlan (&plan)[3], hipStream_t st) {
    ReduceParts3 r;
    int items = 0, most = 0;
    for (int b = 0; b < 3; ++b) {
        r.part[b] = part[b];
        r.nparts[b] = nparts[b];
        r.ut[b] = plan[b].units;
        r.vt[b] = plan[b].vecs;
        int n = plan[b].units.n * 16;
        for (int i = 0; i < plan[b].vecs.n; ++i) n += (plan[b].vecs.len[i] + 255) / 256;
        items = n > items ? n : items;
        most = nparts[b] > most ? nparts[b] : most;
    }
    if (items == 0 || most == 0) return hipSuccess;
    int groups = (most + 15) / 16;
    groups = groups > 8 ? 8 : groups;
    hipLaunchKernelGGL(reduce_parts3_kernel, dim3((unsigned)items, (unsigned)groups, 3), dim3(256), 0, st, r);
    return hipGetLastError();
}

}  // namespace dptm


PH_DEFINE_READER(dptm_debug_phase_cycles_train)
