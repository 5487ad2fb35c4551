// Per-site dense layers of the training step for SMALL batches (at most kSiteMaxRows sites; the reference trains with
// 64, pred/train_PTM/config.py:56): structure branch, ESM branch, trunk (pred/train_PTM/model.py:187-205).
//
// With so few rows a layer is a handful of 32 x 32 MFMA tiles, and the layer-wise kernels (product, column sums,
// BatchNorm, dropout, ... one launch each) spend their time in launch latency.  Here a workgroup owns 32 output
// columns of a layer for ALL rows of the batch, so everything that is local to a column happens in one launch:
//
//   site_fwd_kernel   a = x W + b, y = swish(a), batch statistics of y, z = BatchNorm(y), dropout -> out
//   site_bwd_kernel   upstream gradient of the 32 columns (read, or formed as d(a_next) W_next^T for these columns),
//                     dropout masks, BatchNorm and swish backward -> d(a); then the 32 columns of dW = x^T d(a),
//                     db, dgamma, dbeta.  A layer's input gradient is formed by the kernel of the layer below, which
//                     reads d(a) in full: no atomics, no partial buffers.
//
// Operands come straight from the raw row-major tensors (Keras kernels are [in][out]); the eight wavefronts of a
// workgroup split the contraction and meet in an LDS tile through LDS atomics.  Dropout masks use the same
// (seed, stream, element index) hash as every other path, so results do not depend on which path ran.
#include <hip/hip_runtime.h>

#include "dptm_mfma.h"
#include "dptm_phase.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"

namespace dptm {

namespace {

constexpr int kTilePad = 33;     // floats per row of the [rows][32] LDS tile

__device__ __forceinline__ float drop(float v, const SiteDrop& d, unsigned seed, int64_t r, int c) {
    if (d.thr == 0u) return v;
    return hash_u32(seed, d.stream, (unsigned)(r * d.cols + d.col0 + c)) >= d.thr ? v * d.scale : 0.f;
}

// acc tiles (rows 32rt.., my 32 columns) += partial products of this wavefront's share of the contraction:
//   D[m = column][n = row] += sum_k A[m][k] B[k][n]
// colmat(col, k) / rowmat(row, k) return the four operand values at k .. k+3.  The contraction index is walked in groups
// of 8: the lower half-wave takes k = 8q .. 8q+3, the upper one 8q+4 .. 8q+7 (any pairing of the two halves is a valid
// MFMA step), so that operands that are contiguous along k arrive as 16-byte loads; groups_in_flight() groups are
// requested before the first product of a batch.  The wavefronts' partial tiles are then summed into the LDS tile
// [row][column] (every thread of the workgroup must call this: barriers inside; `part` holds 4096 floats).
template <int RT> constexpr int groups_in_flight() { return RT <= 2 ? 12 : RT <= 4 ? 4 : 2; }

template <int RT, typename FA, typename FB, typename FM>
__device__ __forceinline__ void split_k_product(int K, FA colmat, FB rowmat, FM rowmask, float* tile, float* part, int lane, int wave, int nwaves PH_PARAMS) {
    const int hh = lane >> 5, i = lane & 31;
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rt][rg] = 0.f;
    constexpr int kGroupsInFlight = groups_in_flight<RT>();
    const int groups = K >> 3;
    const int per = (groups + nwaves - 1) / nwaves;
    const int g0 = wave * per, g1 = min(groups, g0 + per);
    for (int gb = g0; gb < g1; gb += kGroupsInFlight) {
        f32x4 a[kGroupsInFlight], b[kGroupsInFlight][RT];
        // every load of the batch is requested before anything waits for one: masks (which need the values) come after
#pragma unroll
        for (int v = 0; v < kGroupsInFlight; ++v) {
            const int k = 8 * min(gb + v, g1 - 1) + 4 * hh;
            a[v] = colmat(i, k);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) b[v][rt] = rowmat(32 * rt + i, k);
        }
        PH_WAIT_VM
        PH_MARK(1)   // operand loads
#pragma unroll
        for (int v = 0; v < kGroupsInFlight; ++v) {
            const int k = 8 * (gb + v) + 4 * hh;
            if (gb + v >= g1) a[v] = f32x4{0.f, 0.f, 0.f, 0.f};      // a group past the end was read again: drop it
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) rowmask(b[v][rt], 32 * rt + i, k);
        }
#pragma unroll
        for (int v = 0; v < kGroupsInFlight; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) mfma(acc[rt], a[v][e], b[v][rt][e]);
        PH_MARK(4)   // masks and products
    }
    // The eight partial tiles meet in LDS, one 32-row block at a time: the upper four wavefronts hand theirs to the
    // lower four, these add their own and hand the sums back, then all threads add the four sums into the tile.
    // (Plain stores and loads: LDS float atomics from eight wavefronts on the same addresses run many times slower.)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float* mine = part + (wave & 3) * 1024 + lane;
        if (wave >= 4) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) mine[rg * 64] = acc[rt][rg];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) mine[rg * 64] += acc[rt][rg];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 1024; o += 512) {
            const int rg = o >> 6, ln = o & 63;
            const int m = (rg & 3) + 8 * (rg >> 2) + 4 * (ln >> 5);
            tile[(32 * rt + (ln & 31)) * kTilePad + m] = (part[o] + part[1024 + o]) + (part[2048 + o] + part[3072 + o]);
        }
        __syncthreads();
    }
    PH_MARK(5)   // LDS reduction
}

// dropout mask of four consecutive elements k .. k+3 of row r
__device__ __forceinline__ void mask4(f32x4& v, int k, const SiteDrop& d, unsigned seed, int64_t r) {
    if (d.thr != 0u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop(v[e], d, seed, r, k + e);
    }
}

// sums over the batch rows of f(r, c) for the 32 columns of the tile, by all 512 threads: thread (c, g) walks rows
// g, g+16, ...; the 16 partial sums of a column meet in LDS.  Results in out1 / out2 [32]; barriers inside.
template <typename F>
__device__ __forceinline__ void column_sums(int B, F f, float* red, float* out1, float* out2) {
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    float s1 = 0.f, s2 = 0.f;
    for (int r = g; r < B; r += 16) {
        float v1, v2;
        f(r, c, v1, v2);
        s1 += v1;
        s2 += v2;
    }
    red[g * 32 + c] = s1;
    red[512 + g * 32 + c] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float* src = red + (threadIdx.x >> 5) * 512 + c;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += src[q * 32];
        (threadIdx.x < 32 ? out1 : out2)[c] = s;
    }
    __syncthreads();
}

}  // namespace

// grid: N / 32 workgroups, block 512
template <int RT>
__global__ __launch_bounds__(512) void site_fwd_kernel(SiteFwdArgs p) {
    __shared__ float tile[RT * 32 * kTilePad];
    __shared__ float part[4096];                   // partial tiles of the product; later the column-sum scratch
    __shared__ float cs1[32], cs2[32], csc[32], csh[32];
    float* red = part;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 32;
    PH_DECL
    const unsigned seed = p.step[kStepSeed];
    // per-column parameters of this thread's column (c = tid & 31 in every loop below): requested before the product
    const int cc = tid & 31;
    const float bias_c = p.bias[n0 + cc];
    const float gamma_c = p.bn ? p.gamma[n0 + cc] : 1.f, beta_c = p.bn ? p.beta[n0 + cc] : 0.f;
    PH_MARK(0)   // start-up
    split_k_product<RT>(
        p.K,
        [&](int c, int k) {
            const float* w = p.W + (int64_t)k * p.N + n0 + c;
            return f32x4{w[0], w[p.N], w[2 * (int64_t)p.N], w[3 * (int64_t)p.N]};
        },
        [&](int r, int k) {      // rows past the batch read the last row again and are zeroed by the mask step
            return *reinterpret_cast<const f32x4*>(p.x + (int64_t)min(r, p.B - 1) * p.ldx + k);
        },
        [&](f32x4& v, int r, int k) {
            if (r < p.B)
                mask4(v, k, p.in_drop, seed, r);
            else
                v = f32x4{0.f, 0.f, 0.f, 0.f};
        },
        tile, part, lane, wave, 8 PH_ARGS);
    __syncthreads();
    PH_MARK(2)   // wait for the other wavefronts
    for (int o = tid; o < p.B * 32; o += 512) {
        const int r = o >> 5, c = o & 31;
        const float v = tile[r * kTilePad + c] + bias_c;
        p.a[(int64_t)r * p.N + n0 + c] = v;
        tile[r * kTilePad + c] = swishf(v);
    }
    __syncthreads();
    PH_MARK(3)   // bias, pre-activation out, swish
    if (p.bn) {
        column_sums(p.B, [&](int r, int c, float& v1, float& v2) {
            const float y = tile[r * kTilePad + c];
            v1 = y;
            v2 = y * y;
        }, red, cs1, cs2);
        if (tid < 32) {
            const float mu = cs1[tid] / (float)p.B;
            float var = cs2[tid] / (float)p.B - mu * mu;
            var = var < 0.f ? 0.f : var;
            const float inv = rsqrtf(var + 1e-3f);
            p.mi[n0 + tid] = mu;
            p.mi[p.N + n0 + tid] = inv;
            p.batch_mean[n0 + tid] = mu;
            p.batch_var[n0 + tid] = var;
            const float sc = inv * gamma_c;
            csc[tid] = sc;
            csh[tid] = beta_c - mu * sc;
        }
        __syncthreads();
    }
    for (int o = tid; o < p.B * 32; o += 512) {
        const int r = o >> 5, c = o & 31;
        float z = tile[r * kTilePad + c];
        if (p.bn) z = fmaf(z, csc[c], csh[c]);
        p.out[(int64_t)r * p.ldo + p.col_off + n0 + c] = drop(z, p.out_drop, seed, r, n0 + c);
    }
    PH_MARK(3)   // batch statistics, normalise, dropout, store
    PH_FLUSH
}

// grid: N / 32 workgroups (N = columns of the layer, or of the plain gradient block when there is no layer), block 512
template <int RT>
__global__ __launch_bounds__(512) void site_bwd_kernel(SiteBwdArgs p) {
    __shared__ float tile[RT * 32 * kTilePad];     // upstream gradient, then d(a)
    __shared__ float ta[RT * 32 * kTilePad];       // pre-activation a of my columns
    __shared__ float part[4096];                   // partial tiles of the product; later the column-sum scratch
    float* red = part;
    __shared__ float cs1[32], cs2[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, i = lane & 31;
    const int n0 = blockIdx.x * 32;
    PH_DECL
    const unsigned seed = p.step[kStepSeed];
    // per-column parameters of this thread's column (c = tid & 31 in every loop below): requested before the product
    const int cc = tid & 31;
    const float mu_c = p.bn ? p.mi[n0 + cc] : 0.f, inv_c = p.bn ? p.mi[p.N + n0 + cc] : 1.f, gamma_c = p.bn ? p.gamma[n0 + cc] : 1.f;
    if (p.part != 2) {
    if (p.a) {
        for (int o = tid; o < p.B * 32; o += 512) {
            const int r = o >> 5, c = o & 31;
            ta[r * kTilePad + c] = p.a[(int64_t)r * p.N + n0 + c];
        }
    }
    // ---- upstream gradient of my columns, behind the next layer's input dropout
    if (p.dzd) {
        for (int o = tid; o < RT * 32 * 32; o += 512) {
            const int r = o >> 5, c = o & 31;
            tile[r * kTilePad + c] = r < p.B ? p.dzd[(int64_t)r * p.ld_dzd + n0 + c] : 0.f;
        }
    } else {
        const int64_t wrow = p.next_row_off + n0;
        split_k_product<RT>(
            p.Nn, [&](int c, int k) { return *reinterpret_cast<const f32x4*>(p.W_next + (wrow + c) * p.Nn + k); },
            [&](int r, int k) { return *reinterpret_cast<const f32x4*>(p.da_next + (int64_t)min(r, p.B - 1) * p.Nn + k); },
            [&](f32x4& v, int r, int k) {
                if (r >= p.B) v = f32x4{0.f, 0.f, 0.f, 0.f};
            },
            tile, part, lane, wave, 8 PH_ARGS);
    }
    PH_MARK(6)   // upstream gradient read
    __syncthreads();
    PH_MARK(2)
    for (int o = tid; o < p.B * 32; o += 512) {
        const int r = o >> 5, c = o & 31;
        float g = drop(tile[r * kTilePad + c], p.mid_drop, seed, r, n0 + c);
        if (p.gout) p.gout[(int64_t)r * p.ld_gout + n0 + c] = g;
        g = drop(g, p.out_drop, seed, r, n0 + c);
        tile[r * kTilePad + c] = g;
    }
    if (!p.a) {            // a plain gradient block: nothing below it in this launch
        PH_MARK(7)
        PH_FLUSH
        return;
    }
    __syncthreads();
    PH_MARK(7)   // dropout masks, gradient out
    // ---- BatchNorm backward (column sums over the batch are local to the workgroup), swish backward
    if (p.bn) {
        column_sums(p.B, [&](int r, int c, float& v1, float& v2) {
            const float dz = tile[r * kTilePad + c];
            const float xh = (swishf(ta[r * kTilePad + c]) - mu_c) * inv_c;
            v1 = dz;
            v2 = dz * xh;
        }, red, cs1, cs2);
        if (tid < 32) {
            p.dgamma[n0 + tid] += cs2[tid];
            p.dbeta[n0 + tid] += cs1[tid];
        }
    }
    PH_MARK(8)   // BatchNorm backward sums
    const float inv_b = 1.0f / (float)p.B;
    for (int o = tid; o < p.B * 32; o += 512) {
        const int r = o >> 5, c = o & 31;
        const float v = ta[r * kTilePad + c];
        const float sg = fast_rcp(1.0f + __expf(-v));
        float dy = tile[r * kTilePad + c];
        if (p.bn) {
            const float xh = (v * sg - mu_c) * inv_c;
            dy = gamma_c * inv_c * (dy - cs1[c] * inv_b - xh * cs2[c] * inv_b);
        }
        const float da = dy * (sg * (1.0f + v * (1.0f - sg)));
        p.da[(int64_t)r * p.N + n0 + c] = da;
        tile[r * kTilePad + c] = da;
    }
    __syncthreads();
    PH_MARK(9)   // d(a)
    {
        float* db1 = cs1;
        column_sums(p.B, [&](int r, int c, float& v1, float& v2) {
            v1 = tile[r * kTilePad + c];
            v2 = 0.f;
        }, red, db1, cs2);
        if (tid < 32) p.db[n0 + tid] += db1[tid];
    }
    if (p.part == 1) {     // the weight gradient follows in a launch of its own
        PH_MARK(10)
        PH_FLUSH
        return;
    }
    } else {
        // d(a) of my columns, as the first launch left it; rows behind the batch hold zeros
        for (int o = tid; o < RT * 32 * 32; o += 512) {
            const int r = o >> 5, c = o & 31;
            tile[r * kTilePad + c] = r < p.B ? p.da[(int64_t)r * p.N + n0 + c] : 0.f;
        }
        __syncthreads();
    }
    // ---- my 32 columns of dW = x^T d(a): 32-row blocks of dW over the wavefronts, contraction over the batch rows
    // (rows 8q + 4hh + e meet in step 4q + e; all loads of a block are requested before its first product)
    PH_MARK(10)  // bias gradient
    // Several blocks per pass, so that their loads are all in flight; this kernel is the only writer of its dW columns.
    // The batch rows are walked in slices of at most 64 (QS groups of 8), so that a pass holds at most 96 operands.
    constexpr int U = RT <= 1 ? 4 : RT == 2 ? 3 : 2;
    constexpr int QS = RT * 4 < 8 ? RT * 4 : 8;
    const int mt = (p.K + 31) / 32;
    for (int t = wave; t < mt; t += 8 * U) {
        f32x16 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[u][rg] = 0.f;
#pragma unroll 1
        for (int q0 = 0; q0 < RT * 4; q0 += QS) {
            f32x4 a[U][QS], b[QS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kc = min(32 * (t + 8 * u) + i, p.K - 1);
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * (q0 + q) + 4 * hh + e;
                        a[u][q][e] = p.x[(int64_t)min(r, p.B - 1) * p.ldx + kc];      // all loads first, masks afterwards
                    }
            }
#pragma unroll
            for (int q = 0; q < QS; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[q][e] = tile[(8 * (q0 + q) + 4 * hh + e) * kTilePad + i];   // rows >= B hold zeros
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t + 8 * u >= mt) break;
                const int k = 32 * (t + 8 * u) + i;
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * (q0 + q) + 4 * hh + e;
                        a[u][q][e] = (r < p.B && k < p.K) ? drop(a[u][q][e], p.in_drop, seed, r, k) : 0.f;
                    }
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mfma(acc[u], a[u][q][e], b[q][e]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t + 8 * u >= mt) break;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int m = 32 * (t + 8 * u) + (rg & 3) + 8 * (rg >> 2) + 4 * hh;
                if (m < p.K) {
                    const int64_t at = (int64_t)m * p.N + n0 + i;
                    p.dW[at] = p.l2c != 0.f ? fmaf(p.l2c, p.W[at], acc[u][rg]) : acc[u][rg];
                }
            }
        }
    }
    PH_MARK(11)  // weight gradient
    PH_FLUSH
}

hipError_t t_site_fwd(const SiteFwdArgs& a, hipStream_t st) {
    if (a.B < 1 || a.B > kSiteMaxRows || a.N % 32 != 0 || a.K % 8 != 0 || a.ldx % 4 != 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.N / 32)), block(512);
    const int rt = (a.B + 31) / 32;
    if (rt == 1)
        hipLaunchKernelGGL(site_fwd_kernel<1>, grid, block, 0, st, a);
    else if (rt == 2)
        hipLaunchKernelGGL(site_fwd_kernel<2>, grid, block, 0, st, a);
    else if (rt <= 4)
        hipLaunchKernelGGL(site_fwd_kernel<4>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(site_fwd_kernel<8>, grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t t_site_bwd(const SiteBwdArgs& a, hipStream_t st) {
    if (a.B < 1 || a.B > kSiteMaxRows || a.N % 32 != 0 || (a.part != 2 && !a.dzd && a.Nn % 8 != 0)) return hipErrorInvalidValue;
    if (a.part < 0 || a.part > 2 || (a.part != 0 && (!a.a || !a.da || !a.dW))) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.N / 32)), block(512);
    const int rt = (a.B + 31) / 32;
    if (rt == 1)
        hipLaunchKernelGGL(site_bwd_kernel<1>, grid, block, 0, st, a);
    else if (rt == 2)
        hipLaunchKernelGGL(site_bwd_kernel<2>, grid, block, 0, st, a);
    else if (rt <= 4)
        hipLaunchKernelGGL(site_bwd_kernel<4>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(site_bwd_kernel<8>, grid, block, 0, st, a);
    return hipGetLastError();
}

}  // namespace dptm

PH_DEFINE_READER(dptm_debug_phase_cycles_site)
