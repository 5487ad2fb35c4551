// Half-precision-MFMA variant of the fused sequence branch (gfx950).
//
// Same graph and the same fp32 "chain layout" of activations as branch_kernel in
// dptm_kernels.hip, but every contraction -- dense layers, conv taps AND the attention
// products -- runs on v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation:
//   MODE_BF16   operands rounded to bfloat16, one MFMA per product
//   MODE_F16    operands rounded to float16, one MFMA per product
//   MODE_F16X3  each operand split x = hi + lo (two float16 values, |lo| <= ulp(hi)/2),
//               products hi*hi + hi*lo + lo*hi: 3 MFMAs, ~2^-22 relative error, i.e.
//               fp32-class results at 3/16 of the fp32 MFMA cost.  gfx950 keeps fp16
//               subnormals in MFMA operands, so small lo parts are not flushed.
// Attention on MFMA: S^T = K Q^T per (aligned 32-key tile, head) is ONE k=16 MFMA; keys
// outside the query's own site are masked before the softmax; O^T = V^T P^T takes the
// softmax tile straight from the accumulator registers as the next B operand.
#include <hip/hip_runtime.h>

#include "dptm_kernels.h"
#include "dptm_layout.h"
#include "dptm_phase.h"
#include "dptm_stage.h"

namespace dptm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int MODE> struct HP;
template <> struct HP<kModeBF16> { typedef b8 frag; typedef __bf16 elem; static constexpr bool kSplit = false; };
template <> struct HP<kModeF16> { typedef h8 frag; typedef _Float16 elem; static constexpr bool kSplit = false; };
template <> struct HP<kModeF16X3> { typedef h8 frag; typedef _Float16 elem; static constexpr bool kSplit = true; };

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(b8 a, b8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int MODE>
struct Frag {                    // one MFMA operand: 8 k-elements per lane (hi, and lo in the split mode)
    typename HP<MODE>::frag hi, lo;
};

// x - (float)half, the half taken from the low / high 16 bits of a packed pair: one mixed-precision FMA
// (v_fma_mix_f32 reads the f16 operand directly) instead of a conversion and a subtraction
template <int HIGH>
__device__ __forceinline__ float sub_half(float x, unsigned pair) {
    float d;
    if (HIGH)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(x));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(x));
    return d;
}

// 8 fp32 values (two chain-layout groups) -> operand fragment
template <int MODE>
__device__ __forceinline__ Frag<MODE> make_frag(const f32x4& x0, const f32x4& x1) {
    typedef typename HP<MODE>::elem E;
    Frag<MODE> f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.hi[e] = (E)x0[e];
        f.hi[4 + e] = (E)x1[e];
    }
    if (HP<MODE>::kSplit) {
        const uint4 hp = __builtin_bit_cast(uint4, f.hi);
        const unsigned pr[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f.lo[2 * q] = (E)sub_half<0>(x0[2 * q], pr[q]);
            f.lo[2 * q + 1] = (E)sub_half<1>(x0[2 * q + 1], pr[q]);
            f.lo[4 + 2 * q] = (E)sub_half<0>(x1[2 * q], pr[2 + q]);
            f.lo[5 + 2 * q] = (E)sub_half<1>(x1[2 * q + 1], pr[2 + q]);
        }
    }
    return f;
}

template <int MODE>
__device__ __forceinline__ void mac(f32x16& acc, const Frag<MODE>& a, const Frag<MODE>& b) {
    acc = mfma16(a.hi, b.hi, acc);
    if (HP<MODE>::kSplit) {
        acc = mfma16(a.hi, b.lo, acc);
        acc = mfma16(a.lo, b.hi, acc);
    }
}

template <int MODE>
__device__ __forceinline__ typename HP<MODE>::frag as_frag(const uint4& v) {
    return __builtin_bit_cast(typename HP<MODE>::frag, v);
}
template <int MODE>
__device__ __forceinline__ uint4 as_u4(const typename HP<MODE>::frag& f) { return __builtin_bit_cast(uint4, f); }

// weight fragment (t, ks) of a packed half matrix with KS k-steps per out tile
template <int MODE>
__device__ __forceinline__ Frag<MODE> load_w(const uint4* __restrict__ hi, const uint4* __restrict__ lo, int idx) {
    Frag<MODE> f;
    f.hi = as_frag<MODE>(hi[idx]);
    if (HP<MODE>::kSplit) f.lo = as_frag<MODE>(lo[idx]);
    return f;
}

__device__ __forceinline__ float exp2_h(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp_h(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float swish_h(float x) { return x * fast_rcp_h(1.0f + __expf(-x)); }
// sum / max over the two half-wave lanes with the same j: v_permlane32_swap on two copies leaves (own, other)
__device__ __forceinline__ float xhalf_sum_h(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhalf_max_h(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int NT>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[NT], const f32x4* __restrict__ vec, int h, int groups_per_half) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = vec[h * groups_per_half + 4 * t + g];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
        }
}

template <int NT>
__device__ __forceinline__ void acc_x(const f32x16 (&acc)[NT], f32x4 (&x)[4 * NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * t + g][e] = acc[t][4 * g + e];
}

// acc[t] += W(t)^T X^T, X given as KS operand fragments; all weight fragments of the layer are
// fetched up front (a 64-wide layer is only KS = 4 steps deep, shorter than one L2 round trip)
template <int MODE, int KS, int NT>
__device__ __forceinline__ void gemm_h(f32x16 (&acc)[NT], const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                       const Frag<MODE> (&x)[KS], int lane) {
    Frag<MODE> a[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) a[ks][t] = load_w<MODE>(whi, wlo, (t * KS + ks) * 64 + lane);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) mac<MODE>(acc[t], a[ks][t], x[ks]);
}

__device__ __forceinline__ void layernorm64_h(f32x4 (&x)[8], const f32x4* __restrict__ g, const f32x4* __restrict__ b, int h) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (x[u][0] + x[u][1]) + (x[u][2] + x[u][3]);
    const float mean = xhalf_sum_h(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = x[u][e] - mean;
            x[u][e] = d;
            q = fmaf(d, d, q);
        }
    const float inv = rsqrtf(xhalf_sum_h(q) * (1.0f / 64.0f) + 1e-6f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f32x4 gg = g[h * 8 + u], bb = b[h * 8 + u];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[u][e] = fmaf(x[u][e] * inv, gg[e], bb[e]);
    }
}

// LDS images (byte offsets inside one 73,728-byte region; hi at 0, lo at kLoOff)
constexpr int kRowB = 144;                    // fragment image: [row][4 steps][2 halves] x 16 B + 16 B pad
constexpr int kLoOff = 256 * kRowB;           // 36,864
constexpr int kVtRowB = 528;                  // V^T image: [feature][256 keys] x 2 B + 16 B pad
constexpr int kVtLoOff = 64 * kVtRowB;        // 33,792
constexpr int kRegionB = 2 * 256 * kRowB;     // 73,728 >= 2*64*528 and >= 256*68*4

template <int MODE>
__device__ __forceinline__ void store_frag(char* region, int row, int slot, int h, const Frag<MODE>& f) {
    char* p = region + row * kRowB + (slot * 2 + h) * 16;
    *reinterpret_cast<uint4*>(p) = as_u4<MODE>(f.hi);
    if (HP<MODE>::kSplit) *reinterpret_cast<uint4*>(p + kLoOff) = as_u4<MODE>(f.lo);
}
template <int MODE>
__device__ __forceinline__ Frag<MODE> load_frag(const char* region, int row, int slot, int h) {
    const char* p = region + row * kRowB + (slot * 2 + h) * 16;
    Frag<MODE> f;
    f.hi = as_frag<MODE>(*reinterpret_cast<const uint4*>(p));
    if (HP<MODE>::kSplit) f.lo = as_frag<MODE>(*reinterpret_cast<const uint4*>(p + kLoOff));
    return f;
}
template <int MODE>
__device__ __forceinline__ Frag<MODE> zero_frag() {
    Frag<MODE> f;
    const uint4 z = {0u, 0u, 0u, 0u};
    f.hi = as_frag<MODE>(z);
    f.lo = as_frag<MODE>(z);
    return f;
}

// Register buffer for the weight fragments of one product: KS k-steps (16 inputs each) x NT out tiles.
// load(): fragment (t, ks) = index ((t0 + t)*stride + ks0 + ks)*64 + lane of a packed half image whose
// out tiles are `stride` k-steps apart.
template <int MODE, int KS, int NT>
struct WBuf {
    Frag<MODE> a[KS][NT];
    __device__ __forceinline__ void load(const uint4* __restrict__ hi, const uint4* __restrict__ lo, int stride, int t0, int ks0, int lane) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) a[ks][t] = load_w<MODE>(hi, lo, ((t0 + t) * stride + ks0 + ks) * 64 + lane);
    }
    __device__ __forceinline__ void mm(f32x16 (&acc)[NT], const Frag<MODE> (&x)[KS]) const {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) mac<MODE>(acc[t], a[ks][t], x[ks]);
    }
};

template <int W, int MODE>
__device__ __forceinline__ void branch_h_body(const f32x4* __restrict__ P, const uint4* __restrict__ Whi,
                                              const uint4* __restrict__ Wlo, const uint8_t* __restrict__ tok,
                                              int64_t n_sites, float* __restrict__ out, int ldo, int col_off) {
    typedef typename HP<MODE>::elem E;
    constexpr int S = kTileRows / W;
    constexpr int MAXNB = (W == 51) ? 5 : 4;
    extern __shared__ __attribute__((aligned(16))) char lds_h[];
    char* regA = lds_h;                 // X0 fragments | K fragments | final activations (fp32 [256][68])
    char* regB = lds_h + kRegionB;      // y1 fragments | V^T
    float* score = reinterpret_cast<float*>(lds_h + 2 * kRegionB);
    // bias / scale / LayerNorm vectors of the whole branch, copied once: a vector read costs the texture path
    // as much as a weight fragment, and there are more of them than weight fragments
    f32x4* vecs = reinterpret_cast<f32x4*>(lds_h + 2 * kRegionB + kTileRows * sizeof(float));
    // one-hot "site of this key" operand fragments, [S + 1 sites][2 halves] x 16 B (site S = padding rows)
    uint4* site_frag = reinterpret_cast<uint4*>(lds_h + 2 * kRegionB + kTileRows * sizeof(float) + kVecsF4 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    PH_DECL
    // vectors and both embedding tables -> LDS: the thread's loads (and its token) are requested before the first store
    VecsStage<f32x4, 512> vst;
    CopyStage<f32x4, (kCodes + W) * 16, 512> tst;
    vst.load(P, tid);
    tst.load(P + kOffEmb, tid);                                                      // kOffPos == kOffEmb + 21*16
    if (tid < (S + 1) * 2) {
        typename HP<MODE>::frag f;
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) f[idx] = (8 * (tid & 1) + idx == (tid >> 1)) ? (E)1.0f : (E)0.0f;
        site_frag[tid] = as_u4<MODE>(f);
    }
    const int64_t site0 = (int64_t)blockIdx.x * S;
    const int sites_here = (int)min((int64_t)S, n_sites - site0);
    const int rows_valid = sites_here * W;

    const int row = wave * 32 + j;
    const bool row_ok = row < rows_valid;
    const int site_l = row_ok ? row / W : 0;
    const int pos_l = row_ok ? row - site_l * W : 0;
    const bool tap0_ok = row_ok && pos_l > 0;
    const bool tap2_ok = row_ok && pos_l < W - 1;
    // aligned 32-key tiles this wavefront needs (wave-uniform)
    int b_lo, nb;
    {
        const int r_first = wave * 32;
        if (r_first >= rows_valid) {
            b_lo = wave;
            nb = 1;
        } else {
            const int r_last = min(r_first + 31, rows_valid - 1);
            const int kfirst = (r_first / W) * W;
            const int klast = (r_last / W) * W + W - 1;
            b_lo = kfirst >> 5;
            nb = (min(klast, kTileRows - 1) >> 5) - b_lo + 1;
        }
    }
    // Keys outside the query's own site are pushed out of the softmax by one extra MFMA per score tile:
    // key operand = one-hot of the key's site over 16 "site" dimensions, query operand = 0 on the query's
    // own site and -60000 elsewhere, so their product adds 0 or -60000 to the score (scores are O(10)).
    // Padding rows form a site of their own.  LDS byte offset of the key fragment per tile:
    int site_frag_off[MAXNB];
#pragma unroll
    for (int bi = 0; bi < MAXNB; ++bi) {
        const int krow = 32 * (b_lo + bi) + j;
        const int ks = krow < rows_valid ? krow / W : S;
        site_frag_off[bi] = (ks * 2 + h) * 16;
    }
    typename HP<MODE>::frag qsite;      // query-side operand, fixed for the kernel
    {
        const int qs = row_ok ? site_l : S;
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) qsite[idx] = (8 * h + idx == qs) ? (E)0.0f : (E)(-60000.0f);
    }

    // ---- embedding + position embedding, this lane's row in the chain layout (model.py:157-158)
    // Both tables (21 + W rows of 64 floats) are first copied into the still unused second LDS region: a
    // per-lane gather from memory touches up to 64 cache lines per instruction and dominated the prologue.
    f32x4* tabs = reinterpret_cast<f32x4*>(regB);
    static_assert(kOffPos == kOffEmb + kCodes * 16, "embedding and position tables must be adjacent");
    int tk = tok[site0 * W + (row_ok ? row : 0)];
    vst.store(vecs, tid);
    tst.store(tabs, tid);
    tk = tk > kCodes - 1 ? kCodes - 1 : tk;
    __syncthreads();
    f32x4 x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row_ok) v = tabs[tk * 16 + 2 * u + h] + tabs[(kCodes + pos_l) * 16 + 2 * u + h];
        x1[u] = v;
    }
    Frag<MODE> fx[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fx[ks] = make_frag<MODE>(x1[2 * ks], x1[2 * ks + 1]);
        store_frag<MODE>(regA, row, ks, h, fx[ks]);
    }
    PH_MARK(9)    // set-up, embedding
    __syncthreads();
    PH_MARK(10)   // barrier wait

    // ---- residual conv block (model.py:129-141)
    {
        const f32x4* cv = vecs;
        f32x16 acc_s[2], acc_a[2];
        acc_bias<2>(acc_s, cv + kVScB * kVec64, h, 8);
        acc_bias<2>(acc_a, cv + kVC3aB * kVec64, h, 8);
        const int rl = tap0_ok ? row - 1 : row, rr = tap2_ok ? row + 1 : row;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Frag<MODE> bl = load_frag<MODE>(regA, rl, ks, h), br = load_frag<MODE>(regA, rr, ks, h);
            if (!tap0_ok) bl = zero_frag<MODE>();
            if (!tap2_ok) br = zero_frag<MODE>();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = (t * 4 + ks) * 64 + lane;
                mac<MODE>(acc_s[t], load_w<MODE>(Whi + kOffScW / 2, Wlo + kOffScW / 2, idx), fx[ks]);
                mac<MODE>(acc_a[t], load_w<MODE>(Whi + kOffC3aW / 2, Wlo + kOffC3aW / 2, idx), bl);
                mac<MODE>(acc_a[t], load_w<MODE>(Whi + (kOffC3aW + kMat64) / 2, Wlo + (kOffC3aW + kMat64) / 2, idx), fx[ks]);
                mac<MODE>(acc_a[t], load_w<MODE>(Whi + (kOffC3aW + 2 * kMat64) / 2, Wlo + (kOffC3aW + 2 * kMat64) / 2, idx), br);
            }
        }
        f32x4 y[8];
        acc_x<2>(acc_a, y);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 sc = cv[kVBn1S * kVec64 + h * 8 + u], sh = cv[kVBn1T * kVec64 + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[u][e] = fmaf(swish_h(y[u][e]), sc[e], sh[e]);
        }
        Frag<MODE> fy[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fy[ks] = make_frag<MODE>(y[2 * ks], y[2 * ks + 1]);
            store_frag<MODE>(regB, row, ks, h, fy[ks]);
        }
        PH_MARK(11)   // shortcut + first conv, activation
        __syncthreads();
        f32x16 acc_b[2];
        acc_bias<2>(acc_b, cv + kVC3bB * kVec64, h, 8);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Frag<MODE> bl = load_frag<MODE>(regB, rl, ks, h), br = load_frag<MODE>(regB, rr, ks, h);
            if (!tap0_ok) bl = zero_frag<MODE>();
            if (!tap2_ok) br = zero_frag<MODE>();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = (t * 4 + ks) * 64 + lane;
                mac<MODE>(acc_b[t], load_w<MODE>(Whi + kOffC3bW / 2, Wlo + kOffC3bW / 2, idx), bl);
                mac<MODE>(acc_b[t], load_w<MODE>(Whi + (kOffC3bW + kMat64) / 2, Wlo + (kOffC3bW + kMat64) / 2, idx), fy[ks]);
                mac<MODE>(acc_b[t], load_w<MODE>(Whi + (kOffC3bW + 2 * kMat64) / 2, Wlo + (kOffC3bW + 2 * kMat64) / 2, idx), br);
            }
        }
        f32x4 s[8];
        acc_x<2>(acc_b, x1);
        acc_x<2>(acc_s, s);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 sc = cv[kVBn2S * kVec64 + h * 8 + u], sh = cv[kVBn2T * kVec64 + h * 8 + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) x1[u][e] = fmaf(swish_h(x1[u][e]), sc[e], sh[e]) + s[u][e];
        }
    }

    PH_MARK(0)   // barrier + second conv, activation, residual
    // ---- two transformer blocks (model.py:93-118, 172-173)
    // Weight fragments travel through two register buffers (wa, wb) that are refilled one
    // layer ahead of their use, so the L2 latency of a layer's weights hides behind the
    // previous layer's MFMAs and epilogue; sched_barrier pins that issue order.
#pragma unroll 1
    for (int blk = 0; blk < 2; ++blk) {
        const int To = kOffTb + blk * kTbSize;
        const f32x4* tv = vecs + kVecsConv + blk * kVecsTb;
        const uint4* Thi = Whi + To / 2;
        const uint4* Tlo = Wlo + To / 2;
        // K, V, Q projections in pieces of NTQ out tiles (one tile in the split mode, whose fragments are
        // twice as large); piece u+1 is in flight while piece u multiplies
        constexpr int NTQ = HP<MODE>::kSplit ? 1 : 2;
        constexpr int NP = 2 / NTQ;
        WBuf<MODE, 4, NTQ> wq[2];
        WBuf<MODE, 4, 2> wo;
        wq[0].load(Thi + kTbWk / 2, Tlo + kTbWk / 2, 4, 0, 0, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fx[ks] = make_frag<MODE>(x1[2 * ks], x1[2 * ks + 1]);
        PH_MARK(1)   // fragments of x1, first weight loads
        __syncthreads();   // every wave is done with regA/regB (conv taps, or the previous block's K / V^T)
        PH_MARK(2)   // barrier wait
        Frag<MODE> fq[4];   // Q operand per head, scaled by log2(e)/sqrt(16): the softmax below works in base 2
#pragma unroll
        for (int u = 0; u < 3 * NP; ++u) {
            constexpr int kMatOff[3] = {kTbWk, kTbWv, kTbWq};
            constexpr int kVecOff[3] = {kVBk, kVBv, kVBq};
            const int pr = u / NP, t0 = (u % NP) * NTQ;
            if (u + 1 < 3 * NP) {
                const int pn = (u + 1) / NP, tn = ((u + 1) % NP) * NTQ;
                wq[(u + 1) & 1].load(Thi + kMatOff[pn] / 2, Tlo + kMatOff[pn] / 2, 4, tn, 0, lane);
            } else if (!HP<MODE>::kSplit) {
                wo.load(Thi + kTbWo / 2, Tlo + kTbWo / 2, 4, 0, 0, lane);   // stays in flight across the attention
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[NTQ];
            acc_bias<NTQ>(acc, tv + kVecOff[pr] * kVec64 + 4 * t0, h, 8);
            wq[u & 1].mm(acc, fx);
            __builtin_amdgcn_sched_barrier(0);
            if (pr == 0) {          // K -> fragment image [row][head][half]
                f32x4 kx[4 * NTQ];
                acc_x<NTQ>(acc, kx);
#pragma unroll
                for (int hd = 0; hd < 2 * NTQ; ++hd)
                    store_frag<MODE>(regA, row, 2 * t0 + hd, h, make_frag<MODE>(kx[2 * hd], kx[2 * hd + 1]));
            } else if (pr == 1) {   // V -> transposed image [feature][key], keys permuted inside each aligned
                // 32-key tile so that the 8 keys one lane of the A operand needs (16s + 8c + 4h' + e) are
                // contiguous: position (s, h', c, e)
                const int kpos = (wave * 4 + ((j >> 4) & 1) * 2 + ((j >> 2) & 1)) * 8 + ((j >> 3) & 1) * 4 + (j & 3);
                char* vb = regB + kpos * 2;
#pragma unroll
                for (int t = 0; t < NTQ; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int feat = 32 * (t0 + t) + 8 * (r >> 2) + 4 * h + (r & 3);
                        const float v = acc[t][r];
                        const E vh = (E)v;
                        *reinterpret_cast<E*>(vb + feat * kVtRowB) = vh;
                        if (HP<MODE>::kSplit)
                            *reinterpret_cast<E*>(vb + kVtLoOff + feat * kVtRowB) =
                                (E)sub_half<0>(v, (unsigned)__builtin_bit_cast(unsigned short, vh));
                    }
            } else {
                f32x4 qx[4 * NTQ];
                acc_x<NTQ>(acc, qx);
                constexpr float kQScale = 0.25f * 1.4426950408889634f;
#pragma unroll
                for (int hd = 0; hd < 2 * NTQ; ++hd) fq[2 * t0 + hd] = make_frag<MODE>(qx[2 * hd] * kQScale, qx[2 * hd + 1] * kQScale);
            }
        }
        PH_MARK(3)   // K, V, Q projections and their LDS images
        __syncthreads();   // K and V^T of the whole tile are in LDS
        PH_MARK(4)   // barrier wait

        // attention, one head at a time, keys walked in aligned 32-key tiles with a running maximum
        f32x4 o[8];        // attention output of this lane's row, chain layout
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            float m = -3.0e38f, den = 0.f;
            f32x16 ao = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            Frag<MODE> kf = load_frag<MODE>(regA, 32 * b_lo + j, hd, h);
#pragma unroll
            for (int bi = 0; bi < MAXNB; ++bi) {
                if (bi < nb) {
                    // S^T tile: rows = keys of tile b_lo+bi, columns = this wave's 32 queries
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const typename HP<MODE>::frag ksite = as_frag<MODE>(
                        *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(site_frag) + site_frag_off[bi]));
                    f32x16 st = mfma16(ksite, qsite, zero);
                    mac<MODE>(st, kf, fq[hd]);
                    // operands of the next steps are fetched while the MFMA runs
                    const char* vp = regB + (32 * (hd >> 1) + j) * kVtRowB + ((b_lo + bi) * 4 + h) * 16;
                    Frag<MODE> vf0, vf1;
                    vf0.hi = as_frag<MODE>(*reinterpret_cast<const uint4*>(vp));
                    vf1.hi = as_frag<MODE>(*reinterpret_cast<const uint4*>(vp + 32));
                    if (HP<MODE>::kSplit) {
                        vf0.lo = as_frag<MODE>(*reinterpret_cast<const uint4*>(vp + kVtLoOff));
                        vf1.lo = as_frag<MODE>(*reinterpret_cast<const uint4*>(vp + kVtLoOff + 32));
                    }
                    if (bi + 1 < MAXNB && bi + 1 < nb) kf = load_frag<MODE>(regA, 32 * (b_lo + bi + 1) + j, hd, h);
                    float tm = st[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, st[r]);
                    // the first tile starts the running sums: nothing to rescale (and the first O^T product then takes a
                    // literal zero as its accumulator instead of sixteen cleared and rescaled registers)
                    const float mn = bi == 0 ? xhalf_max_h(tm) : fmaxf(m, xhalf_max_h(tm));
                    if (bi != 0) {
                        const float alpha = exp2_h(m - mn);
                        den *= alpha;
                        ao *= alpha;
                    }
                    m = mn;
                    f32x4 p[4];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = exp2_h(st[r] - mn);
                        p[r >> 2][r & 3] = e;
                        den += e;
                    }
                    // O^T (features of head pair hd>>1 x queries) += V^T(tile, step) * P^T(step)
                    mac<MODE>(ao, vf0, make_frag<MODE>(p[0], p[1]));
                    mac<MODE>(ao, vf1, make_frag<MODE>(p[2], p[3]));
                }
            }
            const float rden = 1.0f / xhalf_sum_h(den);
            // rows of ao = features 32*(hd>>1) + 8g + 4h + e; this head owns g = 2*(hd&1), 2*(hd&1)+1
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * (hd & 1) + gg;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * (hd >> 1) + g][e] = ao[4 * g + e] * rden;
            }
        }

        PH_MARK(5)   // attention
        // output projection, residual, LayerNorm 1
        // FFN 64 -> 256 (swish) -> 64 walks the hidden units in chunks of HT 32-wide tiles (one tile in the
        // split mode, whose fragments are twice as large), residual, LayerNorm 2
        constexpr int HT = HP<MODE>::kSplit ? 1 : 2;
        WBuf<MODE, 4, HT> w1;          // ffn1 packed [8 out tiles][4 k-steps]: chunk c = out tiles HT*c ..
        WBuf<MODE, 2 * HT, 2> w2;      // ffn2 packed [2 out tiles][16 k-steps]: chunk c = k-steps 2*HT*c ..
        f32x4 hres[8];
        {
            Frag<MODE> fo[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fo[ks] = make_frag<MODE>(o[2 * ks], o[2 * ks + 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (HP<MODE>::kSplit) wo.load(Thi + kTbWo / 2, Tlo + kTbWo / 2, 4, 0, 0, lane);    // no registers to spare earlier
            w1.load(Thi + kTbF1 / 2, Tlo + kTbF1 / 2, 4, 0, 0, lane);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[2];
            acc_bias<2>(acc, tv + kVBo * kVec64, h, 8);
            wo.mm(acc, fo);
            __builtin_amdgcn_sched_barrier(0);
            w2.load(Thi + kTbF2 / 2, Tlo + kTbF2 / 2, 16, 0, 0, lane);
            __builtin_amdgcn_sched_barrier(0);
            acc_x<2>(acc, hres);
#pragma unroll
            for (int u = 0; u < 8; ++u) hres[u] += x1[u];
            layernorm64_h(hres, tv + kVLn1G * kVec64, tv + kVLn1B * kVec64, h);
        }
        PH_MARK(6)   // output projection, LayerNorm 1
        {
            Frag<MODE> fh[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fh[ks] = make_frag<MODE>(hres[2 * ks], hres[2 * ks + 1]);
            f32x16 acc_o[2];
            acc_bias<2>(acc_o, tv + kVB2 * kVec64, h, 8);
            constexpr int NCH = 8 / HT;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                f32x16 acc_h[HT];
                acc_bias<HT>(acc_h, tv + 9 * kVec64 + 4 * HT * c, h, 32);
                w1.mm(acc_h, fh);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NCH) w1.load(Thi + kTbF1 / 2, Tlo + kTbF1 / 2, 4, HT * (c + 1), 0, lane);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 hid[4 * HT];
                acc_x<HT>(acc_h, hid);
#pragma unroll
                for (int u = 0; u < 4 * HT; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hid[u][e] = swish_h(hid[u][e]);
                Frag<MODE> fhid[2 * HT];
#pragma unroll
                for (int ks = 0; ks < 2 * HT; ++ks) fhid[ks] = make_frag<MODE>(hid[2 * ks], hid[2 * ks + 1]);
                w2.mm(acc_o, fhid);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NCH) w2.load(Thi + kTbF2 / 2, Tlo + kTbF2 / 2, 16, 0, 2 * HT * (c + 1), lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc_x<2>(acc_o, x1);
#pragma unroll
            for (int u = 0; u < 8; ++u) x1[u] += hres[u];
            layernorm64_h(x1, tv + kVLn2G * kVec64, tv + kVLn2B * kVec64, h);
        }
        PH_MARK(7)   // FFN, LayerNorm 2
    }

    // ---- attention pooling (model.py:176-182), fp32 on the vector units
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
        const float ew = __expf(tanhf(xhalf_sum_h(part) + vecs[kVecsPool + kVec64][0]));
        __syncthreads();   // all attention reads of regA are finished
        float* bufX = reinterpret_cast<float*>(regA);
        float* xb = bufX + row * kLdsStride + 4 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(xb + 8 * u) = x1[u];
        if (h == 0) score[row] = ew;
        __syncthreads();
        for (int oi = tid; oi < sites_here * 64; oi += 512) {
            const int s = oi >> 6, f = oi & 63;
            const float* sp = score + s * W;
            const float* xp = bufX + (s * W) * kLdsStride + f;
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
// Per-site dense layer on the half-precision MFMA: out[r][col_off + n] = epi(sum_k X[src(r)][k] W[k][n] + b[n]).
// Same tiling as dense_kernel in dptm_kernels.hip (one wavefront = 64 rows x NT*32 columns, four wavefronts
// of a workgroup share the weight stream of one column slab); X stays fp32 in memory and is rounded (or
// split) to operand fragments on the way in, so the layers around it keep their fp32 buffers.
// ------------------------------------------------------------------------------------
// Variant that fetches the operand order straight from memory (one lane per row).  Kept for the split mode,
// where the staged variant below measures slower (its two-part fragments make the LDS round trip and the
// conversions the bottleneck, not the fetch).
template <int NT, int EPI, int MODE>
__global__ __launch_bounds__(256, 2) void dense_kernel_h_direct(DenseArgs p, const uint4* __restrict__ whi, const uint4* __restrict__ wlo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (r0 >= p.m) return;
    const int slab = blockIdx.y;
    const int KS = p.k >> 4;
    const size_t wbase = (size_t)slab * NT * KS * 64 + lane;

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

    Frag<MODE> a_n[NT];
    f32x4 b_n[2][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) a_n[t] = load_w<MODE>(whi + wbase, wlo + wbase, t * KS * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        b_n[i][0] = *reinterpret_cast<const f32x4*>(xrow[i]);
        b_n[i][1] = *reinterpret_cast<const f32x4*>(xrow[i] + 8);
    }
    for (int ks = 0; ks < KS; ++ks) {
        Frag<MODE> a[NT], b[2];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = a_n[t];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            b[i] = make_frag<MODE>(ok[i] ? b_n[i][0] : z, ok[i] ? b_n[i][1] : z);
        }
        if (ks + 1 < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a_n[t] = load_w<MODE>(whi + wbase, wlo + wbase, (t * KS + ks + 1) * 64);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                b_n[i][0] = *reinterpret_cast<const f32x4*>(xrow[i] + 16 * (ks + 1));
                b_n[i][1] = *reinterpret_cast<const f32x4*>(xrow[i] + 16 * (ks + 1) + 8);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            mac<MODE>(acc[0][t], a[t], b[0]);
            mac<MODE>(acc[1][t], a[t], b[1]);
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + 32 * i + j;
        if (r >= p.m) continue;
        float* orow = p.out + r * p.ldo + p.col_off + slab * NT * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                const int vi = h * (p.n_pad >> 3) + slab * NT * 4 + 4 * t + g;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = swish_h(v[e]);
                if (EPI == kEpiSwishAffine) {
                    const f32x4 sc = p.scale[vi], sh = p.shift[vi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                }
                *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * g) = v;
            }
    }
}

constexpr int kStageStride = 36;   // floats per staged row: 32 + 4 keeps the fragment reads spread over the banks

template <int NT, int EPI, int MODE>
__global__ __launch_bounds__(256, 2) void dense_kernel_h(DenseArgs p, const uint4* __restrict__ whi, const uint4* __restrict__ wlo) {
    // X is staged through LDS in blocks of 32 inputs: a wavefront fetches its 64 rows with every load
    // instruction covering whole 128-byte lines (8 lanes per row), where a lane-per-row fetch of the MFMA
    // operand order would touch 32 lines per instruction and use a quarter of each.  The staging area is
    // private to the wavefront, so no workgroup barrier is needed (LDS operations of a wavefront stay ordered).
    __shared__ __attribute__((aligned(16))) float stage_all[4][64 * kStageStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (r0 >= p.m) return;
    float* stage = stage_all[wave];
    const int slab = blockIdx.y;
    const int KS = p.k >> 4, KB = p.k >> 5;
    const size_t wbase = (size_t)slab * NT * KS * 64 + lane;

    // fetch pattern: load i covers rows 8i .. 8i+7 of the tile, lane l -> row 8i + (l >> 3), floats 4*(l & 7)
    const float* xsrc[8];
    bool ok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t r = r0 + 8 * i + (lane >> 3);
        if (r > p.m - 1) r = p.m - 1;
        int64_t src = r;
        ok[i] = true;
        if (p.row_index) {
            const int32_t s = p.row_index[r];
            ok[i] = s >= 0;
            src = s >= 0 ? s : 0;
        }
        xsrc[i] = p.x + src * p.ldx + 4 * (lane & 7);
    }
    float* st_w = stage + (lane >> 3) * kStageStride + 4 * (lane & 7);
    const float* st_r[2] = {stage + j * kStageStride + 4 * h, stage + (32 + j) * kStageStride + 4 * h};

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

    // weight fragments one step ahead, except in the split mode whose fragments are twice as large
    constexpr bool kAhead = !HP<MODE>::kSplit;
    Frag<MODE> a_n[NT];
    f32x4 xg[8];
    if (kAhead) {
#pragma unroll
        for (int t = 0; t < NT; ++t) a_n[t] = load_w<MODE>(whi + wbase, wlo + wbase, t * KS * 64);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xg[i] = *reinterpret_cast<const f32x4*>(xsrc[i]);
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(st_w + 8 * i * kStageStride) = ok[i] ? xg[i] : z;
        }
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xg[i] = *reinterpret_cast<const f32x4*>(xsrc[i] + 32 * (kb + 1));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ks = 2 * kb + kk;
            Frag<MODE> a[NT], b[2];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = kAhead ? a_n[t] : load_w<MODE>(whi + wbase, wlo + wbase, (t * KS + ks) * 64);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                b[i] = make_frag<MODE>(*reinterpret_cast<const f32x4*>(st_r[i] + 16 * kk),
                                       *reinterpret_cast<const f32x4*>(st_r[i] + 16 * kk + 8));
            if (kAhead && ks + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a_n[t] = load_w<MODE>(whi + wbase, wlo + wbase, (t * KS + ks + 1) * 64);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                mac<MODE>(acc[0][t], a[t], b[0]);
                mac<MODE>(acc[1][t], a[t], b[1]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + 32 * i + j;
        if (r >= p.m) continue;
        float* orow = p.out + r * p.ldo + p.col_off + slab * NT * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                const int vi = h * (p.n_pad >> 3) + slab * NT * 4 + 4 * t + g;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = swish_h(v[e]);
                if (EPI == kEpiSwishAffine) {
                    const f32x4 sc = p.scale[vi], sh = p.shift[vi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                }
                *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * g) = v;
            }
    }
}

// The same layer when X already holds the mode's 16-bit operand values (rows that crossed PCIe as bfloat16 / float16
// instead of fp32: half the bytes, and nothing to convert here).  Staged through LDS like dense_kernel_h, 64 inputs
// per block: a fetch instruction covers whole 128-byte lines (8 lanes per row), the operand fragments are 16-byte
// reads of the staged rows.  One-operand modes only (the split mode needs fp32 inputs); k a multiple of 64.
constexpr int kStageStrideH = 72;   // 16-bit values per staged row: 64 + 8 keeps the fragment reads spread over the banks

template <int NT, int EPI, int MODE>
__global__ __launch_bounds__(256, 2) void dense_kernel_hx(DenseArgs p, const uint4* __restrict__ whi) {
    static_assert(!HP<MODE>::kSplit, "half inputs carry one operand value each");
    __shared__ __attribute__((aligned(16))) unsigned short stage_all[4][64 * kStageStrideH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (r0 >= p.m) return;
    unsigned short* stage = stage_all[wave];
    const int slab = blockIdx.y;
    const int KS = p.k >> 4, KB = p.k >> 6;
    const size_t wbase = (size_t)slab * NT * KS * 64 + lane;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(p.x);

    // fetch pattern: load i covers rows 8i .. 8i+7 of the tile, lane l -> row 8i + (l >> 3), values 8*(l & 7) .. +7
    const unsigned short* xsrc[8];
    bool ok[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t r = r0 + 8 * i + (lane >> 3);
        if (r > p.m - 1) r = p.m - 1;
        int64_t src = r;
        ok[i] = true;
        if (p.row_index) {
            const int32_t s = p.row_index[r];
            ok[i] = s >= 0;
            src = s >= 0 ? s : 0;
        }
        xsrc[i] = xh + src * p.ldx + 8 * (lane & 7);
    }
    unsigned short* st_w = stage + (lane >> 3) * kStageStrideH + 8 * (lane & 7);
    // operand order of a k-step (the chain layout, as make_frag builds it from fp32): the lane of half-wave h takes the
    // inputs 16 ks + 4h .. + 3 and 16 ks + 8 + 4h .. + 3 -- two 8-byte pieces of the staged row
    const unsigned short* st_r[2] = {stage + j * kStageStrideH + 4 * h, stage + (32 + j) * kStageStrideH + 4 * h};

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

    typename HP<MODE>::frag a_n[NT];
    uint4 xg[8];
#pragma unroll
    for (int t = 0; t < NT; ++t) a_n[t] = as_frag<MODE>(whi[wbase + t * KS * 64]);
#pragma unroll
    for (int i = 0; i < 8; ++i) xg[i] = *reinterpret_cast<const uint4*>(xsrc[i]);
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4*>(st_w + 8 * i * kStageStrideH) = ok[i] ? xg[i] : z;
        }
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xg[i] = *reinterpret_cast<const uint4*>(xsrc[i] + 64 * (kb + 1));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ks = 4 * kb + kk;
            typename HP<MODE>::frag a[NT], b[2];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = a_n[t];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint2 lo = *reinterpret_cast<const uint2*>(st_r[i] + 16 * kk), hi = *reinterpret_cast<const uint2*>(st_r[i] + 16 * kk + 8);
                b[i] = as_frag<MODE>(uint4{lo.x, lo.y, hi.x, hi.y});
            }
            if (ks + 1 < KS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a_n[t] = as_frag<MODE>(whi[wbase + (t * KS + ks + 1) * 64]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = mfma16(a[t], b[0], acc[0][t]);
                acc[1][t] = mfma16(a[t], b[1], acc[1][t]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t r = r0 + 32 * i + j;
        if (r >= p.m) continue;
        float* orow = p.out + r * p.ldo + p.col_off + slab * NT * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[i][t][4 * g], acc[i][t][4 * g + 1], acc[i][t][4 * g + 2], acc[i][t][4 * g + 3]};
                const int vi = h * (p.n_pad >> 3) + slab * NT * 4 + 4 * t + g;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = swish_h(v[e]);
                if (EPI == kEpiSwishAffine) {
                    const f32x4 sc = p.scale[vi], sh = p.shift[vi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                }
                *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * g) = v;
            }
    }
}

template <int MODE>
static hipError_t launch_dense_hx_m(const DenseArgs& a, int epi, const void* whi, hipStream_t st) {
    const unsigned gx = (unsigned)((a.m + 255) / 256), gy = (unsigned)(a.n_pad / 128);
    const uint4* hi = reinterpret_cast<const uint4*>(whi);
    if (epi == kEpiSwish)
        hipLaunchKernelGGL((dense_kernel_hx<4, kEpiSwish, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi);
    else if (epi == kEpiSwishAffine)
        hipLaunchKernelGGL((dense_kernel_hx<4, kEpiSwishAffine, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_dense_h_m(const DenseArgs& a, int epi, const void* whi, const void* wlo, hipStream_t st) {
    const unsigned gx = (unsigned)((a.m + 255) / 256), gy = (unsigned)(a.n_pad / 128);
    const uint4* hi = reinterpret_cast<const uint4*>(whi);
    const uint4* lo = reinterpret_cast<const uint4*>(wlo);
    constexpr bool kDirect = HP<MODE>::kSplit;
    if (epi == kEpiSwish) {
        if (kDirect)
            hipLaunchKernelGGL((dense_kernel_h_direct<4, kEpiSwish, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi, lo);
        else
            hipLaunchKernelGGL((dense_kernel_h<4, kEpiSwish, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi, lo);
    } else if (epi == kEpiSwishAffine) {
        if (kDirect)
            hipLaunchKernelGGL((dense_kernel_h_direct<4, kEpiSwishAffine, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi, lo);
        else
            hipLaunchKernelGGL((dense_kernel_h<4, kEpiSwishAffine, MODE>), dim3(gx, gy), dim3(256), 0, st, a, hi, lo);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_dense_h(int mode, const DenseArgs& a, int epi, const void* whi, const void* wlo, hipStream_t st) {
    if (a.m == 0) return hipSuccess;
    if (a.n_pad % 128 != 0 || a.k % 32 != 0) return hipErrorInvalidValue;
    if (a.x_half) {      // 16-bit inputs: one-operand modes, rows of 8-value groups on 16-byte boundaries
        if (a.k % 64 != 0 || a.ldx % 8 != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0) return hipErrorInvalidValue;
        switch (mode) {
            case kModeBF16: return launch_dense_hx_m<kModeBF16>(a, epi, whi, st);
            case kModeF16: return launch_dense_hx_m<kModeF16>(a, epi, whi, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (mode) {
        case kModeBF16: return launch_dense_h_m<kModeBF16>(a, epi, whi, wlo, st);
        case kModeF16: return launch_dense_h_m<kModeF16>(a, epi, whi, wlo, st);
        case kModeF16X3: return launch_dense_h_m<kModeF16X3>(a, epi, whi, wlo, st);
        default: return hipErrorInvalidValue;
    }
}

template <int W, int MODE>
__global__ __launch_bounds__(512) void branch_kernel_h(const f32x4* __restrict__ P, const uint4* __restrict__ Whi,
                                                       const uint4* __restrict__ Wlo, const uint8_t* __restrict__ tok,
                                                       int64_t n_sites, float* __restrict__ out, int ldo, int col_off) {
    branch_h_body<W, MODE>(P, Whi, Wlo, tok, n_sites, out, ldo, col_off);
}

// the three window sizes of a small call in one launch (grid.y = 0: W = 51), as branch3_kernel does in fp32
struct Branch3ArgsH {
    const f32x4* P[3];          // windows 21 | 33 | 51
    const uint4* Whi[3];
    const uint4* Wlo[3];
    const uint8_t* tok[3];
    int64_t n_sites;
    float* out;
    int ldo;
};
template <int MODE>
__global__ __launch_bounds__(512) void branch3_kernel_h(Branch3ArgsH p) {
    if (blockIdx.y == 2) {
        if ((int64_t)blockIdx.x * (kTileRows / 21) < p.n_sites)
            branch_h_body<21, MODE>(p.P[0], p.Whi[0], p.Wlo[0], p.tok[0], p.n_sites, p.out, p.ldo, 0);
    } else if (blockIdx.y == 1) {
        if ((int64_t)blockIdx.x * (kTileRows / 33) < p.n_sites)
            branch_h_body<33, MODE>(p.P[1], p.Whi[1], p.Wlo[1], p.tok[1], p.n_sites, p.out, p.ldo, kD);
    } else {
        if ((int64_t)blockIdx.x * (kTileRows / 51) < p.n_sites)
            branch_h_body<51, MODE>(p.P[2], p.Whi[2], p.Wlo[2], p.tok[2], p.n_sites, p.out, p.ldo, 2 * kD);
    }
}

static constexpr size_t kBranchLdsH = 2 * (size_t)kRegionB + kTileRows * sizeof(float) + kVecsF4 * 16 + 13 * 2 * 16;

// the LDS opt-in of one instance on the current device (per device: remembered for every device of this process)
template <int W, int MODE>
static hipError_t branch_h_opt_in() {
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    bool& attr_done = attr_done_dev[dev];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&branch_kernel_h<W, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBranchLdsH);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    return hipSuccess;
}

template <int MODE>
static hipError_t branch3_h_opt_in() {
    static bool attr_done_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_done_dev[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&branch3_kernel_h<MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBranchLdsH);
    if (e == hipSuccess) attr_done_dev[dev] = true;
    return e;
}

template <int MODE>
static hipError_t branch_h_opt_in_mode() {
    hipError_t e = branch_h_opt_in<21, MODE>();
    if (e == hipSuccess) e = branch_h_opt_in<33, MODE>();
    if (e == hipSuccess) e = branch_h_opt_in<51, MODE>();
    if (e == hipSuccess) e = branch3_h_opt_in<MODE>();
    return e;
}

// Per-device kernel attributes of one half-precision mode ahead of its first launch, so that no launch -- one under a
// caller's graph capture, say -- has to set them.
hipError_t prepare_kernels_h(int mode) {
    switch (mode) {
        case kModeBF16: return branch_h_opt_in_mode<kModeBF16>();
        case kModeF16: return branch_h_opt_in_mode<kModeF16>();
        case kModeF16X3: return branch_h_opt_in_mode<kModeF16X3>();
        default: return hipErrorInvalidValue;
    }
}

template <int W, int MODE>
static hipError_t launch_branch_h_t(const float* params, const void* whi, const void* wlo, const uint8_t* tok, int64_t n,
                                    float* out, int ldo, int col_off, hipStream_t st) {
    if (hipError_t e = branch_h_opt_in<W, MODE>(); e != hipSuccess) return e;
    constexpr int S = kTileRows / W;
    const int64_t blocks = (n + S - 1) / S;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((branch_kernel_h<W, MODE>), dim3((unsigned)blocks), dim3(512), kBranchLdsH, st,
                       reinterpret_cast<const f32x4*>(params), reinterpret_cast<const uint4*>(whi),
                       reinterpret_cast<const uint4*>(wlo), tok, n, out, ldo, col_off);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_branch_h_m(int w, const float* params, const void* whi, const void* wlo, const uint8_t* tok,
                                    int64_t n, float* out, int ldo, int col_off, hipStream_t st) {
    switch (w) {
        case 21: return launch_branch_h_t<21, MODE>(params, whi, wlo, tok, n, out, ldo, col_off, st);
        case 33: return launch_branch_h_t<33, MODE>(params, whi, wlo, tok, n, out, ldo, col_off, st);
        case 51: return launch_branch_h_t<51, MODE>(params, whi, wlo, tok, n, out, ldo, col_off, st);
        default: return hipErrorInvalidValue;
    }
}

template <int MODE>
static hipError_t launch_branch3_h_m(const float* const (&params)[3], const void* const (&whi)[3], const void* const (&wlo)[3],
                                     const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (hipError_t e = branch3_h_opt_in<MODE>(); e != hipSuccess) return e;
    Branch3ArgsH p;
    for (int b = 0; b < 3; ++b) {
        p.P[b] = reinterpret_cast<const f32x4*>(params[b]);
        p.Whi[b] = reinterpret_cast<const uint4*>(whi[b]);
        p.Wlo[b] = reinterpret_cast<const uint4*>(wlo[b]);
        p.tok[b] = tok[b];
    }
    p.n_sites = n;
    p.out = out;
    p.ldo = ldo;
    constexpr int S51 = kTileRows / 51;
    const int64_t blocks = (n + S51 - 1) / S51;      // W = 51 has the most tiles
    hipLaunchKernelGGL(branch3_kernel_h<MODE>, dim3((unsigned)blocks, 3), dim3(512), kBranchLdsH, st, p);
    return hipGetLastError();
}

hipError_t launch_branch3_h(int mode, const float* const (&params)[3], const void* const (&whi)[3], const void* const (&wlo)[3],
                            const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo, hipStream_t st) {
    switch (mode) {
        case kModeBF16: return launch_branch3_h_m<kModeBF16>(params, whi, wlo, tok, n, out, ldo, st);
        case kModeF16: return launch_branch3_h_m<kModeF16>(params, whi, wlo, tok, n, out, ldo, st);
        case kModeF16X3: return launch_branch3_h_m<kModeF16X3>(params, whi, wlo, tok, n, out, ldo, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_branch_h(int mode, int w, const float* params, const void* whi, const void* wlo, const uint8_t* tok,
                           int64_t n, float* out, int ldo, int col_off, hipStream_t st) {
    switch (mode) {
        case kModeBF16: return launch_branch_h_m<kModeBF16>(w, params, whi, wlo, tok, n, out, ldo, col_off, st);
        case kModeF16: return launch_branch_h_m<kModeF16>(w, params, whi, wlo, tok, n, out, ldo, col_off, st);
        case kModeF16X3: return launch_branch_h_m<kModeF16X3>(w, params, whi, wlo, tok, n, out, ldo, col_off, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dptm

PH_DEFINE_READER(dptm_debug_phase_cycles_half)
