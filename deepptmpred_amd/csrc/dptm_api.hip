// C ABI of libdptm.so (declared in include/dptm.h): weight packing, device memory,
// chunked launch sequences and HIP-event profiling around the kernels of
// dptm_kernels.hip.  Host code only; no kernel lives in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/dptm.h"
#include "dptm_kernels.h"
#include "dptm_layout.h"

using namespace dptm;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// A C++ exception must not leave an extern "C" function: every entry point is a function-try-block
// whose handler lands here, turns the exception into a return code and records the message.
int on_exception(const char* where) noexcept {
    int code = DPTM_EHIP;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            code = DPTM_ENOMEM;
            g_err = std::string(where) + ": out of host memory";
        } catch (const std::exception& e) {
            g_err = std::string(where) + ": " + e.what();
        } catch (...) {
            g_err = std::string(where) + ": unknown C++ exception";
        }
    } catch (...) {   // even the message could not be stored
    }
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(DPTM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

constexpr double kBnEps = 1e-3;  // keras BatchNormalization default epsilon
constexpr int64_t kChunkSites = 32768;    // sites per pass of the host-buffer entry points (size of their staging slots)
// Sites per pass of the device-resident entry points.  A branch kernel holds one workgroup per CU (its LDS), so a
// launch runs in rounds of 256 tiles and the last round is partly empty; 107,520 = 84 x 5 x 256 = 60 x 7 x 256 =
// 35 x 12 x 256 sites fill whole rounds of all three window sizes (5 / 7 / 12 sites per tile), and a launch three
// times as long loses a third as much to its ragged end: 3.61e6 -> 3.68e6 sites/s over a million resident sites.
constexpr int64_t kPassSites = 107520;

// ---- host-side packing into the layouts of dptm_layout.h

// W is a Keras kernel [K][N] (row-major).  dst gets n_pad/32 * K/8 * 64 float4.
void pack_matrix(const float* W, int K, int N, int n_pad, float* dst) {
    const int k8 = K / 8;
    for (int t = 0; t < n_pad / 32; ++t)
        for (int u = 0; u < k8; ++u)
            for (int lane = 0; lane < 64; ++lane) {
                const int h = lane >> 5, j = lane & 31;
                for (int e = 0; e < 4; ++e) {
                    const int kk = 8 * u + 4 * h + e, nn = 32 * t + j;
                    dst[(((size_t)t * k8 + u) * 64 + lane) * 4 + e] =
                        nn < N ? W[(size_t)kk * N + nn] : 0.0f;
                }
            }
}

inline uint16_t f32_to_bf16(float f) {   // round to nearest even; NaN stays NaN
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
inline float f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// Half-precision image of a Keras kernel W [K][N] for v_mfma_f32_32x32x16: 16-byte unit
// ((t*(K/16) + ks)*64 + lane) holds, for idx = 0..7, W[8*(2ks + (idx>>2)) + 4h + (idx&3)][32t + j].
// kind 0: bfloat16;  1: float16 (hi);  2: float16 of the remainder W - hi (lo).
void pack_matrix_half(const float* W, int K, int N, int kind, uint16_t* dst) {
    const int KS = K / 16;
    for (int t = 0; t < N / 32; ++t)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int h = lane >> 5, j = lane & 31;
                for (int idx = 0; idx < 8; ++idx) {
                    const int kk = 8 * (2 * ks + (idx >> 2)) + 4 * h + (idx & 3);
                    const float w = W[(size_t)kk * N + 32 * t + j];
                    uint16_t v;
                    if (kind == 0) v = f32_to_bf16(w);
                    else if (kind == 1) v = f32_to_f16_bits(w);
                    else v = f32_to_f16_bits(w - f16_bits_to_f32(f32_to_f16_bits(w)));
                    dst[((((size_t)t * KS + ks) * 64 + lane) * 8) + idx] = v;
                }
            }
}

// v has F features; dst gets f_pad/4 floats grouped [half][group][4].
void pack_vector(const float* v, int F, int f_pad, float* dst) {
    const int g8 = f_pad / 8;
    for (int h = 0; h < 2; ++h)
        for (int u = 0; u < g8; ++u)
            for (int e = 0; e < 4; ++e) {
                const int f = 8 * u + 4 * h + e;
                dst[((size_t)h * g8 + u) * 4 + e] = f < F ? v[f] : 0.0f;
            }
}

// BatchNorm at inference as y = x*scale + shift (gamma, beta, mean, var consecutive)
void bn_affine(const float* g, const float* b, const float* mu, const float* var, int n,
               std::vector<float>& scale, std::vector<float>& shift) {
    scale.resize(n);
    shift.resize(n);
    for (int i = 0; i < n; ++i) {
        const double s = (double)g[i] / std::sqrt((double)var[i] + kBnEps);
        scale[i] = (float)s;
        shift[i] = (float)((double)b[i] - (double)mu[i] * s);
    }
}

struct Reader {
    const float* p;
    const float* take(size_t n) {
        const float* r = p;
        p += n;
        return r;
    }
};

struct DenseLayer {
    int k = 0, n = 0, n_pad = 0;
    float* d_w = nullptr;
    float* d_bias = nullptr;
    float* d_scale = nullptr;
    float* d_shift = nullptr;
    uint16_t* d_w_half[3] = {nullptr, nullptr, nullptr};   // bf16 | f16 hi | f16 lo images (k % 16 == 0, n % 128 == 0 only)
};

struct ProfRec {
    int cls;
    hipEvent_t a, b;
};

// per-pass activations of the per-site layers: concat [m][576], h1 [m][512], t1 [m][512], t2 [m][256]
struct Workspace {
    float* concat = nullptr;
    float* h1 = nullptr;
    float* t1 = nullptr;
    float* t2 = nullptr;
};
constexpr int64_t kWorkspaceFloatsPerSite = kConcat + kEsm1 + kT1 + kT2;   // 1856

// Workspace of dptm_score_device_noalloc: the per-site layers' activations for at most kNoAllocMaxSites
// sites (30.4 MB), resident on every GPU the library runs on until the process ends.  In host code the
// array's name is only a stand-in; its address on a device is asked for once per device (dptm_create) and
// kept in g_scratch_ptr, which is what kernels and copies are given.
constexpr int64_t kNoAllocMaxSites = 4096;
static __device__ __align__(256) float scratch[kNoAllocMaxSites * kWorkspaceFloatsPerSite];
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};

}  // namespace

struct dptm_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    float* d_branch[3] = {nullptr, nullptr, nullptr};
    uint16_t* d_branch_half[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // [branch][bf16 | f16 hi | f16 lo]
    int precision = kModeF32;
    std::vector<float> blob;              // canonical weights, kept to build the half-precision images on demand
    bool half_ready = false;
    DenseLayer strct, esm1, esm2, t1, t2, head;
    Workspace ws;                         // per-pass workspace (ws_cap rows: kChunkSites, or kPassSites once a call needed more)
    int64_t ws_cap = 0;
    bool plain = false;                   // set for the duration of a call that may create no stream or event
    // small host-array calls (at most kDenseSmallRows sites): ONE pinned block and its device twin, so that the five
    // input arrays cross PCIe as one copy from pinned memory instead of five from pageable memory
    uint8_t *small_pin = nullptr, *small_dev = nullptr;
    float* d_zero_row = nullptr;
    // per-chunk staging for assembled windows
    uint8_t* d_tok[3] = {nullptr, nullptr, nullptr};
    int32_t* d_esm_row = nullptr;
    // staging for the host-buffer entry points (allocated on first use)
    float* s_strct = nullptr;
    float* s_esm = nullptr;
    float* s_out = nullptr;
    uint8_t* s_tok[3] = {nullptr, nullptr, nullptr};
    uint8_t* s_codes = nullptr;
    int64_t s_codes_cap = 0;
    float* s_table = nullptr;
    int64_t s_table_cap = 0;
    int32_t* s_pos = nullptr;
    // whole-call staging of the multi-protein host entry point (grow-only)
    int64_t* s_seq_off = nullptr;
    int64_t* s_esm_off = nullptr;
    int64_t s_seq_off_cap = 0, s_esm_off_cap = 0;
    int32_t* s_site_pos = nullptr;
    int32_t* s_site_prot = nullptr;
    int64_t s_site_pos_cap = 0, s_site_prot_cap = 0;
    // second staging slot, copy stream and events of the pipelined host-buffer entry point (dptm_score)
    float* s2_strct = nullptr;
    float* s2_esm = nullptr;
    float* s2_out = nullptr;
    uint8_t* s2_tok[3] = {nullptr, nullptr, nullptr};
    float* pin_out[2] = {nullptr, nullptr};     // pinned host copies of a chunk's probabilities
    uint8_t* pin_in[2] = {nullptr, nullptr};    // pinned host copies of a chunk's inputs (esm | struct | tok21 | tok33 | tok51)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr};
    hipEvent_t ev_scored[2] = {nullptr, nullptr};
    bool prof_armed = false;
    std::vector<ProfRec> prof;
};

namespace {

const int kWin[3] = {21, 33, 51};

bool spec_supported(const dptm_spec* s) {
    return s && s->n_windows == 3 && s->window_sizes[0] == 21 && s->window_sizes[1] == 33 &&
           s->window_sizes[2] == 51 && s->max_features == kCodes && s->embed_dim == kD &&
           s->num_heads == kHeads && s->ff_dim == kFF && s->esm_dim == kEsm &&
           s->struct_feature_dim == kStruct && s->struct_units == kStructUnits &&
           s->esm_units1 == kEsm1 && s->esm_units2 == kEsm2 && s->trunk_units1 == kT1 &&
           s->trunk_units2 == kT2 && s->n_classes == 2;
}

int64_t weight_count() {
    int64_t per_branch_fixed = kCodes * kD + (kD * kD + kD) + 2 * (3 * kD * kD + kD + 4 * kD) +
                               2 * (4 * (kD * kD + kD) + kD * kFF + kFF + kFF * kD + kD + 4 * kD) +
                               kD + 1;
    int64_t total = 3 * per_branch_fixed + (21 + 33 + 51) * kD;
    total += kStruct * kStructUnits + kStructUnits + 4 * kStructUnits;
    total += (int64_t)kEsm * kEsm1 + kEsm1 + 4 * kEsm1;
    total += kEsm1 * kEsm2 + kEsm2;
    total += kConcat * kT1 + kT1 + 4 * kT1;
    total += kT1 * kT2 + kT2 + 4 * kT2;
    total += kT2 * 2 + 2;
    return total;
}

inline bool misaligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

int upload(const std::vector<float>& v, float** d) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(d), v.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return DPTM_OK;
}

// Packs one sequence branch (tensor order of spec.py:tensor_table) into the buffer
// layout of dptm_layout.h.
void pack_branch(Reader& rd, int W, std::vector<float>& buf, std::vector<uint16_t> (&half)[3], bool want_half) {
    buf.assign((size_t)kBranchF4 * 4, 0.0f);
    if (want_half)
        for (auto& hbuf : half) hbuf.assign((size_t)kBranchF4 * 4, 0);   // same offsets, in 16-bit units: float4 offset/2 = uint4 offset
    float* B = buf.data();
    auto f4 = [&](int off) { return B + (size_t)off * 4; };
    // a matrix at float4 offset `off` has its half images at uint4 offset off/2 = 16-bit offset off*4
    auto pack_matrix = [&](const float* Wm, int K, int N, int n_pad, float* dst) {
        if (!want_half) {
            ::pack_matrix(Wm, K, N, n_pad, dst);
            return;
        }
        const size_t off16 = (size_t)(dst - B);   // floats from the start == 16-bit units of the half image
        for (int kind = 0; kind < 3; ++kind) pack_matrix_half(Wm, K, N, kind, half[kind].data() + off16);
    };
    std::memcpy(f4(kOffEmb), rd.take(kCodes * kD), sizeof(float) * kCodes * kD);
    std::memcpy(f4(kOffPos), rd.take((size_t)W * kD), sizeof(float) * W * kD);
    pack_matrix(rd.take(kD * kD), kD, kD, kD, f4(kOffScW));
    pack_vector(rd.take(kD), kD, kD, f4(kOffConvV + kVScB * kVec64));
    std::vector<float> sc, sh;
    {
        const float* k = rd.take(3 * kD * kD);
        for (int t = 0; t < 3; ++t) pack_matrix(k + (size_t)t * kD * kD, kD, kD, kD, f4(kOffC3aW + t * kMat64));
        pack_vector(rd.take(kD), kD, kD, f4(kOffConvV + kVC3aB * kVec64));
        const float* g = rd.take(kD); const float* b = rd.take(kD);
        const float* mu = rd.take(kD); const float* var = rd.take(kD);
        bn_affine(g, b, mu, var, kD, sc, sh);
        pack_vector(sc.data(), kD, kD, f4(kOffConvV + kVBn1S * kVec64));
        pack_vector(sh.data(), kD, kD, f4(kOffConvV + kVBn1T * kVec64));
    }
    {
        const float* k = rd.take(3 * kD * kD);
        for (int t = 0; t < 3; ++t) pack_matrix(k + (size_t)t * kD * kD, kD, kD, kD, f4(kOffC3bW + t * kMat64));
        pack_vector(rd.take(kD), kD, kD, f4(kOffConvV + kVC3bB * kVec64));
        const float* g = rd.take(kD); const float* b = rd.take(kD);
        const float* mu = rd.take(kD); const float* var = rd.take(kD);
        bn_affine(g, b, mu, var, kD, sc, sh);
        pack_vector(sc.data(), kD, kD, f4(kOffConvV + kVBn2S * kVec64));
        pack_vector(sh.data(), kD, kD, f4(kOffConvV + kVBn2T * kVec64));
    }
    for (int blk = 0; blk < 2; ++blk) {
        const int T = kOffTb + blk * kTbSize;
        const int wm[4] = {kTbWq, kTbWk, kTbWv, kTbWo};
        const int bv[4] = {kVBq, kVBk, kVBv, kVBo};
        for (int i = 0; i < 4; ++i) {
            pack_matrix(rd.take(kD * kD), kD, kD, kD, f4(T + wm[i]));
            pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + bv[i] * kVec64));
        }
        pack_matrix(rd.take(kD * kFF), kD, kFF, kFF, f4(T + kTbF1));
        pack_vector(rd.take(kFF), kFF, kFF, f4(T + kTbB1));
        pack_matrix(rd.take(kFF * kD), kFF, kD, kD, f4(T + kTbF2));
        pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + kVB2 * kVec64));
        pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + kVLn1G * kVec64));
        pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + kVLn1B * kVec64));
        pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + kVLn2G * kVec64));
        pack_vector(rd.take(kD), kD, kD, f4(T + kTbVec + kVLn2B * kVec64));
    }
    pack_vector(rd.take(kD), kD, kD, f4(kOffPoolV));
    f4(kOffPoolB)[0] = *rd.take(1);
}

int make_dense(Reader& rd, int K, int N, int n_pad, bool bn, DenseLayer& L) {
    L.k = K;
    L.n = N;
    L.n_pad = n_pad;
    std::vector<float> w((size_t)K * n_pad), v((size_t)n_pad);
    const float* Wm = rd.take((size_t)K * N);
    pack_matrix(Wm, K, N, n_pad, w.data());
    int rc = upload(w, &L.d_w);
    if (rc) return rc;
    (void)Wm;
    pack_vector(rd.take(N), N, n_pad, v.data());
    rc = upload(v, &L.d_bias);
    if (rc) return rc;
    if (bn) {
        const float* g = rd.take(N); const float* b = rd.take(N);
        const float* mu = rd.take(N); const float* var = rd.take(N);
        std::vector<float> sc, sh;
        bn_affine(g, b, mu, var, N, sc, sh);
        pack_vector(sc.data(), N, n_pad, v.data());
        rc = upload(v, &L.d_scale);
        if (rc) return rc;
        pack_vector(sh.data(), N, n_pad, v.data());
        rc = upload(v, &L.d_shift);
        if (rc) return rc;
    }
    return DPTM_OK;
}

// half-precision images of a dense layer's kernel (layers with k % 16 == 0 and n % 128 == 0 only)
int make_dense_half(const float* Wm, int K, int N, DenseLayer& L) {
    if (K % 32 != 0 || N % 128 != 0) return DPTM_OK;
    std::vector<uint16_t> hw((size_t)K * N);
    for (int kind = 0; kind < 3; ++kind) {
        pack_matrix_half(Wm, K, N, kind, hw.data());
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&L.d_w_half[kind]), hw.size() * sizeof(uint16_t)));
        HIP_TRY(hipMemcpy(L.d_w_half[kind], hw.data(), hw.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    return DPTM_OK;
}

void free_dense(DenseLayer& L) {
    if (L.d_w) (void)hipFree(L.d_w);
    if (L.d_bias) (void)hipFree(L.d_bias);
    if (L.d_scale) (void)hipFree(L.d_scale);
    if (L.d_shift) (void)hipFree(L.d_shift);
    for (auto* pw : L.d_w_half)
        if (pw) (void)hipFree(pw);
    L = DenseLayer();
}

struct Prof {
    dptm_handle* h;
    hipStream_t st;
    int cls;
    bool on;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(dptm_handle* h_, hipStream_t st_, int cls_) : h(h_), st(st_), cls(cls_), on(h_->prof_armed && !h_->plain) {
        if (on) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, st);
        }
    }
    ~Prof() {
        if (on) {
            (void)hipEventRecord(b, st);
            h->prof.push_back({cls, a, b});
        }
    }
};

DenseArgs dense_args(const DenseLayer& L, const float* x, int64_t ldx, const int32_t* row_index, int64_t m, float* out,
                     int64_t ldo, int col_off, bool x_half = false) {
    DenseArgs a;
    a.x = x;
    a.ldx = ldx;
    a.x_half = x_half ? 1 : 0;
    a.row_index = row_index;
    a.m = m;
    a.k = L.k;
    a.n_pad = L.n_pad;
    a.w = reinterpret_cast<const f32x4_t*>(L.d_w);
    a.bias = reinterpret_cast<const f32x4_t*>(L.d_bias);
    a.scale = reinterpret_cast<const f32x4_t*>(L.d_scale);
    a.shift = reinterpret_cast<const f32x4_t*>(L.d_shift);
    a.out = out;
    a.ldo = ldo;
    a.col_off = col_off;
    return a;
}

int run_dense(dptm_handle* h, const DenseLayer& L, int epi, const float* x, int64_t ldx,
              const int32_t* row_index, int64_t m, float* out, int64_t ldo, int col_off,
              hipStream_t st, int cls, bool x_half = false) {
    const DenseArgs a = dense_args(L, x, ldx, row_index, m, out, ldo, col_off, x_half);
    Prof p(h, st, cls);
    if (x_half && (h->precision == kModeF32 || h->precision == kModeF16X3 || !L.d_w_half[0]))
        return fail(DPTM_EINVAL, "16-bit input rows need the bf16 or f16 mode");
    if (h->precision != kModeF32 && L.d_w_half[0] && epi != kEpiHead && (m > kDenseSmallRows || x_half)) {
        HIP_TRY(launch_dense_h(h->precision, a, epi, L.d_w_half[h->precision == kModeBF16 ? 0 : 1], L.d_w_half[2], st));
    } else {
        HIP_TRY(launch_dense(a, epi, st));   // fp32: the 8-feature structure layer, the 2-class head, small calls
    }
    return DPTM_OK;
}

// One chunk (m <= kChunkSites) of the forward pass, everything device-resident.
// esm_half: the ESM rows hold the mode's bfloat16 / float16 values (what the host pipelines send in those modes)
int score_chunk(dptm_handle* h, const Workspace& ws, const uint8_t* const tok[3], const float* strct, const float* esm,
                int64_t esm_ld, const int32_t* esm_row, int64_t m, float* out, hipStream_t st, bool esm_half = false) {
    // A small call cannot fill the chip with any one kernel, and its duration is one wavefront chain per branch: its
    // three sequence branches are ONE launch (grid.y = branch).  (As three launches on three side streams the call
    // paid two hand-overs between streams, and the chain twice over whenever two of those streams landed on one hardware
    // queue: 362 us for 53 sites against 202 us now.)  A large call starts a launch per branch, on one stream.
    const bool merged = m <= kDenseSmallRows && !h->prof_armed;
    bool front = false;      // the structure and first ESM layer went out with the branches
    if (merged) {
        const float* const params[3] = {h->d_branch[0], h->d_branch[1], h->d_branch[2]};
        const uint8_t* const toks[3] = {tok[0], tok[1], tok[2]};
        if (h->precision == kModeF32 && !esm_half) {
            // ... together with the two layers that depend on the inputs only (structure, first ESM layer)
            front = true;
            HIP_TRY(launch_front5(params, toks, m, ws.concat, kConcat,
                                  dense_args(h->strct, strct, kStruct, nullptr, m, ws.concat, kConcat, 3 * kD),
                                  dense_args(h->esm1, esm, esm_ld, esm_row, m, ws.h1, kEsm1, 0), st));
        } else if (h->precision == kModeF32) {
            HIP_TRY(launch_branch3(params, toks, m, ws.concat, kConcat, st));
        } else {
            const int hi = h->precision == kModeBF16 ? 0 : 1;
            const void* const whi[3] = {h->d_branch_half[0][hi], h->d_branch_half[1][hi], h->d_branch_half[2][hi]};
            const void* const wlo[3] = {h->d_branch_half[0][2], h->d_branch_half[1][2], h->d_branch_half[2][2]};
            HIP_TRY(launch_branch3_h(h->precision, params, whi, wlo, toks, m, ws.concat, kConcat, st));
        }
    }
    for (int b = 0; b < 3 && !merged; ++b) {
        Prof p(h, st, DPTM_PROF_BRANCH21 + b);
        if (h->precision == kModeF32) {
            HIP_TRY(launch_branch(kWin[b], h->d_branch[b], tok[b], m, ws.concat, kConcat, b * kD, st));
        } else {
            const uint16_t* hi = h->d_branch_half[b][h->precision == kModeBF16 ? 0 : 1];
            HIP_TRY(launch_branch_h(h->precision, kWin[b], h->d_branch[b], hi, h->d_branch_half[b][2], tok[b], m,
                                    ws.concat, kConcat, b * kD, st));
        }
    }
    int rc;
    if (!front && (rc = run_dense(h, h->strct, kEpiSwishAffine, strct, kStruct, nullptr, m, ws.concat, kConcat,
                                  3 * kD, st, DPTM_PROF_STRUCT)))
        return rc;
    if (!front && (rc = run_dense(h, h->esm1, kEpiSwishAffine, esm, esm_ld, esm_row, m, ws.h1, kEsm1, 0, st,
                                  DPTM_PROF_ESM1, esm_half)))
        return rc;
    if ((rc = run_dense(h, h->esm2, kEpiSwish, ws.h1, kEsm1, nullptr, m, ws.concat, kConcat,
                        3 * kD + kStructUnits, st, DPTM_PROF_ESM2)))
        return rc;
    if ((rc = run_dense(h, h->t1, kEpiSwishAffine, ws.concat, kConcat, nullptr, m, ws.t1, kT1, 0, st,
                        DPTM_PROF_TRUNK1)))
        return rc;
    if ((rc = run_dense(h, h->t2, kEpiSwishAffine, ws.t1, kT1, nullptr, m, ws.t2, kT2, 0, st,
                        DPTM_PROF_TRUNK2)))
        return rc;
    return run_dense(h, h->head, kEpiHead, ws.t2, kT2, nullptr, m, out, 2, 0, st, DPTM_PROF_HEAD);
}

// Per-pass device buffers (activations of the per-site layers, assembled windows, ESM row indices) for passes of up to
// `sites` sites: kChunkSites rows from dptm_create on, kPassSites rows from the first call that scores more than that
// in one go (grow-only; the old buffers are released once nothing can be using them).
int ensure_workspace(dptm_handle* h, int64_t sites) {
    if (sites <= h->ws_cap) return DPTM_OK;
    const int64_t cap = sites > kChunkSites ? kPassSites : kChunkSites;
    if (h->ws_cap) HIP_TRY(hipDeviceSynchronize());
    void** bufs[] = {(void**)&h->ws.concat, (void**)&h->ws.h1, (void**)&h->ws.t1, (void**)&h->ws.t2, (void**)&h->d_tok[0],
                     (void**)&h->d_tok[1], (void**)&h->d_tok[2], (void**)&h->d_esm_row};
    const size_t bytes[] = {cap * (size_t)kConcat * 4, cap * (size_t)kEsm1 * 4, cap * (size_t)kT1 * 4, cap * (size_t)kT2 * 4,
                            cap * (size_t)kWin[0], cap * (size_t)kWin[1], cap * (size_t)kWin[2], cap * sizeof(int32_t)};
    h->ws_cap = 0;
    for (int i = 0; i < 8; ++i) {
        if (*bufs[i]) (void)hipFree(*bufs[i]);
        *bufs[i] = nullptr;
    }
    for (int i = 0; i < 8; ++i)
        if (hipMalloc(bufs[i], bytes[i]) != hipSuccess) return fail(DPTM_ENOMEM, "device allocation of the per-pass workspace failed");
    h->ws_cap = cap;
    return DPTM_OK;
}

int ensure_host_staging(dptm_handle* h) {
    if (h->s_out) return DPTM_OK;
    for (int b = 0; b < 3; ++b) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_tok[b]), kChunkSites * kWin[b]));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_strct), kChunkSites * kStruct * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_esm), kChunkSites * (size_t)kEsm * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_pos), kChunkSites * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_out), kChunkSites * 2 * sizeof(float)));
    return DPTM_OK;
}

// layout of one pinned input slot
constexpr size_t kPinEsm = 0;
constexpr size_t kPinStruct = kPinEsm + (size_t)kChunkSites * kEsm * sizeof(float);
constexpr size_t kPinTok21 = kPinStruct + (size_t)kChunkSites * kStruct * sizeof(float);
constexpr size_t kPinTok33 = kPinTok21 + (size_t)kChunkSites * 21;
constexpr size_t kPinTok51 = kPinTok33 + (size_t)kChunkSites * 33;
constexpr size_t kPinInBytes = kPinTok51 + (size_t)kChunkSites * 51;

// threads of the host-side gather / conversion: one core moves ~10 GB/s, PCIe takes 50+ (DPTM_HOST_THREADS overrides)
unsigned host_threads() {
    static const unsigned v = [] {
        if (const char* e = std::getenv("DPTM_HOST_THREADS")) {
            const long x = std::atol(e);
            if (x >= 1 && x <= 256) return (unsigned)x;
        }
        const unsigned hw = std::thread::hardware_concurrency();
        return std::min<unsigned>(hw ? hw : 1u, 12u);
    }();
    return v;
}

// work(a, b) over [0, n) in slices of at least `grain` on a few threads; the first slice runs on the caller's thread
template <typename F>
void parallel_slices(size_t n, size_t grain, F work) {
    size_t nt = std::min<size_t>(host_threads(), n / std::max<size_t>(grain, 1));
    if (nt <= 1) {
        work((size_t)0, n);
        return;
    }
    const size_t per = (n + nt - 1) / nt;
    std::vector<std::thread> pool;
    pool.reserve(nt);
    size_t done = std::min(per, n);
    for (size_t t = 1; t < nt && done < n; ++t) {
        const size_t a = done, b = std::min(n, a + per);
        try {
            pool.emplace_back(work, a, b);
        } catch (...) {   // no more threads to be had: the rest runs here
            work(a, n);
            done = n;
            break;
        }
        done = b;
    }
    work((size_t)0, std::min(per, n));
    for (auto& th : pool) th.join();
}

// memcpy split over a few threads
void parallel_copy(void* dst, const void* src, size_t bytes) {
    constexpr size_t kMinPerThread = 8u << 20;
    parallel_slices(bytes >> 12, kMinPerThread >> 12, [=](size_t p0, size_t p1) {      // in 4 KiB pages
        const size_t lo = p0 << 12, hi = p1 == (bytes >> 12) ? bytes : p1 << 12;
        std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo);
    });
    if ((bytes >> 12) == 0) std::memcpy(dst, src, bytes);
}

// ---- fp32 -> bfloat16 / float16 on the host, round to nearest even: the values the kernels' own conversions give, so a
// row converted here and a row converted on the device score alike bit for bit.  In the one-operand half-precision
// modes the host pipelines send the ESM rows in this form: half the bytes over PCIe.
inline uint16_t bf16_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);     // NaN stays NaN (quiet)
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}
inline uint16_t f16_rne(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                        // NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                       // rounds to infinity (>= 65520)
    if (x < 0x38800000u) {                                                         // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;                                // below half of the smallest subnormal
        const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t shift = 126u - e;                                           // 14 .. 24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = x - 0x38000000u;                                                  // rebias 127 -> 15
    r += 0xfffu + ((r >> 13) & 1u);
    return (uint16_t)(sign | (r >> 13));
}
void convert_rows_scalar(uint16_t* dst, const float* src, size_t n, int mode) {
    if (mode == kModeBF16)
        for (size_t i = 0; i < n; ++i) dst[i] = bf16_rne(src[i]);
    else
        for (size_t i = 0; i < n; ++i) dst[i] = f16_rne(src[i]);
}
#if defined(__x86_64__)
__attribute__((target("avx2,f16c"))) void convert_rows_avx2(uint16_t* dst, const float* src, size_t n, int mode) {
    size_t i = 0;
    if (mode == kModeBF16) {
        const __m256i bias = _mm256_set1_epi32(0x7fff), one = _mm256_set1_epi32(1), absm = _mm256_set1_epi32(0x7fffffff),
                      inf = _mm256_set1_epi32(0x7f800000), quiet = _mm256_set1_epi32(0x40);
        for (; i + 16 <= n; i += 16) {
            __m256i r[2];
            for (int k = 0; k < 2; ++k) {
                const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 8 * k));
                const __m256i rounded = _mm256_srli_epi32(_mm256_add_epi32(x, _mm256_add_epi32(bias, _mm256_and_si256(_mm256_srli_epi32(x, 16), one))), 16);
                const __m256i is_nan = _mm256_cmpgt_epi32(_mm256_and_si256(x, absm), inf);
                r[k] = _mm256_blendv_epi8(rounded, _mm256_or_si256(_mm256_srli_epi32(x, 16), quiet), is_nan);
            }
            const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi32(r[0], r[1]), 0xd8);
            _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), packed);
        }
    } else {
        for (; i + 8 <= n; i += 8)
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i),
                             _mm256_cvtps_ph(_mm256_loadu_ps(src + i), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
    }
    convert_rows_scalar(dst + i, src + i, n - i, mode);
}
#endif
void convert_rows(uint16_t* dst, const float* src, size_t n, int mode) {
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c");
    if (fast) return convert_rows_avx2(dst, src, n, mode);
#endif
    convert_rows_scalar(dst, src, n, mode);
}

// m rows of 1280 floats -> the mode's 16-bit values, on a few threads
void parallel_convert(uint16_t* dst, const float* src, int64_t m, int mode) {
    parallel_slices((size_t)m, 1024, [=](size_t a, size_t b) { convert_rows(dst + a * kEsm, src + a * kEsm, (b - a) * kEsm, mode); });
}

int ensure_pipeline(dptm_handle* h) {
    if (h->ev_scored[1]) return DPTM_OK;
    for (int b = 0; b < 3; ++b) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s2_tok[b]), kChunkSites * kWin[b]));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s2_strct), kChunkSites * kStruct * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s2_esm), kChunkSites * (size_t)kEsm * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s2_out), kChunkSites * 2 * sizeof(float)));
    for (int s = 0; s < 2; ++s) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_out[s]), kChunkSites * 2 * sizeof(float), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pin_in[s]), kPinInBytes, hipHostMallocDefault));
    }
    HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int s = 0; s < 2; ++s) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_copied[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_scored[s], hipEventDisableTiming));
    }
    return DPTM_OK;
}

// bf16 / f16 hi / f16 lo images of every matrix the half-precision kernels read, built the first time a
// half-precision mode is selected (the fp32 default never pays for them)
int build_half_images(dptm_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    Reader rd{h->blob.data()};
    std::vector<float> buf;
    std::vector<uint16_t> half[3];
    for (int b = 0; b < 3; ++b) {
        pack_branch(rd, kWin[b], buf, half, true);
        for (int kind = 0; kind < 3; ++kind) {
            const size_t bytes = half[kind].size() * sizeof(uint16_t);
            if (!h->d_branch_half[b][kind]) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_branch_half[b][kind]), bytes));
            HIP_TRY(hipMemcpy(h->d_branch_half[b][kind], half[kind].data(), bytes, hipMemcpyHostToDevice));
        }
    }
    struct { int k, n; bool bn; DenseLayer* L; } layers[] = {
        {kStruct, kStructUnits, true, &h->strct}, {kEsm, kEsm1, true, &h->esm1}, {kEsm1, kEsm2, false, &h->esm2},
        {kConcat, kT1, true, &h->t1}, {kT1, kT2, true, &h->t2}};
    for (auto& l : layers) {
        const float* Wm = rd.take((size_t)l.k * l.n);
        rd.take(l.n);                                   // bias
        if (l.bn) rd.take(4 * (size_t)l.n);             // gamma, beta, moving mean, moving variance
        if (!l.L->d_w_half[0]) {
            int rc = make_dense_half(Wm, l.k, l.n, *l.L);
            if (rc) return rc;
        }
    }
    h->half_ready = true;
    return DPTM_OK;
}

// The host-buffer entry points enqueue copies out of the caller's arrays.  On any early return the
// work already queued is waited for, so that nothing reads those arrays (or the staging slots a later
// call will refill) after the call has reported its error.
struct DrainOnError {
    dptm_handle* h;
    bool armed = true;
    ~DrainOnError() {
        if (!armed) return;
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    }
};

// rows of the concatenated ESM tables that the sites [i0, i1) need, gathered into dst [i1-i0][1280]
// (zero row when the protein has no table or the position lies beyond it: predict.py:493,498-500);
// half_mode != 0: dst receives the rows as that mode's 16-bit values (uint16 [i1-i0][1280])
void gather_esm_rows(void* dst, const float* tables, const int64_t* esm_off, const int32_t* positions,
                     const int32_t* site_protein, int64_t i0, int64_t i1, int half_mode = 0) {
    parallel_slices((size_t)(i1 - i0), 2048, [=](size_t a, size_t b) {
        for (int64_t i = i0 + (int64_t)a; i < i0 + (int64_t)b; ++i) {
            const int64_t p = site_protein[i], r = (int64_t)positions[i] - 1;
            const bool have = tables && r < esm_off[p + 1] - esm_off[p];
            const float* srow = have ? tables + (esm_off[p] + r) * (int64_t)kEsm : nullptr;
            if (half_mode) {
                uint16_t* d = static_cast<uint16_t*>(dst) + (i - i0) * (int64_t)kEsm;
                if (have)
                    convert_rows(d, srow, kEsm, half_mode);
                else
                    std::memset(d, 0, kEsm * sizeof(uint16_t));
            } else {
                float* d = static_cast<float*>(dst) + (i - i0) * (int64_t)kEsm;
                if (have)
                    std::memcpy(d, srow, kEsm * sizeof(float));
                else
                    std::memset(d, 0, kEsm * sizeof(float));
            }
        }
    });
}

// grow-only device buffer
template <typename T>
int ensure_capacity(T** buf, int64_t* cap, int64_t want) {
    if (want <= *cap) return DPTM_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(buf), (size_t)want * sizeof(T)));
    *cap = want;
    return DPTM_OK;
}

int check_protein_tables(const char* who, const int64_t* seq_off, const int64_t* esm_off, int64_t n_proteins,
                         const int32_t* positions, const int32_t* site_protein, int64_t n) {
    char msg[192];
    if (seq_off[0] != 0 || esm_off[0] != 0) return fail(DPTM_EINVAL, std::string(who) + ": offsets must start at 0");
    for (int64_t p = 0; p < n_proteins; ++p)
        if (seq_off[p + 1] < seq_off[p] || esm_off[p + 1] < esm_off[p]) {
            std::snprintf(msg, sizeof msg, "%s: offsets decrease at protein %lld", who, (long long)p);
            return fail(DPTM_EINVAL, msg);
        }
    if (esm_off[n_proteins] > INT32_MAX)
        return fail(DPTM_EINVAL, std::string(who) + ": more than 2^31-1 ESM table rows in one call");
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = site_protein[i];
        if (p < 0 || p >= n_proteins) {
            std::snprintf(msg, sizeof msg, "%s: site %lld names protein %lld of %lld", who, (long long)i, (long long)p,
                          (long long)n_proteins);
            return fail(DPTM_EINVAL, msg);
        }
        const int64_t L = seq_off[p + 1] - seq_off[p];
        if (positions[i] < 1 || positions[i] > L) {
            std::snprintf(msg, sizeof msg, "%s: position %d (site %lld, protein %lld) outside 1..%lld", who, positions[i],
                          (long long)i, (long long)p, (long long)L);
            return fail(DPTM_EINVAL, msg);
        }
    }
    return DPTM_OK;
}

}  // namespace

extern "C" {

const char* dptm_last_error(void) { return g_err.c_str(); }

int64_t dptm_weight_count(const dptm_spec* spec) {
    if (!spec_supported(spec)) return fail(DPTM_EINVAL, "dptm_weight_count: unsupported model spec");
    return weight_count();
}

void dptm_destroy(dptm_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& r : h->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (int b = 0; b < 3; ++b) {
        if (h->d_branch[b]) (void)hipFree(h->d_branch[b]);
        for (int kind = 0; kind < 3; ++kind)
            if (h->d_branch_half[b][kind]) (void)hipFree(h->d_branch_half[b][kind]);
        if (h->d_tok[b]) (void)hipFree(h->d_tok[b]);
        if (h->s_tok[b]) (void)hipFree(h->s_tok[b]);
        if (h->s2_tok[b]) (void)hipFree(h->s2_tok[b]);
    }
    if (h->small_pin) (void)hipHostFree(h->small_pin);
    if (h->small_dev) (void)hipFree(h->small_dev);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    for (int s = 0; s < 2; ++s) {
        if (h->ev_copied[s]) (void)hipEventDestroy(h->ev_copied[s]);
        if (h->ev_scored[s]) (void)hipEventDestroy(h->ev_scored[s]);
        if (h->pin_out[s]) (void)hipHostFree(h->pin_out[s]);
        if (h->pin_in[s]) (void)hipHostFree(h->pin_in[s]);
    }
    free_dense(h->strct);
    free_dense(h->esm1);
    free_dense(h->esm2);
    free_dense(h->t1);
    free_dense(h->t2);
    free_dense(h->head);
    void* bufs[] = {h->ws.concat, h->ws.h1, h->ws.t1, h->ws.t2, h->d_zero_row, h->d_esm_row, h->s_strct,
                    h->s_esm, h->s_out, h->s_codes, h->s_table, h->s_pos, h->s2_strct, h->s2_esm, h->s2_out,
                    h->s_seq_off, h->s_esm_off, h->s_site_pos, h->s_site_prot};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int dptm_create(const void* weights, size_t nbytes, const dptm_spec* spec, int device, dptm_handle** out) try {
    if (!out) return fail(DPTM_EINVAL, "dptm_create: out is NULL");
    *out = nullptr;
    if (!weights) return fail(DPTM_EINVAL, "dptm_create: weights is NULL");
    if (!spec_supported(spec))
        return fail(DPTM_EINVAL, "dptm_create: unsupported model spec (only the reference shape "
                                 "windows 21/33/51, embed 64, 4 heads, ff 256, esm 1280 is built)");
    const int64_t want = weight_count();
    if ((int64_t)(nbytes / sizeof(float)) != want || nbytes % sizeof(float) != 0) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "dptm_create: weight blob has %zu bytes, expected %lld", nbytes,
                      (long long)(want * (int64_t)sizeof(float)));
        return fail(DPTM_EINVAL, msg);
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DPTM_EINVAL, "dptm_create: no such device");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DPTM_EINVAL, std::string("dptm_create: kernels are built for gfx950, device is ") +
                                     prop.gcnArchName);

    dptm_handle* h = new (std::nothrow) dptm_handle();
    if (!h) return fail(DPTM_ENOMEM, "dptm_create: out of host memory");
    std::unique_ptr<dptm_handle, void (*)(dptm_handle*)> owner(h, dptm_destroy);   // released on success only
    h->device = device;
    int rc = DPTM_OK;
    auto bail = [&](int code) { return code; };   // `owner` frees the half-built handle on every early return
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(DPTM_EHIP, "dptm_create: hipStreamCreate failed"));

    h->blob.assign(static_cast<const float*>(weights), static_cast<const float*>(weights) + want);
    Reader rd{h->blob.data()};
    std::vector<float> buf;
    std::vector<uint16_t> half[3];
    for (int b = 0; b < 3; ++b) {
        pack_branch(rd, kWin[b], buf, half, false);
        if ((rc = upload(buf, &h->d_branch[b]))) return bail(rc);
    }
    if ((rc = make_dense(rd, kStruct, kStructUnits, kStructUnits, true, h->strct))) return bail(rc);
    if ((rc = make_dense(rd, kEsm, kEsm1, kEsm1, true, h->esm1))) return bail(rc);
    if ((rc = make_dense(rd, kEsm1, kEsm2, kEsm2, false, h->esm2))) return bail(rc);
    if ((rc = make_dense(rd, kConcat, kT1, kT1, true, h->t1))) return bail(rc);
    if ((rc = make_dense(rd, kT1, kT2, kT2, true, h->t2))) return bail(rc);
    if ((rc = make_dense(rd, kT2, 2, 32, false, h->head))) return bail(rc);
    if (rd.p - h->blob.data() != want)
        return bail(fail(DPTM_EINVAL, "dptm_create: internal weight-order mismatch"));

    auto dalloc = [&](void** p, size_t bytes) -> int {
        if (hipMalloc(p, bytes) != hipSuccess) return fail(DPTM_ENOMEM, "dptm_create: device allocation failed");
        return DPTM_OK;
    };
    if ((rc = ensure_workspace(h, 1))) return bail(rc);
    if (device >= kMaxDevices) return bail(fail(DPTM_EINVAL, "dptm_create: device index beyond the library's table"));
    if (!g_scratch_ptr[device]) {     // this device's copy of the static workspace (device `device` is current)
        float* sp = nullptr;
        hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&sp), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sp)
            return bail(fail(DPTM_EHIP, std::string("dptm_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)));
        g_scratch_ptr[device] = sp;
    }
    if (hipError_t e = prepare_kernels(); e != hipSuccess)
        return bail(fail(DPTM_EHIP, std::string("dptm_create: kernel set-up failed: ") + hipGetErrorString(e)));
    if ((rc = dalloc((void**)&h->d_zero_row, kEsm * 4))) return bail(rc);
    if (hipMemset(h->d_zero_row, 0, kEsm * 4) != hipSuccess) return bail(fail(DPTM_EHIP, "hipMemset failed"));
    if (hipDeviceSynchronize() != hipSuccess) return bail(fail(DPTM_EHIP, "dptm_create: device sync failed"));
    *out = owner.release();
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_create");
}

int64_t dptm_chunk_sites(const dptm_handle* h) { return h ? kChunkSites : 0; }

int dptm_set_precision(dptm_handle* h, int mode) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_set_precision: NULL handle");
    if (mode != kModeF32 && mode != kModeBF16 && mode != kModeF16 && mode != kModeF16X3)
        return fail(DPTM_EINVAL, "dptm_set_precision: unknown mode");
    if (mode != kModeF32 && !h->half_ready) {
        int rc = build_half_images(h);
        if (rc) return rc;
    }
    if (mode != kModeF32) {
        // kernel attributes now, not at the mode's first launch (which may sit inside a caller's graph capture)
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(prepare_kernels_h(mode));
    }
    h->precision = mode;
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_set_precision");
}

int dptm_convert_rows(const float* src, int64_t n, int mode, uint16_t* dst) try {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(DPTM_EINVAL, "dptm_convert_rows: bad argument");
    if (mode != kModeBF16 && mode != kModeF16) return fail(DPTM_EINVAL, "dptm_convert_rows: mode must be bf16 (1) or f16 (2)");
    parallel_slices((size_t)n, (size_t)1 << 20, [=](size_t a, size_t b) { convert_rows(dst + a, src + a, b - a, mode); });
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_convert_rows");
}

int dptm_sync(dptm_handle* h) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_sync: NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_sync");
}

int dptm_score_device(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                      const float* strct, const float* esm, int64_t n, float* out, void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_device: NULL handle");
    if (n < 0) return fail(DPTM_EINVAL, "dptm_score_device: n < 0");
    if (n == 0) return DPTM_OK;
    if (!tok21 || !tok33 || !tok51 || !strct || !esm || !out)
        return fail(DPTM_EINVAL, "dptm_score_device: NULL buffer");
    // the kernels read rows with 16-byte vector loads and store probability pairs with 8-byte stores
    if (misaligned(strct, 16) || misaligned(esm, 16) || misaligned(out, 8))
        return fail(DPTM_EINVAL, "dptm_score_device: struct / esm must be 16-byte aligned, out 8-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = ensure_workspace(h, std::min(n, kPassSites))) return rc;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    for (int64_t c0 = 0; c0 < n; c0 += kPassSites) {
        const int64_t m = n - c0 < kPassSites ? n - c0 : kPassSites;
        const uint8_t* tok[3] = {tok21 + c0 * 21, tok33 + c0 * 33, tok51 + c0 * 51};
        int rc = score_chunk(h, h->ws, tok, strct + c0 * kStruct, esm + c0 * (int64_t)kEsm, kEsm, nullptr, m,
                             out + c0 * 2, st);
        if (rc) return rc;
    }
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_score_device");
}

int64_t dptm_noalloc_max_sites(void) { return kNoAllocMaxSites; }

int dptm_score_device_noalloc(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                              const float* strct, const float* esm, int64_t n, float* out, void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_device_noalloc: NULL handle");
    if (n < 0) return fail(DPTM_EINVAL, "dptm_score_device_noalloc: n < 0");
    if (n > kNoAllocMaxSites) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "dptm_score_device_noalloc: %lld sites, the static workspace holds at most %lld per call",
                      (long long)n, (long long)kNoAllocMaxSites);
        return fail(DPTM_EINVAL, msg);
    }
    if (n == 0) return DPTM_OK;
    if (!tok21 || !tok33 || !tok51 || !strct || !esm || !out)
        return fail(DPTM_EINVAL, "dptm_score_device_noalloc: NULL buffer");
    if (misaligned(strct, 16) || misaligned(esm, 16) || misaligned(out, 8))
        return fail(DPTM_EINVAL, "dptm_score_device_noalloc: struct / esm must be 16-byte aligned, out 8-byte aligned");
    float* base = g_scratch_ptr[h->device];
    if (!base) return fail(DPTM_EHIP, "dptm_score_device_noalloc: no workspace address for this device");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    // every row the layers read was written earlier in the same call (rows past n are never read), so nothing in
    // the workspace has to start at zero
    Workspace ws;
    ws.concat = base;
    ws.h1 = ws.concat + kNoAllocMaxSites * kConcat;
    ws.t1 = ws.h1 + kNoAllocMaxSites * kEsm1;
    ws.t2 = ws.t1 + kNoAllocMaxSites * kT1;
    struct Plain {       // no side streams, no timing events: this call creates nothing
        dptm_handle* h;
        ~Plain() { h->plain = false; }
    } plain{h};
    h->plain = true;
    const uint8_t* tok[3] = {tok21, tok33, tok51};
    return score_chunk(h, ws, tok, strct, esm, kEsm, nullptr, n, out, st);
} catch (...) {
    return on_exception("dptm_score_device_noalloc");
}

int dptm_assemble_device(dptm_handle* h, const uint8_t* codes, int64_t L, int64_t esm_rows,
                         const int32_t* positions, int64_t n, uint8_t* tok21, uint8_t* tok33,
                         uint8_t* tok51, int32_t* esm_row, void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_assemble_device: NULL handle");
    if (n < 0 || L < 0 || esm_rows < 0) return fail(DPTM_EINVAL, "dptm_assemble_device: negative size");
    if (n == 0) return DPTM_OK;
    if (!codes || !positions || !tok21 || !tok33 || !tok51 || !esm_row)
        return fail(DPTM_EINVAL, "dptm_assemble_device: NULL buffer");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    AssembleArgs a{};
    a.codes = codes;
    a.L = L;
    a.esm_rows = esm_rows;
    a.positions = positions;
    a.n = n;
    a.t21 = tok21;
    a.t33 = tok33;
    a.t51 = tok51;
    a.esm_row = esm_row;
    Prof p(h, st, DPTM_PROF_ASSEMBLE);
    HIP_TRY(launch_assemble(a, st));
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_assemble_device");
}

int dptm_score_protein_device(dptm_handle* h, const uint8_t* codes, int64_t L, const float* esm_table,
                              int64_t esm_rows, const int32_t* positions, const float* strct, int64_t n,
                              float* out, void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_protein_device: NULL handle");
    if (n < 0 || L < 0 || esm_rows < 0) return fail(DPTM_EINVAL, "dptm_score_protein_device: negative size");
    if (n == 0) return DPTM_OK;
    if (!codes || !positions || !strct || !out || (esm_rows > 0 && !esm_table))
        return fail(DPTM_EINVAL, "dptm_score_protein_device: NULL buffer");
    if (misaligned(strct, 16) || (esm_rows > 0 && misaligned(esm_table, 16)) || misaligned(out, 8) || misaligned(positions, 4))
        return fail(DPTM_EINVAL, "dptm_score_protein_device: struct / esm table must be 16-byte aligned, out 8-byte, positions 4-byte");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    const float* table = esm_rows > 0 ? esm_table : h->d_zero_row;
    if (int rc = ensure_workspace(h, std::min(n, kPassSites))) return rc;
    for (int64_t c0 = 0; c0 < n; c0 += kPassSites) {
        const int64_t m = n - c0 < kPassSites ? n - c0 : kPassSites;
        int rc = dptm_assemble_device(h, codes, L, esm_rows, positions + c0, m, h->d_tok[0], h->d_tok[1],
                                      h->d_tok[2], h->d_esm_row, st);
        if (rc) return rc;
        const uint8_t* tok[3] = {h->d_tok[0], h->d_tok[1], h->d_tok[2]};
        rc = score_chunk(h, h->ws, tok, strct + c0 * kStruct, table, kEsm, h->d_esm_row, m, out + c0 * 2, st);
        if (rc) return rc;
    }
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_score_protein_device");
}

// dptm_score and dptm_score_rows16: `esm` holds fp32 rows, or (rows16) the 16-bit operand values of the handle's
// one-operand half-precision mode
static int score_host_rows(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                           const float* strct, const void* esm_rows, bool rows16, int64_t n, float* out) {
    if (!h) return fail(DPTM_EINVAL, "dptm_score: NULL handle");
    if (n < 0) return fail(DPTM_EINVAL, "dptm_score: n < 0");
    if (n == 0) return DPTM_OK;
    if (!tok21 || !tok33 || !tok51 || !strct || !esm_rows || !out) return fail(DPTM_EINVAL, "dptm_score: NULL buffer");
    if (rows16 && h->precision != kModeBF16 && h->precision != kModeF16)
        return fail(DPTM_EINVAL, "dptm_score_rows16: 16-bit input rows need the bf16 or f16 mode (dptm_set_precision)");
    const float* esm = static_cast<const float*>(esm_rows);            // when !rows16
    const uint16_t* esm16 = static_cast<const uint16_t*>(esm_rows);    // when rows16
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_host_staging(h);
    if (rc) return rc;
    DrainOnError drain{h};
    if (n <= kDenseSmallRows && !rows16) {
        // a protein's worth of sites: the call is latency -- the inputs are packed into one pinned block and cross
        // PCIe as ONE copy (five copies from pageable memory cost ~60 us more), the probabilities come back through it
        const size_t o_struct = (size_t)n * kEsm * sizeof(float), o_tok0 = o_struct + (size_t)n * kStruct * sizeof(float);
        const size_t o_tok1 = (o_tok0 + (size_t)n * kWin[0] + 15) & ~(size_t)15, o_tok2 = (o_tok1 + (size_t)n * kWin[1] + 15) & ~(size_t)15;
        const size_t o_out = (o_tok2 + (size_t)n * kWin[2] + 15) & ~(size_t)15, total = o_out + (size_t)n * 2 * sizeof(float);
        constexpr size_t kSmallBytes = (size_t)kDenseSmallRows * (kEsm * 4 + kStruct * 4 + 21 + 33 + 51 + 8) + 64;
        if (!h->small_pin) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->small_pin), kSmallBytes, hipHostMallocDefault));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->small_dev), kSmallBytes));
        }
        if (total > kSmallBytes) return fail(DPTM_EINVAL, "dptm_score: small-call staging");
        uint8_t *pin = h->small_pin, *dev = h->small_dev;
        std::memcpy(pin, esm, o_struct);
        std::memcpy(pin + o_struct, strct, (size_t)n * kStruct * sizeof(float));
        std::memcpy(pin + o_tok0, tok21, (size_t)n * kWin[0]);
        std::memcpy(pin + o_tok1, tok33, (size_t)n * kWin[1]);
        std::memcpy(pin + o_tok2, tok51, (size_t)n * kWin[2]);
        HIP_TRY(hipMemcpyAsync(dev, pin, o_out, hipMemcpyHostToDevice, h->stream));
        const uint8_t* tok[3] = {dev + o_tok0, dev + o_tok1, dev + o_tok2};
        float* d_out = reinterpret_cast<float*>(dev + o_out);
        if ((rc = score_chunk(h, h->ws, tok, reinterpret_cast<const float*>(dev + o_struct), reinterpret_cast<const float*>(dev), kEsm,
                              nullptr, n, d_out, h->stream)))
            return rc;
        HIP_TRY(hipMemcpyAsync(pin + o_out, d_out, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::memcpy(out, pin + o_out, (size_t)n * 2 * sizeof(float));
        drain.armed = false;
        return DPTM_OK;
    }
    if (n <= kChunkSites) {
        // one pass: nothing to overlap, and the pinned staging blocks of the pipeline (2 x 172 MB) are not
        // worth allocating for a protein's worth of sites
        for (int b = 0; b < 3; ++b)
            HIP_TRY(hipMemcpyAsync(h->s_tok[b], (b == 0 ? tok21 : b == 1 ? tok33 : tok51), n * kWin[b], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_strct, strct, n * kStruct * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_esm, esm_rows, n * (size_t)kEsm * (rows16 ? sizeof(uint16_t) : sizeof(float)), hipMemcpyHostToDevice,
                               h->stream));
        const uint8_t* tok[3] = {h->s_tok[0], h->s_tok[1], h->s_tok[2]};
        if ((rc = score_chunk(h, h->ws, tok, h->s_strct, h->s_esm, kEsm, nullptr, n, h->s_out, h->stream, rows16))) return rc;
        HIP_TRY(hipMemcpyAsync(out, h->s_out, n * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        drain.armed = false;
        return DPTM_OK;
    }
    if ((rc = ensure_pipeline(h))) return rc;
    // Two staging slots.  Per chunk: the host gathers the caller's (pageable) rows into a pinned block with a few
    // threads, the copy stream moves that block over PCIe, the handle's stream scores it.  The gather of chunk i+1
    // and its transfer run while chunk i is being scored; probabilities come back through pinned memory.
    const uint8_t* tk[3] = {tok21, tok33, tok51};
    const size_t pin_tok[3] = {kPinTok21, kPinTok33, kPinTok51};
    uint8_t* const* s_tok[2] = {h->s_tok, h->s2_tok};
    float* const s_strct[2] = {h->s_strct, h->s2_strct};
    float* const s_esm[2] = {h->s_esm, h->s2_esm};
    float* const s_out[2] = {h->s_out, h->s2_out};
    auto retire = [&](int slot, int64_t c0, int64_t m) -> int {
        HIP_TRY(hipEventSynchronize(h->ev_scored[slot]));
        std::memcpy(out + c0 * 2, h->pin_out[slot], (size_t)m * 2 * sizeof(float));
        return DPTM_OK;
    };
    // one-operand half-precision modes: the ESM rows cross PCIe as the mode's 16-bit values (converted while they are
    // gathered), half the bytes; the split mode and fp32 need the fp32 rows
    const bool esm_half = h->precision == kModeBF16 || h->precision == kModeF16;
    const size_t esm_elem = esm_half ? sizeof(uint16_t) : sizeof(float);
    int64_t prev_c0 = 0, prev_m = 0;
    int64_t i = 0;
    for (int64_t c0 = 0; c0 < n; c0 += kChunkSites, ++i) {
        const int64_t m = n - c0 < kChunkSites ? n - c0 : kChunkSites;
        const int slot = (int)(i & 1);
        uint8_t* pin = h->pin_in[slot];
        if (i >= 2) HIP_TRY(hipEventSynchronize(h->ev_copied[slot]));   // the pinned block has crossed PCIe
        if (rows16)
            parallel_copy(pin + kPinEsm, esm16 + c0 * (int64_t)kEsm, (size_t)m * kEsm * sizeof(uint16_t));
        else if (esm_half)
            parallel_convert(reinterpret_cast<uint16_t*>(pin + kPinEsm), esm + c0 * (int64_t)kEsm, m, h->precision);
        else
            parallel_copy(pin + kPinEsm, esm + c0 * (int64_t)kEsm, (size_t)m * kEsm * sizeof(float));
        std::memcpy(pin + kPinStruct, strct + c0 * kStruct, (size_t)m * kStruct * sizeof(float));
        for (int b = 0; b < 3; ++b) std::memcpy(pin + pin_tok[b], tk[b] + c0 * kWin[b], (size_t)m * kWin[b]);
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_scored[slot], 0));   // device slot free again
        HIP_TRY(hipMemcpyAsync(s_esm[slot], pin + kPinEsm, (size_t)m * kEsm * esm_elem, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipMemcpyAsync(s_strct[slot], pin + kPinStruct, (size_t)m * kStruct * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
        for (int b = 0; b < 3; ++b)
            HIP_TRY(hipMemcpyAsync(s_tok[slot][b], pin + pin_tok[b], (size_t)m * kWin[b], hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipEventRecord(h->ev_copied[slot], h->copy_stream));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
        const uint8_t* tok[3] = {s_tok[slot][0], s_tok[slot][1], s_tok[slot][2]};
        rc = score_chunk(h, h->ws, tok, s_strct[slot], s_esm[slot], kEsm, nullptr, m, s_out[slot], h->stream, esm_half);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(h->pin_out[slot], s_out[slot], m * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(h->ev_scored[slot], h->stream));
        if (i >= 1 && (rc = retire(slot ^ 1, prev_c0, prev_m))) return rc;
        prev_c0 = c0;
        prev_m = m;
    }
    if ((rc = retire((int)((i - 1) & 1), prev_c0, prev_m))) return rc;
    drain.armed = false;
    return DPTM_OK;
}

int dptm_score(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
               const float* strct, const float* esm, int64_t n, float* out) try {
    return score_host_rows(h, tok21, tok33, tok51, strct, esm, false, n, out);
} catch (...) {
    return on_exception("dptm_score");
}

int dptm_score_rows16(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                      const float* strct, const uint16_t* esm16, int64_t n, float* out) try {
    return score_host_rows(h, tok21, tok33, tok51, strct, esm16, true, n, out);
} catch (...) {
    return on_exception("dptm_score_rows16");
}

int dptm_score_protein(dptm_handle* h, const uint8_t* codes, int64_t L, const float* esm_table,
                       int64_t esm_rows, const int32_t* positions, const float* strct, int64_t n,
                       float* out) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_protein: NULL handle");
    if (n < 0 || L < 0 || esm_rows < 0) return fail(DPTM_EINVAL, "dptm_score_protein: negative size");
    if (n == 0) return DPTM_OK;
    if (!codes || !positions || !strct || !out || (esm_rows > 0 && !esm_table))
        return fail(DPTM_EINVAL, "dptm_score_protein: NULL buffer");
    for (int64_t i = 0; i < n; ++i)
        if (positions[i] < 1 || positions[i] > L) {
            char msg[128];
            std::snprintf(msg, sizeof msg, "dptm_score_protein: position %d (index %lld) outside 1..%lld",
                          positions[i], (long long)i, (long long)L);
            return fail(DPTM_EINVAL, msg);
        }
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_host_staging(h);
    if (rc) return rc;
    if (L > h->s_codes_cap) {
        if (h->s_codes) (void)hipFree(h->s_codes);
        h->s_codes = nullptr;
        h->s_codes_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_codes), (size_t)L));
        h->s_codes_cap = L;
    }
    if (esm_rows > h->s_table_cap) {
        if (h->s_table) (void)hipFree(h->s_table);
        h->s_table = nullptr;
        h->s_table_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->s_table), (size_t)esm_rows * kEsm * sizeof(float)));
        h->s_table_cap = esm_rows;
    }
    DrainOnError drain{h};
    HIP_TRY(hipMemcpyAsync(h->s_codes, codes, (size_t)L, hipMemcpyHostToDevice, h->stream));
    if (esm_rows > 0)
        HIP_TRY(hipMemcpyAsync(h->s_table, esm_table, (size_t)esm_rows * kEsm * sizeof(float), hipMemcpyHostToDevice, h->stream));
    for (int64_t c0 = 0; c0 < n; c0 += kChunkSites) {
        const int64_t m = n - c0 < kChunkSites ? n - c0 : kChunkSites;
        HIP_TRY(hipMemcpyAsync(h->s_pos, positions + c0, m * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_strct, strct + c0 * kStruct, m * kStruct * sizeof(float), hipMemcpyHostToDevice, h->stream));
        rc = dptm_score_protein_device(h, h->s_codes, L, h->s_table, esm_rows, h->s_pos, h->s_strct, m,
                                       h->s_out, h->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + c0 * 2, h->s_out, m * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    drain.armed = false;
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_score_protein");
}

int dptm_assemble_proteins_device(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets,
                                  const int64_t* esm_offsets, const int32_t* positions, const int32_t* site_protein,
                                  int64_t n, uint8_t* tok21, uint8_t* tok33, uint8_t* tok51, int32_t* esm_row,
                                  void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_assemble_proteins_device: NULL handle");
    if (n < 0) return fail(DPTM_EINVAL, "dptm_assemble_proteins_device: n < 0");
    if (n == 0) return DPTM_OK;
    if (!codes || !seq_offsets || !esm_offsets || !positions || !site_protein || !tok21 || !tok33 || !tok51 || !esm_row)
        return fail(DPTM_EINVAL, "dptm_assemble_proteins_device: NULL buffer");
    if (misaligned(seq_offsets, 8) || misaligned(esm_offsets, 8) || misaligned(positions, 4) || misaligned(site_protein, 4) ||
        misaligned(esm_row, 4))
        return fail(DPTM_EINVAL, "dptm_assemble_proteins_device: offsets must be 8-byte aligned, positions / site_protein / esm_row 4-byte");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    AssembleArgs a{};
    a.codes = codes;
    a.seq_off = seq_offsets;
    a.esm_off = esm_offsets;
    a.site_protein = site_protein;
    a.positions = positions;
    a.n = n;
    a.t21 = tok21;
    a.t33 = tok33;
    a.t51 = tok51;
    a.esm_row = esm_row;
    Prof p(h, st, DPTM_PROF_ASSEMBLE);
    HIP_TRY(launch_assemble(a, st));
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_assemble_proteins_device");
}

int dptm_score_proteins_device(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets,
                               const float* esm_tables, const int64_t* esm_offsets, int64_t n_proteins,
                               const int32_t* positions, const int32_t* site_protein, const float* strct, int64_t n,
                               float* out, void* stream) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_proteins_device: NULL handle");
    if (n < 0 || n_proteins < 0) return fail(DPTM_EINVAL, "dptm_score_proteins_device: negative size");
    if (n == 0) return DPTM_OK;
    if (n_proteins == 0) return fail(DPTM_EINVAL, "dptm_score_proteins_device: sites but no proteins");
    if (!codes || !seq_offsets || !esm_offsets || !positions || !site_protein || !strct || !out)
        return fail(DPTM_EINVAL, "dptm_score_proteins_device: NULL buffer");
    if (misaligned(strct, 16) || (esm_tables && misaligned(esm_tables, 16)) || misaligned(out, 8))
        return fail(DPTM_EINVAL, "dptm_score_proteins_device: struct / esm tables must be 16-byte aligned, out 8-byte");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    // esm_tables == NULL: the caller states that no protein has a table (all esm_offsets equal), so no row is read
    const float* table = esm_tables ? esm_tables : h->d_zero_row;
    if (int rc = ensure_workspace(h, std::min(n, kPassSites))) return rc;
    for (int64_t c0 = 0; c0 < n; c0 += kPassSites) {
        const int64_t m = n - c0 < kPassSites ? n - c0 : kPassSites;
        int rc = dptm_assemble_proteins_device(h, codes, seq_offsets, esm_offsets, positions + c0, site_protein + c0, m,
                                               h->d_tok[0], h->d_tok[1], h->d_tok[2], h->d_esm_row, st);
        if (rc) return rc;
        const uint8_t* tok[3] = {h->d_tok[0], h->d_tok[1], h->d_tok[2]};
        rc = score_chunk(h, h->ws, tok, strct + c0 * kStruct, table, kEsm, h->d_esm_row, m, out + c0 * 2, st);
        if (rc) return rc;
    }
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_score_proteins_device");
}

int dptm_score_proteins(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets, const float* esm_tables,
                        const int64_t* esm_offsets, int64_t n_proteins, const int32_t* positions,
                        const int32_t* site_protein, const float* strct, int64_t n, float* out) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_score_proteins: NULL handle");
    if (n < 0 || n_proteins < 0) return fail(DPTM_EINVAL, "dptm_score_proteins: negative size");
    if (n == 0) return DPTM_OK;
    if (n_proteins == 0) return fail(DPTM_EINVAL, "dptm_score_proteins: sites but no proteins");
    if (!codes || !seq_offsets || !esm_offsets || !positions || !site_protein || !strct || !out)
        return fail(DPTM_EINVAL, "dptm_score_proteins: NULL buffer");
    int rc = check_protein_tables("dptm_score_proteins", seq_offsets, esm_offsets, n_proteins, positions, site_protein, n);
    if (rc) return rc;
    if (!esm_tables && esm_offsets[n_proteins] != 0)
        return fail(DPTM_EINVAL, "dptm_score_proteins: esm_offsets name table rows but esm_tables is NULL");
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = ensure_host_staging(h))) return rc;
    const int64_t residues = seq_offsets[n_proteins];
    if ((rc = ensure_capacity(&h->s_codes, &h->s_codes_cap, std::max<int64_t>(residues, 1)))) return rc;
    if ((rc = ensure_capacity(&h->s_seq_off, &h->s_seq_off_cap, n_proteins + 1))) return rc;
    if ((rc = ensure_capacity(&h->s_esm_off, &h->s_esm_off_cap, n_proteins + 1))) return rc;
    if ((rc = ensure_capacity(&h->s_site_pos, &h->s_site_pos_cap, n))) return rc;
    if ((rc = ensure_capacity(&h->s_site_prot, &h->s_site_prot_cap, n))) return rc;
    DrainOnError drain{h};
    // Everything but the ESM rows is small and goes up whole: residue codes (1 B per residue), offsets, positions.
    // ESM rows cross PCIe gathered -- one 5 KB row per SITE, not per residue -- and the windows are assembled on the device.
    HIP_TRY(hipMemcpyAsync(h->s_codes, codes, (size_t)residues, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_seq_off, seq_offsets, (size_t)(n_proteins + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_esm_off, esm_offsets, (size_t)(n_proteins + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_site_pos, positions, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_site_prot, site_protein, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    auto assemble = [&](int64_t c0, int64_t m) {
        return dptm_assemble_proteins_device(h, h->s_codes, h->s_seq_off, h->s_esm_off, h->s_site_pos + c0,
                                             h->s_site_prot + c0, m, h->d_tok[0], h->d_tok[1], h->d_tok[2], h->d_esm_row,
                                             h->stream);
    };
    const uint8_t* tok[3] = {h->d_tok[0], h->d_tok[1], h->d_tok[2]};
    if (n <= kChunkSites) {
        std::vector<float> rows((size_t)n * kEsm);
        gather_esm_rows(rows.data(), esm_tables, esm_offsets, positions, site_protein, 0, n);
        HIP_TRY(hipMemcpyAsync(h->s_esm, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_strct, strct, (size_t)n * kStruct * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if ((rc = assemble(0, n))) return rc;
        if ((rc = score_chunk(h, h->ws, tok, h->s_strct, h->s_esm, kEsm, nullptr, n, h->s_out, h->stream))) return rc;
        HIP_TRY(hipMemcpyAsync(out, h->s_out, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        drain.armed = false;
        return DPTM_OK;
    }
    if ((rc = ensure_pipeline(h))) return rc;
    // the two-slot pipeline of dptm_score: gather of chunk i+1 into pinned memory and its transfer overlap the
    // scoring of chunk i (the token windows of a chunk are produced on the scoring stream, in order)
    float* const s_strct[2] = {h->s_strct, h->s2_strct};
    float* const s_esm[2] = {h->s_esm, h->s2_esm};
    float* const s_out[2] = {h->s_out, h->s2_out};
    auto retire = [&](int slot, int64_t c0, int64_t m) -> int {
        HIP_TRY(hipEventSynchronize(h->ev_scored[slot]));
        std::memcpy(out + c0 * 2, h->pin_out[slot], (size_t)m * 2 * sizeof(float));
        return DPTM_OK;
    };
    // one-operand half-precision modes: the gathered rows cross PCIe as the mode's 16-bit values
    const bool esm_half = h->precision == kModeBF16 || h->precision == kModeF16;
    const size_t esm_elem = esm_half ? sizeof(uint16_t) : sizeof(float);
    int64_t prev_c0 = 0, prev_m = 0, i = 0;
    for (int64_t c0 = 0; c0 < n; c0 += kChunkSites, ++i) {
        const int64_t m = n - c0 < kChunkSites ? n - c0 : kChunkSites;
        const int slot = (int)(i & 1);
        uint8_t* pin = h->pin_in[slot];
        if (i >= 2) HIP_TRY(hipEventSynchronize(h->ev_copied[slot]));   // the pinned block has crossed PCIe
        gather_esm_rows(pin + kPinEsm, esm_tables, esm_offsets, positions, site_protein, c0, c0 + m, esm_half ? h->precision : 0);
        std::memcpy(pin + kPinStruct, strct + c0 * kStruct, (size_t)m * kStruct * sizeof(float));
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_scored[slot], 0));   // device slot free again
        HIP_TRY(hipMemcpyAsync(s_esm[slot], pin + kPinEsm, (size_t)m * kEsm * esm_elem, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipMemcpyAsync(s_strct[slot], pin + kPinStruct, (size_t)m * kStruct * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(hipEventRecord(h->ev_copied[slot], h->copy_stream));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
        if ((rc = assemble(c0, m))) return rc;
        if ((rc = score_chunk(h, h->ws, tok, s_strct[slot], s_esm[slot], kEsm, nullptr, m, s_out[slot], h->stream, esm_half))) return rc;
        HIP_TRY(hipMemcpyAsync(h->pin_out[slot], s_out[slot], (size_t)m * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(h->ev_scored[slot], h->stream));
        if (i >= 1 && (rc = retire(slot ^ 1, prev_c0, prev_m))) return rc;
        prev_c0 = c0;
        prev_m = m;
    }
    if ((rc = retire((int)((i - 1) & 1), prev_c0, prev_m))) return rc;
    drain.armed = false;
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_score_proteins");
}

int dptm_profile_begin(dptm_handle* h) try {
    if (!h) return fail(DPTM_EINVAL, "dptm_profile_begin: NULL handle");
    for (auto& r : h->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    h->prof.clear();
    h->prof_armed = true;
    return DPTM_OK;
} catch (...) {
    return on_exception("dptm_profile_begin");
}

int dptm_profile_end(dptm_handle* h, float* ms, int64_t* launches) try {
    if (!h || !ms || !launches) return fail(DPTM_EINVAL, "dptm_profile_end: NULL argument");
    h->prof_armed = false;
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < DPTM_PROF_COUNT; ++i) {
        ms[i] = 0.f;
        launches[i] = 0;
    }
    int rc = DPTM_OK;
    for (auto& r : h->prof) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
        if (e != hipSuccess && rc == DPTM_OK)
            rc = fail(DPTM_EHIP, std::string("dptm_profile_end: ") + hipGetErrorString(e));
        ms[r.cls] += t;
        launches[r.cls] += 1;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    h->prof.clear();
    return rc;
} catch (...) {
    return on_exception("dptm_profile_end");
}

}  // extern "C"
