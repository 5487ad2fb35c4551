// Training step behind the C ABI (dptm_trainer_* in include/dptm.h): parameter layout,
// activation arena, the forward/backward launch sequence of the graph of
// pred/train_PTM/model.py:143-210 in training mode, the loss of trainer.py:36-54 and the
// AdamW update of trainer.py:305-306.  Host code only; kernels are in dptm_train_kernels.hip.
#include <hip/hip_runtime.h>

#include <memory>
#include <new>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dptm.h"
#include "dptm_layout.h"
#include "dptm_train_device.h"
#include "dptm_train_kernels.h"

using namespace dptm;

namespace {

thread_local std::string g_terr;
int tfail(int code, const std::string& msg) {
    g_terr = msg;
    return code;
}
#define T_TRY(expr)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return tfail(DPTM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int kAlign = 64;  // every tensor starts on a 256-byte boundary of the flat buffers

struct Tn {          // one tensor of the canonical order
    int64_t off;     // offset in the aligned flat buffers (floats)
    int64_t size;
    int64_t canon;   // offset in the canonical blob
    bool trainable;
    bool l2;
};

struct BnP { int gamma, beta, mean, var, slot; };   // slot: index of the layer's column-sum buffers
struct TbP { int wq, bq, wk, bk, wv, bv, wo, bo, f1k, f1b, f2k, f2b, ln1g, ln1b, ln2g, ln2b; };
struct BranchP { int emb, pos, sck, scb, c3ak, c3ab; BnP bn1; int c3bk, c3bb; BnP bn2; TbP tb[2]; int poolk, poolb; };
struct DenseP { int k, b; BnP bn; };

struct PMap {
    std::vector<Tn> t;
    int64_t total = 0, canon_total = 0;
    BranchP br[3];
    DenseP strct, esm1, esm2, t1, t2, head;
    int add(int64_t size, bool trainable = true, bool l2 = false) {
        Tn x{total, size, canon_total, trainable, l2};
        t.push_back(x);
        total += (size + kAlign - 1) / kAlign * kAlign;
        canon_total += size;
        return (int)t.size() - 1;
    }
    int n_bn = 0;
    BnP add_bn(int n) { BnP b; b.gamma = add(n); b.beta = add(n); b.mean = add(n, false); b.var = add(n, false); b.slot = n_bn++; return b; }
    void build() {
        const int W[3] = {21, 33, 51};
        for (int b = 0; b < 3; ++b) {
            BranchP& p = br[b];
            p.emb = add(kCodes * kD); p.pos = add((int64_t)W[b] * kD);
            p.sck = add(kD * kD, true, true); p.scb = add(kD);
            p.c3ak = add(3 * kD * kD, true, true); p.c3ab = add(kD); p.bn1 = add_bn(kD);
            p.c3bk = add(3 * kD * kD, true, true); p.c3bb = add(kD); p.bn2 = add_bn(kD);
            for (int i = 0; i < 2; ++i) {
                TbP& q = p.tb[i];
                q.wq = add(kD * kD); q.bq = add(kD); q.wk = add(kD * kD); q.bk = add(kD);
                q.wv = add(kD * kD); q.bv = add(kD); q.wo = add(kD * kD); q.bo = add(kD);
                q.f1k = add(kD * kFF); q.f1b = add(kFF); q.f2k = add(kFF * kD); q.f2b = add(kD);
                q.ln1g = add(kD); q.ln1b = add(kD); q.ln2g = add(kD); q.ln2b = add(kD);
            }
            p.poolk = add(kD); p.poolb = add(1);
        }
        // column-sum slots by stage, not by branch: the three branches' sums of one stage are then one contiguous range,
        // which synchronised BatchNorm sums over the replicas with ONE collective
        for (int b = 0; b < 3; ++b) {
            br[b].bn1.slot = b;
            br[b].bn2.slot = 3 + b;
        }
        strct.k = add(kStruct * kStructUnits); strct.b = add(kStructUnits); strct.bn = add_bn(kStructUnits);
        esm1.k = add((int64_t)kEsm * kEsm1); esm1.b = add(kEsm1); esm1.bn = add_bn(kEsm1);
        esm2.k = add(kEsm1 * kEsm2); esm2.b = add(kEsm2);
        t1.k = add(kConcat * kT1, true, true); t1.b = add(kT1); t1.bn = add_bn(kT1);
        t2.k = add(kT1 * kT2, true, true); t2.b = add(kT2); t2.bn = add_bn(kT2);
        head.k = add(kT2 * 2); head.b = add(2);
    }
};

// bump allocator over one device block
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    float* f(size_t n) {
        size_t bytes = (n * sizeof(float) + 255) / 256 * 256;
        if (used + bytes > cap) return nullptr;
        float* p = reinterpret_cast<float*>(base + used);
        used += bytes;
        return p;
    }
};

struct BranchAct {
    int wlen;
    ConvSave cs;                       // convolution block: shortcut output and the two pre-activations
    TbSave sv[2];                      // what the fused TransformerBlock kernels keep for the backward pass
    f32x2_tr* pool;                    // per row: tanh score and pooling weight
};

}  // namespace

struct dptm_trainer {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t max_batch = 0;
    PMap pm;
    float *W = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;  // weights, grads (+ batch BN stats), Adam moments
    float* P = nullptr;                   // weight images in MFMA fragment order, rebuilt from W every step
    PackJob* pack_jobs = nullptr;         // device table of the images to build
    int n_pack_jobs = 0;
    int64_t img_fwd[3], img_bwd[3];       // per branch: packed forward image (dptm_layout.h) and transposed images, floats into P
    uint8_t* trainable = nullptr;
    Arena arena;
    BranchAct ba[3];
    // per-site tensors
    float *sa, *sy, *smi, *ea, *ey, *emi, *e1, *e2a, *concat, *t1a, *t1y, *t1mi, *t1z, *t2a, *t2y, *t2mi, *t2z, *proba;
    float* d_loss = nullptr;
    // column sums of every BatchNorm, forward [slot] and backward [kBnLayers + slot], 1024 floats each: zeroed once per step
    float* bn_slots = nullptr;
    // synchronised BatchNorm: the backward sums of the WHOLE batch, [kBnLayers][1024] like the second half of bn_slots
    // (which keeps this replica's share for the parameter gradients)
    float* bn_all = nullptr;
    hipEvent_t ev_sync[4][4] = {};        // stage joins of the branch streams and the per-site lane around a packed collective:
                                          // [point][W=21 | W=33 | main stream after the sum | per-site lane]
    // Scratch of one concurrent lane of the step: the three sequence branches run on side streams next to
    // the per-site layers, so each owns its gradient scratch ([rows][64] / [rows][256]) and BatchNorm sums.
    struct Scratch {
        f32x4_tr *dx1 = nullptr, *dz1 = nullptr;   // chain-tiled gradients handed from one backward kernel to the next
        float* part = nullptr;            // partial parameter gradients of the fused backward kernel, one slab per workgroup
        float* row_part = nullptr;        // the same for the row-wise launches of a large batch: [2 blocks][row_parts][kRowPartStride]
        size_t row_parts = 0;             // (none when max_batch never reaches the eight-wavefront tiles)
    };
    Scratch sc[4];                        // [0..2] sequence branches, [3] per-site layers (BatchNorm sums only)
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    int lanes_independent = 0;            // how many of the four streams the queue probe found on queues of their own
    int64_t loss_slot = -1;               // a padding float of the gradient buffer (dptm_trainer_stage_loss)
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipEvent_t ev_caller[2] = {nullptr, nullptr};   // the caller's stream hands over to / takes over from the step's streams
    float *dconcat, *d512a, *d512b, *d256a, *d256b, *d128;
    int64_t step = 0;
    int64_t last_n = 0;
    // staging of dptm_trainer_forward_backward_rows (one allocation, made by its first call): max_batch rows of
    // [esm 1280 f | struct 8 f | label i32 | codes 21 + 33 + 51]
    void* stage = nullptr;
    unsigned* d_step = nullptr;           // per-step scalars (StepParam)
    int l2_slot = 0;                      // which of the two l2-value slots the step in hand uses
    bool pooled_grad_in_kernel = false;   // this step's backward tiles form d(pooled rows) themselves (see TbBwdArgs::up_da)
    // captured graph of one forward_backward: replayed while the call's shape, buffers and options repeat
    struct GraphKey {
        int64_t n = -1;
        const void* ptr[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        float rate[4] = {0, 0, 0, 0};
        float l2 = 0, l2_scale = 0;
        hipStream_t st = nullptr;
        bool operator==(const GraphKey& o) const {
            return n == o.n && std::memcmp(ptr, o.ptr, sizeof ptr) == 0 && std::memcmp(rate, o.rate, sizeof rate) == 0 &&
                   l2 == o.l2 && l2_scale == o.l2_scale && st == o.st;
        }
    };
    GraphKey gkey;
    int gkey_hits = 0;                    // consecutive eager calls with gkey
    hipGraphExec_t gexec = nullptr;
    // synchronised BatchNorm: sums the column sums of every BatchNorm over the replicas (NULL: per-replica statistics)
    dptm_allreduce_fn ar_fn = nullptr;
    void* ar_user = nullptr;
    int64_t n_local = 1, n_global = 1;    // sites of the current call here / over all replicas
    bool serial = false;                  // DPTM_TRAIN_SERIAL=1: branches on the caller's stream too (per-kernel profiling)
    bool use_graph = false;               // opt-in (DPTM_TRAIN_GRAPH=1): replay measures slower than eager multi-stream issue
    // per-kernel clocks of one step (dptm_trainer_profile_begin / _end): event pairs around the six launches of
    // every sequence branch, slot = branch * 6 + {conv_a_fwd, conv_b_fwd, tb_fwd, tb_bwd, conv_b_bwd, conv_a_bwd}
    bool prof_on = false, prof_serial_before = false;
    hipEvent_t prof_ev[DPTM_TRAIN_PROFILE_SLOTS][2] = {};
    bool prof_used[DPTM_TRAIN_PROFILE_SLOTS] = {};
};

namespace {

const int kW[3] = {21, 33, 51};
constexpr int kBnLayers = 10;   // 2 per sequence branch + struct, ESM, two trunk layers

// workgroups of the row-wise backward launches that a batch of up to B sites can start: none unless the batch reaches
// the eight-wavefront tiles (the only ones that take the five-launch form), two 128-row tiles per 256-row tile
size_t row_parts_for(int64_t B, int wlen) {
    const TileCfg tc = t_tile_cfg(B, wlen);
    if (tc.NW != 8) return 0;
    const int64_t tiles = 2 * tc.ntiles;
    return (size_t)(tiles < kRowMaxParts ? tiles : kRowMaxParts);
}

int alloc_acts(dptm_trainer* t) {
    const int64_t B = t->max_batch;
    Arena& A = t->arena;
    auto need = [&](float*& p, size_t n) -> bool { p = A.f(n); return p != nullptr; };
    bool ok = true;
    for (int b = 0; b < 3; ++b) {
        BranchAct& a = t->ba[b];
        a.wlen = kW[b];
        // chain-tiled arrays cover the padded rows of the largest tiling a batch up to max_batch can take
        const size_t RP = (size_t)t_tile_rows_max(B, kW[b]);
        auto need4 = [&](f32x4_tr*& p4, size_t rows) -> bool { float* f = nullptr; const bool k = need(f, rows * 64); p4 = reinterpret_cast<f32x4_tr*>(f); return k; };
        ok &= need4(a.cs.s, RP) && need4(a.cs.a1, RP) && need4(a.cs.a2, RP);
        for (int i = 0; i < 2; ++i) {
            TbSave& q = a.sv[i];
            float *st = nullptr, *iv = nullptr;
            ok &= need4(q.q, RP) && need4(q.k, RP) && need4(q.v, RP) && need4(q.o, RP) && need4(q.xh1, RP) && need4(q.xh2, RP) &&
                  need(st, RP * 2 * 4) && need(iv, RP * 2 * 2);
            q.stat = reinterpret_cast<f32x4_tr*>(st);
            q.inv = reinterpret_cast<f32x2_tr*>(iv);
        }
        {
            float* pl = nullptr;
            ok &= need(pl, RP * 2 * 2);
            a.pool = reinterpret_cast<f32x2_tr*>(pl);
        }
    }
    ok &= need(t->sa, B * 128) && need(t->sy, B * 128) && need(t->smi, 256) && need(t->ea, B * 512) && need(t->ey, B * 512) &&
          need(t->emi, 1024) && need(t->e1, B * 512) && need(t->e2a, B * 256) && need(t->concat, B * 576) &&
          need(t->t1a, B * 512) && need(t->t1y, B * 512) && need(t->t1mi, 1024) && need(t->t1z, B * 512) &&
          need(t->t2a, B * 256) && need(t->t2y, B * 256) && need(t->t2mi, 512) && need(t->t2z, B * 256) &&
          need(t->proba, B * 2) && need(t->d_loss, 64) && need(t->bn_slots, 2 * kBnLayers * 1024) &&
          need(t->bn_all, kBnLayers * 1024);
    ok = ok && t->bn_slots == t->d_loss + 64;      // one fill clears both every step
    for (int b = 0; b < 4; ++b) {
        dptm_trainer::Scratch& s = t->sc[b];
        if (b == 3) break;
        const size_t RP = (size_t)t_tile_rows_max(B, kW[b]);
        float *d1 = nullptr, *d2 = nullptr;
        ok &= need(d1, RP * 64) && need(d2, RP * 64) && need(s.part, (size_t)kTbMaxParts * kTbPartStride);
        s.row_parts = row_parts_for(B, kW[b]);
        if (s.row_parts) ok &= need(s.row_part, 2 * s.row_parts * kRowPartStride);
        s.dx1 = reinterpret_cast<f32x4_tr*>(d1);
        s.dz1 = reinterpret_cast<f32x4_tr*>(d2);
    }
    ok &= need(t->dconcat, B * 576) && need(t->d512a, B * 512) && need(t->d512b, B * 512) && need(t->d256a, B * 256) &&
          need(t->d256b, B * 256) && need(t->d128, B * 128);
    return ok ? DPTM_OK : tfail(DPTM_ENOMEM, "dptm_trainer_create: activation arena too small");
}

size_t arena_bytes(int64_t B) {
    size_t f = 0;
    for (int b = 0; b < 3; ++b) {
        const size_t RP = (size_t)t_tile_rows_max(B, kW[b]);
        f += RP * 64 * 3;                                           // conv block activations
        f += 2 * (RP * 64 * 6 + RP * 8 + RP * 4) + RP * 4;          // saved TransformerBlock activations, pooling
        f += RP * 64 * 2 + (size_t)kTbMaxParts * kTbPartStride;     // gradient hand-over, partial gradients
        f += 2 * row_parts_for(B, kW[b]) * kRowPartStride;          // partial gradients of the row-wise launches
    }
    f += (size_t)B * (128 * 2 + 512 * 3 + 256 + 576 + 512 * 3 + 256 * 3 + 2);
    f += (4 + 3 * 10) * 1024 + 64 + (size_t)B * (576 + 1024 + 512 + 128);
    return f * sizeof(float) + (size_t)640 * 256 + (1 << 20);   // + per-allocation rounding slack
}

struct Ctx {
    dptm_trainer* t;
    hipStream_t st;
    float* W;
    float* G;
    dptm_trainer::Scratch* s;
    float* w(int id) const { return W + t->pm.t[id].off; }
    float* g(int id) const { return G + t->pm.t[id].off; }
};

// rows of the whole (all-replica) batch that correspond to `rows` local rows
int64_t global_rows(const dptm_trainer* t, int64_t rows) { return t->ar_fn ? rows / t->n_local * t->n_global : rows; }

int replica_sum(const Ctx& c, float* buf, int64_t count) {
    if (!c.t->ar_fn) return DPTM_OK;
    if (c.t->ar_fn(c.t->ar_user, buf, count, c.st) != 0)
        return tfail(DPTM_EHIP, "dptm_trainer_forward_backward: the all-reduce callback reported an error");
    return DPTM_OK;
}

// Which half of a BatchNorm call to run.  kBnSums / kBnApply: the column sums only / the layer itself from sums that the
// caller has summed over the replicas in between (synchronised BatchNorm packs several layers into one collective).
enum BnPart { kBnWhole, kBnSums, kBnApply };

// BatchNorm in training mode on y [rows][cols] -> z (may live inside a wider buffer)
int bn_train_fwd(const Ctx& c, const float* y, float* z, int64_t ldz, int zoff, int64_t rows, int cols, const BnP& bn, float* mi,
                 BnPart part = kBnWhole) {
    float* stats = c.t->bn_slots + (size_t)bn.slot * 1024;
    if (part != kBnApply) T_TRY(t_colsum2(y, nullptr, rows, cols, stats, c.st));
    if (part == kBnWhole) {
        int rc = replica_sum(c, stats, 2 * cols);
        if (rc) return rc;
    }
    // the batch statistics ride in the (otherwise unused) moving-stat slots of the gradient buffer
    if (part != kBnSums)
        T_TRY(t_bn_fwd(y, z, rows, cols, stats, global_rows(c.t, rows), mi, c.g(bn.mean), c.g(bn.var), c.w(bn.gamma), c.w(bn.beta), ldz,
                       zoff, c.st));
    return DPTM_OK;
}

// Backward of the same layer: dz (inside a wider buffer) -> dy, gamma / beta gradients accumulated.  The
// input gradient needs the sums over the WHOLE batch, the parameter gradients this replica's share only
// (the gradient all-reduce adds the shares up).
int bn_train_bwd(const Ctx& c, const float* dz, int64_t lddz, int dzoff, const float* y, float* dy, int64_t rows, int cols,
                 const float* mi, const BnP& bn, BnPart part = kBnWhole) {
    float* sums = c.t->bn_slots + (size_t)(kBnLayers + bn.slot) * 1024;
    if (part != kBnApply) T_TRY(t_bn_bwd_reduce(dz, lddz, dzoff, y, rows, cols, mi, sums, c.st));
    if (part == kBnSums) return DPTM_OK;
    const float* all = sums;
    if (c.t->ar_fn) {
        float* whole = c.t->bn_all + (size_t)bn.slot * 1024;
        if (part == kBnWhole) {
            T_TRY(hipMemcpyAsync(whole, sums, 2 * cols * sizeof(float), hipMemcpyDeviceToDevice, c.st));
            int rc = replica_sum(c, whole, 2 * cols);
            if (rc) return rc;
        }
        all = whole;
    }
    T_TRY(t_bn_bwd_apply(dz, lddz, dzoff, y, dy, rows, cols, mi, c.w(bn.gamma), all, global_rows(c.t, rows), sums,
                         c.g(bn.gamma), c.g(bn.beta), c.st));
    return DPTM_OK;
}

GemmArgs ga(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* out, int64_t ldo, int col_off,
            int64_t m, int k, int cols, int shift = 0, int wlen = 0, int acc = 0) {
    GemmArgs a;
    a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo; a.col_off = col_off;
    a.m = m; a.k = k; a.cols = cols; a.shift = shift; a.wlen = wlen; a.accumulate = acc;
    a.taps = 1; a.tap_stride = 0; a.tap_dir = 1; a.x_tap_stride = 0;
    a.residual = nullptr; a.ldr = 0; a.act_out = nullptr; a.lda = 0;
    a.batch = 1; a.bx = a.bw = a.bb = a.bo = 0;
    return a;
}
GemmTnArgs gt(const float* x, int64_t ldx, int xoff, const float* dy, int64_t ldy, int dyoff, float* dw, int64_t ldw, float* db,
              int64_t m, int k, int n, int shift = 0, int wlen = 0) {
    GemmTnArgs a;
    a.x = x; a.ldx = ldx; a.x_col_off = xoff; a.dy = dy; a.ldy = ldy; a.dy_col_off = dyoff; a.dw = dw; a.ldw = ldw; a.db = db;
    a.m = m; a.k = k; a.n = n; a.shift = shift; a.wlen = wlen; a.rows_per_slab = 0;
    a.slabs = 1; a.batch = 1; a.bx = a.bdy = a.bdw = a.bdb = 0; a.bshift_dir = 0;
    return a;
}

struct Rates { float embed, conv, tb, dense; };

// While a profile is armed, brackets one launch with an event pair on the launch's own stream.
struct ProfScope {
    dptm_trainer* t;
    hipStream_t st;
    int slot;
    ProfScope(const Ctx& c, int slot_) : t(c.t), st(c.st), slot(slot_) {
        if (t->prof_on) (void)hipEventRecord(t->prof_ev[slot][0], st);
    }
    ~ProfScope() {
        if (t->prof_on) {
            (void)hipEventRecord(t->prof_ev[slot][1], st);
            t->prof_used[slot] = true;
        }
    }
};

// Tile counter of backward launch k (0: TransformerBlocks, 1: second convolution, 2: first convolution) of branch b.
// The counters sit in the unused part of the 64-float loss slot, which the step's first launch clears together with
// the BatchNorm column sums.
// A launch with a workgroup per tile needs none (nullptr: tiles by workgroup index).
unsigned* tile_queue(dptm_trainer* t, int b, int k, int64_t ntiles) {
    return ntiles > kTbMaxParts ? reinterpret_cast<unsigned*>(t->d_loss + 32) + b * 3 + k : nullptr;
}

// per-branch arguments shared by the four convolution-block kernels
ConvArgs conv_args(const Ctx& c, int b, const uint8_t* tok, int64_t B, const unsigned* seed, const Rates& rt, const TileCfg& tc) {
    dptm_trainer* t = c.t;
    BranchAct& a = t->ba[b];
    const BranchP& p = t->pm.br[b];
    const float inv_rows = 1.0f / (float)global_rows(t, B * a.wlen);
    ConvArgs f;
    f.P = reinterpret_cast<const f32x4_tr*>(t->P + t->img_fwd[b]);
    f.PT = reinterpret_cast<const f32x4_tr*>(t->P + t->img_bwd[b]);
    f.tok = tok;
    f.cs = a.cs;
    f.stats1 = t->bn_slots + (size_t)p.bn1.slot * 1024;
    f.stats2 = t->bn_slots + (size_t)p.bn2.slot * 1024;
    // the batch statistics ride in the (otherwise unused) moving-stat slots of the gradient buffer
    f.bn1 = BnFwd{f.stats1, inv_rows, c.w(p.bn1.gamma), c.w(p.bn1.beta), c.g(p.bn1.mean), c.g(p.bn1.var)};
    float* own1 = t->bn_slots + (size_t)(kBnLayers + p.bn1.slot) * 1024;
    float* own2 = t->bn_slots + (size_t)(kBnLayers + p.bn2.slot) * 1024;
    // with synchronised BatchNorm the sums of the whole batch live in a second buffer (filled by the all-reduce)
    const float* all1 = t->ar_fn ? t->bn_all + (size_t)p.bn1.slot * 1024 : own1;
    const float* all2 = t->ar_fn ? t->bn_all + (size_t)p.bn2.slot * 1024 : own2;
    f.bb1 = BnBwd{f.stats1, inv_rows, c.w(p.bn1.gamma), c.w(p.bn1.beta), all1, own1, c.g(p.bn1.gamma), c.g(p.bn1.beta)};
    f.bb2 = BnBwd{f.stats2, inv_rows, c.w(p.bn2.gamma), c.w(p.bn2.beta), all2, own2, c.g(p.bn2.gamma), c.g(p.bn2.beta)};
    f.bn1_sums = own1;
    f.dx1 = c.s->dx1;
    f.dz1 = c.s->dz1;
    f.part = c.s->part;
    f.queue_b = tile_queue(t, b, 1, tc.ntiles);
    f.queue_a = tile_queue(t, b, 2, tc.ntiles);
    f.n_sites = B;
    f.S = tc.S;
    f.ntiles = tc.ntiles;
    f.step = seed;
    f.stream_embed = (unsigned)(b * 8 + 0);
    f.thr_embed = dropout_threshold(rt.embed);
    f.scale_embed = rt.embed > 0.f ? 1.0f / (1.0f - rt.embed) : 1.0f;
    f.stream_conv = (unsigned)(b * 8 + 1);
    f.thr_conv = dropout_threshold(rt.conv);
    f.scale_conv = rt.conv > 0.f ? 1.0f / (1.0f - rt.conv) : 1.0f;
    return f;
}

BnFwd bn2_fwd(const Ctx& c, int b, int64_t B) {
    dptm_trainer* t = c.t;
    const BranchP& p = t->pm.br[b];
    return BnFwd{t->bn_slots + (size_t)p.bn2.slot * 1024, 1.0f / (float)global_rows(t, B * t->ba[b].wlen), c.w(p.bn2.gamma),
                 c.w(p.bn2.beta), c.g(p.bn2.mean), c.g(p.bn2.var)};
}

// BN2 + shortcut add, both TransformerBlocks and the attention pooling (model.py:138-141, 172-182): arguments of branch b
TbFwdArgs tb_fwd_args(const Ctx& c, int b, const ConvArgs& cf, int64_t B, const unsigned* seed, const Rates& rt, const TileCfg& tc) {
    dptm_trainer* t = c.t;
    BranchAct& a = t->ba[b];
    TbFwdArgs f;
    f.P = cf.P;
    f.cs = a.cs;
    f.bn2 = bn2_fwd(c, b, B);
    f.sv[0] = a.sv[0];
    f.sv[1] = a.sv[1];
    f.pool = a.pool;
    f.out = t->concat;
    f.ldo = kConcat;
    f.col_off = b * 64;
    f.n_sites = B;
    f.S = tc.S;
    f.ntiles = tc.ntiles;
    f.step = seed;
    f.stream_base = (unsigned)(b * 8 + 2);
    f.thr = dropout_threshold(rt.tb);
    f.scale = rt.tb > 0.f ? 1.0f / (1.0f - rt.tb) : 1.0f;
    return f;
}

// One sequence branch forward (model.py:157-182) in three launches; the cuts sit where a BatchNorm needs the
// statistics of the whole batch (and, with synchronised BatchNorm, of all replicas).  Launches first..last of the three
// (synchronised BatchNorm runs the branches stage by stage and sums the three branches' column sums in between).
int branch_fwd(const Ctx& c, int b, const uint8_t* tok, int64_t B, const unsigned* seed, const Rates& rt, int first = 0,
               int last = 2) {
    dptm_trainer* t = c.t;
    BranchAct& a = t->ba[b];
    const int wl = a.wlen;
    const TileCfg tc = t_tile_cfg(B, wl);
    const ConvArgs cf = conv_args(c, b, tok, B, seed, rt, tc);
    if (first <= 0 && last >= 0) {
        ProfScope ps(c, b * 6 + 0);
        T_TRY(t_conv_a_fwd(cf, wl, tc.NW, c.st));
    }
    if (first <= 1 && last >= 1) {
        ProfScope ps(c, b * 6 + 1);
        T_TRY(t_conv_b_fwd(cf, wl, tc.NW, c.st));
    }
    if (first <= 2 && last >= 2) {
        const TbFwdArgs f = tb_fwd_args(c, b, cf, B, seed, rt, tc);
        ProfScope ps(c, b * 6 + 2);
        T_TRY(t_tb_fwd(f, wl, tc.NW, c.st));
    }
    return DPTM_OK;
}

// pooling and both TransformerBlocks backward: d(concat columns of this branch) -> d(x1), BN2 column sums
TbBwdArgs tb_bwd_args(const Ctx& c, int b, const ConvArgs& cf, int64_t B, const unsigned* seed, const Rates& rt, const TileCfg& tc) {
    dptm_trainer* t = c.t;
    BranchAct& a = t->ba[b];
    const BranchP& p = t->pm.br[b];
    TbBwdArgs f;
    f.P = cf.P;
    f.PT = cf.PT;
    f.cs = a.cs;
    f.bn2 = bn2_fwd(c, b, B);
    f.bn2_sums = t->bn_slots + (size_t)(kBnLayers + p.bn2.slot) * 1024;
    f.sv[0] = a.sv[0];
    f.sv[1] = a.sv[1];
    f.pool = a.pool;
    f.dout = t->dconcat;
    f.ldd = kConcat;
    f.col_off = b * 64;
    f.dx1 = c.s->dx1;
    f.part = c.s->part;
    f.queue = tile_queue(t, b, 0, tc.ntiles);
    f.n_sites = B;
    f.S = tc.S;
    f.ntiles = tc.ntiles;
    f.step = seed;
    f.stream_base = (unsigned)(b * 8 + 2);
    f.thr = dropout_threshold(rt.tb);
    f.scale = rt.tb > 0.f ? 1.0f / (1.0f - rt.tb) : 1.0f;
    // the split form of large batches: second hand-over array (idle until the convolutions go backward), partial
    // buffers of the row-wise launches, five more tile counters behind the nine of tile_queue
    f.blk = 0;
    f.gin = nullptr;
    f.gout = nullptr;
    f.dtmp = c.s->dz1;
    f.row_part = c.s->row_part;
    f.row_parts = (int)c.s->row_parts;
    f.queues = reinterpret_cast<unsigned*>(t->d_loss + 32) + 9 + b * 5;
    // one-site tiles of a small batch take the gradient of their pooled rows from the first trunk layer themselves
    f.up_da = nullptr;
    f.up_w = nullptr;
    f.up_n = f.up_cols = 0;
    f.up_stream = f.up_thr = 0u;
    f.up_scale = 1.0f;
    if (t->pooled_grad_in_kernel && t_pair_tile(tc.S, tc.NW)) {
        f.up_da = t->d512b;
        f.up_w = c.w(t->pm.t1.k);
        f.up_n = 512;
        f.up_cols = kConcat;
        f.up_stream = 24 + 2;
        f.up_thr = dropout_threshold(rt.dense);
        f.up_scale = rt.dense > 0.f ? 1.0f / (1.0f - rt.dense) : 1.0f;
    }
    return f;
}

TbGrads tb_grads(const Ctx& c, int b) {
    const BranchP& p = c.t->pm.br[b];
    TbGrads g;
    for (int i = 0; i < 2; ++i) {
        const TbP& w = p.tb[i];
        g.wq[i] = c.g(w.wq); g.bq[i] = c.g(w.bq); g.wk[i] = c.g(w.wk); g.bk[i] = c.g(w.bk);
        g.wv[i] = c.g(w.wv); g.bv[i] = c.g(w.bv); g.wo[i] = c.g(w.wo); g.bo[i] = c.g(w.bo);
        g.w1[i] = c.g(w.f1k); g.b1[i] = c.g(w.f1b); g.w2[i] = c.g(w.f2k); g.b2[i] = c.g(w.f2b);
        g.ln1g[i] = c.g(w.ln1g); g.ln1b[i] = c.g(w.ln1b); g.ln2g[i] = c.g(w.ln2g); g.ln2b[i] = c.g(w.ln2b);
    }
    g.poolw = c.g(p.poolk);
    g.poolb = c.g(p.poolb);
    return g;
}

ConvGrads conv_grads(const Ctx& c, int b) {
    const BranchP& p = c.t->pm.br[b];
    return ConvGrads{c.g(p.emb), c.g(p.pos), c.g(p.sck), c.g(p.scb), c.g(p.c3ak), c.g(p.c3ab), c.g(p.c3bk), c.g(p.c3bb)};
}

// d(concat)[:, b*64 .. +64]  ->  all parameter gradients of branch b, in three launches (+ the partial-sum kernels):
// launches first..last of them, `plan` carrying what the launches leave to be summed from one call to the next.
int branch_bwd(const Ctx& c, int b, const uint8_t* tok, int64_t B, const unsigned* seed, const Rates& rt, PartPlan* plan,
               int first = 0, int last = 2) {
    dptm_trainer* t = c.t;
    BranchAct& a = t->ba[b];
    const int wl = a.wlen;
    const TileCfg tc = t_tile_cfg(B, wl);
    const ConvArgs cf = conv_args(c, b, tok, B, seed, rt, tc);
    if (first <= 0 && last >= 0) {
        const TbBwdArgs f = tb_bwd_args(c, b, cf, B, seed, rt, tc);
        const TbGrads g = tb_grads(c, b);
        ProfScope ps(c, b * 6 + 3);
        T_TRY(t_tb_bwd(f, g, wl, tc.NW, plan, c.st));
    }
    const ConvGrads cg = conv_grads(c, b);
    if (first <= 1 && last >= 1) {
        ProfScope ps(c, b * 6 + 4);
        T_TRY(t_conv_b_bwd(cf, cg, wl, tc.NW, plan, c.st));
    }
    if (first <= 2 && last >= 2) {
        ProfScope ps(c, b * 6 + 5);
        T_TRY(t_conv_a_bwd(cf, cg, wl, tc.NW, plan, c.st));
    }
    return DPTM_OK;
}

// ------------------------------------------------------------------------------------ streams of a trainer
// A process has four hardware queues and the runtime maps every new stream to the queue with the fewest streams
// on it, so whether four new streams land on four DIFFERENT queues depends on what else the process has created
// and destroyed; two busy streams on one queue serialise, and the step's lanes then run one after another.
// There is no call that tells a stream's queue, so the trainer finds out by timing: two streams on one queue
// take twice as long over a pair of fixed-duration kernels as two streams on different queues.

__global__ void spin_kernel(unsigned long long ticks) {       // returns after `ticks` of the 100 MHz wall clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// Host noise (other ranks of a node starting at the same time, a descheduled thread) can only lengthen a trial, never
// shorten it: the shortest of three trials decides, so that a busy host does not make two independent streams look
// like one queue.
bool streams_serialise(hipStream_t a, hipStream_t b) {
    constexpr unsigned long long kTicks = 15000;              // 150 us
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    double best = 1e30;
    for (int trial = 0; trial < 3; ++trial) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, kTicks);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, kTicks);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = us < best ? us : best;
        if (best <= 1.6 * 150.0) break;
    }
    return best > 1.6 * 150.0;
}

// t->stream and t->aux[0..2]: four streams on four different hardware queues when the process has that many
hipError_t pick_streams(dptm_trainer* t) {
    constexpr int kCandidates = 8;
    hipStream_t cand[kCandidates] = {};
    hipStream_t chosen[4] = {};
    int n_chosen = 0, n_cand = 0;
    hipError_t err = hipSuccess;
    for (; n_cand < kCandidates && n_chosen < 4; ++n_cand) {
        err = hipStreamCreateWithFlags(&cand[n_cand], hipStreamNonBlocking);
        if (err != hipSuccess) break;
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand[n_cand], 0ull);      // first use binds the queue
        bool fresh = true;
        for (int k = 0; k < n_chosen && fresh; ++k) fresh = !streams_serialise(cand[n_cand], chosen[k]);
        if (fresh) {
            chosen[n_chosen++] = cand[n_cand];
            cand[n_cand] = nullptr;
        }
    }
    t->lanes_independent = n_chosen;       // lanes beyond these share a hardware queue with an earlier one
    // fewer than four queues found: the remaining lanes share streams already chosen, the rest is given back
    for (int k = 0; k < kCandidates; ++k) {
        if (!cand[k]) continue;
        if (n_chosen < 4 && err == hipSuccess)
            chosen[n_chosen++] = cand[k];
        else
            (void)hipStreamDestroy(cand[k]);
    }
    if (n_chosen == 0) return err != hipSuccess ? err : hipErrorUnknown;
    while (n_chosen < 4) {
        if (hipStreamCreateWithFlags(&chosen[n_chosen], hipStreamNonBlocking) != hipSuccess) return hipErrorUnknown;
        ++n_chosen;
    }
    t->stream = chosen[0];
    for (int b = 0; b < 3; ++b) t->aux[b] = chosen[1 + b];
    return hipSuccess;
}

}  // namespace

// a C++ exception must not leave an extern "C" function (see the function-try-blocks below)
static int on_texception(const char* where) noexcept {
    int code = DPTM_EHIP;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            code = DPTM_ENOMEM;
            g_terr = std::string(where) + ": out of host memory";
        } catch (const std::exception& e) {
            g_terr = std::string(where) + ": " + e.what();
        } catch (...) {
            g_terr = std::string(where) + ": unknown C++ exception";
        }
    } catch (...) {
    }
    return code;
}

extern "C" {

const char* dptm_trainer_last_error(void) { return g_terr.c_str(); }

void dptm_trainer_destroy(dptm_trainer* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    (void)hipDeviceSynchronize();
    if (t->gexec) (void)hipGraphExecDestroy(t->gexec);
    for (int b = 0; b < 3; ++b)
        if (t->aux[b]) (void)hipStreamDestroy(t->aux[b]);
    for (int ph = 0; ph < 2; ++ph) {
        if (t->ev_caller[ph]) (void)hipEventDestroy(t->ev_caller[ph]);
        for (int k = 0; k < DPTM_TRAIN_PROFILE_SLOTS; ++k)
            if (t->prof_ev[k][ph]) (void)hipEventDestroy(t->prof_ev[k][ph]);
        if (t->ev_fork[ph]) (void)hipEventDestroy(t->ev_fork[ph]);
        for (int b = 0; b < 3; ++b)
            if (t->ev_join[ph][b]) (void)hipEventDestroy(t->ev_join[ph][b]);
    }
    for (auto& stage : t->ev_sync)
        for (hipEvent_t e : stage)
            if (e) (void)hipEventDestroy(e);
    void* bufs[] = {t->W, t->G, t->M, t->V, t->trainable, t->arena.base, t->d_step, t->P, t->pack_jobs, t->stage};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

int64_t dptm_trainer_param_floats(const dptm_trainer* t) { return t ? t->pm.total : 0; }

int dptm_trainer_create(const void* weights, size_t nbytes, const dptm_spec* spec, int device, int64_t max_batch,
                        dptm_trainer** out) try {
    if (!out) return tfail(DPTM_EINVAL, "dptm_trainer_create: out is NULL");
    *out = nullptr;
    if (!weights || max_batch < 1) return tfail(DPTM_EINVAL, "dptm_trainer_create: bad argument");
    if (dptm_weight_count(spec) < 0) return tfail(DPTM_EINVAL, "dptm_trainer_create: unsupported model spec");
    dptm_trainer* t = new (std::nothrow) dptm_trainer();
    if (!t) return tfail(DPTM_ENOMEM, "dptm_trainer_create: out of host memory");
    std::unique_ptr<dptm_trainer, void (*)(dptm_trainer*)> owner(t, dptm_trainer_destroy);   // released on success only
    t->device = device;
    t->pm.build();
    if ((int64_t)(nbytes / sizeof(float)) != t->pm.canon_total)
        return tfail(DPTM_EINVAL, "dptm_trainer_create: weight blob has the wrong size");
    t->max_batch = max_batch;
    {
        const Tn& last = t->pm.t.back();
        if (last.off + last.size >= t->pm.total) return tfail(DPTM_EINVAL, "dptm_trainer_create: no padding behind the last tensor");
        t->loss_slot = t->pm.total - 1;
    }
    auto bail = [&](int code, const char* msg) { return tfail(code, msg); };   // `owner` frees the half-built trainer
    if (hipSetDevice(device) != hipSuccess) return bail(DPTM_EINVAL, "dptm_trainer_create: no such device");
    if (pick_streams(t) != hipSuccess) return bail(DPTM_EHIP, "hipStreamCreate failed");
    // Events that order the trainer's own streams against each other need no system-scope fence (nothing on the host, and
    // no other device, looks at what they order): DPTM_TRAIN_EVENT_FENCE=1 restores it (A/B)
    const bool sys_fence = [] {
        const char* e = std::getenv("DPTM_TRAIN_EVENT_FENCE");
        return e && e[0] == '1';
    }();
    const unsigned own_flags = hipEventDisableTiming | (sys_fence ? 0u : (unsigned)hipEventDisableSystemFence);
    for (int ph = 0; ph < 2; ++ph) {
        if (hipEventCreateWithFlags(&t->ev_fork[ph], own_flags) != hipSuccess ||
            hipEventCreateWithFlags(&t->ev_caller[ph], hipEventDisableTiming) != hipSuccess)
            return bail(DPTM_EHIP, "hipEventCreate failed");
        for (int b = 0; b < 3; ++b)
            // (the join at the END of the step keeps the fence: behind it the gradients leave for the host or for peers)
            if (hipEventCreateWithFlags(&t->ev_join[ph][b], ph == 1 ? (unsigned)hipEventDisableTiming : own_flags) != hipSuccess)
                return bail(DPTM_EHIP, "hipEventCreate failed");
    }
    for (auto& stage : t->ev_sync)
        for (hipEvent_t& e : stage)
            if (hipEventCreateWithFlags(&e, own_flags) != hipSuccess) return bail(DPTM_EHIP, "hipEventCreate failed");
    const size_t pb = (size_t)t->pm.total * sizeof(float);
    if (hipMalloc((void**)&t->W, pb) != hipSuccess || hipMalloc((void**)&t->G, pb) != hipSuccess ||
        hipMalloc((void**)&t->M, pb) != hipSuccess || hipMalloc((void**)&t->V, pb) != hipSuccess ||
        hipMalloc((void**)&t->trainable, (size_t)t->pm.total) != hipSuccess ||
        hipMalloc((void**)&t->d_step, kStepParams * sizeof(unsigned)) != hipSuccess)
        return bail(DPTM_ENOMEM, "dptm_trainer_create: parameter buffers");
    if (hipMemset(t->d_step, 0, kStepParams * sizeof(unsigned)) != hipSuccess) return bail(DPTM_EHIP, "dptm_trainer_create: step scalars");
    if (const char* env = std::getenv("DPTM_TRAIN_GRAPH")) t->use_graph = std::strcmp(env, "0") != 0;
    if (const char* env = std::getenv("DPTM_TRAIN_SERIAL")) t->serial = std::strcmp(env, "0") != 0;
    std::vector<float> host((size_t)t->pm.total, 0.0f);
    std::vector<uint8_t> tr((size_t)t->pm.total, 0);
    const float* src = static_cast<const float*>(weights);
    for (const Tn& x : t->pm.t) {
        std::memcpy(host.data() + x.off, src + x.canon, (size_t)x.size * sizeof(float));
        if (x.trainable) std::memset(tr.data() + x.off, 1, (size_t)x.size);
    }
    if (hipMemcpy(t->W, host.data(), pb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->trainable, tr.data(), tr.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(t->G, 0, pb) != hipSuccess || hipMemset(t->M, 0, pb) != hipSuccess || hipMemset(t->V, 0, pb) != hipSuccess)
        return bail(DPTM_EHIP, "dptm_trainer_create: parameter upload");
    t->arena.cap = arena_bytes(max_batch);
    if (hipMalloc((void**)&t->arena.base, t->arena.cap) != hipSuccess) return bail(DPTM_ENOMEM, "dptm_trainer_create: activation arena");
    int rc = alloc_acts(t);
    if (rc) return rc;
    {
        // per branch: the packed forward image the fused kernels read (layout of dptm_layout.h) and the transposed
        // images of the backward pass; pack_kernel rebuilds them from W at the start of every step.  The same launch
        // clears what a step accumulates into: the gradient buffer, the loss and the BatchNorm column sums.
        std::vector<PackJob> jobs;
        int64_t off = 0;                       // floats into P
        auto reserve = [&](int64_t floats) {
            const int64_t at = off;
            off += (floats + 63) / 64 * 64;
            return at;
        };
        struct Pending { int tensor; int64_t src_off; int64_t dst; int kind, K, N, tr; };
        std::vector<Pending> pend;
        for (int b = 0; b < 3; ++b) {
            const BranchP& q = t->pm.br[b];
            const int64_t F = t->img_fwd[b] = reserve((int64_t)kBranchF4 * 4);
            const int64_t T = t->img_bwd[b] = reserve((int64_t)kBtTotal * 4);
            auto mat = [&](int tensor, int64_t src_off, int64_t dst_f4, int K, int N, bool tr, int64_t base) {
                pend.push_back(Pending{tensor, src_off, base + dst_f4 * 4, 0, K, N, tr ? 1 : 0});
            };
            auto vec = [&](int tensor, int64_t dst_f4, int n) { pend.push_back(Pending{tensor, 0, F + dst_f4 * 4, 1, n, 0, 0}); };
            auto copy = [&](int tensor, int64_t dst_f4, int n) { pend.push_back(Pending{tensor, 0, F + dst_f4 * 4, 2, n, 0, 0}); };
            copy(q.emb, kOffEmb, kCodes * kD);
            copy(q.pos, kOffPos, kW[b] * kD);
            mat(q.sck, 0, kOffScW, kD, kD, false, F);
            for (int tap = 0; tap < 3; ++tap) {
                mat(q.c3ak, (int64_t)tap * kD * kD, kOffC3aW + tap * kMat64, kD, kD, false, F);
                mat(q.c3bk, (int64_t)tap * kD * kD, kOffC3bW + tap * kMat64, kD, kD, false, F);
            }
            vec(q.scb, kOffConvV + kVScB * kVec64, kD);
            vec(q.c3ab, kOffConvV + kVC3aB * kVec64, kD);
            vec(q.c3bb, kOffConvV + kVC3bB * kVec64, kD);
            for (int i = 0; i < 2; ++i) {
                const TbP& w = q.tb[i];
                const int64_t base = kOffTb + (int64_t)i * kTbSize;
                mat(w.wq, 0, base + kTbWq, kD, kD, false, F);
                mat(w.wk, 0, base + kTbWk, kD, kD, false, F);
                mat(w.wv, 0, base + kTbWv, kD, kD, false, F);
                mat(w.wo, 0, base + kTbWo, kD, kD, false, F);
                mat(w.f1k, 0, base + kTbF1, kD, kFF, false, F);
                mat(w.f2k, 0, base + kTbF2, kFF, kD, false, F);
                const int vid[9] = {w.bq, w.bk, w.bv, w.bo, w.ln1g, w.ln1b, w.f2b, w.ln2g, w.ln2b};   // order of enum TbVec
                for (int v = 0; v < 9; ++v) vec(vid[v], base + kTbVec + v * kVec64, kD);
                vec(w.f1b, base + kTbB1, kFF);
                const int64_t tb = (int64_t)i * kBtSize;
                mat(w.wq, 0, tb + kBtWqT, kD, kD, true, T);
                mat(w.wk, 0, tb + kBtWkT, kD, kD, true, T);
                mat(w.wv, 0, tb + kBtWvT, kD, kD, true, T);
                mat(w.wo, 0, tb + kBtWoT, kD, kD, true, T);
                mat(w.f2k, 0, tb + kBtW2T, kD, kFF, true, T);     // W2^T: M[k][n] = W2[n][k], K = 64, N = 256
                mat(w.f1k, 0, tb + kBtW1T, kFF, kD, true, T);     // W1^T: M[k][n] = W1[n][k], K = 256, N = 64
            }
            vec(q.poolk, kOffPoolV, kD);
            copy(q.poolb, kOffPoolB, 1);
            mat(q.sck, 0, kBtScT, kD, kD, true, T);
            for (int tap = 0; tap < 3; ++tap) {
                mat(q.c3ak, (int64_t)tap * kD * kD, kBtC3aT + tap * 1024, kD, kD, true, T);
                mat(q.c3bk, (int64_t)tap * kD * kD, kBtC3bT + tap * 1024, kD, kD, true, T);
            }
        }
        if (hipMalloc((void**)&t->P, (size_t)off * sizeof(float)) != hipSuccess) return bail(DPTM_ENOMEM, "dptm_trainer_create: weight images");
        if (hipMemset(t->P, 0, (size_t)off * sizeof(float)) != hipSuccess) return bail(DPTM_EHIP, "dptm_trainer_create: weight images");
        for (const Pending& q : pend)
            jobs.push_back(PackJob{t->W + t->pm.t[q.tensor].off + q.src_off, t->P + q.dst, q.kind, q.K, q.N, q.tr});
        // the gradient buffer: 2*lambda*w under the regularised kernels, zero everywhere else (padding included)
        constexpr int64_t kJobFloats = 65536;
        auto span = [&](int64_t lo, int64_t hi, bool l2) {
            for (int64_t at = lo; at < hi; at += kJobFloats)
                jobs.push_back(PackJob{l2 ? t->W + at : nullptr, t->G + at, l2 ? 4 : 3, (int)std::min(kJobFloats, hi - at), 0, 0});
        };
        {
            std::vector<const Tn*> reg;
            for (const Tn& x : t->pm.t)
                if (x.l2) reg.push_back(&x);
            std::sort(reg.begin(), reg.end(), [](const Tn* a, const Tn* b) { return a->off < b->off; });
            int64_t cursor = 0;
            for (const Tn* x : reg) {
                span(cursor, x->off, false);
                span(x->off, x->off + x->size, true);
                cursor = x->off + x->size;
            }
            span(cursor, t->pm.total, false);
        }
        jobs.push_back(PackJob{nullptr, t->d_loss, 3, 64 + 2 * kBnLayers * 1024, 0, 0});   // loss + column sums, back to back
        if (hipMalloc((void**)&t->pack_jobs, jobs.size() * sizeof(PackJob)) != hipSuccess)
            return bail(DPTM_ENOMEM, "dptm_trainer_create: weight image table");
        if (hipMemcpy(t->pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice) != hipSuccess)
            return bail(DPTM_EHIP, "dptm_trainer_create: weight image table");
        t->n_pack_jobs = (int)jobs.size();
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail(DPTM_EHIP, "dptm_trainer_create: sync");
    *out = owner.release();
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_create");
}

int dptm_trainer_loss_buffer(dptm_trainer* t, float** dev_ptr) try {
    if (!t || !dev_ptr) return tfail(DPTM_EINVAL, "dptm_trainer_loss_buffer: NULL argument");
    *dev_ptr = t->d_loss;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_loss_buffer");
}

int dptm_trainer_proba_buffer(dptm_trainer* t, float** dev_ptr) try {
    if (!t || !dev_ptr) return tfail(DPTM_EINVAL, "dptm_trainer_proba_buffer: NULL argument");
    *dev_ptr = t->proba;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_proba_buffer");
}

int dptm_trainer_stage_loss(dptm_trainer* t, int64_t* slot, void* stream) try {
    if (!t || !slot) return tfail(DPTM_EINVAL, "dptm_trainer_stage_loss: NULL argument");
    T_TRY(hipSetDevice(t->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : t->stream;
    T_TRY(hipMemcpyAsync(t->G + t->loss_slot, t->d_loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    *slot = t->loss_slot;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_stage_loss");
}

int dptm_trainer_lanes(dptm_trainer* t, int32_t* independent_out) try {
    if (!t || !independent_out) return tfail(DPTM_EINVAL, "dptm_trainer_lanes: NULL argument");
    *independent_out = t->lanes_independent;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_lanes");
}

int dptm_trainer_stream(dptm_trainer* t, void** stream) try {
    if (!t || !stream) return tfail(DPTM_EINVAL, "dptm_trainer_stream: NULL argument");
    *stream = (void*)t->stream;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_stream");
}

int dptm_trainer_grad_buffer(dptm_trainer* t, float** dev_ptr, int64_t* count) try {
    if (!t || !dev_ptr || !count) return tfail(DPTM_EINVAL, "dptm_trainer_grad_buffer: NULL argument");
    *dev_ptr = t->G;
    *count = t->pm.total;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_grad_buffer");
}

// Enqueues forward (training mode) + loss + backward of one local batch on `st`.  No host synchronisation, no
// allocation: the sequence can be captured into a graph.  Per-step scalars come from t->d_step.
static int enqueue_forward_backward(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                                    const float* strct, const float* esm, const int32_t* labels, int64_t n,
                                    const dptm_train_opts* o, hipStream_t caller) {
    // The step runs on three streams of the trainer's own (created back to back, so that they land on different
    // hardware queues: a caller's stream may share its queue with one of them, and two streams on one queue
    // serialise); the caller's stream only waits for the step's last kernel.
    hipStream_t st = t->serial ? caller : t->stream;
    if (st != caller) {
        T_TRY(hipEventRecord(t->ev_caller[0], caller));
        T_TRY(hipStreamWaitEvent(st, t->ev_caller[0], 0));
    }
    Ctx c{t, st, t->W, t->G, &t->sc[3]};
    const PMap& pm = t->pm;
    const Rates rt{o->rate_embed, o->rate_conv, o->rate_tb, o->rate_dense};
    const unsigned* seed = t->d_step;
    const uint8_t* tok[3] = {tok21, tok33, tok51};
    const int64_t B = n;
    int rc;
    // one launch: weight images of this step's weights; loss accumulator and BatchNorm column sums cleared; the gradient
    // buffer cleared, or set to the gradient of the l2 kernel regularisers (model.py:130-138,200,204), whose value
    // the head / loss kernel adds to the loss
    t->pooled_grad_in_kernel = false;
    // (that launch also writes the step's scalars, unless the launches are being recorded for replay: see forward_backward_on)
    const StepVals sv{o->seed, o->n_global > 0 ? (float)o->n_global : (float)n, (float)o->n_pos_global};
    if (!t->use_graph) t->l2_slot ^= 1;
    T_TRY(t_pack(t->pack_jobs, t->n_pack_jobs, o->l2 > 0.f ? o->l2 * o->l2_scale : 0.f, t->d_step, t->use_graph ? 0 : t->l2_slot,
                 t->use_graph ? nullptr : &sv, st));
    // Small batches without a statistics all-reduce take the per-column kernels of dptm_train_site.hip: one launch per
    // layer and direction instead of a product and half a dozen element-wise passes each.
    static const int64_t site_rows = [] {
        const char* e = std::getenv("DPTM_TRAIN_SITE_ROWS");     // (measurements: another limit, at most kSiteMaxRows)
        const int64_t v = e ? std::atoll(e) : (int64_t)kSiteFusedRows;
        return v < kSiteMaxRows ? v : (int64_t)kSiteMaxRows;
    }();
    const bool small = B <= site_rows && !t->ar_fn &&
                       ((reinterpret_cast<uintptr_t>(strct) | reinterpret_cast<uintptr_t>(esm)) & 15) == 0;
    // ---------------- forward.  Four lanes of work: the W=21 and W=33 branches on side streams, the W=51 branch (the
    // longest) on the step's main stream `st`, the structure / ESM layers of a small batch on a third side stream `ss`.
    // (A process has four hardware queues: a fifth concurrent stream would share one and serialise behind it.)
    // Batches whose tiles of all three branches are on the chip at once (up to ~540 sites) start each of the branches'
    // six launches ONCE for all three branches (grid.y = branch) on the main stream: the step is then a chain of launches
    // on one queue, and no hand-over between streams -- 10 to 25 us each -- sits on its critical path.  (Not under a
    // profile, whose clocks are per branch, nor with synchronised BatchNorm, which sums between the stages.)
    static const bool merge_allowed = [] {
        const char* e = std::getenv("DPTM_TRAIN_MERGE");        // DPTM_TRAIN_MERGE=0: one launch per branch (A/B)
        return !(e && e[0] == '0');
    }();
    bool merged3 = merge_allowed && !t->serial && !t->prof_on && !t->ar_fn;
    const int nw3 = t_tile_cfg(B, kW[2]).NW;
    int64_t wgs3 = 0;
    bool one_site_tiles = true;
    for (int b = 0; b < 3 && merged3; ++b) {
        const TileCfg tcb = t_tile_cfg(B, kW[b]);
        merged3 = tcb.NW == nw3 && (nw3 == 4 || nw3 == 8) && part_sums_deferred(tcb.ntiles);
        wgs3 += tcb.ntiles;
        one_site_tiles = one_site_tiles && tcb.S == 1;
    }
    // ... while all their workgroups are on the chip at once (measured: 512 sites 1.35 -> 1.18 ms; beyond one round of
    // workgroups -- 768, 1,280 sites -- three streams that take turns on the CUs are the faster form again)
    static const int64_t merge_wgs = [] {
        const char* e = std::getenv("DPTM_TRAIN_MERGE_WGS");    // (measurements: another limit)
        return e ? std::atoll(e) : (int64_t)232;                // a little short of the 256 CUs: the per-site lane needs some too
    }();
    // (one-site tiles: wherever they are used, up to 112 sites -- 85 sites: 0.58 ms in one launch against 0.72 on three streams)
    merged3 = merged3 && wgs3 <= (one_site_tiles && !std::getenv("DPTM_TRAIN_MERGE_WGS") ? 336 : merge_wgs);
    hipStream_t bstream[3] = {t->serial || merged3 ? st : t->aux[0], t->serial || merged3 ? st : t->aux[1], st};
    hipStream_t ss = !t->serial ? t->aux[2] : st;      // per-site layers that run next to the branches
    hipStream_t side[3] = {bstream[0], bstream[1], ss};
    auto branches3 = [&](bool forward) -> int {
        ConvArgs cf[3];
        for (int b = 0; b < 3; ++b) {
            Ctx cb{t, st, t->W, t->G, &t->sc[b]};
            cf[b] = conv_args(cb, b, tok[b], B, seed, rt, t_tile_cfg(B, kW[b]));
        }
        if (forward) {
            TbFwdArgs f[3];
            for (int b = 0; b < 3; ++b) {
                Ctx cb{t, st, t->W, t->G, &t->sc[b]};
                f[b] = tb_fwd_args(cb, b, cf[b], B, seed, rt, t_tile_cfg(B, kW[b]));
            }
            T_TRY(t_conv_a_fwd3(cf, nw3, st));
            T_TRY(t_conv_b_fwd3(cf, nw3, st));
            T_TRY(t_tb_fwd3(f, nw3, st));
            return DPTM_OK;
        }
        TbBwdArgs f[3];
        TbGrads g[3];
        ConvGrads cg[3];
        PartPlan plans[3];
        for (int b = 0; b < 3; ++b) {
            Ctx cb{t, st, t->W, t->G, &t->sc[b]};
            f[b] = tb_bwd_args(cb, b, cf[b], B, seed, rt, t_tile_cfg(B, kW[b]));
            g[b] = tb_grads(cb, b);
            cg[b] = conv_grads(cb, b);
        }
        T_TRY(t_tb_bwd3(f, g, nw3, plans, st));
        T_TRY(t_conv_b_bwd3(cf, cg, nw3, plans, st));
        T_TRY(t_conv_a_bwd3(cf, cg, nw3, plans, st));
        return DPTM_OK;
    };
    // the longest branch first: its launches are then the first the host hands to the device
    // Synchronised BatchNorm: the three branches advance stage by stage, and between two stages the main stream sums
    // the column sums all three just produced over the replicas in ONE collective (their slots are contiguous):
    // 2 collectives per direction for the branches instead of 6.  The input gradient of a BatchNorm needs the sums
    // over the WHOLE batch (bn_all), its parameter gradients this replica's share (bn_slots).
    auto stage_sum = [&](int point, int slot0, int nslots, bool backward, bool with_site_lane) -> int {
        for (int b = 0; b < 2; ++b)
            if (bstream[b] != st) {
                T_TRY(hipEventRecord(t->ev_sync[point][b], bstream[b]));
                T_TRY(hipStreamWaitEvent(st, t->ev_sync[point][b], 0));
            }
        if (with_site_lane && ss != st) {
            T_TRY(hipEventRecord(t->ev_sync[point][3], ss));
            T_TRY(hipStreamWaitEvent(st, t->ev_sync[point][3], 0));
        }
        float* buf = t->bn_slots + (size_t)slot0 * 1024;
        if (backward) {
            buf = t->bn_all + (size_t)slot0 * 1024;
            T_TRY(hipMemcpyAsync(buf, t->bn_slots + (size_t)(kBnLayers + slot0) * 1024, (size_t)nslots * 1024 * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
        }
        const int r = replica_sum(c, buf, (int64_t)nslots * 1024);
        if (r) return r;
        T_TRY(hipEventRecord(t->ev_sync[point][2], st));
        for (int b = 0; b < 2; ++b)
            if (bstream[b] != st) T_TRY(hipStreamWaitEvent(bstream[b], t->ev_sync[point][2], 0));
        if (with_site_lane && ss != st) T_TRY(hipStreamWaitEvent(ss, t->ev_sync[point][2], 0));
        return DPTM_OK;
    };
    auto branches = [&](bool forward, int first, int last) -> int {
        if (merged3) return last == 2 ? branches3(forward) : DPTM_OK;     // (all three go out with the call that holds W = 51)
        PartPlan plans[3];
        const bool staged = t->ar_fn != nullptr && first == 0 && last == 2;
        for (int stage = 0; stage < (staged ? 3 : 1); ++stage) {
            for (int b = last; b >= first; --b) {
                Ctx cb{t, bstream[b], t->W, t->G, &t->sc[b]};
                const int s0 = staged ? stage : 0, s1 = staged ? stage : 2;
                const int r = forward ? branch_fwd(cb, b, tok[b], B, seed, rt, s0, s1) : branch_bwd(cb, b, tok[b], B, seed, rt, &plans[b], s0, s1);
                if (r) return r;
            }
            if (staged && stage < 2) {
                // forward: BatchNorm 1 after the first launch, BatchNorm 2 after the second; backward: 2, then 1.  The
                // structure and ESM layers' BatchNorms (slots 6, 7; their sums are on the per-site lane by then) join the
                // first collective of either direction: forward slots 0..7 (3..5 are still zero), backward 3..7.
                const bool first_sum = stage == 0;
                const int slot0 = forward ? 3 * stage : 3 * (1 - stage);
                const int nslots = first_sum ? (forward ? 8 : 5) : 3;
                const int r = stage_sum((forward ? 0 : 2) + stage, slot0, nslots, !forward, first_sum);
                if (r) return r;
            }
        }
        return DPTM_OK;
    };
    auto fork = [&](int phase) -> int {
        T_TRY(hipEventRecord(t->ev_fork[phase], st));
        for (int b = 0; b < 3; ++b)
            if (side[b] != st) T_TRY(hipStreamWaitEvent(side[b], t->ev_fork[phase], 0));
        return DPTM_OK;
    };
    auto join = [&](int phase) -> int {
        for (int b = 0; b < 3; ++b)
            if (side[b] != st) {
                T_TRY(hipEventRecord(t->ev_join[phase][b], side[b]));
                T_TRY(hipStreamWaitEvent(st, t->ev_join[phase][b], 0));
            }
        return DPTM_OK;
    };
    if ((rc = fork(0))) return rc;
    if (small) {
        if ((rc = branches(true, 2, 2))) return rc;       // the longest lane first
        const float sc = rt.dense > 0.f ? 1.0f / (1.0f - rt.dense) : 1.0f;
        const unsigned thr = dropout_threshold(rt.dense);
        auto dropd = [&](int stream, int cols, int col0) { return SiteDrop{(unsigned)stream, thr, sc, cols, col0}; };
        const SiteDrop none{0u, 0u, 1.0f, 1, 0};
        hipStream_t ls = ss;     // stream of the layer launches below
        auto fwd = [&](const float* x, int64_t ldx, int K, SiteDrop in_drop, const DenseP& L, int N, float* a, bool bn, float* mi,
                       SiteDrop out_drop, float* out, int64_t ldo, int col_off) {
            SiteFwdArgs f;
            f.x = x; f.ldx = ldx; f.K = K; f.in_drop = in_drop;
            f.W = c.w(L.k); f.bias = c.w(L.b); f.N = N; f.a = a;
            f.bn = bn ? 1 : 0;
            f.gamma = bn ? c.w(L.bn.gamma) : nullptr; f.beta = bn ? c.w(L.bn.beta) : nullptr; f.mi = mi;
            f.batch_mean = bn ? c.g(L.bn.mean) : nullptr; f.batch_var = bn ? c.g(L.bn.var) : nullptr;
            f.out_drop = out_drop; f.out = out; f.ldo = ldo; f.col_off = col_off;
            f.B = (int)B; f.step = seed;
            return t_site_fwd(f, ls);
        };
        // struct branch (model.py:187-189), ESM branch (model.py:191-194)
        T_TRY(fwd(strct, 8, 8, none, pm.strct, 128, t->sa, true, t->smi, dropd(24 + 0, 128, 0), t->concat, kConcat, 192));
        T_TRY(fwd(esm, kEsm, kEsm, none, pm.esm1, 512, t->ea, true, t->emi, dropd(24 + 1, 512, 0), t->e1, 512, 0));
        T_TRY(fwd(t->e1, 512, 512, none, pm.esm2, 256, t->e2a, false, nullptr, none, t->concat, kConcat, 320));
        if ((rc = branches(true, 0, 1))) return rc;
        if ((rc = join(0))) return rc;              // pooled branch outputs, structure and ESM columns are in concat
        ls = st;
        // trunk (model.py:198-205); the dropout in front of it is applied where the concatenation is read
        T_TRY(fwd(t->concat, kConcat, kConcat, dropd(24 + 2, kConcat, 0), pm.t1, 512, t->t1a, true, t->t1mi, dropd(24 + 3, 512, 0),
                  t->t1z, 512, 0));
        T_TRY(fwd(t->t1z, 512, 512, none, pm.t2, 256, t->t2a, true, t->t2mi, none, t->t2z, 256, 0));
        // head, loss and its gradient (trainer.py:36-54)
        T_TRY(t_head_loss(t->t2z, c.w(pm.head.k), c.w(pm.head.b), labels, B, t->d_step, t->proba, t->d_loss, t->d256a,
                          c.g(pm.head.k), c.g(pm.head.b), st));
        // part (see SiteBwdArgs): with streams to spare the trunk's weight gradients leave the chain that leads to d(concat)
        // and run on the per-site lane next to the branches' backward pass
        // (not when the branches' workgroups fill the chip -- 78 to 85 sites in one-site tiles: 0.56 -> 0.58 ms with the cut)
        int64_t branch_wgs = 0;
        for (int b = 0; b < 3; ++b) branch_wgs += t_tile_cfg(B, kW[b]).ntiles;
        const bool split_dw = ss != st && branch_wgs <= 232;
        auto bwd = [&](const float* dzd, const float* da_next, int Nn, const DenseP* next, int next_row_off, SiteDrop mid_drop,
                       float* gout, SiteDrop out_drop, const DenseP* L, bool bn, const float* mi, const float* a, float* da,
                       const float* x, int64_t ldx, int K, SiteDrop in_drop, int N, int part = 0) {
            SiteBwdArgs f;
            f.part = part;
            f.dzd = dzd; f.ld_dzd = N;
            f.da_next = da_next; f.W_next = next ? c.w(next->k) : nullptr; f.Nn = Nn; f.next_row_off = next_row_off;
            f.mid_drop = mid_drop; f.gout = gout; f.ld_gout = kConcat; f.out_drop = out_drop;
            f.bn = bn ? 1 : 0;
            f.gamma = bn ? c.w(L->bn.gamma) : nullptr; f.mi = mi; f.a = a; f.da = da;
            f.x = x; f.ldx = ldx; f.K = K; f.in_drop = in_drop;
            f.dW = L ? c.g(L->k) : nullptr; f.db = L ? c.g(L->b) : nullptr;
            f.W = L ? c.w(L->k) : nullptr;
            f.l2c = L && o->l2 > 0.f && pm.t[L->k].l2 ? 2.0f * o->l2 * o->l2_scale : 0.f;
            f.dgamma = bn ? c.g(L->bn.gamma) : nullptr; f.dbeta = bn ? c.g(L->bn.beta) : nullptr;
            f.N = N; f.B = (int)B; f.step = seed;
            return t_site_bwd(f, ls);
        };
        T_TRY(bwd(t->d256a, nullptr, 0, nullptr, 0, none, nullptr, none, &pm.t2, true, t->t2mi, t->t2a, t->d256b, t->t1z, 512, 512, none, 256,
                  split_dw ? 1 : 0));
        T_TRY(bwd(nullptr, t->d256b, 256, &pm.t2, 0, none, nullptr, dropd(24 + 3, 512, 0), &pm.t1, true, t->t1mi, t->t1a, t->d512b,
                  t->concat, kConcat, kConcat, dropd(24 + 2, kConcat, 0), 512, split_dw ? 1 : 0));
        // d(concat) of the three pooled branch outputs (columns 0..191), behind the concatenation's dropout -- unless every
        // branch runs one-site tiles, which form their site's 64 values themselves: one launch less on the chain
        t->pooled_grad_in_kernel = true;
        for (int b = 0; b < 3; ++b) {
            const TileCfg tcb = t_tile_cfg(B, kW[b]);
            t->pooled_grad_in_kernel = t->pooled_grad_in_kernel && t_pair_tile(tcb.S, tcb.NW);
        }
        if (!t->pooled_grad_in_kernel)
            T_TRY(bwd(nullptr, t->d512b, 512, &pm.t1, 0, dropd(24 + 2, kConcat, 0), t->dconcat, none, nullptr, false, nullptr, nullptr,
                      nullptr, nullptr, 0, 0, none, 192));
        if ((rc = fork(1))) return rc;
        if ((rc = branches(false, 2, 2))) return rc;      // the longest lane first
        // ESM and structure layers on a stream of their own, side by side with the three branches
        ls = ss;
        if (split_dw) {
            // (ahead of the layers below, which overwrite d256a -- the head's gradient -- but not d256b / d512b)
            T_TRY(bwd(t->d256a, nullptr, 0, nullptr, 0, none, nullptr, none, &pm.t2, true, t->t2mi, t->t2a, t->d256b, t->t1z, 512, 512, none,
                      256, 2));
            T_TRY(bwd(nullptr, t->d256b, 256, &pm.t2, 0, none, nullptr, dropd(24 + 3, 512, 0), &pm.t1, true, t->t1mi, t->t1a, t->d512b,
                      t->concat, kConcat, kConcat, dropd(24 + 2, kConcat, 0), 512, 2));
        }
        T_TRY(bwd(nullptr, t->d512b, 512, &pm.t1, 320, dropd(24 + 2, kConcat, 320), nullptr, none, &pm.esm2, false, nullptr, t->e2a, t->d256a,
                  t->e1, 512, 512, none, 256));
        T_TRY(bwd(nullptr, t->d512b, 512, &pm.t1, 192, dropd(24 + 2, kConcat, 192), nullptr, dropd(24 + 0, 128, 0), &pm.strct, true, t->smi,
                  t->sa, t->d128, strct, 8, 8, none, 128));
        T_TRY(bwd(nullptr, t->d256a, 256, &pm.esm2, 0, none, nullptr, dropd(24 + 1, 512, 0), &pm.esm1, true, t->emi, t->ea, t->d512a, esm, kEsm,
                  kEsm, none, 512));
        if ((rc = branches(false, 0, 1))) return rc;
    } else {
        // structure and ESM layers on the fourth stream, next to the three branches.  With synchronised BatchNorm their
        // column sums are taken BEFORE the branches are enqueued, so that they join the collective behind the branches'
        // first stage, and the layers themselves follow behind it.
        const bool packed = t->ar_fn != nullptr;
        Ctx cf{t, ss, t->W, t->G, &t->sc[3]};
        auto dense_swish = [&](GemmArgs g, float* act, int64_t lda, hipStream_t s) {     // pre-activation to g.out, swish of it to act
            g.act_out = act; g.lda = lda;
            return t_gemm(g, false, s);
        };
        // struct branch (model.py:187-189), ESM branch (model.py:191-194); part 0: up to the column sums, 1: from them on
        auto site_layers_fwd = [&](int part0, int part1) -> int {
            const BnPart bp = part0 == 0 && part1 == 1 ? kBnWhole : part0 == 0 ? kBnSums : kBnApply;
            int r;
            if (part0 == 0) T_TRY(dense_swish(ga(strct, 8, c.w(pm.strct.k), 128, c.w(pm.strct.b), t->sa, 128, 0, B, 8, 128), t->sy, 128, ss));
            if ((r = bn_train_fwd(cf, t->sy, t->concat, kConcat, 192, B, 128, pm.strct.bn, t->smi, bp))) return r;
            if (part1 == 1) T_TRY(t_dropout(t->concat + 192, B, 128, kConcat, 0, seed, 24 + 0, rt.dense, ss));
            if (part0 == 0) T_TRY(dense_swish(ga(esm, kEsm, c.w(pm.esm1.k), 512, c.w(pm.esm1.b), t->ea, 512, 0, B, kEsm, 512), t->ey, 512, ss));
            if ((r = bn_train_fwd(cf, t->ey, t->e1, 512, 0, B, 512, pm.esm1.bn, t->emi, bp))) return r;
            if (part1 == 1) {
                T_TRY(t_dropout(t->e1, B, 512, 512, 0, seed, 24 + 1, rt.dense, ss));
                // swish straight into the concat columns 320..575
                T_TRY(dense_swish(ga(t->e1, 512, c.w(pm.esm2.k), 256, c.w(pm.esm2.b), t->e2a, 256, 0, B, 512, 256), t->concat + 320, kConcat, ss));
            }
            return DPTM_OK;
        };
        if (packed && (rc = site_layers_fwd(0, 0))) return rc;
        if ((rc = branches(true, 0, 2))) return rc;
        if ((rc = packed ? site_layers_fwd(1, 1) : site_layers_fwd(0, 1))) return rc;
        if ((rc = join(0))) return rc;              // pooled branch outputs are in concat
        T_TRY(t_dropout(t->concat, B, kConcat, kConcat, 0, seed, 24 + 2, rt.dense, st));
        // trunk (model.py:200-205)
        T_TRY(dense_swish(ga(t->concat, kConcat, c.w(pm.t1.k), 512, c.w(pm.t1.b), t->t1a, 512, 0, B, kConcat, 512), t->t1y, 512, st));
        if ((rc = bn_train_fwd(c, t->t1y, t->t1z, 512, 0, B, 512, pm.t1.bn, t->t1mi))) return rc;
        T_TRY(t_dropout(t->t1z, B, 512, 512, 0, seed, 24 + 3, rt.dense, st));
        T_TRY(dense_swish(ga(t->t1z, 512, c.w(pm.t2.k), 256, c.w(pm.t2.b), t->t2a, 256, 0, B, 512, 256), t->t2y, 256, st));
        if ((rc = bn_train_fwd(c, t->t2y, t->t2z, 256, 0, B, 256, pm.t2.bn, t->t2mi))) return rc;
        // ---------------- head, loss and its gradient (trainer.py:36-54)
        T_TRY(t_head_loss(t->t2z, c.w(pm.head.k), c.w(pm.head.b), labels, B, t->d_step, t->proba, t->d_loss, t->d256a,
                          c.g(pm.head.k), c.g(pm.head.b), st));
        // ---------------- backward, per-site layers
        // trunk 2: BN -> swish -> dense
        if ((rc = bn_train_bwd(c, t->d256a, 256, 0, t->t2y, t->d256b, B, 256, t->t2mi, pm.t2.bn))) return rc;
        T_TRY(t_swish_bwd(t->t2a, t->d256b, t->d256b, B * 256, st));
        T_TRY(t_gemm(ga(t->d256b, 256, c.w(pm.t2.k), 256, nullptr, t->d512a, 512, 0, B, 256, 512), true, st));
        T_TRY(t_dropout(t->d512a, B, 512, 512, 0, seed, 24 + 3, rt.dense, st));
        if ((rc = bn_train_bwd(c, t->d512a, 512, 0, t->t1y, t->d512b, B, 512, t->t1mi, pm.t1.bn))) return rc;
        T_TRY(t_swish_bwd(t->t1a, t->d512b, t->d512b, B * 512, st));
        T_TRY(t_gemm(ga(t->d512b, 512, c.w(pm.t1.k), 512, nullptr, t->dconcat, kConcat, 0, B, 512, kConcat), true, st));
        T_TRY(t_dropout(t->dconcat, B, kConcat, kConcat, 0, seed, 24 + 2, rt.dense, st));
        if ((rc = fork(1))) return rc;
        // The trunk's two weight gradients wait until here -- nothing on the way to d(concat) needs them -- and run with
        // the ESM / structure layers on the fourth stream, next to the branches instead of holding the whole chip up.
        // (One stream for all of them: the layers below reuse d512b.)  With synchronised BatchNorm the lane goes as far as
        // its two BatchNorms' column sums before the branches are enqueued (they join the collective behind the
        // branches' first backward stage) and finishes behind it.
        Ctx cs{t, ss, t->W, t->G, &t->sc[3]};
        auto site_layers_bwd = [&](int part0, int part1) -> int {
            const BnPart bp = part0 == 0 && part1 == 1 ? kBnWhole : part0 == 0 ? kBnSums : kBnApply;
            int r;
            if (part0 == 0) {
                T_TRY(t_gemm_tn(gt(t->t1z, 512, 0, t->d256b, 256, 0, c.g(pm.t2.k), 256, c.g(pm.t2.b), B, 512, 256), ss));
                T_TRY(t_gemm_tn(gt(t->concat, kConcat, 0, t->d512b, 512, 0, c.g(pm.t1.k), 512, c.g(pm.t1.b), B, kConcat, 512), ss));
                // ESM branch: d(concat[:,320:]) -> swish -> dense2 -> dropout -> BN -> swish -> dense1
                T_TRY(t_copy_cols(t->d256a, 256, 0, t->dconcat, kConcat, 320, B, 256, false, ss));
                T_TRY(t_swish_bwd(t->e2a, t->d256a, t->d256a, B * 256, ss));
                T_TRY(t_gemm_tn(gt(t->e1, 512, 0, t->d256a, 256, 0, c.g(pm.esm2.k), 256, c.g(pm.esm2.b), B, 512, 256), ss));
                T_TRY(t_gemm(ga(t->d256a, 256, c.w(pm.esm2.k), 256, nullptr, t->d512a, 512, 0, B, 256, 512), true, ss));
                T_TRY(t_dropout(t->d512a, B, 512, 512, 0, seed, 24 + 1, rt.dense, ss));
            }
            if (bp != kBnWhole) {
                // struct branch: d(concat[:,192:320]) -> dropout -> BN -> swish -> dense.  (Its gradient lands in d512a, which
                // the ESM BatchNorm still reads: in two parts the sums of both come first.)
                if (part0 == 0) {
                    T_TRY(t_copy_cols(t->d128, 128, 0, t->dconcat, kConcat, 192, B, 128, false, ss));
                    T_TRY(t_dropout(t->d128, B, 128, 128, 0, seed, 24 + 0, rt.dense, ss));
                    if ((r = bn_train_bwd(cs, t->d512a, 512, 0, t->ey, t->d512b, B, 512, t->emi, pm.esm1.bn, kBnSums))) return r;
                    if ((r = bn_train_bwd(cs, t->d128, 128, 0, t->sy, t->d512a, B, 128, t->smi, pm.strct.bn, kBnSums))) return r;
                    return DPTM_OK;
                }
                if ((r = bn_train_bwd(cs, t->d512a, 512, 0, t->ey, t->d512b, B, 512, t->emi, pm.esm1.bn, kBnApply))) return r;
                T_TRY(t_swish_bwd(t->ea, t->d512b, t->d512b, B * 512, ss));
                T_TRY(t_gemm_tn(gt(esm, kEsm, 0, t->d512b, 512, 0, c.g(pm.esm1.k), 512, c.g(pm.esm1.b), B, kEsm, 512), ss));
                if ((r = bn_train_bwd(cs, t->d128, 128, 0, t->sy, t->d512a, B, 128, t->smi, pm.strct.bn, kBnApply))) return r;
                T_TRY(t_swish_bwd(t->sa, t->d512a, t->d512a, B * 128, ss));
                T_TRY(t_gemm_tn(gt(strct, 8, 0, t->d512a, 128, 0, c.g(pm.strct.k), 128, c.g(pm.strct.b), B, 8, 128), ss));
                return DPTM_OK;
            }
            if ((r = bn_train_bwd(cs, t->d512a, 512, 0, t->ey, t->d512b, B, 512, t->emi, pm.esm1.bn))) return r;
            T_TRY(t_swish_bwd(t->ea, t->d512b, t->d512b, B * 512, ss));
            T_TRY(t_gemm_tn(gt(esm, kEsm, 0, t->d512b, 512, 0, c.g(pm.esm1.k), 512, c.g(pm.esm1.b), B, kEsm, 512), ss));
            T_TRY(t_copy_cols(t->d128, 128, 0, t->dconcat, kConcat, 192, B, 128, false, ss));
            T_TRY(t_dropout(t->d128, B, 128, 128, 0, seed, 24 + 0, rt.dense, ss));
            if ((r = bn_train_bwd(cs, t->d128, 128, 0, t->sy, t->d512a, B, 128, t->smi, pm.strct.bn))) return r;
            T_TRY(t_swish_bwd(t->sa, t->d512a, t->d512a, B * 128, ss));
            T_TRY(t_gemm_tn(gt(strct, 8, 0, t->d512a, 128, 0, c.g(pm.strct.k), 128, c.g(pm.strct.b), B, 8, 128), ss));
            return DPTM_OK;
        };
        // d(concat) is complete: the sequence branches go backward side by side with the ESM and structure layers
        // (each owns its parameters' gradients)
        if (packed && (rc = site_layers_bwd(0, 0))) return rc;
        if ((rc = branches(false, 0, 2))) return rc;
        if ((rc = packed ? site_layers_bwd(1, 1) : site_layers_bwd(0, 1))) return rc;
    }
    if ((rc = join(1))) return rc;
    if (st != caller) {
        T_TRY(hipEventRecord(t->ev_caller[1], st));
        T_TRY(hipStreamWaitEvent(caller, t->ev_caller[1], 0));
    }
    return DPTM_OK;
}

// Forward in training mode + loss + backward for one local batch (device pointers).  With
// DPTM_TRAIN_GRAPH=1 the launch sequence (about 300 small kernels over four streams) is captured into a
// graph the second time a call repeats the previous call's batch size, buffers, dropout rates, l2 and
// stream, and replayed from then on.  Off by default: on this stack eager issue is the faster of the two.
static int forward_backward_on(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                               const float* strct, const float* esm, const int32_t* labels, int64_t n,
                               const dptm_train_opts* o, hipStream_t st) {
    const float ng = o->n_global > 0 ? (float)o->n_global : (float)n;
    t->n_local = n;
    t->n_global = o->n_global > 0 ? o->n_global : n;
    // a replayed graph cannot carry this call's scalars: a launch of its own writes them in front of it
    if (t->use_graph) T_TRY(t_set_step(t->d_step, o->seed, ng, (float)o->n_pos_global, st));
    dptm_trainer::GraphKey key;
    key.n = n;
    const void* ptrs[6] = {tok21, tok33, tok51, strct, esm, labels};
    std::memcpy(key.ptr, ptrs, sizeof ptrs);
    const float rates[4] = {o->rate_embed, o->rate_conv, o->rate_tb, o->rate_dense};
    std::memcpy(key.rate, rates, sizeof rates);
    key.l2 = o->l2;
    key.l2_scale = o->l2_scale;
    key.st = st;
    int rc = DPTM_OK;
    if (t->ar_fn) {
        // the callback runs on the host between launches, so this sequence is never captured into a graph
        rc = enqueue_forward_backward(t, tok21, tok33, tok51, strct, esm, labels, n, o, st);
        if (rc) return rc;
    } else if (t->use_graph && t->gexec && key == t->gkey) {
        T_TRY(hipGraphLaunch(t->gexec, st));
    } else {
        if (!(key == t->gkey)) {
            if (t->gexec) {
                (void)hipGraphExecDestroy(t->gexec);
                t->gexec = nullptr;
            }
            t->gkey = key;
            t->gkey_hits = 0;
        }
        const bool capture = t->use_graph && t->gkey_hits >= 1;   // the first call of a configuration runs eagerly
        t->gkey_hits += 1;
        if (capture) {
            T_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            rc = enqueue_forward_backward(t, tok21, tok33, tok51, strct, esm, labels, n, o, st);
            hipGraph_t graph = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            if (e != hipSuccess || !graph) return tfail(DPTM_EHIP, "dptm_trainer_forward_backward: stream capture failed");
            const hipError_t ei = hipGraphInstantiate(&t->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) {
                t->gexec = nullptr;
                return tfail(DPTM_EHIP, "dptm_trainer_forward_backward: graph instantiation failed");
            }
            T_TRY(hipGraphLaunch(t->gexec, st));
        } else {
            rc = enqueue_forward_backward(t, tok21, tok33, tok51, strct, esm, labels, n, o, st);
            if (rc) return rc;
        }
    }
    t->last_n = n;
    return DPTM_OK;
}

int dptm_trainer_forward_backward(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                                  const float* strct, const float* esm, const int32_t* labels, int64_t n,
                                  const dptm_train_opts* o, float* loss_out, void* stream) try {
    if (!t || !o) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward: NULL argument");
    if (n < 1 || n > t->max_batch) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward: batch size outside 1..max_batch");
    if (!tok21 || !tok33 || !tok51 || !strct || !esm || !labels) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward: NULL buffer");
    T_TRY(hipSetDevice(t->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : t->stream;
    const int rc = forward_backward_on(t, tok21, tok33, tok51, strct, esm, labels, n, o, st);
    if (rc) return rc;
    if (loss_out) {
        T_TRY(hipMemcpyAsync(loss_out, t->d_loss, sizeof(float), hipMemcpyDeviceToHost, st));
        T_TRY(hipStreamSynchronize(st));
    }
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_forward_backward");
}

// The step on rows[0..n) of a data set that lives on the device: one gather launch into the trainer's staging buffers,
// the step, one launch that leaves the loss and the class probabilities at the caller's device addresses.  Nothing
// here waits for the GPU.
int dptm_trainer_forward_backward_rows(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                                       const float* strct, const float* esm, const int32_t* labels, int64_t n_data,
                                       const int64_t* rows, int64_t n, const dptm_train_opts* o, float* loss_dev_out,
                                       float* proba_dev_out, void* stream) try {
    if (!t || !o || !rows) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward_rows: NULL argument");
    if (n < 1 || n > t->max_batch) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward_rows: batch size outside 1..max_batch");
    if (n_data < 1) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward_rows: empty data set");
    if (!tok21 || !tok33 || !tok51 || !strct || !esm || !labels) return tfail(DPTM_EINVAL, "dptm_trainer_forward_backward_rows: NULL buffer");
    T_TRY(hipSetDevice(t->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : t->stream;
    const size_t B = (size_t)t->max_batch;
    const size_t off_struct = B * 1280 * sizeof(float), off_lab = off_struct + B * 8 * sizeof(float);
    const size_t off21 = off_lab + B * sizeof(int32_t), off33 = off21 + ((B * 21 + 15) & ~(size_t)15);
    const size_t off51 = off33 + ((B * 33 + 15) & ~(size_t)15), total = off51 + B * 51;
    if (!t->stage && hipMalloc(&t->stage, total) != hipSuccess) {
        t->stage = nullptr;
        return tfail(DPTM_ENOMEM, "dptm_trainer_forward_backward_rows: staging buffers");
    }
    char* base = static_cast<char*>(t->stage);
    GatherRowsArgs g;
    g.tok21 = tok21; g.tok33 = tok33; g.tok51 = tok51; g.strct = strct; g.esm = esm; g.labels = labels;
    g.rows = rows; g.n_data = n_data;
    g.esm_out = reinterpret_cast<float*>(base);
    g.strct_out = reinterpret_cast<float*>(base + off_struct);
    g.labels_out = reinterpret_cast<int32_t*>(base + off_lab);
    g.tok21_out = reinterpret_cast<uint8_t*>(base + off21);
    g.tok33_out = reinterpret_cast<uint8_t*>(base + off33);
    g.tok51_out = reinterpret_cast<uint8_t*>(base + off51);
    T_TRY(t_gather_rows(g, n, st));
    const int rc = forward_backward_on(t, g.tok21_out, g.tok33_out, g.tok51_out, g.strct_out, g.esm_out, g.labels_out, n, o, st);
    if (rc) return rc;
    T_TRY(t_step_outputs(t->d_loss, loss_dev_out, t->proba, proba_dev_out, n, st));
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_forward_backward_rows");
}

// Arms the per-kernel clocks: until dptm_trainer_profile_end the step's branches run one after another on one
// stream (so that a duration is the kernel's own and not its share of a GPU it co-runs on) and every
// sequence-branch launch is bracketed by an event pair on that stream.
int dptm_trainer_profile_begin(dptm_trainer* t) try {
    if (!t) return tfail(DPTM_EINVAL, "dptm_trainer_profile_begin: NULL trainer");
    if (t->use_graph) return tfail(DPTM_EINVAL, "dptm_trainer_profile_begin: not available with graph replay (DPTM_TRAIN_GRAPH)");
    T_TRY(hipSetDevice(t->device));
    for (int k = 0; k < DPTM_TRAIN_PROFILE_SLOTS; ++k) {
        for (int e = 0; e < 2; ++e)
            if (!t->prof_ev[k][e]) T_TRY(hipEventCreate(&t->prof_ev[k][e]));
        t->prof_used[k] = false;
    }
    if (!t->prof_on) t->prof_serial_before = t->serial;
    t->serial = true;
    t->prof_on = true;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_profile_begin");
}

// Waits for the device and writes the milliseconds of the LAST step's launches (0 for a slot the step did not
// use: batches of at most 192 sites take other kernels for the per-site layers, never for the branches).
int dptm_trainer_profile_end(dptm_trainer* t, float* ms_out) try {
    if (!t || !ms_out) return tfail(DPTM_EINVAL, "dptm_trainer_profile_end: NULL argument");
    if (!t->prof_on) return tfail(DPTM_EINVAL, "dptm_trainer_profile_end: no profile armed");
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipDeviceSynchronize());
    t->prof_on = false;
    t->serial = t->prof_serial_before;
    for (int k = 0; k < DPTM_TRAIN_PROFILE_SLOTS; ++k) {
        ms_out[k] = 0.f;
        if (t->prof_used[k]) T_TRY(hipEventElapsedTime(&ms_out[k], t->prof_ev[k][0], t->prof_ev[k][1]));
    }
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_profile_end");
}

int dptm_trainer_set_allreduce(dptm_trainer* t, dptm_allreduce_fn fn, void* user) try {
    if (!t) return tfail(DPTM_EINVAL, "dptm_trainer_set_allreduce: NULL trainer");
    t->ar_fn = fn;
    t->ar_user = user;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_set_allreduce");
}

// tensorflow-addons AdamW step on the flat buffers (gradients are multiplied by grad_scale first).
int dptm_trainer_apply(dptm_trainer* t, float lr, float weight_decay, float grad_scale, float stat_scale, void* stream) try {
    if (!t) return tfail(DPTM_EINVAL, "dptm_trainer_apply: NULL trainer");
    T_TRY(hipSetDevice(t->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : t->stream;
    t->step += 1;
    const double b1 = 0.9, b2 = 0.999;
    const float lr_t = (float)(lr * std::sqrt(1.0 - std::pow(b2, (double)t->step)) / (1.0 - std::pow(b1, (double)t->step)));
    T_TRY(t_adamw(t->W, t->G, t->M, t->V, t->trainable, stat_scale, t->pm.total, lr_t, weight_decay, (float)b1, (float)b2, 1e-7f,
                  grad_scale, st));
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_apply");
}

static int copy_out(dptm_trainer* t, const float* dev, float* host_canonical) {
    std::vector<float> host((size_t)t->pm.total);
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipStreamSynchronize(t->stream));
    T_TRY(hipDeviceSynchronize());
    T_TRY(hipMemcpy(host.data(), dev, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (const Tn& x : t->pm.t) std::memcpy(host_canonical + x.canon, host.data() + x.off, (size_t)x.size * sizeof(float));
    return DPTM_OK;
}

// Weights / gradients in the canonical blob order (host buffers of dptm_weight_count floats).
int dptm_trainer_get_weights(dptm_trainer* t, float* host_out) try {
    if (!t || !host_out) return tfail(DPTM_EINVAL, "dptm_trainer_get_weights: NULL argument");
    return copy_out(t, t->W, host_out);
} catch (...) {
    return on_texception("dptm_trainer_get_weights");
}
int dptm_trainer_get_grads(dptm_trainer* t, float* host_out) try {
    if (!t || !host_out) return tfail(DPTM_EINVAL, "dptm_trainer_get_grads: NULL argument");
    return copy_out(t, t->G, host_out);
} catch (...) {
    return on_texception("dptm_trainer_get_grads");
}
static int copy_in(dptm_trainer* t, float* dev, const float* host_canonical) {
    std::vector<float> host((size_t)t->pm.total, 0.0f);
    for (const Tn& x : t->pm.t) std::memcpy(host.data() + x.off, host_canonical + x.canon, (size_t)x.size * sizeof(float));
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipDeviceSynchronize());
    T_TRY(hipMemcpy(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return DPTM_OK;
}

int dptm_trainer_get_optimizer_state(dptm_trainer* t, float* m_host_out, float* v_host_out, int64_t* step_out) try {
    if (!t || !m_host_out || !v_host_out || !step_out) return tfail(DPTM_EINVAL, "dptm_trainer_get_optimizer_state: NULL argument");
    int rc = copy_out(t, t->M, m_host_out);
    if (rc) return rc;
    if ((rc = copy_out(t, t->V, v_host_out))) return rc;
    *step_out = t->step;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_get_optimizer_state");
}
int dptm_trainer_set_optimizer_state(dptm_trainer* t, const float* m_host, const float* v_host, int64_t step) try {
    if (!t || !m_host || !v_host || step < 0) return tfail(DPTM_EINVAL, "dptm_trainer_set_optimizer_state: bad argument");
    int rc = copy_in(t, t->M, m_host);
    if (rc) return rc;
    if ((rc = copy_in(t, t->V, v_host))) return rc;
    t->step = step;
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_set_optimizer_state");
}
int dptm_trainer_get_proba(dptm_trainer* t, float* host_out, int64_t n) try {
    if (!t || !host_out || n < 0 || n > t->max_batch) return tfail(DPTM_EINVAL, "dptm_trainer_get_proba: bad argument");
    T_TRY(hipSetDevice(t->device));
    T_TRY(hipDeviceSynchronize());
    T_TRY(hipMemcpy(host_out, t->proba, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return DPTM_OK;
} catch (...) {
    return on_texception("dptm_trainer_get_proba");
}

}  // extern "C"
