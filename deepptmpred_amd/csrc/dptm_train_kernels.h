// Launcher interface of the training kernels (dptm_train_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dptm {

struct GemmArgs {
    const float* x;      // [m][ldx]
    int64_t ldx;
    const float* w;      // NN: [k][ldw] ; NT: [cols][ldw]
    int64_t ldw;
    const float* bias;   // [cols] or nullptr
    float* out;          // [m][ldo], written at column col_off
    int64_t ldo;
    int col_off;
    int64_t m;
    int k;               // contraction length (multiple of 8)
    int cols;            // output columns (multiple of 64)
    int shift;           // input row = output row + shift, inside the same window
    int wlen;            // window length of the row stack (0: rows are independent)
    int accumulate;      // out += result
    int taps;            // > 1: a 'same' convolution in one pass -- tap t uses the matrix at w + t*tap_stride and the
                         // row shift tap_dir * (t - taps/2); `shift` is ignored then
    int64_t tap_stride;
    int tap_dir;
    int64_t x_tap_stride;    // floats added to x per tap (taps with tap_dir 0: a sum of products over separate operands)
    const float* residual;   // optional [m][ldr]: added to the result (the same columns as out, without col_off)
    int64_t ldr;
    float* act_out;          // optional [m][lda]: swish(result) is stored here as well
    int64_t lda;
    int batch;               // > 1: that many independent products in one launch (grid.z); operand b of product z is
    int64_t bx, bw, bb, bo;  // x + z*bx, w + z*bw, bias + z*bb, out + z*bo
};

struct GemmTnArgs {
    const float* x;      // [m][ldx], columns x_col_off .. x_col_off+k
    int64_t ldx;
    int x_col_off;
    const float* dy;     // [m][ldy], columns dy_col_off .. dy_col_off+n
    int64_t ldy;
    int dy_col_off;
    float* dw;           // [k][ldw] accumulated with atomics
    int64_t ldw;
    float* db;           // [n] or nullptr
    int64_t m;
    int k;
    int n;               // multiple of 32
    int shift;
    int wlen;
    int64_t rows_per_slab;   // set by the launcher
    int slabs;               // set by the launcher
    int batch;               // > 1: independent products in one launch: x + z*bx, dy + z*bdy, dw + z*bdw, db + z*bdb
    int64_t bx, bdy, bdw, bdb;
    int bshift_dir;          // product z uses the row shift bshift_dir * (z - batch/2) instead of `shift` (taps of a convolution)
};

hipError_t t_gemm(const GemmArgs& a, bool nt, hipStream_t st);
hipError_t t_gemm_tn(GemmTnArgs a, hipStream_t st);
hipError_t t_swish_fwd(const float* a, float* y, int64_t n, hipStream_t st);
hipError_t t_swish_bwd(const float* a, const float* dy, float* da, int64_t n, hipStream_t st);
hipError_t t_add(float* dst, const float* src, int64_t n, hipStream_t st);
hipError_t t_copy_cols(float* dst, int64_t ldd, int dc, const float* src, int64_t lds_, int sc, int64_t rows,
                       int cols, bool add, hipStream_t st);
// Scalars that change every step live in a small device array so that a captured graph of the step can be
// replayed unchanged: [kStepSeed] dropout seed, [kStepNGlobal] / [kStepNPosGlobal] float bits of the global
// batch size and positive count of the loss weights.  t_set_step writes them (stream-ordered).
// [kStepL2 + s], s = [kStepL2Sel]: value of the l2 kernel regularisers of this step -- two slots taken in turn, so that
// the launch that fills one can clear the other for the next step.
enum StepParam { kStepSeed = 0, kStepNGlobal = 1, kStepNPosGlobal = 2, kStepL2 = 3, kStepL2Sel = 5, kStepParams = 6 };
struct StepVals { unsigned seed; float n_global, n_pos_global; };
// (a launch of its own: only where the step's launches are replayed from a graph; otherwise t_pack writes them)
hipError_t t_set_step(unsigned* step_params, unsigned seed, float n_global, float n_pos_global, hipStream_t st);
// x = dropout(src ? src : x); with rate 0 and a src this is a plain copy
hipError_t t_dropout(float* x, int64_t rows, int cols, int64_t ld, int wlen_spatial, const unsigned* step_params, unsigned stream,
                     float rate, hipStream_t st, const float* src = nullptr);
hipError_t t_embed_fwd(const uint8_t* tok, const float* E, const float* P, float* x, int64_t rows, int wlen, hipStream_t st);
hipError_t t_embed_bwd(const uint8_t* tok, const float* dx, float* dE, float* dP, int64_t rows, int wlen, hipStream_t st);
hipError_t t_colsum2(const float* x, const float* y, int64_t rows, int cols, float* stats, hipStream_t st);
// stats: [sum | sum of squares] of the columns of y over rows_global rows (the caller zeroes and fills it)
hipError_t t_bn_fwd(const float* y, float* z, int64_t rows, int cols, const float* stats, int64_t rows_global, float* mean_inv,
                    float* batch_mean, float* batch_var, const float* gamma, const float* beta, int64_t ldz, int z_col_off,
                    hipStream_t st);
// sums must be zero on entry
hipError_t t_bn_bwd_reduce(const float* dz, int64_t lddz, int dz_col_off, const float* y, int64_t rows, int cols,
                           const float* mean_inv, float* sums, hipStream_t st);
hipError_t t_bn_bwd_apply(const float* dz, int64_t lddz, int dz_col_off, const float* y, float* dy, int64_t rows, int cols,
                          const float* mean_inv, const float* gamma, const float* sums, int64_t rows_global,
                          const float* own_sums, float* dgamma, float* dbeta, hipStream_t st);
hipError_t t_ln_fwd(const float* a, const float* b, const float* g, const float* beta, float* out, float* xhat,
                    float* inv, int64_t rows, hipStream_t st);
hipError_t t_ln_bwd(const float* dy, const float* xhat, const float* inv, const float* g, float* dsum, float* dg,
                    float* dbeta, int64_t rows, hipStream_t st);
hipError_t t_attn_fwd(const float* q, const float* k, const float* v, float* o, float* prob, int64_t samples, int wlen, hipStream_t st);
hipError_t t_attn_bwd(const float* q, const float* k, const float* v, const float* prob, const float* d_o, float* dq,
                      float* dk, float* dv, int64_t samples, int wlen, hipStream_t st);
hipError_t t_pool_fwd(const float* x, const float* w, const float* bias, float* out, int64_t ldo, int col_off, float* a_save,
                      float* alpha_save, int64_t samples, int wlen, hipStream_t st);
hipError_t t_pool_bwd(const float* x, const float* w, const float* dout, int64_t ldd, int col_off, const float* a_save,
                      const float* alpha_save, float* dx, float* dw, float* db, int64_t samples, int wlen, hipStream_t st);
struct GatherRowsArgs {
    const uint8_t *tok21, *tok33, *tok51;
    const float *strct, *esm;
    const int32_t* labels;
    const int64_t* rows;
    int64_t n_data;
    uint8_t *tok21_out, *tok33_out, *tok51_out;
    float *strct_out, *esm_out;
    int32_t* labels_out;
};
hipError_t t_gather_rows(const GatherRowsArgs& a, int64_t n, hipStream_t st);
hipError_t t_step_outputs(const float* loss, float* loss_out, const float* proba, float* proba_out, int64_t n, hipStream_t st);
hipError_t t_head_loss(const float* x, const float* W, const float* bias, const int32_t* labels, int64_t n,
                       const unsigned* step_params, float* proba, float* loss, float* dx, float* dW, float* db, hipStream_t st);
hipError_t t_adamw(float* w, const float* g, float* m, float* v, const uint8_t* trainable, float stat_scale, int64_t n,
                   float lr_t, float wd, float b1, float b2, float eps, float gscale, hipStream_t st);


// ---- fused sequence-branch kernels (dptm_train_branch.hip)
typedef float f32x4_tr __attribute__((ext_vector_type(4)));
typedef float f32x2_tr __attribute__((ext_vector_type(2)));

// One entry of the per-step weight-image build (MFMA fragment order, see dptm_layout.h):
//   kind 0  packed matrix of the logical M [K][N]: M = src (raw [K][N]) or, with transpose, M[k][n] = src[n][k]
//   kind 1  packed feature vector of K features
//   kind 2  plain copy of K floats
struct PackJob {
    const float* src;
    float* dst;
    int kind, K, N, transpose;
};
// lambda: coefficient of the l2 kernel regularisers (kind 4 jobs); step_params[kStepL2 + l2_slot] collects lambda * sum w^2
// and the other slot is cleared.  sv: the step's scalars, written by the same launch (NULL: t_set_step has written them).
hipError_t t_pack(const PackJob* jobs_dev, int n_jobs, float lambda, unsigned* step_params, int l2_slot, const StepVals* sv,
                  hipStream_t st);

// transposed weight images of one TransformerBlock for the backward kernel, offsets in float4 units
constexpr int kBtWqT = 0;                    // Wq^T, Wk^T, Wv^T, Wo^T: packed 64 x 64
constexpr int kBtWkT = kBtWqT + 1024;
constexpr int kBtWvT = kBtWkT + 1024;
constexpr int kBtWoT = kBtWvT + 1024;
constexpr int kBtW2T = kBtWoT + 1024;        // W2^T as a [64][256] matrix
constexpr int kBtW1T = kBtW2T + 4096;        // W1^T as a [256][64] matrix
constexpr int kBtSize = kBtW1T + 4096;

// A tile of the fused kernels = S whole sites (S * wlen rows) handled by NW wavefronts of 32 rows each.
struct TileCfg {
    int S, NW;
    int64_t ntiles;
};
TileCfg t_tile_cfg(int64_t n_sites, int wlen);
// rows (incl. padding) that the chain-tiled activation arrays need for any batch up to max_sites
int64_t t_tile_rows_max(int64_t max_sites, int wlen);

// Saved activations of one TransformerBlock in "chain-tiled" order: float4 index ((tile*NW + wave)*8 + u)*64 + lane
// holds features 8u + 4(lane>>5) .. +3 of row wave*32 + (lane&31) of the tile; per-lane arrays use
// (tile*NW + wave)*64 + lane.
struct TbSave {
    f32x4_tr *q, *k, *v, *o, *xh1, *xh2;
    f32x4_tr* stat;        // softmax statistics of heads 2(lane>>5), 2(lane>>5)+1 of the lane's row: {max, 1/den} x 2
    f32x2_tr* inv;         // 1/sigma of the two LayerNormalizations
};

// Activations of the convolution block kept for the backward pass, chain-tiled like TbSave: the shortcut
// convolution s and the pre-activations a1, a2 of the two 3-tap convolutions (model.py:130-138).
struct ConvSave {
    f32x4_tr *s, *a1, *a2;
};
// One BatchNormalization in training mode as the fused kernels see it: column sums [sum | sum of squares] of the
// layer's input over the whole batch (complete before the kernel starts), the row count they cover, the raw
// gamma / beta vectors, and where the first workgroup leaves the batch mean / variance for the moving statistics.
struct BnFwd {
    const float* stats;
    float inv_rows;
    const float *gamma, *beta;
    float *batch_mean, *batch_var;
};
// The same layer in the backward pass: `sums` = [sum dz | sum dz*y] over the whole batch (y the layer's input;
// sum dz*xhat follows as inv * (sum dz*y - mean * sum dz)), `own` this replica's share (the first workgroup adds
// it to the gamma / beta gradients).
struct BnBwd {
    const float* stats;
    float inv_rows;
    const float* gamma;
    const float* beta;
    const float *sums, *own;
    float *dgamma, *dbeta;
};

struct TbFwdArgs {
    const f32x4_tr* P;     // packed branch image (dptm_layout.h), built from this step's weights
    ConvSave cs;           // input of the first block: x1 = BN2(swish(a2)) + s
    BnFwd bn2;
    TbSave sv[2];
    f32x2_tr* pool;        // per lane: {tanh score, pooling weight alpha}
    float* out;            // pooled rows: out[site*ldo + col_off + f]
    int64_t ldo;
    int col_off;
    int64_t n_sites;
    int S;
    int64_t ntiles;
    const unsigned* step;  // per-step scalars (dropout seed)
    unsigned stream_base;  // dropout streams: +2*blk attention output, +2*blk+1 feed-forward output
    unsigned thr;          // keep threshold (0: no dropout)
    float scale;           // 1 / (1 - rate)
};
hipError_t t_tb_fwd(const TbFwdArgs& a, int wlen, int NW, hipStream_t st);
// a tile of ONE site in FOUR wavefronts runs the kernels of dptm_train_pair.h: two wavefronts share a 32-row block
bool t_pair_tile(int S, int NW);

// partial parameter gradients of one workgroup of the backward kernel (floats):
//   24 weight-gradient units of 4096 (MFMA accumulator order) | 30 vectors of 64
constexpr int kTbUnits = 24;
constexpr int kTbVecs = 30;
constexpr int kTbPartFloats = kTbUnits * 4096 + kTbVecs * 64 + 128;   // 100,352: a multiple of 256
// A workgroup's partial buffer has a region per backward launch of the branch -- TransformerBlocks | second convolution
// (3 tap units, bias) | first convolution (3 tap units, shortcut, embedding table; two bias vectors; position table) --
// so that ONE launch sums all of them after the last backward kernel.
constexpr int kMaxWin = 51;
constexpr int kCbBase = kTbPartFloats, kCbFloats = 3 * 4096 + 256;
constexpr int kCaBase = kCbBase + kCbFloats, kCaFloats = 5 * 4096 + 256 + kMaxWin * 64;
constexpr int kTbPartStride = (kCaBase + kCaFloats + 255) / 256 * 256;
constexpr int kTbMaxParts = 256;
struct TbBwdArgs {
    const f32x4_tr* P;
    const f32x4_tr* PT;    // [2][kBtSize] transposed images
    ConvSave cs;
    BnFwd bn2;             // (batch_mean / batch_var unused)
    float* bn2_sums;       // [sum d(x1) | sum d(x1) * swish(a2)] of this replica, added with atomics (zeroed per step)
    TbSave sv[2];
    const f32x2_tr* pool;
    const float* dout;     // gradient of the pooled rows: dout[site*ldd + col_off + f]
    int64_t ldd;
    int col_off;
    f32x4_tr* dx1;         // gradient w.r.t. x1, chain-tiled
    float* part;           // [grid][kTbPartStride]
    unsigned* queue;       // tile counter of this launch, zero when the launch starts (see next_tile)
    int64_t n_sites;
    int S;
    int64_t ntiles;
    const unsigned* step;
    unsigned stream_base, thr;
    float scale;
    // ---- the split form of large batches (t_tb_bwd decides): the row-wise half of every block -- LayerNorm 2, feed-forward
    // layers, LayerNorm 1 -- runs as tb_row_bwd_kernel, this kernel does the pooling (part 1) or the attention half of
    // block `blk` (part 2), and the gradient travels between the launches through two chain-tiled arrays
    int blk;               // part 2: the block in hand
    const f32x4_tr* gin;   // part 2: d(x + attention output) of the block, from the row-wise kernel
    f32x4_tr* gout;        // part 1, part 2 with blk = 1: gradient w.r.t. the output of the block below
    f32x4_tr* dtmp;        // second hand-over array (free while the TransformerBlocks go backward)
    float* row_part;       // [2 blocks][row_parts][kRowPartStride] partial gradients of the row-wise launches (may be NULL)
    int row_parts;         // workgroups a row-wise launch may start (<= kRowMaxParts)
    unsigned* queues;      // five tile counters for the split form's launches, zero when the step starts
    // One-site tiles may form the gradient of their pooled row themselves instead of reading dout (which then needs no
    // launch of its own in front of this one): dout[f] = dropout(sum_n up_da[site][n] * up_w[(col_off + f) * up_n + n]),
    // the layer above being a dense layer of up_n units whose input dropout (element index site * up_cols + column) is
    // up_stream / up_thr / up_scale.  up_da == nullptr: read dout.
    const float *up_da, *up_w;
    int up_n, up_cols;
    unsigned up_stream, up_thr;
    float up_scale;
};
// Row-wise half of one TransformerBlock's backward pass (no site structure: tiles of any 128 rows).
constexpr int kRowUnits = 8;       // dW1 chunks 0..3, dW2 chunks 0..3
constexpr int kRowVecs = 10;       // LayerNorm 2 gamma / beta, b2, b1 chunks 0..3, LayerNorm 1 gamma / beta, (spare)
constexpr int kRowPartStride = kRowUnits * 4096 + kRowVecs * 64;      // 33,408 floats, a multiple of 64
constexpr int kRowMaxParts = 512;
struct TbRowBwdArgs {
    const f32x4_tr* P;     // packed image of the block (p.P + kOffTb + blk * kTbSize)
    const f32x4_tr* PT;    // transposed images of the block (p.PT + blk * kBtSize)
    const f32x4_tr *xh1, *xh2;
    const f32x2_tr* inv;
    const f32x4_tr* din;   // gradient w.r.t. the block's output
    f32x4_tr* dout;        // d(x + attention output): input of the attention half
    float* part;           // [grid][kRowPartStride]
    unsigned* queue;
    int64_t n_sites;       // validity of a row: the forward tiling (S sites of wlen rows in NWf wavefronts per tile)
    int S, wlen, NWf;
    int64_t nblocks;       // 32-row blocks of the chain-tiled arrays (forward tiles x NWf)
    const unsigned* step;
    unsigned stream, thr;  // dropout behind the feed-forward layers
    float scale;
};
// gradient tensors (raw layouts) that the partial sums are added to
struct TbGrads {
    float *wq[2], *bq[2], *wk[2], *bk[2], *wv[2], *bv[2], *wo[2], *bo[2];
    float *w1[2], *b1[2], *w2[2], *b2[2], *ln1g[2], *ln1b[2], *ln2g[2], *ln2b[2];
    float *poolw, *poolb;
};
struct PartPlan;       // below: what the branch's backward launches leave to be summed
hipError_t t_tb_bwd(const TbBwdArgs& a, const TbGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st);

// Sum of the workgroups' partial buffers (stride kTbPartStride) into gradient tensors: weight-gradient units
// (unit u at float u*4096 of a buffer, MFMA accumulator order; element (k, n) goes to dst[u][k*ld[u] + n]) and
// plain vectors (len floats at float off of a buffer).  Results are ADDED to the destinations.
constexpr int kPartMaxUnits = 32, kPartMaxVecs = 36;
struct PartUnits {
    int n;
    float* dst[kPartMaxUnits];
    int off[kPartMaxUnits];     // float offset of the unit inside a buffer
    int ld[kPartMaxUnits];
    int rows[kPartMaxUnits];    // rows k >= rows[u] of a unit are dropped (64: the whole unit)
    PartUnits() : n(0) {
        for (int i = 0; i < kPartMaxUnits; ++i) { dst[i] = nullptr; off[i] = i * 4096; ld[i] = 64; rows[i] = 64; }
    }
};
struct PartVecs {
    int n;
    float* dst[kPartMaxVecs];
    int off[kPartMaxVecs], len[kPartMaxVecs];
    PartVecs() : n(0) {}
};
// What the backward launches of one branch leave to be summed: each launch appends its units and vectors.  A step with a
// tile per workgroup is a chain of short launches, and ONE summing launch after the last backward kernel (the first
// convolution's) keeps two launches off that chain; with many tiles per workgroup every kernel's sums are added right
// behind it, where the launch overlaps with the other branches instead of ending the step (part_sums_deferred).
struct PartPlan {
    PartUnits units;
    PartVecs vecs;
};
inline bool part_sums_deferred(int64_t ntiles) { return ntiles <= kTbMaxParts; }
hipError_t t_reduce_parts(const float* part, int nparts, const PartUnits& u, const PartVecs& v, hipStream_t st,
                          int part_stride = kTbPartStride);

// ---- fused embedding + residual convolution block (dptm_train_conv.hip); transposed images of the convolution
// kernels follow the two blocks' images in PT: shortcut^T, conv3a taps^T, conv3b taps^T (packed 64 x 64 each)
constexpr int kBtConv = 2 * kBtSize;
constexpr int kBtScT = kBtConv;
constexpr int kBtC3aT = kBtScT + 1024;
constexpr int kBtC3bT = kBtC3aT + 3 * 1024;
constexpr int kBtTotal = kBtC3bT + 3 * 1024;

struct ConvArgs {
    const f32x4_tr* P;
    const f32x4_tr* PT;
    const uint8_t* tok;        // [sites][wlen] residue codes
    ConvSave cs;
    BnFwd bn1;                 // forward kernels
    float *stats1, *stats2;    // column sums a forward kernel adds to (atomics; zeroed per step)
    BnBwd bb1, bb2;            // backward kernels
    float* bn1_sums;           // [sum d(z1) | sum d(z1) * swish(a1)] of this replica (atomics; zeroed per step)
    const f32x4_tr* dx1;       // from the TransformerBlock backward kernel
    f32x4_tr* dz1;             // gradient behind the first BatchNorm + dropout, chain-tiled (conv_b_bwd -> conv_a_bwd)
    float* part;               // per-workgroup partial gradients, as in TbBwdArgs
    unsigned *queue_b, *queue_a;   // tile counters of conv_b_bwd / conv_a_bwd, zero when the launch starts
    int64_t n_sites;
    int S;
    int64_t ntiles;
    const unsigned* step;
    unsigned stream_embed, thr_embed, stream_conv, thr_conv;   // spatial dropouts: one mask value per (site, channel)
    float scale_embed, scale_conv;
};
struct ConvGrads {
    float *emb, *pos, *sck, *scb, *c3ak, *c3ab, *c3bk, *c3bb;
};
hipError_t t_conv_a_fwd(const ConvArgs& a, int wlen, int NW, hipStream_t st);   // tokens -> s, a1, stats1
hipError_t t_conv_b_fwd(const ConvArgs& a, int wlen, int NW, hipStream_t st);   // a1 -> a2, stats2
hipError_t t_conv_b_bwd(const ConvArgs& a, const ConvGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st);   // dx1 -> dz1, conv3b / BN2 gradients
hipError_t t_conv_a_bwd(const ConvArgs& a, const ConvGrads& g, int wlen, int NW, PartPlan* plan, hipStream_t st);   // dz1, dx1 -> conv3a / shortcut / BN1 / embedding gradients


// ---- the same launch of the three sequence branches (windows 21 | 33 | 51, in that order) as ONE launch, for batches
// with a tile per workgroup in every branch (tiles of NW = 4 or 8 wavefronts): such a batch's step is then a chain of
// launches on ONE stream, with no hand-over between streams on its critical path.  t_conv_a_bwd3 ends with the one
// launch that adds the three branches' partial sums to the gradients.
hipError_t t_conv_a_fwd3(const ConvArgs (&a)[3], int NW, hipStream_t st);
hipError_t t_conv_b_fwd3(const ConvArgs (&a)[3], int NW, hipStream_t st);
hipError_t t_tb_fwd3(const TbFwdArgs (&a)[3], int NW, hipStream_t st);
hipError_t t_tb_bwd3(const TbBwdArgs (&a)[3], const TbGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st);
hipError_t t_conv_b_bwd3(const ConvArgs (&a)[3], const ConvGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st);
hipError_t t_conv_a_bwd3(const ConvArgs (&a)[3], const ConvGrads (&g)[3], int NW, PartPlan (&plan)[3], hipStream_t st);
hipError_t t_reduce_parts3(const float* const (&part)[3], const int (&nparts)[3], const PartPlan (&plan)[3], hipStream_t st);

// ---- per-site dense layers for small batches (dptm_train_site.hip)
constexpr int kSiteMaxRows = 256;       // what the kernels can take
constexpr int kSiteFusedRows = 192;     // what a step gives them: a layer is N / 32 workgroups however many rows there are, so
                                        // beyond this the layer-wise kernels, which spread over the chip, are the faster form
// inverted dropout on the stateless hash: element (row r, column c) has index r*cols + col0 + c; thr 0: none
struct SiteDrop {
    unsigned stream, thr;
    float scale;
    int cols, col0;
};
struct SiteFwdArgs {
    const float* x;          // [B][ldx], K columns used, in_drop applied on load
    int64_t ldx;
    int K;
    SiteDrop in_drop;
    const float *W, *bias;   // raw [K][N], [N]
    int N;
    float* a;                // [B][N] pre-activation, kept for the backward pass
    int bn;                  // BatchNormalization (batch statistics) behind the swish
    const float *gamma, *beta;
    float* mi;               // [mean | 1/sigma] of the batch, 2N floats
    float *batch_mean, *batch_var;
    SiteDrop out_drop;
    float* out;              // out[r*ldo + col_off + c]
    int64_t ldo;
    int col_off;
    int B;
    const unsigned* step;
};
hipError_t t_site_fwd(const SiteFwdArgs& a, hipStream_t st);
struct SiteBwdArgs {
    // gradient w.r.t. the layer's output as the next layer saw it: read from dzd[r*ld_dzd + c], or (dzd == nullptr)
    // formed as sum_n da_next[r][n] * W_next[(next_row_off + c)*Nn + n]
    const float* dzd;
    int64_t ld_dzd;
    const float *da_next, *W_next;
    int Nn, next_row_off;
    SiteDrop mid_drop;       // the next layer's input dropout (columns counted from this layer's column 0)
    float* gout;             // optional: the gradient behind mid_drop is stored here (gout[r*ld_gout + c])
    int64_t ld_gout;
    SiteDrop out_drop;       // this layer's own output dropout
    int bn;
    const float *gamma, *mi;
    const float* a;          // nullptr: no layer below the gradient (only gout is written)
    float* da;               // [B][N] gradient w.r.t. the pre-activation, read by the layer below
    const float* x;          // the layer's input, as in the forward pass
    int64_t ldx;
    int K;
    SiteDrop in_drop;
    float *dW, *db, *dgamma, *dbeta;   // gradients (raw layouts): dW is written, the vectors are added to
    const float* W;          // the layer's own kernel [K][N] and the coefficient of its l2 regulariser:
    float l2c;               // dW = x^T d(a) + l2c * W (l2c = 0: no regulariser, W is not read)
    int N, B;
    const unsigned* step;
    // 0: the whole layer.  The weight gradient is needed by nobody on the way down, so a step may cut the launch in two:
    // 1 = everything but dW (d(a), bias / BatchNorm gradients), 2 = dW alone from the d(a) that launch left, on any stream
    int part;
};
hipError_t t_site_bwd(const SiteBwdArgs& a, hipStream_t st);

}  // namespace dptm
