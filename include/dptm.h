/*
 * dptm.h -- C ABI of libdptm.so: DeepPTMPred's site-scoring path on MI355X (gfx950).
 *
 * The library replaces what the reference reaches through Keras at
 *   pred/train_PTM/predict.py:569-577  load_model(path, custom_objects=...)   -> dptm_create
 *   pred/train_PTM/predict.py:598-599  self.model.predict(model_inputs)[:, 1] -> dptm_score
 *   pred/train_PTM/case.py:468-476,495 (same two calls)
 *   pred/train_PTM/trainer.py:62       model.predict(X, batch_size=...)       -> dptm_score
 * and, for the fused form, the host-side feature assembly in front of that call:
 *   pred/train_PTM/predict.py:487-501  _prepare_esm_features   (ESM row gather)
 *   pred/train_PTM/predict.py:529-560  _get_sequence_windows + _encode_sequences
 *                                                                -> dptm_score_protein
 *
 * Conventions
 *   - Every entry point returns 0 on success and a negative DPTM_E* code on failure;
 *     dptm_last_error() returns a thread-local message for the last failure.
 *   - The caller owns every input and output buffer.  The library copies the weights
 *     at create time and owns all device memory it allocates.  No pointer outlives a
 *     call except the handle.
 *   - A handle is bound to one GPU and is not thread-safe.  One handle per GPU.
 *   - "tok" arrays hold residue codes 0..20 (predict.py:549-553; 20 = '*' padding and
 *     any unknown letter), one byte each, row-major [n][W].
 *   - Probabilities come back as float32 [n][2], rows in input order; column 1 is the
 *     "modified" class the reference thresholds at >= 0.5 (predict.py:599,609).
 */
#ifndef DPTM_H
#define DPTM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPTM_OK 0
#define DPTM_EINVAL (-1)   /* bad argument / unsupported spec / bad position */
#define DPTM_EHIP (-2)     /* a HIP runtime call failed */
#define DPTM_ENOMEM (-3)   /* host or device allocation failed */

#define DPTM_MAX_WINDOWS 3

/* Model dimensions (pred/train_PTM/config.py:49-76 = predict.py:163-169).  Only the
 * reference's shape -- windows {21,33,51}, 21 codes, embed 64, 4 heads, ff 256,
 * esm 1280, struct 8, units 128/512/256/512/256/2 -- is implemented; any other value
 * makes dptm_create fail with DPTM_EINVAL. */
typedef struct dptm_spec {
    int32_t n_windows;
    int32_t window_sizes[DPTM_MAX_WINDOWS];
    int32_t max_features;
    int32_t embed_dim;
    int32_t num_heads;
    int32_t ff_dim;
    int32_t esm_dim;
    int32_t struct_feature_dim;
    int32_t struct_units;
    int32_t esm_units1;
    int32_t esm_units2;
    int32_t trunk_units1;
    int32_t trunk_units2;
    int32_t n_classes;
} dptm_spec;

typedef struct dptm_handle dptm_handle;

/* Number of fp32 values dptm_create expects for `spec` (1,620,229 for the reference
 * shape), or a negative error code.  The blob is every weight tensor of
 * create_ptm_model (model.py:143-210) in the order documented in
 * deepptmpred_amd/spec.py:tensor_table, each C-contiguous in its Keras shape. */
int64_t dptm_weight_count(const dptm_spec* spec);

/* Replaces load_model (predict.py:569-577): builds a scorer on GPU `device` from a
 * host blob of `nbytes` bytes of little-endian fp32 weights. */
int dptm_create(const void* weights, size_t nbytes, const dptm_spec* spec, int device,
                dptm_handle** out);

void dptm_destroy(dptm_handle* h);

/* Replaces model.predict([x21, x33, x51, struct, esm]) (predict.py:598-599).  All
 * pointers are HOST buffers; the call returns after the results are in `out`.
 *   tok21/tok33/tok51: uint8 [n][21|33|51]; strct: float32 [n][8];
 *   esm: float32 [n][1280]; out: float32 [n][2].  n = 0 is allowed. */
int dptm_score(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
               const float* strct, const float* esm, int64_t n, float* out);

/* Fused form of prepare_protein_data + predict (predict.py:462-485, 598-599): window
 * assembly and the ESM row gather run on the device.  HOST buffers.
 *   codes: uint8 [L] residue codes of the protein sequence;
 *   esm_table: float32 [esm_rows][1280] per-residue table (may be NULL with
 *              esm_rows = 0: every site then gets a zero row, predict.py:498-500);
 *              a site whose position-1 >= esm_rows also gets zeros (predict.py:493);
 *   positions: int32 [n], 1-based; a position outside 1..L fails with DPTM_EINVAL
 *              (the reference raises IndexError at predict.py:535);
 *   strct: float32 [n][8]; out: float32 [n][2]. */
int dptm_score_protein(dptm_handle* h, const uint8_t* codes, int64_t L, const float* esm_table,
                       int64_t esm_rows, const int32_t* positions, const float* strct, int64_t n,
                       float* out);

/* Device-resident forms: every pointer is DEVICE memory on the handle's GPU.  The work
 * is enqueued on `stream` (a hipStream_t passed as void*; NULL = the handle's own
 * stream) and the call returns without waiting.  These are what the batch scheduler
 * and bench.py drive; they have no counterpart in the reference, which only ever hands
 * host arrays to Keras.  The kernels fetch rows as 16-byte vectors and store the two
 * probabilities as one pair: strct / esm / esm_table must be 16-byte aligned, out 8-byte,
 * positions 4-byte (any hipMalloc or framework allocation is; a pointer offset into a
 * row is not) — otherwise DPTM_EINVAL and nothing is enqueued. */
int dptm_score_device(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33,
                      const uint8_t* tok51, const float* strct, const float* esm, int64_t n,
                      float* out, void* stream);

/* dptm_score_device for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments,
 * same kernels and bit-identical results, but the per-site layers' activations live in a static
 * device array of the library (30.4 MB per GPU, held until the process ends) instead of the handle's
 * buffers, and the call creates no stream or event and never allocates or frees.  At most
 * dptm_noalloc_max_sites() = 4096 sites per call; more fails with DPTM_EINVAL and nothing is enqueued.
 * All calls on one GPU -- from any handle -- share that array: calls that overlap in time on one GPU
 * race; order them on one stream or with events.  Per-kernel profiling does not see these calls. */
int dptm_score_device_noalloc(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33,
                              const uint8_t* tok51, const float* strct, const float* esm, int64_t n,
                              float* out, void* stream);
int64_t dptm_noalloc_max_sites(void);

int dptm_score_protein_device(dptm_handle* h, const uint8_t* codes, int64_t L,
                              const float* esm_table, int64_t esm_rows, const int32_t* positions,
                              const float* strct, int64_t n, float* out, void* stream);

/* Window assembly alone (predict.py:529-560 on the device).  DEVICE pointers.
 * Writes tok21/tok33/tok51 uint8 [n][W] and esm_row int32 [n] (= position-1, or -1 when
 * the site gets a zero ESM row).  Positions must already be validated (1..L). */
int dptm_assemble_device(dptm_handle* h, const uint8_t* codes, int64_t L, int64_t esm_rows,
                         const int32_t* positions, int64_t n, uint8_t* tok21, uint8_t* tok33,
                         uint8_t* tok51, int32_t* esm_row, void* stream);

/* Many proteins per call: replaces the per-protein loop a proteome run makes around
 * predict_ptm_sites (pred/train_PTM/predict.py:584-612; window assembly predict.py:529-560, ESM row
 * gather predict.py:487-501) with ONE call whose sites may belong to any number of proteins.
 *   codes:        uint8 [seq_offsets[P]]  residue codes of all proteins, concatenated;
 *   seq_offsets:  int64 [P+1], seq_offsets[0] = 0: protein p owns codes[seq_offsets[p] .. seq_offsets[p+1]);
 *   esm_tables:   float32 [esm_offsets[P]][1280]  per-residue tables, concatenated (NULL if none has one);
 *   esm_offsets:  int64 [P+1], esm_offsets[0] = 0: first table row of protein p; a protein without a
 *                 table has esm_offsets[p+1] == esm_offsets[p] (its sites get zero rows, predict.py:498-500),
 *                 and a site beyond its protein's table gets a zero row (predict.py:493);
 *   positions:    int32 [n], 1-based within the site's protein;  site_protein: int32 [n], 0..P-1;
 *   strct: float32 [n][8];  out: float32 [n][2], rows in input order.
 * Sites need not be grouped by protein.  Results equal those of dptm_score_protein called per
 * protein.  The host form checks every offset, protein index and position (DPTM_EINVAL, nothing
 * scored) and moves only the ESM rows the sites use across PCIe, double-buffered; the device
 * form (DEVICE pointers, same alignment rules as above, offsets 8-byte aligned, enqueued on
 * `stream`) expects validated input and gathers rows from the resident tables inside the ESM layer. */
int dptm_score_proteins(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets,
                        const float* esm_tables, const int64_t* esm_offsets, int64_t n_proteins,
                        const int32_t* positions, const int32_t* site_protein, const float* strct,
                        int64_t n, float* out);

int dptm_score_proteins_device(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets,
                               const float* esm_tables, const int64_t* esm_offsets, int64_t n_proteins,
                               const int32_t* positions, const int32_t* site_protein, const float* strct,
                               int64_t n, float* out, void* stream);

/* Window assembly alone for sites of many proteins (predict.py:529-560, 487-501 on the device).
 * DEVICE pointers, validated input.  Writes tok21/tok33/tok51 uint8 [n][W] and esm_row int32 [n]
 * (row of the concatenated tables, or -1 for a zero row).  One launch covers all n sites.  The arrays may
 * start at any byte; the kernel is at its fastest when codes starts on a 4-byte boundary (residues are then
 * fetched as 32-bit words) and the three outputs on 16-byte boundaries (16-byte stores).
 * READ EXTENT of `codes` in every device form (this call, dptm_assemble_device, dptm_score_protein_device,
 * dptm_score_proteins_device): when codes starts on a 4-byte boundary the kernel fetches whole 32-bit words, so
 * the bytes up to the next 4-byte boundary behind the last residue are READ (never used: they are replaced by the
 * padding code).  The array must be readable that far -- any hipMalloc'd buffer is; a sub-range that ends inside
 * another allocation's last word is too; a sub-range that ends exactly at the end of a mapping whose size is not a
 * multiple of 4 is not, and must be passed from an address that is not 4-byte aligned (byte path) or padded. */
int dptm_assemble_proteins_device(dptm_handle* h, const uint8_t* codes, const int64_t* seq_offsets,
                                  const int64_t* esm_offsets, const int32_t* positions,
                                  const int32_t* site_protein, int64_t n, uint8_t* tok21, uint8_t* tok33,
                                  uint8_t* tok51, int32_t* esm_row, void* stream);

/* fp32 -> the 16-bit operand values of a one-operand half-precision mode (mode 1 = bfloat16, 2 = float16; round to
 * nearest even, the kernels' own conversion), n values, on a few host threads; no GPU involved.  This is what the
 * host-buffer entry points (dptm_score, dptm_score_proteins) do to the ESM rows of a call larger than one pass while
 * they gather them into pinned memory in those modes, so that the rows cross PCIe at 2 bytes per value: results are
 * bit-identical to converting on the device.  No counterpart in the reference (Keras feeds fp32, predict.py:598-599). */
int dptm_convert_rows(const float* src, int64_t n, int mode, uint16_t* dst);

/* dptm_score for a caller that KEEPS its ESM rows as the 16-bit values dptm_convert_rows makes (uint16 [n][1280], half
 * the memory and half the host reads of fp32 rows): the handle must be in the matching one-operand mode
 * (DPTM_PREC_BF16 / DPTM_PREC_F16, else DPTM_EINVAL).  Results are bit-identical to dptm_score on the fp32 rows the
 * values were converted from for calls of more than 512 sites (smaller calls of dptm_score take fp32 dense kernels).
 * No counterpart in the reference (Keras feeds fp32, predict.py:598-599). */
int dptm_score_rows16(dptm_handle* h, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                      const float* strct, const uint16_t* esm16, int64_t n, float* out);

/* Blocks until everything enqueued on the handle's own stream has finished. */
int dptm_sync(dptm_handle* h);

/* Per-kernel timing of the device-resident calls, measured with HIP events on the
 * stream the kernels run on.  dptm_profile_begin arms it; every dptm_score*_device call
 * after that records events around each kernel launch; dptm_profile_end waits for the
 * stream, fills `ms` (DPTM_PROF_COUNT floats: total milliseconds per kernel class since
 * begin) and `launches` (same length: launches per class), and disarms it. */
enum {
    DPTM_PROF_ASSEMBLE = 0,
    DPTM_PROF_BRANCH21 = 1,
    DPTM_PROF_BRANCH33 = 2,
    DPTM_PROF_BRANCH51 = 3,
    DPTM_PROF_STRUCT = 4,
    DPTM_PROF_ESM1 = 5,
    DPTM_PROF_ESM2 = 6,
    DPTM_PROF_TRUNK1 = 7,
    DPTM_PROF_TRUNK2 = 8,
    DPTM_PROF_HEAD = 9,
    DPTM_PROF_COUNT = 10
};
int dptm_profile_begin(dptm_handle* h);
int dptm_profile_end(dptm_handle* h, float* ms, int64_t* launches);

/* Arithmetic of the matrix products in the fused sequence branches and in the ESM / trunk dense
 * layers (the 8-feature structure layer and the 2-class head always run in fp32).  Everything
 * else -- normalisation, softmax, activations, accumulation -- is fp32 in every mode.  No counterpart in the reference (Keras runs the
 * graph in fp32).
 *   DPTM_PREC_F32    fp32 MFMA (default); probabilities within 1e-4 of the fp32 reference
 *   DPTM_PREC_BF16   operands rounded to bfloat16
 *   DPTM_PREC_F16    operands rounded to float16
 *   DPTM_PREC_F16X3  operands split into two float16 parts, three MFMAs per product:
 *                    fp32-class accuracy (same 1e-4 bound) on the half-precision matrix cores */
#define DPTM_PREC_F32 0
#define DPTM_PREC_BF16 1
#define DPTM_PREC_F16 2
#define DPTM_PREC_F16X3 3
int dptm_set_precision(dptm_handle* h, int mode);

/* Sites per internal pass of the HOST-buffer calls (dptm_score, dptm_score_protein, dptm_score_proteins): 32,768, the
 * size of their staging slots.  The device-resident calls walk their sites in passes of 107,520 (whole rounds of
 * tiles for all three window sizes); their per-pass buffers grow to that size on the first call that scores more
 * than 32,768 sites at once (one allocation, kept). */
int64_t dptm_chunk_sites(const dptm_handle* h);

const char* dptm_last_error(void);

/* Solvent-accessible surface area per atom, Shrake-Rupley with `probe` (Angstrom) and the caller's
 * `n_points` unit-sphere test points: replaces the per-residue SASA the reference takes from
 * PyRosetta (pred/train_PTM/predict.py:210-241, calc_per_atom_sasa with probe 1.4) in the
 * structure-feature stage in front of the scoring path.  Host buffers: xyz [n][3], radii [n]
 * (<= 0: atom ignored), unit_points [n_points][3], area_out [n], all double.  Counts are exact
 * integer work, so results are bit-identical to the NumPy restatement in
 * deepptmpred_amd/structure.py.  Errors through dptm_sasa_last_error(). */
int dptm_sasa(const double* xyz, const double* radii, int64_t n, double probe, const double* unit_points,
              int32_t n_points, int32_t device, double* area_out);
const char* dptm_sasa_last_error(void);

/* ------------------------------------------------------------------------------------
 * Training step.  Replaces what model.fit runs per batch in the reference
 *   pred/train_PTM/trainer.py:304-309  compile(optimizer=AdamW(lr, weight_decay), loss=focal_loss)
 *   pred/train_PTM/trainer.py:332-339  model.fit(...)   (forward in training mode, loss of
 *                                      trainer.py:36-54, backward, optimizer update)
 * for the graph of model.py:143-210.  One trainer per GPU (data-parallel replica); the
 * caller all-reduces the flat gradient buffer between forward_backward and apply.
 * ------------------------------------------------------------------------------------ */
typedef struct dptm_trainer dptm_trainer;

typedef struct dptm_train_opts {
    uint32_t seed;          /* dropout stream seed for this step (and rank) */
    float rate_embed;       /* SpatialDropout1D after the position embedding (model.py:159: 1e-4) */
    float rate_conv;        /* SpatialDropout1D in the conv block (model.py:136: 0.3) */
    float rate_tb;          /* TransformerBlock dropouts (model.py:94,108-109: 0.1) */
    float rate_dense;       /* Dropout on struct/ESM/merged/trunk (model.py:189,193,198,202: 0.3) */
    int64_t n_global;       /* N of trainer.py:39 over the whole (global) batch, >= 1 */
    int64_t n_pos_global;   /* N_pos of trainer.py:40 over the whole batch */
    float l2;               /* kernel regulariser factor (config.py:53: 1e-4); 0 disables */
    float l2_scale;         /* multiplies the regulariser value and gradient (1/world for DP) */
} dptm_train_opts;

/* weights: the same fp32 blob dptm_create takes.  max_batch bounds the sites per call. */
int dptm_trainer_create(const void* weights, size_t nbytes, const dptm_spec* spec, int device,
                        int64_t max_batch, dptm_trainer** out);
void dptm_trainer_destroy(dptm_trainer* t);

/* Forward (training mode) + loss + backward for n sites.  Inputs are DEVICE pointers laid
 * out as for dptm_score_device; labels int32 [n] (0/1 = argmax of the one-hot y_true,
 * trainer.py:38).  Gradients land in the trainer's flat gradient buffer.  If loss_out (a
 * HOST pointer) is not NULL the call waits and stores this replica's loss share there. */
int dptm_trainer_forward_backward(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33,
                                  const uint8_t* tok51, const float* strct, const float* esm,
                                  const int32_t* labels, int64_t n, const dptm_train_opts* opts,
                                  float* loss_out, void* stream);

/* Device address and length (floats) of the flat gradient buffer: one contiguous fp32
 * array holding every tensor's gradient (256-byte aligned slots, padding stays zero but for the last float, see
 * dptm_trainer_stage_loss; the
 * slots of the BatchNorm moving statistics hold the batch statistics), the single bucket
 * a data-parallel job all-reduces per step. */
int dptm_trainer_grad_buffer(dptm_trainer* t, float** dev_ptr, int64_t* count);
int64_t dptm_trainer_param_floats(const dptm_trainer* t);

/* tensorflow-addons AdamW (trainer.py:305-306): var -= weight_decay*var, then Adam with
 * beta1 0.9, beta2 0.999, eps 1e-7 on grad_scale * gradient.  The BatchNorm moving
 * statistics are not trained: their slots of the gradient buffer carry the batch mean /
 * variance of the last forward (so the one all-reduce also sums them over replicas) and
 * they move as stat = 0.99*stat + 0.01 * stat_scale * slot (stat_scale = 1/replicas). */
int dptm_trainer_apply(dptm_trainer* t, float lr, float weight_decay, float grad_scale,
                       float stat_scale, void* stream);

/* Synchronised BatchNorm for data-parallel jobs (no counterpart in the reference, which trains on
 * one device: trainer.py:332).  By default every replica normalises with the statistics of its own
 * shard.  With a callback set, forward_backward hands it the BatchNorms' column sums -- forward:
 * [sum x | sum x^2], backward: [sum dz | sum dz*xhat], `count` floats at DEVICE address `buf` --
 * right after the kernels that produced them were enqueued, on `stream` (a hipStream_t).  The callback
 * must sum the buffer in place over all replicas, ordered on that stream (e.g. an RCCL all-reduce),
 * and return 0; every replica makes the same sequence of calls: 8 per step -- the ten BatchNorms sit on
 * four dependent levels per direction (first / second convolution of the three branches, with the structure
 * and first ESM layer riding in the first collective of either direction; the two trunk layers), and the
 * sums of a level are one contiguous buffer of 1,024-float slots (3 to 8 K floats).  The layer then uses
 * the GLOBAL batch (opts->n_global sites) for mean, variance and the backward terms, so N replicas
 * reproduce one device run on the whole batch.  fn = NULL restores per-replica statistics. */
typedef int (*dptm_allreduce_fn)(void* user, float* buf, int64_t count, void* stream);
int dptm_trainer_set_allreduce(dptm_trainer* t, dptm_allreduce_fn fn, void* user);

/* Host copies in the canonical blob order (dptm_weight_count floats); they wait for the GPU. */
int dptm_trainer_get_weights(dptm_trainer* t, float* host_out);
int dptm_trainer_get_grads(dptm_trainer* t, float* host_out);
int dptm_trainer_get_proba(dptm_trainer* t, float* host_out, int64_t n);
/* The optimizer's state -- Adam's first and second moments in the canonical blob order (dptm_weight_count floats each)
 * and the count of steps taken (it sets the bias correction) -- so that a trainer that is rebuilt, for a larger
 * batch say, carries on where the old one stopped.  The reference keeps ONE optimizer object for the whole of
 * model.fit (pred/train_PTM/trainer.py:304-309, 332-339). */
int dptm_trainer_get_optimizer_state(dptm_trainer* t, float* m_host_out, float* v_host_out, int64_t* step_out);
int dptm_trainer_set_optimizer_state(dptm_trainer* t, const float* m_host, const float* v_host, int64_t step);
/* Device address of the step's loss: one float that dptm_trainer_forward_backward writes in stream order (data
 * term + l2 regularisers, what loss_out receives).  A training loop that copies it device-to-device after every
 * step and reads the copies once per epoch need not wait for the GPU at every step, as Keras' model.fit does not
 * (pred/train_PTM/trainer.py:332-339: the per-batch losses only surface as the epoch's mean). */
int dptm_trainer_loss_buffer(dptm_trainer* t, float** dev_ptr);
/* Device address of the step's class probabilities: [n][2] floats (n = the step's sites; room for max_batch) that
 * dptm_trainer_forward_backward writes in stream order -- the training-mode outputs (dropout on, batch statistics)
 * from which Keras accumulates the `accuracy` and `auc` it logs per epoch (pred/train_PTM/trainer.py:308: compile
 * with metrics=['accuracy', AUC(name='auc')]).  A loop copies them device-to-device after each step and reads the
 * copies once per epoch; dptm_trainer_get_proba is the host copy that waits. */
int dptm_trainer_proba_buffer(dptm_trainer* t, float** dev_ptr);

/* dptm_trainer_forward_backward on rows[0..n) (int64 indices in DEVICE memory) of a data set that lives on the device:
 * tok21 [n_data][21] ... labels [n_data] as in dptm_trainer_forward_backward, n <= max_batch.  One launch gathers the
 * rows into staging buffers of the trainer's (allocated by the first call: 5.3 KB per site of max_batch), the step
 * runs on them, and one more launch leaves the step's loss (1 float) at loss_dev_out and its class probabilities
 * (n x 2 floats, training-mode outputs) at proba_dev_out -- DEVICE addresses, either may be NULL.  Everything is
 * ordered on `stream`; nothing waits for the GPU.  An index outside [0, n_data) is clamped into it.  This is the
 * body of Keras' model.fit loop (pred/train_PTM/trainer.py:332-339: shuffled batches of a resident training set,
 * per-batch losses and metrics that only surface per epoch) without a host round trip or a per-batch upload. */
int dptm_trainer_forward_backward_rows(dptm_trainer* t, const uint8_t* tok21, const uint8_t* tok33, const uint8_t* tok51,
                                       const float* strct, const float* esm, const int32_t* labels, int64_t n_data,
                                       const int64_t* rows, int64_t n, const dptm_train_opts* opts, float* loss_dev_out,
                                       float* proba_dev_out, void* stream);

/* Data-parallel jobs: copies the step's loss (dptm_trainer_loss_buffer) into a float of the gradient buffer that no
 * parameter uses -- the last float of the buffer, padding behind the head bias -- in stream order, and returns that
 * float's index in `slot`.  The ONE all-reduce of the gradient bucket then also sums the replicas' loss shares, and a
 * training loop needs no collective or host round trip of its own for the loss.  The matching float of the weights is
 * padding too; dptm_trainer_apply leaves a moving average of the staged value there, which nothing reads.  No
 * counterpart in the reference (one device: trainer.py:332). */
int dptm_trainer_stage_loss(dptm_trainer* t, int64_t* slot, void* stream);

/* How many of the step's four lanes (main stream + three side streams) the trainer's queue probe found on hardware
 * queues of their own when it was created: 4 = every lane runs concurrently; fewer = lanes beyond that number share
 * a queue with an earlier lane and serialise behind it (a performance diagnostic, results do not depend on it). */
int dptm_trainer_lanes(dptm_trainer* t, int32_t* independent_out);

/* The stream the step's main lane runs on (a hipStream_t the trainer owns; it dies with the trainer).  The step
 * always runs on the trainer's own streams, one per hardware queue of the process; a call made on another stream
 * hands over to them and back through two events.  A training loop that enqueues its own work -- batch staging,
 * the gradient all-reduce, dptm_trainer_apply -- on THIS stream and passes it as `stream` has no hand-over, and no
 * further busy stream that shares a hardware queue with one of the step's.  No counterpart in the reference
 * (TensorFlow owns its streams). */
int dptm_trainer_stream(dptm_trainer* t, void** stream);

/* Per-kernel clocks of one training step, the trainer's counterpart of dptm_profile_begin / _end (the
 * reference has none; Keras' own tooling would be the TensorBoard profiler callback).  Between _begin and
 * _end the three sequence branches run one after another on one stream and each of their six launches is
 * bracketed by HIP events on that stream.  _end waits for the device and writes DPTM_TRAIN_PROFILE_SLOTS
 * floats, the milliseconds of the last step's launches: slot = branch * 6 + kernel with branch 0/1/2 =
 * window 21/33/51 and kernel 0 embedding+conv-a forward, 1 conv-b forward, 2 TransformerBlocks+pooling
 * forward, 3 TransformerBlocks+pooling backward, 4 conv-b backward, 5 conv-a+embedding backward. */
#define DPTM_TRAIN_PROFILE_SLOTS 18
int dptm_trainer_profile_begin(dptm_trainer* t);
int dptm_trainer_profile_end(dptm_trainer* t, float* ms_out);
const char* dptm_trainer_last_error(void);


#ifdef __cplusplus
}
#endif
#endif /* DPTM_H */
