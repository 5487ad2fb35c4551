// Launcher interface between the C ABI (dptm_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dptm {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

enum DenseEpilogue { kEpiSwish = 0, kEpiSwishAffine = 1, kEpiHead = 2 };

// calls of at most this many rows (a protein's worth of sites) use the latency-oriented kernels and streams
constexpr int64_t kDenseSmallRows = 512;

// arithmetic of the matrix products in the fused sequence branch (DPTM_PREC_* in dptm.h)
enum PrecisionMode { kModeF32 = 0, kModeBF16 = 1, kModeF16 = 2, kModeF16X3 = 3 };

struct DenseArgs {
    const float* x;            // [rows][ldx] fp32 activations (x_half: bfloat16 / float16 values of the mode, 2 bytes each)
    int64_t ldx;               // elements per row of x
    int x_half = 0;            // half-precision modes only: x already holds the mode's 16-bit operand values
    const int32_t* row_index;  // optional gather: source row of output row r, <0 = zero row
    int64_t m;                 // output rows
    int k;                     // contraction length, multiple of 8
    int n_pad;                 // out features padded to the packed width
    const f32x4_t* w;          // packed weights [n_pad/32][k/8][64] float4
    const f32x4_t* bias;       // packed vector (n_pad)
    const f32x4_t* scale;      // packed vector, kEpiSwishAffine only
    const f32x4_t* shift;
    float* out;                // [rows][ldo]
    int64_t ldo;
    int col_off;
};

// the three sequence branches of a small call in one launch: columns 0 | 64 | 128 of out (windows 21 | 33 | 51)
hipError_t launch_branch3(const float* const (&params)[3], const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo,
                          hipStream_t st);
// ... together with the structure layer and the first ESM layer (DenseArgs of m = n rows, swish + affine epilogue)
hipError_t launch_front5(const float* const (&params)[3], const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo,
                         const DenseArgs& strct, const DenseArgs& esm1, hipStream_t st);
hipError_t launch_branch(int w, const float* params, const uint8_t* tok, int64_t n, float* out,
                         int ldo, int col_off, hipStream_t st);
// half-precision-MFMA branch (dptm_kernels_h.hip): whi/wlo = packed half weight images of the branch
hipError_t launch_branch3_h(int mode, const float* const (&params)[3], const void* const (&whi)[3], const void* const (&wlo)[3],
                            const uint8_t* const (&tok)[3], int64_t n, float* out, int ldo, hipStream_t st);
hipError_t launch_branch_h(int mode, int w, const float* params, const void* whi, const void* wlo, const uint8_t* tok,
                           int64_t n, float* out, int ldo, int col_off, hipStream_t st);
hipError_t launch_dense(const DenseArgs& a, int epi, hipStream_t st);
// half-precision-MFMA dense layer (swish epilogues only; k a multiple of 16): whi/wlo = packed half images of a.w
hipError_t launch_dense_h(int mode, const DenseArgs& a, int epi, const void* whi, const void* wlo, hipStream_t st);
// Window assembly for n sites (see assemble_kernel).  seq_off / esm_off / site_protein are nullptr for the
// one-protein form, which takes L and esm_rows instead.
struct AssembleArgs {
    const uint8_t* codes;          // residue codes, all proteins concatenated
    const int64_t* seq_off;        // [P+1] first residue of each protein in codes
    const int64_t* esm_off;        // [P+1] first ESM table row of each protein (equal neighbours: no table)
    const int32_t* site_protein;   // [n] protein of each site
    int64_t L;                     // one-protein form: sequence length
    int64_t esm_rows;              // one-protein form: table rows
    const int32_t* positions;      // [n] 1-based, protein-local, already validated
    int64_t n;
    uint8_t* t21;                  // [n][21]
    uint8_t* t33;                  // [n][33]
    uint8_t* t51;                  // [n][51]
    int32_t* esm_row;              // [n] table row of the site, or -1 (zero row)
    bool words;                    // set by launch_assemble: every output array is 16-byte aligned
};
hipError_t launch_assemble(const AssembleArgs& a, hipStream_t st);
hipError_t prepare_kernels();
hipError_t prepare_kernels_h(int mode);   // LDS opt-in of the half-precision branch kernels of one mode

}  // namespace dptm
