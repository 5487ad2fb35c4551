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
    const float* x;            // [rows][ldx] fp32 activations
    int64_t ldx;               // floats per row of x
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

hipError_t launch_branch(int w, const float* params, const uint8_t* tok, int64_t n, float* out,
                         int ldo, int col_off, hipStream_t st);
// half-precision-MFMA branch (dptm_kernels_h.hip): whi/wlo = packed half weight images of the branch
hipError_t launch_branch_h(int mode, int w, const float* params, const void* whi, const void* wlo, const uint8_t* tok,
                           int64_t n, float* out, int ldo, int col_off, hipStream_t st);
hipError_t launch_dense(const DenseArgs& a, int epi, hipStream_t st);
// half-precision-MFMA dense layer (swish epilogues only; k a multiple of 16): whi/wlo = packed half images of a.w
hipError_t launch_dense_h(int mode, const DenseArgs& a, int epi, const void* whi, const void* wlo, hipStream_t st);
hipError_t launch_assemble(const uint8_t* codes, int64_t L, int64_t esm_rows,
                           const int32_t* positions, int64_t n, uint8_t* t21, uint8_t* t33,
                           uint8_t* t51, int32_t* esm_row, hipStream_t st);

}  // namespace dptm
