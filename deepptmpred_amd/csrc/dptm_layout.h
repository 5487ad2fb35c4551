// Shared constants: fixed model dimensions and the layout of the packed per-branch
// parameter buffer the fused sequence-branch kernel reads.
//
// "Chain layout" (used for every activation vector held in registers): a 32-row tile
// is owned by one wavefront; lane l = 32*h + j holds, for data row j, the features
//     feat(R, h) = 8*(R>>2) + 4*h + (R&3),   R = 0 .. F/2-1
// i.e. for every group of 8 consecutive features the lower half-wave holds the first
// four and the upper half-wave the last four.  This is exactly the C/D register map of
// v_mfma_f32_32x32x2_f32 (row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31) when the
// MFMA computes Y^T = W^T * X^T, and at the same time the B-operand order of the NEXT
// MFMA (k = lane>>5 picks the half-wave), so dense layers chain register to register.
//
// Packed weight matrix [K][N] (Keras kernel [in,out]):  float4 index
//     ((t*(K/8) + u)*64 + lane)  holds  W[8u + 4h + e][32t + j], e = 0..3
// one coalesced 1 KiB wave-load feeds four MFMA steps of out-tile t.
// Packed feature vector of F features: float4 index (h*(F/8) + u) holds v[8u+4h+e].
#pragma once
#include <cstdint>

namespace dptm {

constexpr int kD = 64;          // embed_dim
constexpr int kHeads = 4;
constexpr int kFF = 256;
constexpr int kCodes = 21;
constexpr int kEsm = 1280;
constexpr int kStruct = 8;
constexpr int kStructUnits = 128;
constexpr int kEsm1 = 512;
constexpr int kEsm2 = 256;
constexpr int kConcat = 3 * kD + kStructUnits + kEsm2;  // 576
constexpr int kT1 = 512;
constexpr int kT2 = 256;
constexpr int kMaxW = 51;

constexpr int kTileRows = 256;   // rows (site positions) per workgroup of the branch kernel
constexpr int kLdsStride = 68;   // floats per LDS row: 64 + 4 keeps ds_read_b128 conflict-free

// ---- per-branch packed buffer, offsets in float4 units
constexpr int kMat64 = 64 * 64 / 4;        // 1024 float4 per packed 64x64 matrix
constexpr int kMatFF = 64 * 256 / 4;       // 4096
constexpr int kVec64 = 16;                 // float4 per packed 64-feature vector
constexpr int kVec256 = 64;

constexpr int kOffEmb = 0;                              // [21][16] natural layout
constexpr int kOffPos = kOffEmb + kCodes * 16;          // [W][16] natural layout (room for 51)
constexpr int kOffScW = kOffPos + kMaxW * 16;
constexpr int kOffC3aW = kOffScW + kMat64;              // 3 taps
constexpr int kOffC3bW = kOffC3aW + 3 * kMat64;
constexpr int kOffConvV = kOffC3bW + 3 * kMat64;        // 7 vectors, see enum below
constexpr int kOffTb = kOffConvV + 7 * kVec64;

enum ConvVec { kVScB = 0, kVC3aB, kVBn1S, kVBn1T, kVC3bB, kVBn2S, kVBn2T };

// one transformer block
constexpr int kTbWq = 0;
constexpr int kTbWk = kTbWq + kMat64;
constexpr int kTbWv = kTbWk + kMat64;
constexpr int kTbWo = kTbWv + kMat64;
constexpr int kTbF1 = kTbWo + kMat64;
constexpr int kTbF2 = kTbF1 + kMatFF;
constexpr int kTbVec = kTbF2 + kMatFF;                  // 9 vectors of 64, see enum below
constexpr int kTbB1 = kTbVec + 9 * kVec64;              // ffn1 bias, 256 features
constexpr int kTbSize = kTbB1 + kVec256;

enum TbVec { kVBq = 0, kVBk, kVBv, kVBo, kVLn1G, kVLn1B, kVB2, kVLn2G, kVLn2B };

constexpr int kOffPoolV = kOffTb + 2 * kTbSize;         // pooling weight vector (64)
constexpr int kOffPoolB = kOffPoolV + kVec64;           // pooling bias in .x
constexpr int kBranchF4 = kOffPoolB + 1;

// LDS copy of a branch's bias / scale / LayerNorm vectors (float4 units): conv (7 x 64) | block 0
// (9 x 64 + 256) | block 1 | pooling (64 + 1).  vecs_source(i) = float4 index in the packed buffer.
constexpr int kVecsConv = 7 * kVec64;
constexpr int kVecsTb = 9 * kVec64 + kVec256;
constexpr int kVecsPool = kVecsConv + 2 * kVecsTb;
constexpr int kVecsF4 = kVecsPool + kVec64 + 1;
static_assert(kTbB1 == kTbVec + 9 * kVec64 && kOffPoolB == kOffPoolV + kVec64, "vector groups must be contiguous");
constexpr int vecs_source(int i) {
    return i < kVecsConv ? kOffConvV + i
         : i < kVecsConv + kVecsTb ? kOffTb + kTbVec + (i - kVecsConv)
         : i < kVecsPool ? kOffTb + kTbSize + kTbVec + (i - kVecsConv - kVecsTb)
                         : kOffPoolV + (i - kVecsPool);
}

}  // namespace dptm
