// Small device helpers shared by the training kernels (dptm_train_kernels.hip, dptm_train_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dptm {

// stateless dropout stream: lowbias32 of the element index mixed with (seed, stream)
__device__ __forceinline__ unsigned hash_u32(unsigned seed, unsigned stream, unsigned idx) {
    unsigned x = idx ^ (seed * 0x9E3779B9u);
    x += stream * 0x85EBCA6Bu;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// orders a wavefront's own LDS writes before its own later LDS reads (and the reverse)
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dropout threshold on the 32-bit hash for a rate in [0, 1): keep when hash >= thr
inline unsigned dropout_threshold(float rate) {
    if (rate <= 0.f) return 0u;
    const double t = (double)rate * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

}  // namespace dptm
