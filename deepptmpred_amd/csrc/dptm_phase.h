// Debug builds only (-DDPTM_PHASE_TIMING): cycles per phase summed over all wavefronts of a branch kernel
// (phase ids are the PH_MARK arguments), read back with dptm_debug_phase_cycles().  s_memtime reads the
// clock; the sums leave through vector atomics at the end of the kernel.  Slots 16..23 / 24..31 hold the barrier
// waits (phases 2 + 4) / the attention time (phase 5) per wavefront index of the workgroup.
#pragma once
#ifdef DPTM_PHASE_TIMING
namespace dptm { static __device__ unsigned long long g_phase_cycles[32]; }   // one per translation unit
#define PH_DECL unsigned long long ph_t0 = __builtin_amdgcn_s_memtime(); unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(id) { const unsigned long long ph_t1 = __builtin_amdgcn_s_memtime(); ph_acc[id] += ph_t1 - ph_t0; ph_t0 = ph_t1; }
#define PH_FLUSH if ((threadIdx.x & 63) == 0) { for (int ph_i = 0; ph_i < 12; ++ph_i) atomicAdd(&dptm::g_phase_cycles[ph_i], ph_acc[ph_i]); \
    atomicAdd(&dptm::g_phase_cycles[16 + ((threadIdx.x >> 6) & 7)], ph_acc[2] + ph_acc[4]); atomicAdd(&dptm::g_phase_cycles[24 + ((threadIdx.x >> 6) & 7)], ph_acc[5]); }
// for device functions that mark phases of the kernel they are inlined into
#define PH_PARAMS , unsigned long long& ph_t0, unsigned long long* ph_acc
#define PH_ARGS , ph_t0, ph_acc
#define PH_WAIT_VM __builtin_amdgcn_s_waitcnt(0x0F70);
#else
#define PH_DECL
#define PH_MARK(id)
#define PH_FLUSH
#define PH_PARAMS
#define PH_ARGS
#define PH_WAIT_VM
#endif

#ifdef DPTM_PHASE_TIMING
// defines the host-side reader of this translation unit's counters under the given C name
#define PH_DEFINE_READER(name)                                                                                      \
    extern "C" int name(unsigned long long* out32, int reset) {                                                     \
        if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(dptm::g_phase_cycles), 32 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        if (reset) {                                                                                                \
            unsigned long long z[32] = {};                                                                          \
            if (hipMemcpyToSymbol(HIP_SYMBOL(dptm::g_phase_cycles), z, sizeof z) != hipSuccess) return -1;          \
        }                                                                                                           \
        return 0;                                                                                                   \
    }
#else
#define PH_DEFINE_READER(name)
#endif
