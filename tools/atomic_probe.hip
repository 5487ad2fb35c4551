// Prices the ways a backward workgroup can add its 64 x 64 weight-gradient units (4096 floats each, 24 per tile) to
// memory: read-modify-write of a private slab per workgroup (what tb_bwd_kernel does) against float atomics into K
// shared slabs (K = 1 ... 256).  Same shape as a batch-8192 launch of the W = 51 backward kernel: 256 workgroups of
// 8 wavefronts, 1639 tiles, every wavefront adds 16 registers per unit (RMW: the four lower wavefronts only).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe && tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kUnits = 24, kUnitFloats = 4096, kSlab = kUnits * kUnitFloats;

__global__ __launch_bounds__(512) void rmw_kernel(float* slabs, int tiles, unsigned* queue) {
    __shared__ int slot;
    float* part = slabs + (size_t)blockIdx.x * kSlab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) slot = (int)atomicAdd(queue, 1u);
        __syncthreads();
        if (slot >= tiles) break;
        for (int u = 0; u < kUnits; ++u) {
            if (wave < 4) {
                float* dst = part + u * kUnitFloats + wave * 1024 + lane;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = dst[r * 64];
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r * 64] = v[r] + 1.0f;
            }
        }
    }
}

template <int SCOPE>
__global__ __launch_bounds__(512) void atomic_kernel(float* slabs, int nslabs, int tiles, unsigned* queue) {
    __shared__ int slot;
    float* part = slabs + (size_t)(blockIdx.x % nslabs) * kSlab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) slot = (int)atomicAdd(queue, 1u);
        __syncthreads();
        if (slot >= tiles) break;
        for (int u = 0; u < kUnits; ++u) {
            float* dst = part + u * kUnitFloats + (wave & 3) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) (void)__hip_atomic_fetch_add(dst + r * 64, 0.5f, __ATOMIC_RELAXED, SCOPE);
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int tiles = 1639, wgs = 256;
    float* slabs;
    unsigned* queue;
    CK(hipMalloc(&slabs, (size_t)wgs * kSlab * sizeof(float)));
    CK(hipMalloc(&queue, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(slabs, 0, (size_t)wgs * kSlab * sizeof(float)));
            CK(hipMemset(queue, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        float first = 0;
        CK(hipMemcpy(&first, slabs, 4, hipMemcpyDeviceToHost));
        printf("%-44s %8.3f ms   (slab[0][0] = %.1f)\n", name, best, first);
        return 0;
    };
    if (run("rmw, private slab per workgroup", [&] { hipLaunchKernelGGL(rmw_kernel, dim3(wgs), dim3(512), 0, 0, slabs, tiles, queue); })) return 1;
    for (int k : {256, 32, 8, 1}) {
        char name[96];
        snprintf(name, sizeof name, "atomics, agent scope, %d shared slabs", k);
        if (run(name, [&] { hipLaunchKernelGGL(atomic_kernel<__HIP_MEMORY_SCOPE_AGENT>, dim3(wgs), dim3(512), 0, 0, slabs, k, tiles, queue); })) return 1;
    }
    for (int k : {256, 8}) {
        char name[96];
        snprintf(name, sizeof name, "atomics, workgroup scope, %d slabs (timing only)", k);
        if (run(name, [&] { hipLaunchKernelGGL(atomic_kernel<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(wgs), dim3(512), 0, 0, slabs, k, tiles, queue); })) return 1;
    }
    return 0;
}
