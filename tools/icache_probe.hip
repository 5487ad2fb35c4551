// How fast does straight-line code run when it is not in the instruction cache?  A kernel of N dependent v_fma_f32
// (8 bytes each), one wavefront per CU on 64 CUs, timed (a) launched back to back (code stays cached), (b) alternating
// with a second, 96 KB kernel that evicts it, (c) as a rolled loop of the same length (a few hundred bytes of code).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/icache_probe tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int TAG>
__global__ __launch_bounds__(64) void straight(float* p, float a, float b) {
    float v = p[threadIdx.x];
#pragma unroll
    for (int i = 0; i < N; ++i) v = __builtin_fmaf(v, a, b);
    p[blockIdx.x * 64 + threadIdx.x] = v + TAG;
}

__global__ __launch_bounds__(64) void rolled(float* p, float a, float b, int n) {
    float v = p[threadIdx.x];
#pragma unroll 1
    for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, a, b);
    p[blockIdx.x * 64 + threadIdx.x] = v;
}

template <typename F>
float timed(F launch, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

template <int N>
void row(float* d) {
    const dim3 g(64), blk(64);
    const float hot = timed([&] { hipLaunchKernelGGL((straight<N, 0>), g, blk, 0, 0, d, 1.0001f, 0.5f); }, 50);
    const float evict = timed([&] { hipLaunchKernelGGL((straight<12288, 1>), g, blk, 0, 0, d, 1.0001f, 0.5f); }, 50);
    const float both = timed([&] {
        hipLaunchKernelGGL((straight<N, 0>), g, blk, 0, 0, d, 1.0001f, 0.5f);
        hipLaunchKernelGGL((straight<12288, 1>), g, blk, 0, 0, d, 1.0001f, 0.5f);
    }, 50);
    const float loop = timed([&] { hipLaunchKernelGGL(rolled, g, blk, 0, 0, d, 1.0001f, 0.5f, N); }, 50);
    printf("%6d fma (%3d KB): cached %.1f us   evicted between launches %.1f us   rolled loop %.1f us\n", N, N * 8 / 1024, hot,
           both - evict, loop);
}

int main() {
    float* d; (void)hipMalloc(&d, 64 * 64 * 4); (void)hipMemset(d, 0, 64 * 64 * 4);
    row<1024>(d);
    row<2048>(d);
    row<4096>(d);
    row<7168>(d);
    return 0;
}
