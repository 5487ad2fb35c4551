// Issue-rate probe for gfx950: packed vs scalar fp32 FMA, the exponential unit, and ds_read_b128
// (distinct vs broadcast addresses).  Every kernel runs ITER trips of an unrolled body on 8 waves
// per workgroup, one workgroup per CU x 4; rates are printed per CU per clock at the measured
// shader clock.     hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

__global__ __launch_bounds__(512) void k_pk_fma(float* out) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k_fma(float* out) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = (float)threadIdx.x + i;
    float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], m, c);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k_exp(float* out) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x & 7) * 0.01f + i * 0.001f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]) - 1.0f;
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// MODE 0: every lane its own 16 bytes (conflict-free); 1: whole wave reads one address;
// 2: 8 distinct addresses per wave (the attention pattern: 4 heads x 2 sites)
template <int MODE>
__global__ __launch_bounds__(512) void k_lds(float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int base = MODE == 0 ? lane * 4 : MODE == 1 ? 0 : (lane & 3) * 16 + ((lane >> 5) & 1) * 68 * 20;
    f32x4 acc = {0, 0, 0, 0};
    const float* src = lds + base;
    for (int it = 0; it < ITER; ++it) {
        asm volatile("" ::: "memory");          // the loads are re-issued every trip
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * 272);
            asm volatile("" ::"v"(v));
            if (r == 31) acc += v;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;  // Hz
    const int blocks = cus * 4;
    float* out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    const double waves_per_cu = 4.0 * 8;   // sequentially, 8 resident at a time
    auto report = [&](const char* name, float ms, double instr_per_wave, double unit_per_instr, const char* unit) {
        const double clks = ms * 1e-3 * clk;
        const double per_cu = waves_per_cu * instr_per_wave;
        printf("%-28s %8.3f ms  %.2f clk per wave-instruction per SIMD  %.1f %s/clk/CU\n", name, ms,
               clks / (per_cu / 4.0), per_cu * unit_per_instr / clks, unit);
    };
    printf("%s, %d CUs, %.0f MHz\n", p.name, cus, clk / 1e6);
    const double n = (double)ITER * 32;
    report("v_pk_fma_f32", time_ms([&] { k_pk_fma<<<blocks, 512>>>(out); }), n, 256, "flop");
    report("v_fma_f32", time_ms([&] { k_fma<<<blocks, 512>>>(out); }), n, 128, "flop");
    report("v_exp_f32 (+v_sub)", time_ms([&] { k_exp<<<blocks, 512>>>(out); }), n, 64, "exp");
    report("ds_read_b128 distinct", time_ms([&] { k_lds<0><<<blocks, 512>>>(out); }), n, 1024, "B");
    report("ds_read_b128 one address", time_ms([&] { k_lds<1><<<blocks, 512>>>(out); }), n, 1024, "B");
    report("ds_read_b128 8 addresses", time_ms([&] { k_lds<2><<<blocks, 512>>>(out); }), n, 1024, "B");
    hipFree(out);
    return 0;
}
