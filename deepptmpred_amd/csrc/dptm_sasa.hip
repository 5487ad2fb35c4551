// Solvent-accessible surface area per atom (Shrake-Rupley) on the GPU, for the structure-feature
// stage in front of the scoring path (pred/train_PTM/predict.py:210-241 computes it in PyRosetta
// with a 1.4 A probe).  Integer work at heart: for every atom, count the test points on its
// expanded sphere that lie outside every neighbour's expanded sphere.  All comparisons are done in
// fp64 with contraction off and in the same operation order as the NumPy restatement
// (deepptmpred_amd/structure.py: shrake_rupley), so the counts -- and therefore the areas -- are
// bit-identical to it.
//
// One workgroup per atom: the wavefronts first collect the atom's neighbours into LDS (all-pairs
// scan of the coordinates: a protein is a few thousand atoms, 48 KB of coordinates that stay in
// L2), then one thread per test point walks the neighbour list with early exit.
#include <hip/hip_runtime.h>

#include <new>

#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/dptm.h"

namespace {

constexpr int kMaxNeighbours = 1024;   // an expanded sphere of <= ~7 A holds far fewer atoms
constexpr int kThreads = 256;

#pragma clang fp contract(off)
__global__ __launch_bounds__(kThreads) void sasa_kernel(const double* __restrict__ xyz, const double* __restrict__ radii,
                                                        int n, double probe, const double* __restrict__ unit, int n_points,
                                                        double* __restrict__ area, int* __restrict__ overflow) {
    __shared__ int nb[kMaxNeighbours];
    __shared__ int nb_count;
    __shared__ int exposed_count;
    const int i = blockIdx.x;
    const double ri = radii[i];
    if (threadIdx.x == 0) {
        nb_count = 0;
        exposed_count = 0;
    }
    __syncthreads();
    if (ri <= 0.0) {                       // hydrogens and unknown atoms carry no surface
        if (threadIdx.x == 0) area[i] = 0.0;
        return;
    }
    const double xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2];
    const double ei = ri + probe;
    for (int j = threadIdx.x; j < n; j += kThreads) {
        const double rj = radii[j];
        if (j == i || rj <= 0.0) continue;
        const double dx = xyz[3 * j] - xi, dy = xyz[3 * j + 1] - yi, dz = xyz[3 * j + 2] - zi;
        const double d = sqrt((dx * dx + dy * dy) + dz * dz);
        if (d < ei + (rj + probe)) {
            const int slot = atomicAdd(&nb_count, 1);
            if (slot < kMaxNeighbours) nb[slot] = j;
        }
    }
    __syncthreads();
    const int cnt = nb_count;
    if (cnt > kMaxNeighbours) {
        if (threadIdx.x == 0) atomicAdd(overflow, 1);
        return;
    }
    int mine = 0;
    for (int p = threadIdx.x; p < n_points; p += kThreads) {
        const double px = xi + ei * unit[3 * p], py = yi + ei * unit[3 * p + 1], pz = zi + ei * unit[3 * p + 2];
        bool ok = true;
        for (int q = 0; q < cnt && ok; ++q) {
            const int j = nb[q];
            const double ej = radii[j] + probe;
            const double dx = px - xyz[3 * j], dy = py - xyz[3 * j + 1], dz = pz - xyz[3 * j + 2];
            ok = ((dx * dx + dy * dy) + dz * dz) >= ej * ej;
        }
        mine += ok ? 1 : 0;
    }
    if (mine) atomicAdd(&exposed_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double frac = (double)exposed_count / (double)n_points;
        area[i] = 4.0 * M_PI * (ei * ei) * frac;
    }
}

thread_local std::string g_sasa_err;

}  // namespace

extern "C" {

const char* dptm_sasa_last_error(void) { return g_sasa_err.c_str(); }

int dptm_sasa(const double* xyz, const double* radii, int64_t n, double probe, const double* unit_points, int32_t n_points,
              int32_t device, double* area_out) try {
    auto fail = [](int code, const std::string& msg) {
        g_sasa_err = msg;
        return code;
    };
    if (n < 0 || n_points < 1 || n > (1 << 24)) return fail(DPTM_EINVAL, "dptm_sasa: bad sizes");
    if (n == 0) return DPTM_OK;
    if (!xyz || !radii || !unit_points || !area_out) return fail(DPTM_EINVAL, "dptm_sasa: NULL buffer");
    if (hipSetDevice(device) != hipSuccess) return fail(DPTM_EINVAL, "dptm_sasa: no such device");
    // device buffers of this call, freed on every way out
    struct DevBufs {
        void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~DevBufs() {
            for (void* q : p)
                if (q) (void)hipFree(q);
        }
    } buf;
    const size_t bytes[5] = {(size_t)n * 3 * sizeof(double), (size_t)n * sizeof(double), (size_t)n_points * 3 * sizeof(double),
                             (size_t)n * sizeof(double), sizeof(int)};
    for (int i = 0; i < 5; ++i) {
        const hipError_t e = hipMalloc(&buf.p[i], bytes[i]);
        if (e != hipSuccess) return fail(DPTM_ENOMEM, std::string("dptm_sasa: hipMalloc: ") + hipGetErrorString(e));
    }
    double* d_xyz = static_cast<double*>(buf.p[0]);
    double* d_r = static_cast<double*>(buf.p[1]);
    double* d_u = static_cast<double*>(buf.p[2]);
    double* d_a = static_cast<double*>(buf.p[3]);
    int* d_over = static_cast<int*>(buf.p[4]);
    auto check = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        g_sasa_err = std::string("dptm_sasa: ") + what + ": " + hipGetErrorString(e);
        return false;
    };
    if (!check(hipMemcpy(d_xyz, xyz, bytes[0], hipMemcpyHostToDevice), "copy in")) return DPTM_EHIP;
    if (!check(hipMemcpy(d_r, radii, bytes[1], hipMemcpyHostToDevice), "copy in")) return DPTM_EHIP;
    if (!check(hipMemcpy(d_u, unit_points, bytes[2], hipMemcpyHostToDevice), "copy in")) return DPTM_EHIP;
    if (!check(hipMemset(d_over, 0, sizeof(int)), "memset")) return DPTM_EHIP;
    hipLaunchKernelGGL(sasa_kernel, dim3((unsigned)n), dim3(kThreads), 0, 0, d_xyz, d_r, (int)n, probe, d_u, (int)n_points, d_a, d_over);
    if (!check(hipGetLastError(), "launch")) return DPTM_EHIP;
    int over = 0;
    if (!check(hipMemcpy(&over, d_over, sizeof(int), hipMemcpyDeviceToHost), "copy out")) return DPTM_EHIP;
    if (over) return fail(DPTM_EINVAL, "dptm_sasa: an atom has more than 1024 neighbours (coordinates not in Angstrom?)");
    if (!check(hipMemcpy(area_out, d_a, bytes[3], hipMemcpyDeviceToHost), "copy out")) return DPTM_EHIP;
    return DPTM_OK;
} catch (const std::bad_alloc&) {
    return DPTM_ENOMEM;
} catch (...) {
    return DPTM_EHIP;
}

}  // extern "C"
