// cem_kernels.hip -- CEM bookkeeping for M independent planning problems on gfx950 (MI355X).
//
// Replaces the host-side NumPy/SciPy part of CEMOptimizer.obtain_solution
// (recovery_rl/optimizers.py:73-124): truncated-normal sampling and the elite mean/variance
// update.  The reference plans for ONE env per call with scipy.stats.truncnorm on the CPU; here
// every env that needs a recovery action is one row of a batch and nothing leaves HBM.
//   cem_sample_kernel : one lane per sample element, Philox + Box-Muller, redraw while |z| > 2
//   cem_update_kernel : one workgroup per env; bitonic sort of (cost, index) in LDS, then one lane
//                       per solution dimension accumulates the elites in sorted order (fixed
//                       order => bit-reproducible)
#include <hip/hip_runtime.h>

#include "rrl_device.hpp"
#include "rrl_host.hpp"

#pragma clang fp contract(off)

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

__device__ __forceinline__ double truncnorm2(uint64_t seed, uint32_t row, uint64_t ctr, uint32_t d) {
    for (uint32_t attempt = 0;; ++attempt) {
        double z0, z1;
        rrl::normal_at(seed, row, rrl::kStreamCem, (ctr << 20) | (uint64_t(d) << 8) | attempt, z0, z1);
        if ((z0 >= -2.0 && z0 <= 2.0) || attempt >= 255) return z0 < -2.0 ? -2.0 : (z0 > 2.0 ? 2.0 : z0);
    }
}

__global__ __launch_bounds__(kBlock) void cem_sample_kernel(int64_t M, int pop, int dim,
                                                            const double* mean, const double* var,
                                                            const double* lb, const double* ub,
                                                            double epsilon, int sticky, uint8_t* active,
                                                            uint64_t seed, uint64_t counter,
                                                            uint64_t* counter_dev, uint64_t counter_inc,
                                                            float* samples) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    const int64_t total = M * pop * dim;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
        const int d = int(e % dim);
        const int64_t row = e / dim;          // m * pop + i
        const int64_t m = row / pop;
        const double* v = var + m * dim;
        double vmax = v[0];
        for (int k = 1; k < dim; ++k) vmax = v[k] > vmax ? v[k] : vmax;
        bool act = vmax > epsilon;            // while-condition of optimizers.py:94
        if (sticky && !active[m]) act = false;
        if (d == 0 && row == m * pop) active[m] = uint8_t(act);
        if (!act) continue;
        const double mu = mean[m * dim + d];
        const double lo = (mu - lb[d]) / 2.0, hi = (ub[d] - mu) / 2.0;
        double cv = lo * lo < hi * hi ? lo * lo : hi * hi;
        cv = v[d] < cv ? v[d] : cv;
        const double z = truncnorm2(seed, uint32_t(row), ctr, uint32_t(d));
        samples[e] = float(z * sqrt(cv) + mu);
    }
    rrl::advance_counter(counter_dev, counter_inc);
}

__device__ __forceinline__ bool key_less(float ca, int ia, float cb, int ib) {
    return (ca < cb) | ((ca == cb) & (ia < ib));
}

__global__ __launch_bounds__(kBlock) void cem_update_kernel(int pop, int dim, int num_elites, int padded,
                                                            double alpha, const float* samples,
                                                            const float* costs, double* mean, double* var,
                                                            const uint8_t* active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* kc = (float*)smem;            // [padded] costs
    int* ki = (int*)(kc + padded);       // [padded] sample indices
    const int64_t m = blockIdx.x;
    if (active && !active[m]) return;
    for (int i = threadIdx.x; i < padded; i += kBlock) {
        float c = __int_as_float(0x7f800000);  // +inf padding sorts last
        if (i < pop) {
            c = costs[m * pop + i];
            c = (c != c) ? 1e6f : c;           // NaN -> 1e6 (MPC.py:415)
        }
        kc[i] = c;
        ki[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < padded; i += kBlock) {
                const int partner = i ^ j;
                if (partner > i) {
                    const bool up = (i & k) == 0;
                    const float ca = kc[i], cb = kc[partner];
                    const int ia = ki[i], ib = ki[partner];
                    const bool swap = up ? key_less(cb, ib, ca, ia) : key_less(ca, ia, cb, ib);
                    if (swap) {
                        kc[i] = cb; kc[partner] = ca;
                        ki[i] = ib; ki[partner] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    const int d = threadIdx.x;
    if (d < dim) {
        const float* base = samples + m * int64_t(pop) * dim;
        double sum = 0.0;
        for (int e = 0; e < num_elites; ++e) sum += double(base[int64_t(ki[e]) * dim + d]);
        const double em = sum / double(num_elites);
        double sq = 0.0;
        for (int e = 0; e < num_elites; ++e) {
            const double dv = double(base[int64_t(ki[e]) * dim + d]) - em;
            sq += dv * dv;
        }
        const double ev = sq / double(num_elites);
        mean[m * dim + d] = alpha * mean[m * dim + d] + (1.0 - alpha) * em;
        var[m * dim + d] = alpha * var[m * dim + d] + (1.0 - alpha) * ev;
    }
}

}  // namespace

extern "C" {

int rrl_cem_sample(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                   const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                   uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                   float* samples, void* stream) {
    if (!mean || !var || !lb || !ub || !active || !samples) return RRL_EINVAL;
    if (M < 0 || pop <= 0 || pop > 1024 || dim <= 0 || dim > 64) return RRL_ERANGE;
    if (M * pop > 0xffffffffLL) return RRL_ERANGE;
    if (M == 0) return RRL_OK;
    hipLaunchKernelGGL(cem_sample_kernel, dim3(grid_for(M * pop * dim)), dim3(kBlock), 0,
                       (hipStream_t)stream, M, pop, dim, mean, var, lb, ub, epsilon, sticky, active, seed,
                       counter, counter_dev, counter_inc, samples);
    return check_launch();
}

int rrl_cem_update(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                   const float* samples, const float* costs, double* mean, double* var,
                   const uint8_t* active, void* stream) {
    if (!samples || !costs || !mean || !var) return RRL_EINVAL;
    if (M < 0 || pop <= 0 || pop > 1024 || dim <= 0 || dim > 64) return RRL_ERANGE;
    if (num_elites <= 0 || num_elites > pop) return RRL_EINVAL;   // optimizers.py:66-68 raises ValueError
    if (M == 0) return RRL_OK;
    int padded = 1;
    while (padded < pop) padded <<= 1;
    hipLaunchKernelGGL(cem_update_kernel, dim3((unsigned)M), dim3(kBlock), size_t(padded) * 8,
                       (hipStream_t)stream, pop, dim, num_elites, padded, alpha, samples, costs, mean, var,
                       active);
    return check_launch();
}

}  // extern "C"
