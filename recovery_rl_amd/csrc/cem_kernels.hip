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
                                                            float* samples, const int32_t* m_dev) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    if (m_dev) M = m_dev[0];                 // number of planning problems decided on the device (<= the launch bound)
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
    // an EMPTY device-counted planning set leaves the tick alone, as the host-count path (MPC.act returns before it draws)
    rrl::advance_counter(counter_dev, (m_dev && M == 0) ? 0 : counter_inc);
}

__device__ __forceinline__ bool key_less(float ca, int ia, float cb, int ib) {
    return (ca < cb) | ((ca == cb) & (ia < ib));
}

__global__ __launch_bounds__(kBlock) void cem_update_kernel(int pop, int dim, int num_elites, int padded,
                                                            double alpha, const float* samples,
                                                            const float* costs, double* mean, double* var,
                                                            const uint8_t* active, const int32_t* m_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* kc = (float*)smem;            // [padded] costs
    int* ki = (int*)(kc + padded);       // [padded] sample indices
    const int64_t m = blockIdx.x;
    if (m_dev && m >= m_dev[0]) return;
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

// ---- planning set decided on the device (no host round trip in MPC.act) ----------------------------------------
// cem_begin: ONE workgroup.  idx[0..count) = the rows with mask != 0 in ascending order (what mask.nonzero() lists),
// count[0] = their number; then the planner's inputs are gathered for the compacted problems j < count:
// mean[j] = prev_sol[idx[j]], var[j] = init_var, cur_obs[j] = obs[idx[j]], active[j] = 1  (MPC.py:336-341).
constexpr int kBeginBlock = 1024;
__global__ __launch_bounds__(kBeginBlock) void cem_begin_kernel(int64_t n, const uint8_t* __restrict__ mask, int dim,
                                                                const double* __restrict__ prev_sol,
                                                                const double* __restrict__ init_var,
                                                                const float2* __restrict__ obs, int32_t* __restrict__ idx,
                                                                int32_t* __restrict__ count, double* __restrict__ mean,
                                                                double* __restrict__ var, float2* __restrict__ cur_obs,
                                                                uint8_t* __restrict__ active) {
    __shared__ int wave_cnt[kBeginBlock / 64];
    __shared__ int base_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_sh = 0;
    __syncthreads();
    for (int64_t first = 0; first < n; first += kBeginBlock) {
        const int64_t i = first + threadIdx.x;
        const bool on = i < n && mask[i] != 0;
        const unsigned long long b = __ballot(on);
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int before = base_sh, total = 0;
        for (int w = 0; w < kBeginBlock / 64; ++w) {
            const int c = wave_cnt[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (on) idx[before + __popcll(b & ((1ULL << lane) - 1))] = int32_t(i);
        __syncthreads();
        if (threadIdx.x == 0) base_sh += total;
        __syncthreads();
    }
    const int m = base_sh;
    if (threadIdx.x == 0) count[0] = m;
    __threadfence_block();
    __syncthreads();
    for (int64_t e = threadIdx.x; e < int64_t(m) * dim; e += kBeginBlock) {
        const int64_t j = e / dim;
        const int d = int(e % dim);
        mean[e] = prev_sol[int64_t(idx[j]) * dim + d];
        var[e] = init_var[d];
    }
    for (int j = threadIdx.x; j < m; j += kBeginBlock) {
        cur_obs[j] = obs[idx[j]];
        active[j] = 1;
    }
}

// cem_finish: action[i] = the first `du` entries of the solution of env i's problem (0 for envs that did not plan);
// prev_sol[i] = the solution shifted by one step, zero-filled (MPC.py:342-344).
__global__ __launch_bounds__(kBlock) void cem_finish_kernel(int64_t n, const uint8_t* __restrict__ mask, int dim, int du,
                                                            const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ count,
                                                            const double* __restrict__ mean, double* __restrict__ prev_sol,
                                                            float* __restrict__ action) {
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < n * du; e += stride)
        if (!mask[e / du]) action[e] = 0.f;
    const int64_t total = int64_t(count[0]) * dim;
    for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e / dim;
        const int d = int(e % dim);
        const int64_t i = idx[j];
        if (d < du) action[i * du + d] = float(mean[e]);
        prev_sol[i * dim + d] = d + du < dim ? mean[e + du] : 0.0;
    }
}

}  // namespace

extern "C" {

static int cem_sample_impl(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                           const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                           uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                           float* samples, const int32_t* m_dev, void* stream) {
    if (!mean || !var || !lb || !ub || !active || !samples) return RRL_EINVAL;
    if (M < 0 || pop <= 0 || pop > 1024 || dim <= 0 || dim > 64) return RRL_ERANGE;
    if (M * pop > 0xffffffffLL) return RRL_ERANGE;
    if (M == 0) return RRL_OK;
    hipLaunchKernelGGL(cem_sample_kernel, dim3(grid_for(M * pop * dim)), dim3(kBlock), 0,
                       (hipStream_t)stream, M, pop, dim, mean, var, lb, ub, epsilon, sticky, active, seed,
                       counter, counter_dev, counter_inc, samples, m_dev);
    return check_launch();
}

static int cem_update_impl(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                           const float* samples, const float* costs, double* mean, double* var,
                           const uint8_t* active, const int32_t* m_dev, void* stream) {
    if (!samples || !costs || !mean || !var) return RRL_EINVAL;
    if (M < 0 || pop <= 0 || pop > 1024 || dim <= 0 || dim > 64) return RRL_ERANGE;
    if (num_elites <= 0 || num_elites > pop) return RRL_EINVAL;   // optimizers.py:66-68 raises ValueError
    if (M == 0) return RRL_OK;
    int padded = 1;
    while (padded < pop) padded <<= 1;
    hipLaunchKernelGGL(cem_update_kernel, dim3((unsigned)M), dim3(kBlock), size_t(padded) * 8,
                       (hipStream_t)stream, pop, dim, num_elites, padded, alpha, samples, costs, mean, var,
                       active, m_dev);
    return check_launch();
}

int rrl_cem_sample(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                   const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                   uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                   float* samples, void* stream) {
    return cem_sample_impl(M, pop, dim, mean, var, lb, ub, epsilon, sticky, active, seed, counter, counter_dev,
                           counter_inc, samples, nullptr, stream);
}

int rrl_cem_update(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                   const float* samples, const float* costs, double* mean, double* var,
                   const uint8_t* active, void* stream) {
    return cem_update_impl(M, pop, dim, num_elites, alpha, samples, costs, mean, var, active, nullptr, stream);
}

int rrl_cem_sample_n(const int32_t* m_dev, int64_t m_max, int32_t pop, int32_t dim, const double* mean,
                     const double* var, const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                     uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* samples,
                     void* stream) {
    if (!m_dev) return RRL_EINVAL;
    return cem_sample_impl(m_max, pop, dim, mean, var, lb, ub, epsilon, sticky, active, seed, counter, counter_dev,
                           counter_inc, samples, m_dev, stream);
}

int rrl_cem_update_n(const int32_t* m_dev, int64_t m_max, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                     const float* samples, const float* costs, double* mean, double* var, const uint8_t* active,
                     void* stream) {
    if (!m_dev) return RRL_EINVAL;
    return cem_update_impl(m_max, pop, dim, num_elites, alpha, samples, costs, mean, var, active, m_dev, stream);
}

int rrl_cem_begin(int64_t n, const uint8_t* mask, int32_t dim, const double* prev_sol, const double* init_var,
                  const float* obs, int32_t* idx, int32_t* count, double* mean, double* var, float* cur_obs,
                  uint8_t* active, void* stream) {
    if (!mask || !prev_sol || !init_var || !obs || !idx || !count || !mean || !var || !cur_obs || !active)
        return RRL_EINVAL;
    if (n <= 0 || n > 0x7fffffffLL || dim <= 0 || dim > 64) return RRL_ERANGE;
    hipLaunchKernelGGL(cem_begin_kernel, dim3(1), dim3(kBeginBlock), 0, (hipStream_t)stream, n, mask, dim, prev_sol,
                       init_var, (const float2*)obs, idx, count, mean, var, (float2*)cur_obs, active);
    return check_launch();
}

int rrl_cem_finish(int64_t n, const uint8_t* mask, int32_t dim, int32_t du, const int32_t* idx, const int32_t* count,
                   const double* mean, double* prev_sol, float* action, void* stream) {
    if (!mask || !idx || !count || !mean || !prev_sol || !action) return RRL_EINVAL;
    if (n <= 0 || n > 0x7fffffffLL || dim <= 0 || dim > 64 || du <= 0 || du > dim) return RRL_ERANGE;
    hipLaunchKernelGGL(cem_finish_kernel, dim3(grid_for(n * dim)), dim3(kBlock), 0, (hipStream_t)stream, n, mask, dim,
                       du, idx, count, mean, prev_sol, action);
    return check_launch();
}

}  // extern "C"
