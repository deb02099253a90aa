// log_kernels.hip -- device-side episode log: what the reference keeps as per-step info dicts in
// run_stats.pkl (recovery_rl/experiment.py:421,456-461,540-543) reduced, per finished episode, to the
// quantities its plotting code derives from them (plotting/plot_runs.py:194-235): length, return, last
// reward, number of constraint steps, number of recovery steps, final flags.  gfx950 only.
#include "rrl_host.hpp"

using namespace rrl_host;

namespace {

// state = {count, iteration, ticket}.  Records are appended in completion order (atomic slot); the
// host sorts by (iteration, env), which is unique, so the drained table is deterministic.
__global__ __launch_bounds__(kBlock) void episode_log_kernel(
    int64_t n, const float* __restrict__ reward, const uint8_t* __restrict__ constraint,
    const uint8_t* __restrict__ success, const uint8_t* __restrict__ ep_done,
    const uint8_t* __restrict__ recovery, int32_t* __restrict__ ep_len, double* __restrict__ ep_ret,
    int32_t* __restrict__ ep_viol, int32_t* __restrict__ ep_rec, int32_t* __restrict__ rec_i32,
    double* __restrict__ rec_f64, int64_t cap, int64_t* __restrict__ state) {
    const int64_t iteration = state[1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_iter = (n + stride - 1) / stride;       // uniform trip count: the ballot needs whole waves
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t i = it * stride + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
        const bool live = i < n;
        float r = 0.f;
        int c = 0, rec = 0, len = 0, viol = 0, recs = 0;
        double ret = 0.0;
        bool finished = false;
        if (live) {
            r = reward[i];
            c = constraint[i] != 0;
            rec = recovery ? (recovery[i] != 0) : 0;
            len = ep_len[i] + 1;
            ret = ep_ret[i] + (double)r;
            viol = ep_viol[i] + c;
            recs = ep_rec[i] + rec;
            finished = ep_done[i] != 0;
        }
        // one atomic per wave reserves the slots of all its finished episodes (one per lane serialised at large n)
        const unsigned long long bal = __ballot(finished);
        long long base = 0;
        if (bal) {
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
            if (lane == leader) base = (long long)atomicAdd((unsigned long long*)&state[0], (unsigned long long)__popcll(bal));
            base = __shfl(base, leader, 64);
            if (finished) {
                const long long slot = base + __popcll(bal & ((1ULL << lane) - 1ULL));
                if (slot < cap) {
                    int32_t* ri = rec_i32 + slot * RRL_EPLOG_I32;
                    ri[0] = (int32_t)i;
                    ri[1] = (int32_t)iteration;
                    ri[2] = len;
                    ri[3] = viol;
                    ri[4] = recs;
                    ri[5] = (success[i] ? 1 : 0) | (c ? 2 : 0) | (rec ? 4 : 0);
                    rec_f64[slot * 2 + 0] = ret;
                    rec_f64[slot * 2 + 1] = (double)r;
                }
            }
        }
        if (live) {
            ep_len[i] = finished ? 0 : len;
            ep_ret[i] = finished ? 0.0 : ret;
            ep_viol[i] = finished ? 0 : viol;
            ep_rec[i] = finished ? 0 : recs;
        }
    }
    // iteration += 1 by the last workgroup to finish (every workgroup read it before its ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd((unsigned long long*)&state[2], 1ULL);
        if (ticket == gridDim.x - 1) {
            state[1] = iteration + 1;
            state[2] = 0;
        }
    }
}

}  // namespace

extern "C" {

int rrl_episode_log_append(int64_t n, const float* reward, const uint8_t* constraint, const uint8_t* success,
                           const uint8_t* ep_done, const uint8_t* recovery, int32_t* ep_len, double* ep_ret,
                           int32_t* ep_viol, int32_t* ep_rec, const rrl_episode_log_t* log, void* stream) {
    if (!reward || !constraint || !success || !ep_done || !ep_len || !ep_ret || !ep_viol || !ep_rec || !log ||
        !log->rec_i32 || !log->rec_f64 || !log->state || log->cap <= 0 || n <= 0)
        return RRL_EINVAL;
    hipLaunchKernelGGL(episode_log_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reward,
                       constraint, success, ep_done, recovery, ep_len, ep_ret, ep_viol, ep_rec, log->rec_i32,
                       log->rec_f64, log->cap, log->state);
    return check_launch();
}

}  // extern "C"
