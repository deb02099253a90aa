// step_push.hpp -- the lock-step iteration tail as ONE kernel, shared by the navigation and maze envs:
// env.step + reward penalty + bootstrap mask + memory.push + recovery_memory.push + episode counters
// (recovery_rl/experiment.py:420-461).  ENV supplies the transition and the reset draw:
//   static Outcome ENV::step(const StepArgs&, int64_t i, uint64_t ctr, double2 pos, float2 action, int32_t t_after)
//   static void    ENV::reset(const StepArgs&, int64_t i, uint64_t ctr, double& x, double& y)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "replay_device.hpp"
#include "rrl_device.hpp"
#include "rrl_host.hpp"

struct StepArgs {
    int64_t n;
    double2* pos;
    const float2* action;
    const double2* noise;
    uint64_t seed;
    uint64_t counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    float2* next_obs;
    float2* obs;
    float* reward;
    uint8_t* done;
    uint8_t* constraint;
    uint8_t* success;
    uint8_t* ep_done;
    int32_t* t;
    int32_t horizon;
    int32_t auto_reset;
};


namespace rrl_step {

struct Outcome {
    double x, y;        // next position (before an auto-reset)
    float reward;
    bool constraint, success, done;   // done = the env's own termination (the horizon is added by the kernel)
};

// Counterpart of the body of recovery_rl/experiment.py:420-461 for n envs: env.step, reward penalty,
// mask = not done (before the horizon check), memory.push, recovery_memory.push, episode counters.
struct StepPushArgs {
    StepArgs step;            // obs = observation buffer: read as the pre-step state, then overwritten
    const float2* task_action;
    const uint8_t* recovery;  // nullable
    float reward_penalty;
    int push_real_action;     // disable_action_relabeling (experiment.py:437-441)
    rrl_replay_t memory;
    rrl_replay_t recovery_memory;
    int use_recovery_memory;
    unsigned long long* stats;   // env_steps, episodes, num_viols, viol_and_recovery, viol_and_no_recovery,
                                 // num_successes, recovery_steps, constraint_steps
    double* reward_sums;         // {sum of rewards, sum of finished-episode returns}
    float* ep_reward;            // [n] running episode return
};

// Episode counters: every lane keeps its own tallies over the grid-stride loop; they are added up per wave
// (ballot-free shuffles), then per workgroup in LDS, and ONE atomic per workgroup and counter reaches memory.
// (One atomic per wave and iteration serialised 115 k atomics on 7 addresses at 2^20 envs: 719 us.)
constexpr int kCounters = 7;

template <class ENV>
__global__ __launch_bounds__(rrl_host::kBlock) void step_push_kernel(StepPushArgs p) {
    constexpr int kBlock = rrl_host::kBlock;
    const StepArgs& a = p.step;
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t mpos = p.memory.state[0], msize = p.memory.state[1];
    int64_t rpos = 0, rsize = 0;
    if (p.use_recovery_memory) { rpos = p.recovery_memory.state[0]; rsize = p.recovery_memory.state[1]; }
    double rsum = 0.0, retsum = 0.0;
    unsigned cnt[kCounters] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int64_t n_iter = (a.n + stride - 1) / stride;   // uniform trip count: ballots need whole waves
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t i = it * stride + int64_t(blockIdx.x) * kBlock + threadIdx.x;
        const bool live = i < a.n;
        bool cons = false, succ = false, epd = false, rec = false;
        if (live) {
            const double2 pp = a.pos[i];
            const float2 act = a.action[i];
            const float2 prev = a.obs[i];
            int32_t ti = a.t[i];
            ti += 1;
            const Outcome out = ENV::step(a, i, ctr, pp, act, ti);
            double nx = out.x, ny = out.y;
            cons = out.constraint;
            succ = out.success;
            const bool dn = out.done;
            epd = dn | (ti == a.horizon);
            rec = p.recovery ? p.recovery[i] != 0 : false;
            const float2 nobs = make_float2(float(nx), float(ny));
            const float rew = out.reward;
            a.next_obs[i] = nobs;
            a.reward[i] = rew;
            a.done[i] = uint8_t(dn);
            a.constraint[i] = uint8_t(cons);
            a.success[i] = uint8_t(succ);
            if (a.ep_done) a.ep_done[i] = uint8_t(epd);
            // replay rows (experiment.py:431-448)
            const float mask = dn ? 0.0f : 1.0f;
            const float prew = rew - (cons ? p.reward_penalty : 0.0f);
            const float2 stored = p.push_real_action ? act : p.task_action[i];
            rrl_replay::store_values(p.memory, (mpos + i) % p.memory.cap, msize, prev, stored, prew, nobs, mask);
            if (p.use_recovery_memory)
                rrl_replay::store_values(p.recovery_memory, (rpos + i) % p.recovery_memory.cap, rsize, prev, act,
                                         cons ? 1.0f : 0.0f, nobs, mask);
            // episode accounting
            const float er = p.ep_reward[i] + rew;
            rsum += double(rew);
            if (epd) retsum += double(er);
            p.ep_reward[i] = epd ? 0.0f : er;
            if (a.auto_reset && epd) {
                ENV::reset(a, i, ctr, nx, ny);
                ti = 0;
            }
            a.pos[i] = make_double2(nx, ny);
            a.t[i] = ti;
            a.obs[i] = make_float2(float(nx), float(ny));
        }
        const bool end_viol = epd & cons;
        cnt[0] += epd;
        cnt[1] += end_viol;
        cnt[2] += end_viol & rec;
        cnt[3] += end_viol & !rec;
        cnt[4] += epd & succ;
        cnt[5] += live & rec;
        cnt[6] += cons;
    }
    __shared__ unsigned block_cnt[kCounters];
    __shared__ double block_sum[2][kBlock / 64];
    if (threadIdx.x < kCounters) block_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        rsum += __shfl_down(rsum, off);
        retsum += __shfl_down(retsum, off);
#pragma unroll
        for (int k = 0; k < kCounters; ++k) cnt[k] += __shfl_down(cnt[k], off);
    }
    if ((threadIdx.x & 63) == 0) {
        block_sum[0][threadIdx.x >> 6] = rsum;
        block_sum[1][threadIdx.x >> 6] = retsum;
#pragma unroll
        for (int k = 0; k < kCounters; ++k)
            if (cnt[k]) atomicAdd(&block_cnt[k], cnt[k]);
    }
    __syncthreads();
    if (threadIdx.x < kCounters) {
        if (block_cnt[threadIdx.x]) atomicAdd(p.stats + 1 + threadIdx.x, (unsigned long long)block_cnt[threadIdx.x]);
    } else if (threadIdx.x < kCounters + 2) {
        const int w = threadIdx.x - kCounters;
        double sum = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) sum += block_sum[w][k];       // fixed order inside the workgroup
        if (sum != 0.0) atomicAdd(p.reward_sums + w, sum);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(p.stats, (unsigned long long)a.n);
    rrl_replay::advance_ring(p.memory, mpos, msize, a.n);
    if (p.use_recovery_memory) rrl_replay::advance_ring(p.recovery_memory, rpos, rsize, a.n);
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}


}  // namespace rrl_step
