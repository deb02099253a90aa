// maze_kernels.hip -- batched Maze env for gfx950 (MI355X).
//
// Counterpart of env/maze.py:139-232.  The reference advances MuJoCo 1.50 (500 sim steps per
// env step); that third-party engine is not available, so the kernels run the documented
// kinematic surrogate: straight-line displacement GAIN * a, stopped at the
// first of 64 sub-steps where the disc touches a wall rectangle or an arena plane.  Same I/O
// layout and launch shape as the navigation kernels: one lane per env, SoA, HBM-bound.
#include <hip/hip_runtime.h>

#include "maze_device.hpp"
#include "replay_device.hpp"
#include "rrl_device.hpp"
#include "rrl_host.hpp"
#include "step_push.hpp"

#pragma clang fp contract(off)

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

using namespace rrl_maze;

struct MazeArgs {
    int64_t n;
    double2* pos;
    const float2* action;
    uint64_t seed, counter;
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
    int32_t horizon, auto_reset;
};

__global__ __launch_bounds__(kBlock) void maze_step_kernel(MazeArgs a) {
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < a.n; i += stride) {
        const double2 p = a.pos[i];
        const float2 act = a.action[i];
        double x = p.x, y = p.y;
        move(x, y, double(act.x), double(act.y));
        int32_t ti = a.t[i] + 1;
        const bool cons = in_contact(x, y);
        const double d = goal_distance(x, y);
        const bool dn = (ti >= a.horizon) | cons | (d < kGoalThresh);
        const double rew = -d;
        const bool succ = rew > -0.03;
        const bool epd = dn | (ti == a.horizon);
        a.next_obs[i] = make_float2(float(x), float(y));
        a.reward[i] = float(rew);
        a.done[i] = uint8_t(dn);
        a.constraint[i] = uint8_t(cons);
        a.success[i] = uint8_t(succ);
        if (a.ep_done) a.ep_done[i] = uint8_t(epd);
        if (a.auto_reset && epd) {
            reset_one(a.seed, uint32_t(i), ctr, 0, true, x, y);
            ti = 0;
        }
        a.pos[i] = make_double2(x, y);
        a.t[i] = ti;
        if (a.obs) a.obs[i] = make_float2(float(x), float(y));
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

__global__ __launch_bounds__(kBlock) void maze_reset_kernel(int64_t n, double2* pos, float2* obs,
                                                            int32_t* t, const uint8_t* mask, int mode,
                                                            int check, uint64_t seed, uint64_t counter,
                                                            const uint64_t* counter_dev) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i]) continue;
        double x, y;
        reset_one(seed, uint32_t(i), ctr, mode, check != 0, x, y);
        pos[i] = make_double2(x, y);
        if (t) t[i] = 0;
        if (obs) obs[i] = make_float2(float(x), float(y));
    }
}

// one lane per 20-step segment (env/maze.py:41-52: reset every 20 transitions)
__global__ __launch_bounds__(kBlock) void maze_offline_kernel(int64_t half, int64_t n_seg, uint64_t seed,
                                                              float2* s, float2* a, float* c, float2* s2,
                                                              float* m) {
    const int64_t g_all = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (g_all >= 2 * n_seg) return;
    const int part = g_all >= n_seg;
    const int64_t g = g_all - part * n_seg;
    const uint32_t row = uint32_t(g_all);
    const rrl::Bits128 b = rrl::philox_at(seed, row, rrl::kStreamOffline, 0);
    const double sample = rrl::unit_open(b.lo);
    const int mode = sample < 0.3 ? 1 : (sample < 0.6 ? 2 : 0);
    double x, y;
    reset_one(seed, row, 1ULL << 40, mode, false, x, y);
    const int64_t len = (g == n_seg - 1) ? half - 20 * g : 20;
    int64_t w = part * half + 20 * g;
    int steps = 0;
    for (int64_t j = 0; j < len; ++j, ++w) {
        double ax, ay;
        if (part == 0) {
            const rrl::Bits128 u = rrl::philox_at(seed, row, rrl::kStreamOffline, uint64_t(1 + j));
            ax = double(float(-0.1 + 0.2 * rrl::unit_open(u.lo)));   // action_space.sample() is float32 (:58)
            ay = double(float(-0.1 + 0.2 * rrl::unit_open(u.hi)));
        } else {
            expert_action(x, y, ax, ay);                             // float64 into step() (:88-89)
        }
        const float axf = float(ax), ayf = float(ay);
        double nx = x, ny = y;
        move(nx, ny, ax, ay);
        steps += 1;
        const bool cons = in_contact(nx, ny);
        const bool dn = (steps >= 100) | cons | (goal_distance(nx, ny) < kGoalThresh);
        s[w] = make_float2(float(x), float(y));
        a[w] = make_float2(axf, ayf);
        c[w] = cons ? 1.0f : 0.0f;
        s2[w] = make_float2(float(nx), float(ny));
        m[w] = dn ? 0.0f : 1.0f;
        x = nx;
        y = ny;
    }
}

// the maze transition plugged into the fused step + push kernel (step_push.hpp)
struct MazeEnv {
    static __device__ __forceinline__ rrl_step::Outcome step(const StepArgs& a, int64_t, uint64_t, double2 p,
                                                             float2 act, int32_t t_after) {
        double x = p.x, y = p.y;
        move(x, y, double(act.x), double(act.y));
        const double d = goal_distance(x, y);
        rrl_step::Outcome o;
        o.x = x;
        o.y = y;
        o.reward = float(-d);
        o.constraint = in_contact(x, y);
        o.success = -d > -0.03;
        o.done = (t_after >= a.horizon) | o.constraint | (d < kGoalThresh);      // env/maze.py:207-213
        return o;
    }
    static __device__ __forceinline__ void reset(const StepArgs& a, int64_t i, uint64_t ctr, double& x, double& y) {
        reset_one(a.seed, uint32_t(i), ctr, 0, true, x, y);
    }
};

}  // namespace

extern "C" {

int rrl_maze_step(int64_t n, double* pos, const float* action, uint64_t seed, uint64_t counter,
                  uint64_t* counter_dev, uint64_t counter_inc, float* next_obs, float* obs,
                  float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                  uint8_t* ep_done, int32_t* t, int32_t horizon, int auto_reset, void* stream) {
    if (n < 0 || n > 0xffffffffLL) return RRL_ERANGE;
    if (!pos || !action || !next_obs || !reward || !done || !constraint || !success || !t)
        return RRL_EINVAL;
    if (n == 0) return RRL_OK;
    MazeArgs a{n, (double2*)pos, (const float2*)action, seed, counter, counter_dev, counter_inc,
               (float2*)next_obs, (float2*)obs, reward, done, constraint, success, ep_done, t, horizon,
               auto_reset};
    hipLaunchKernelGGL(maze_step_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return check_launch();
}

int rrl_maze_reset(int64_t n, double* pos, float* obs, int32_t* t, const uint8_t* mask, int mode,
                   int check_constraint, uint64_t seed, uint64_t counter,
                   const uint64_t* counter_dev, void* stream) {
    if (n < 0 || n > 0xffffffffLL) return RRL_ERANGE;
    if (!pos || mode < 0 || mode > 3) return RRL_EINVAL;
    if (n == 0) return RRL_OK;
    hipLaunchKernelGGL(maze_reset_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n,
                       (double2*)pos, (float2*)obs, t, mask, mode, check_constraint, seed, counter,
                       counter_dev);
    return check_launch();
}

int rrl_maze_offline(int64_t num_transitions, uint64_t seed, float* s, float* a, float* c, float* s2,
                     float* m, int64_t capacity, void* stream) {
    if (num_transitions < 0 || !s || !a || !c || !s2 || !m) return RRL_EINVAL;
    const int64_t half = num_transitions / 2;
    if (2 * half > capacity) return RRL_ERANGE;
    if (half == 0) return RRL_OK;
    const int64_t n_seg = (half + 19) / 20;
    hipLaunchKernelGGL(maze_offline_kernel, dim3((unsigned)((2 * n_seg + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, half, n_seg, seed, (float2*)s, (float2*)a, c,
                       (float2*)s2, m);
    return check_launch();
}

int rrl_maze_step_push(int64_t n, double* pos, int32_t* t, float* obs, const float* task_action,
                       const float* real_action, const uint8_t* recovery, uint64_t seed, uint64_t counter,
                       uint64_t* counter_dev, uint64_t counter_inc, int32_t horizon, int auto_reset,
                       float reward_penalty, int push_real_action, const rrl_replay_t* memory,
                       const rrl_replay_t* recovery_memory, float* next_obs, float* reward, uint8_t* done,
                       uint8_t* constraint, uint8_t* success, uint8_t* ep_done, uint64_t* stats, double* reward_sums,
                       float* ep_reward, void* stream) {
    rrl_step::StepPushArgs p;
    const int rc = rrl_step::fill_args(p, n, pos, t, obs, task_action, 2, real_action, recovery, nullptr, seed, counter,
                                       counter_dev, counter_inc, horizon, auto_reset, reward_penalty, push_real_action,
                                       memory, recovery_memory, next_obs, reward, done, constraint, success, ep_done,
                                       stats, reward_sums, ep_reward);
    if (rc != RRL_OK || n == 0) return rc;
    // latency regime: the reset draw (one Philox call + the contact test of the candidate; the start region is clear of
    // the walls, so the rejection loop runs once) beside the move instead of after it, and the early cursor ticket
    rrl_step::launch<MazeEnv>(p, n, (hipStream_t)stream);
    return check_launch();
}

int rrl_maze_step_push_select(int64_t n, double* pos, int32_t* t, float* obs, const float* task_action, int ld_task,
                              const float* z, int z_n_part, long long z_part_stride, float eps_safe, const float* rec_action, const rrl_policy_head_t* rec_head,
                             float* real_action,
                              uint8_t* recovery, uint64_t seed, uint64_t counter, uint64_t* counter_dev,
                              uint64_t counter_inc, int32_t horizon, int auto_reset, float reward_penalty,
                              int push_real_action, const rrl_replay_t* memory, const rrl_replay_t* recovery_memory,
                              float* next_obs, float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                              uint8_t* ep_done, uint64_t* stats, double* reward_sums, float* ep_reward, void* stream) {
    rrl_step::StepPushArgs p;
    const rrl_step::SelectIn sel{z, z_n_part, z_part_stride, eps_safe, rec_action, rec_head, real_action, recovery};
    const int rc = rrl_step::fill_args(p, n, pos, t, obs, task_action, ld_task, nullptr, nullptr, &sel, seed, counter,
                                       counter_dev, counter_inc, horizon, auto_reset, reward_penalty, push_real_action,
                                       memory, recovery_memory, next_obs, reward, done, constraint, success, ep_done,
                                       stats, reward_sums, ep_reward);
    if (rc != RRL_OK || n == 0) return rc;
    // latency regime: the reset draw (one Philox call + the contact test of the candidate; the start region is clear of
    // the walls, so the rejection loop runs once) beside the move instead of after it, and the early cursor ticket
    rrl_step::launch<MazeEnv>(p, n, (hipStream_t)stream);
    return check_launch();
}

int rrl_maze_step_push_packed(int S, const rrl_step_push_t* a, void* stream) {
    if (S <= 0 || S > rrl_pack::kMaxSeeds || !a) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_maze_step_push_x(&a[0], stream);
    rrl_pack::Key key;
    key.pod(7);
    key.pod(S);
    for (int s = 0; s < S; ++s) {
        key.pod(a[s]);
        if (a[s].memory) key.pod(*a[s].memory);
        if (a[s].recovery_memory) key.pod(*a[s].recovery_memory);
        if (a[s].sel_rec_head) key.pod(*a[s].sel_rec_head);
    }
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<rrl_step::StepPushArgs> ps(S);
        rrl_pack::Idx ix;
        ix.S = S;
        ix.first[0] = 0;
        const int regime = rrl_step::regime_of(a[0].n);
        for (int s = 0; s < S; ++s) {
            const int rc = rrl_step::fill_args(ps[s], &a[s]);
            if (rc != RRL_OK) return rc;
            if (a[s].n <= 0 || rrl_step::regime_of(a[s].n) != regime) return RRL_EINVAL;
            ix.first[s + 1] = ix.first[s] + rrl_step::grid_cover(a[s].n);
        }
        for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
        plan = rrl_pack::store(key, ps.data(), sizeof(rrl_step::StepPushArgs) * S, st);
        if (!plan) return rrl_pack::store_error();
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        plan->i0 = regime;
    }
    const auto* dev = (const rrl_step::StepPushArgs*)plan->dev;
    rrl_step::launch_pack<MazeEnv>(dev, plan->ix, plan->grid, plan->i0, st);
    return check_launch();
}

int rrl_maze_step_push_x(const rrl_step_push_t* a, void* stream) {
    rrl_step::StepPushArgs p;
    const int rc = rrl_step::fill_args(p, a);
    if (rc != RRL_OK || a->n == 0) return rc;
    rrl_step::launch<MazeEnv>(p, a->n, (hipStream_t)stream);
    return check_launch();
}

}  // extern "C"
