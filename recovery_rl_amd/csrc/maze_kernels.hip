// maze_kernels.hip -- batched Maze env for gfx950 (MI355X).
//
// Counterpart of env/maze.py:139-232.  The reference advances MuJoCo 1.50 (500 sim steps per
// env step); that third-party engine is not available, so the kernels run the documented
// kinematic surrogate: straight-line displacement GAIN * a, stopped at the
// first of 64 sub-steps where the disc touches a wall rectangle or an arena plane.  Same I/O
// layout and launch shape as the navigation kernels: one lane per env, SoA, HBM-bound.
#include <hip/hip_runtime.h>

#include "rrl_device.hpp"
#include "rrl_host.hpp"

#pragma clang fp contract(off)

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

constexpr double kGain = 0.24667750873451577;  // m per unit control per env step (500 x 2 ms from rest)
constexpr double kRadius = 0.025;              // simple_maze.xml:28
constexpr double kLim = 0.3;                   // arena planes / joint range
constexpr double kMaxForce = 0.1;              // env/maze.py:17
constexpr double kGoalX = 0.25, kGoalY = 0.0;  // env/maze.py:135-137
constexpr double kGoalThresh = 0.03;           // env/maze.py:19
constexpr int kSubsteps = 64;

__device__ __forceinline__ bool touches_wall(double x, double y, double cx, double cy) {
    double dx = fabs(x - cx) - 0.005, dy = fabs(y - cy) - 0.2;  // half sizes, simple_maze.xml:22-25
    dx = dx < 0.0 ? 0.0 : dx;
    dy = dy < 0.0 ? 0.0 : dy;
    return dx * dx + dy * dy <= kRadius * kRadius;
}

// ncon > 3  <=>  the disc touches an arena plane or one of the four walls (env/maze.py:199-206)
__device__ __forceinline__ bool in_contact(double x, double y) {
    const bool plane = (kLim - x <= kRadius) | (x + kLim <= kRadius) | (kLim - y <= kRadius) |
                       (y + kLim <= kRadius);
    return plane | touches_wall(x, y, -0.1, 0.42) | touches_wall(x, y, 0.1, 0.48) |
           touches_wall(x, y, -0.1, -0.33) | touches_wall(x, y, 0.1, -0.17);
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ double goal_distance(double x, double y) {
    const double ex = kGoalX - x, ey = kGoalY - y;
    return sqrt((ex * ex + ey * ey) / 2.0);  // sqrt(mean(sq)), env/maze.py:219
}

__device__ __forceinline__ void move(double& x, double& y, double ax, double ay) {
    ax = clampd(ax, -kMaxForce, kMaxForce);
    ay = clampd(ay, -kMaxForce, kMaxForce);
    if (in_contact(x, y)) return;  // env/maze.py:144-147: no sim steps while in contact
    const double dx = kGain * ax, dy = kGain * ay;
    double qx = x, qy = y;
    for (int k = 1; k <= kSubsteps; ++k) {
        const double f = double(k) * (1.0 / kSubsteps);
        qx = clampd(x + dx * f, -kLim, kLim);
        qy = clampd(y + dy * f, -kLim, kLim);
        if (in_contact(qx, qy)) break;
    }
    x = qx;
    y = qy;
}

__device__ __forceinline__ void reset_one(uint64_t seed, uint32_t row, uint64_t counter, int mode,
                                          bool check, double& x, double& y) {
    for (uint32_t r = 0;; ++r) {
        const rrl::Bits128 b = rrl::philox_at(seed, row, rrl::kStreamReset, counter | (uint64_t(r) << 48));
        const double u0 = rrl::unit_open(b.lo), u1 = rrl::unit_open(b.hi);
        if (mode == 1) x = 0.14 + 0.08 * u0;
        else if (mode == 2) x = -0.04 + 0.08 * u0;
        else if (mode == 3) x = -0.27 + 0.54 * u0;
        else x = -0.22 + 0.09 * u0;
        y = -0.22 + 0.44 * u1;
        if (!check || !in_contact(x, y) || r >= 1000) return;
    }
}

__device__ __forceinline__ void expert_action(double x, double y, double& ax, double& ay) {
    double tx, ty;  // env/maze.py:222-232
    if (x <= -0.151) { tx = -0.15; ty = -0.125; }
    else if (x <= 0.149) { tx = 0.15; ty = 0.125; }
    else { tx = kGoalX; ty = kGoalY; }
    ax = 1.05 * (tx - x);
    ay = 1.05 * (ty - y);
}

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
            ax = -0.1 + 0.2 * rrl::unit_open(u.lo);
            ay = -0.1 + 0.2 * rrl::unit_open(u.hi);
        } else {
            expert_action(x, y, ax, ay);
        }
        const float axf = float(ax), ayf = float(ay);
        double nx = x, ny = y;
        move(nx, ny, double(axf), double(ayf));
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

}  // extern "C"
