// maze_device.hpp -- device-side Maze surrogate shared by the stand-alone maze kernels and
// the fused step + push kernel.  Counterpart of env/maze.py:139-232.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrl_device.hpp"

#pragma clang fp contract(off)

namespace rrl_maze {

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
    // four sub-steps per iteration: their positions and contact tests are independent (4x the instruction-level
    // parallelism of the one-by-one scan), the first one in contact wins -- same values, same result
    for (int k0 = 1; k0 <= kSubsteps; k0 += 4) {
        double px[4], py[4];
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double f = double(k0 + u) * (1.0 / kSubsteps);
            px[u] = clampd(x + dx * f, -kLim, kLim);
            py[u] = clampd(y + dy * f, -kLim, kLim);
            hit[u] = in_contact(px[u], py[u]);
        }
        const int first = hit[0] ? 0 : (hit[1] ? 1 : (hit[2] ? 2 : 3));
        qx = px[first];
        qy = py[first];
        if (hit[0] | hit[1] | hit[2] | hit[3]) break;
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

}  // namespace rrl_maze
