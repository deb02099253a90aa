// maze_device.hpp -- device-side Maze surrogate shared by the stand-alone maze kernels and
// the fused step + push kernel.  Counterpart of env/maze.py:139-232.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrl_device.hpp"

#pragma clang fp contract(off)

// the geometry below is plain IEEE-double C++: it also compiles for the host, where tests/test_maze_host_geometry.py
// runs it against the CPU checker's sequential scan (no GPU needed)
#define RRL_HD __host__ __device__ __forceinline__

namespace rrl_maze {

constexpr double kGain = 0.24667750873451577;  // m per unit control per env step (500 x 2 ms from rest)
constexpr double kRadius = 0.025;              // simple_maze.xml:28
constexpr double kLim = 0.3;                   // arena planes / joint range
constexpr double kMaxForce = 0.1;              // env/maze.py:17
constexpr double kGoalX = 0.25, kGoalY = 0.0;  // env/maze.py:135-137
constexpr double kGoalThresh = 0.03;           // env/maze.py:19
constexpr int kSubsteps = 64;

// wall rectangles after reset() moved them (env/maze.py:199-206; geoms 5..8 = wall1A, wall2A, wall1B, wall2B):
// y centres 0.5 + w1, 0.4 + w2, -0.25 + w1, -0.25 + w2 with w1 = -0.08, w2 = 0.08 -- the sums as the reference forms them
// (0.4 + 0.08 is not the double nearest to 0.48); half sizes 0.005 x 0.2 (simple_maze.xml:22-25)
constexpr double kWallHX = 0.005, kWallHY = 0.2;
RRL_HD double wall_x(int j) { return (j & 1) ? 0.1 : -0.1; }
RRL_HD double wall_y(int j) {
    return j == 0 ? 0.5 + -0.08 : (j == 1 ? 0.4 + 0.08 : (j == 2 ? -0.25 + -0.08 : -0.25 + 0.08));
}

RRL_HD bool touches_wall(double x, double y, double cx, double cy) {
    double dx = fabs(x - cx) - kWallHX, dy = fabs(y - cy) - kWallHY;
    dx = dx < 0.0 ? 0.0 : dx;
    dy = dy < 0.0 ? 0.0 : dy;
    return dx * dx + dy * dy <= kRadius * kRadius;
}

RRL_HD bool touches_plane(double x, double y) {
    return (kLim - x <= kRadius) | (x + kLim <= kRadius) | (kLim - y <= kRadius) | (y + kLim <= kRadius);
}

// ncon > 3  <=>  the disc touches an arena plane or one of the four walls
RRL_HD bool in_contact(double x, double y) {
    return touches_plane(x, y) | touches_wall(x, y, wall_x(0), wall_y(0)) | touches_wall(x, y, wall_x(1), wall_y(1)) |
           touches_wall(x, y, wall_x(2), wall_y(2)) | touches_wall(x, y, wall_x(3), wall_y(3));
}

RRL_HD double clampd(double v, double lo, double hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

RRL_HD double goal_distance(double x, double y) {
    const double ex = kGoalX - x, ey = kGoalY - y;
    return sqrt((ex * ex + ey * ey) / 2.0);  // sqrt(mean(sq)), env/maze.py:219
}

// Sub-step k of the move (k = 1..64): the position the sequential scan of the specification visits
RRL_HD void substep_pos(double x, double y, double dx, double dy, int k, double& px, double& py) {
    const double f = double(k) * (1.0 / kSubsteps);
    px = clampd(x + dx * f, -kLim, kLim);
    py = clampd(y + dy * f, -kLim, kLim);
}

// The bracket [t_in, t_out] and the jump lengths are ESTIMATES that only have to err on the safe side by less than the
// one or two sub-steps of margin the scan keeps (the exact f64 predicate decides every visited sub-step), so they are
// computed in f32 with v_rcp_f32 / v_sqrt_f32 (a handful of instructions instead of the ~25-instruction f64 division and
// square-root sequences: with one wave per SIMD the step kernel's time IS this dependent chain).  The slabs are widened by
// 1e-6 (f32 rounding of 0.3-sized coordinates is 3e-8) and a slightly negative discriminant counts as a grazing root.
RRL_HD void slab_interval_f32(float p, float d, float c, float h, float& lo, float& hi) {
    const float a = (c - h) - p, b = (c + h) - p;
    if (d == 0.0f) {
        const bool inside = (a <= 0.0f) & (b >= 0.0f);
        lo = inside ? -1e30f : 1e30f;
        hi = inside ? 1e30f : -1e30f;
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const float inv = __builtin_amdgcn_rcpf(d);
#else
    const float inv = 1.0f / d;
#endif
    const float t0 = a * inv, t1 = b * inv;
    lo = t0 < t1 ? t0 : t1;
    hi = t0 < t1 ? t1 : t0;
}

// Entry parameter (in sub-steps, may be < 0 or > 64) of the segment p + t d, t in [0, 64], into the slab |v - c| <= h
// of one coordinate; lo/hi = the parameter interval inside the slab.
RRL_HD void slab_interval(double p, double d, double c, double h, double& lo, double& hi) {
    const double a = (c - h) - p, b = (c + h) - p;
    if (d == 0.0) {
        const bool inside = (a <= 0.0) & (b >= 0.0);
        lo = inside ? -1e300 : 1e300;
        hi = inside ? 1e300 : -1e300;
        return;
    }
    const double t0 = a / d, t1 = b / d;
    lo = t0 < t1 ? t0 : t1;
    hi = t0 < t1 ? t1 : t0;
}

// The specification scans 64 equal sub-steps one by one and stops
// at the first that is in contact.  Every obstacle (a wall rectangle inflated by the disc radius, a half plane) is
// convex, so along the segment its contact set is one interval of sub-steps: the first sub-step in contact with
// obstacle j is ceil(entry_j) give or take rounding, where entry_j comes from a slab test against the inflated
// bounding box (a superset of the rounded rectangle: it can only be early, never late).  The kernel evaluates the
// EXACT predicate of the specification on the few candidates from each entry point onward (until the first hit, at
// most to the exit of the box) instead of on all 64 positions: same positions, same predicate, same result.
RRL_HD void move(double& x, double& y, double ax, double ay) {
    ax = clampd(ax, -kMaxForce, kMaxForce);
    ay = clampd(ay, -kMaxForce, kMaxForce);
    if (in_contact(x, y)) return;  // env/maze.py:144-147: no sim steps while in contact
    const double dx = kGain * ax, dy = kGain * ay;
    const double sx = dx * (1.0 / kSubsteps), sy = dy * (1.0 / kSubsteps);   // per sub-step, for the interval estimate only
    int first = kSubsteps + 1;                                             // first sub-step in contact with anything
    // arena planes: contact where |v| >= kLim - kRadius, i.e. outside the slab |v| < 0.275
    {
        float lo, hi, t_out = 1e30f;
        constexpr float kNarrow = 1e-6f;        // the slab shrunk: the estimated exit can only be early
        slab_interval_f32(float(x), float(sx), 0.0f, float(kLim - kRadius) - kNarrow, lo, hi);   // inside for t in [lo, hi]
        t_out = hi < t_out ? hi : t_out;
        slab_interval_f32(float(y), float(sy), 0.0f, float(kLim - kRadius) - kNarrow, lo, hi);
        t_out = hi < t_out ? hi : t_out;
        // the move leaves the free square at t_out: candidates from two sub-steps before it
        int k = t_out >= float(kSubsteps + 2) ? kSubsteps + 1 : (t_out <= 2.0f ? 1 : int(t_out) - 2);
        k = k < 1 ? 1 : k;
        for (; k <= kSubsteps && k < first; ++k) {
            double px, py;
            substep_pos(x, y, dx, dy, k, px, py);
            if (touches_plane(px, py)) { first = k; break; }
        }
    }
    // every sub-step position lies in the box spanned by the start and the end of the move (rounding is monotone in k, and
    // the start is inside the arena, so clamping keeps it there): a wall whose inflated rectangle misses that box in x or
    // in y cannot be touched at any sub-step (tested with a slack of 1e-9).  A move is at most 0.025 long, so this settles three or four of the four
    // walls with comparisons alone -- no divisions, no candidate loop.
    const double ex = clampd(x + dx, -kLim, kLim), ey = clampd(y + dy, -kLim, kLim);
    const double bx0 = x < ex ? x : ex, bx1 = x < ex ? ex : x, by0 = y < ey ? y : ey, by1 = y < ey ? ey : y;
    // The walls are 0.2 apart in x and 0.3 apart in y, a move is 0.025 long: at most ONE wall survives the box test, so
    // the search below exists once, with that wall's centre as data, instead of four times behind four branches (with one
    // wave per SIMD the kernel's time is the number of instructions the wave walks through: every copy any lane needs).
    // The loop form keeps it correct for any geometry: it simply runs once per surviving wall.
    unsigned survivors = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr double kSlack = 1e-9;     // far above any rounding of the predicate, far below the geometry
        const bool out = wall_x(j) + (kWallHX + kRadius + kSlack) < bx0 || wall_x(j) - (kWallHX + kRadius + kSlack) > bx1 ||
                         wall_y(j) + (kWallHY + kRadius + kSlack) < by0 || wall_y(j) - (kWallHY + kRadius + kSlack) > by1;
        survivors |= out ? 0u : (1u << j);
    }
    while (survivors) {
        const int j = __builtin_ctz(survivors);
        survivors &= survivors - 1;
        const double wcx = wall_x(j), wcy = wall_y(j);
        float lox, hix, loy, hiy;
        constexpr float kWide = 1e-6f;
        slab_interval_f32(float(x), float(sx), float(wcx), float(kWallHX + kRadius) + kWide, lox, hix);
        slab_interval_f32(float(y), float(sy), float(wcy), float(kWallHY + kRadius) + kWide, loy, hiy);
        const float t_in = lox > loy ? lox : loy, t_out = hix < hiy ? hix : hiy;
        if (!(t_in <= t_out) || t_out < -1.0f || t_in > float(kSubsteps + 1)) continue;   // the segment misses the box
        int k = t_in <= 3.0f ? 1 : int(t_in) - 2;
        int k_end = t_out >= float(kSubsteps - 1) ? kSubsteps : int(t_out) + 3;
        k_end = k_end > kSubsteps ? kSubsteps : k_end;
        // A candidate inside the inflated box that does not touch is in one of its four corner zones (beside the
        // rectangle in x AND in y): from there contact starts either where the path enters the circle around that
        // corner or where it leaves the zone into a face zone -- both are roots of the straight-line motion, so the scan
        // jumps there instead of walking (paths past a wall END cross a corner zone for up to 64 sub-steps: the slowest
        // lane's walk was 11 of the step kernel's 20 us at 4096 envs).  Every visited k gets the exact predicate; a
        // jump stops one sub-step short of the earliest root, and the clamped coordinates it ignores only matter beyond
        // an arena plane, where `first` already ends the scan.
        while (k <= k_end && k < first) {
            double px, py;
            substep_pos(x, y, dx, dy, k, px, py);
            if (touches_wall(px, py, wcx, wcy)) { first = k; break; }
            int skip = 1;
            const double ox = px - wcx, oy = py - wcy;
            const double ax_ = fabs(ox) - kWallHX, ay_ = fabs(oy) - kWallHY;
            if (ax_ > 0.0 && ay_ > 0.0) {                      // corner zone (or still outside the box)
                // position relative to the corner and motion per sub-step, f32 (estimates, see slab_interval_f32)
                const float rx = float(ox > 0.0 ? ax_ : -ax_), ry = float(oy > 0.0 ? ay_ : -ay_);
                const float fx = float(sx), fy = float(sy);
                float tau = 1e30f;                              // sub-steps until contact can begin
                // circle of radius r around the corner: |rel + tau s|^2 = r^2
                const float qa = fx * fx + fy * fy, qb = rx * fx + ry * fy;
                const float qc = rx * rx + ry * ry - float(kRadius * kRadius);
                const float disc = qb * qb - qa * qc;
                if (qb < 0.0f && disc >= -1e-4f * qb * qb) tau = (-qb - sqrtf(disc > 0.0f ? disc : 0.0f)) / qa;
                // leaving the corner zone across the rectangle's edge lines (into a face zone of the box)
                if (rx * fx < 0.0f) { const float tx = -rx / fx; tau = tx < tau ? tx : tau; }
                if (ry * fy < 0.0f) { const float ty = -ry / fy; tau = ty < tau ? ty : tau; }
                skip = tau >= float(kSubsteps) ? kSubsteps : int(tau) - 2;
                skip = skip < 1 ? 1 : skip;
            }
            k += skip;
        }
    }
    const int k_stop = first <= kSubsteps ? first : kSubsteps;
    double qx, qy;
    substep_pos(x, y, dx, dy, k_stop, qx, qy);
    x = qx;
    y = qy;
}

// reset ranges (env/maze.py:188-196): np.random.uniform(lo, hi) = lo + (hi - lo) * u with the difference taken in
// double, as numpy does (0.22 - 0.14 is not the double nearest to 0.08)
RRL_HD void reset_xy(int mode, double u0, double u1, double& x, double& y) {
    if (mode == 1) x = 0.14 + (0.22 - 0.14) * u0;
    else if (mode == 2) x = -0.04 + (0.04 - -0.04) * u0;
    else if (mode == 3) x = -0.27 + (0.27 - -0.27) * u0;
    else x = -0.22 + (-0.13 - -0.22) * u0;
    y = -0.22 + (0.22 - -0.22) * u1;
}

__device__ __forceinline__ void reset_one(uint64_t seed, uint32_t row, uint64_t counter, int mode,
                                          bool check, double& x, double& y) {
    for (uint32_t r = 0;; ++r) {
        const rrl::Bits128 b = rrl::philox_at(seed, row, rrl::kStreamReset, counter | (uint64_t(r) << 48));
        reset_xy(mode, rrl::unit_open(b.lo), rrl::unit_open(b.hi), x, y);
        if (!check || !in_contact(x, y) || r >= 1000) return;
    }
}

RRL_HD void expert_action(double x, double y, double& ax, double& ay) {
    double tx, ty;  // env/maze.py:222-232
    if (x <= -0.151) { tx = -0.15; ty = -0.125; }
    else if (x <= 0.149) { tx = 0.15; ty = 0.125; }
    else { tx = kGoalX; ty = kGoalY; }
    ax = 1.05 * (tx - x);
    ay = 1.05 * (ty - y);
}

}  // namespace rrl_maze
