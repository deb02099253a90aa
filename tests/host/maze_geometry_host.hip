// Host-side check of the kernels' Maze geometry (recovery_rl_amd/csrc/maze_device.hpp, compiled for the HOST by hipcc)
// against the oracle's sequential 64-sub-step scan (oracle/rrl_oracle.c maze_move via rrl_oracle_maze_step64):
// the kernel replaces the scan by a bracketed search and must land on the same sub-step, bit for bit.
// Usage: maze_geometry_host <n_random> <seed>; prints "mismatches <k> of <n>".
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "maze_device.hpp"

extern "C" int rrl_oracle_maze_step64(double* x, double* y, double ax, double ay, int32_t* steps, int32_t horizon,
                                      double* reward, int* done, int* constraint, int* success);
extern "C" int rrl_oracle_maze_contact(double x, double y);

static long g_hits = 0, g_stuck = 0;

static long check(double x, double y, double ax, double ay, bool verbose) {
    double ox = x, oy = y, rew;
    int32_t steps = 0;
    int dn, cons, succ;
    rrl_oracle_maze_step64(&ox, &oy, ax, ay, &steps, 100, &rew, &dn, &cons, &succ);
    if (rrl_oracle_maze_contact(x, y)) ++g_stuck;
    else if (cons) ++g_hits;
    double kx = x, ky = y;
    rrl_maze::move(kx, ky, ax, ay);
    const bool same = std::memcmp(&kx, &ox, 8) == 0 && std::memcmp(&ky, &oy, 8) == 0 &&
                      rrl_maze::in_contact(kx, ky) == (cons != 0) &&
                      rrl_maze::in_contact(x, y) == (rrl_oracle_maze_contact(x, y) != 0);
    if (!same && verbose)
        std::printf("MISMATCH pos (%.17g, %.17g) act (%.17g, %.17g): kernel (%.17g, %.17g) oracle (%.17g, %.17g)\n", x, y,
                    ax, ay, kx, ky, ox, oy);
    return same ? 0 : 1;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 1000000;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 1);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long bad = 0, total = 0;
    const double wx[4] = {-0.1, 0.1, -0.1, 0.1}, wy[4] = {0.42, 0.48, -0.33, -0.17};
    for (long i = 0; i < n; ++i) {
        double x, y, ax, ay;
        const int kind = int(i % 8);
        if (kind < 3) {                                      // anywhere in the arena
            x = -0.3 + 0.6 * U(rng); y = -0.3 + 0.6 * U(rng);
        } else if (kind < 6) {                               // a ring around a wall (faces, ends and corners)
            const int j = int(rng() % 4);
            const double d = 0.02501 + 0.02 * U(rng);
            const double sx = U(rng) < 0.5 ? -1.0 : 1.0, sy = U(rng) < 0.5 ? -1.0 : 1.0;
            if (U(rng) < 0.5) { x = wx[j] + sx * (0.005 + d); y = wy[j] + (2 * U(rng) - 1) * 0.26; }
            else { x = wx[j] + (2 * U(rng) - 1) * 0.05; y = wy[j] + sy * (0.2 + d); }
            if (y > 0.27 || y < -0.27) y = (2 * U(rng) - 1) * 0.27;   // the walls reach beyond the arena
        } else if (kind == 6) {                              // next to an arena plane
            const double d = 0.27 + 0.01 * U(rng);
            if (U(rng) < 0.5) { x = (U(rng) < 0.5 ? -d : d); y = -0.3 + 0.6 * U(rng); }
            else { y = (U(rng) < 0.5 ? -d : d); x = -0.3 + 0.6 * U(rng); }
        } else {                                             // wall corners exactly at grazing distance
            const int j = int(rng() % 4);
            const double ang = 6.283185307179586 * U(rng), rad = 0.025 + (U(rng) - 0.02) * 1e-3;
            x = wx[j] + (U(rng) < 0.5 ? -0.005 : 0.005) + rad * std::cos(ang);
            y = wy[j] + (U(rng) < 0.5 ? -0.2 : 0.2) + rad * std::sin(ang);
        }
        const int ak = int((i / 8) % 6);
        if (ak == 0) { ax = -0.15 + 0.3 * U(rng); ay = -0.15 + 0.3 * U(rng); }
        else if (ak == 1) { ax = (U(rng) - 0.5) * 0.2; ay = 0.0; }
        else if (ak == 2) { ax = 0.0; ay = (U(rng) - 0.5) * 0.2; }
        else if (ak == 3) { ax = (U(rng) - 0.5) * 2e-3; ay = (U(rng) - 0.5) * 2e-3; }       // tiny moves
        else if (ak == 4) { ax = (U(rng) - 0.5) * 1e-12; ay = (U(rng) - 0.5) * 0.2; }
        else { ax = double(float(-0.1 + 0.2 * U(rng))); ay = double(float(-0.1 + 0.2 * U(rng))); }
        bad += check(x, y, ax, ay, bad < 10);
        ++total;
    }
    std::printf("mismatches %ld of %ld (moves that ran into something: %ld, started in contact: %ld)\n", bad, total, g_hits,
                g_stuck);
    return bad ? 1 : 0;
}
