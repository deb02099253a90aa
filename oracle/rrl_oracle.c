/*
 * rrl_oracle.c -- CPU ORACLE (test infrastructure, see rrl_oracle.h).
 *
 * Plain C, scalar, one thread.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off).
 * All env arithmetic is IEEE double with no FMA contraction so that the HIP kernels
 * (which are written independently against the same specification, DESIGN.md
 * "Deterministic math") can be compared bit-for-bit.
 */
#include "rrl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) -- counter-based RNG.  The reference draws from the
 * global numpy MT19937 stream (env/navigation1.py:93,103); a sequential stream cannot be
 * vectorised, so the build keys a counter RNG by (seed | env index, stream, step counter).
 * ---------------------------------------------------------------------------------------- */
void rrl_oracle_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_bits(uint64_t seed, uint32_t idx, uint32_t stream, uint64_t counter,
                        uint64_t* b0, uint64_t* b1)
{
    uint32_t w[4];
    rrl_oracle_philox4x32(idx, stream, (uint32_t)counter, (uint32_t)(counter >> 32),
                          (uint32_t)seed, (uint32_t)(seed >> 32), w);
    *b0 = ((uint64_t)w[1] << 32) | w[0];
    *b1 = ((uint64_t)w[3] << 32) | w[2];
}

/* u = (k + 0.5) * 2^-52 with k the top 52 bits: exact, strictly inside (0,1). */
double rrl_oracle_uniform01(uint64_t bits)
{
    return ((double)(bits >> 12) + 0.5) * 0x1.0p-52;
}

/* log(u), u in (0,1): exponent split + atanh series, basic IEEE ops only. */
static double det_log(double u)
{
    uint64_t b;
    memcpy(&b, &u, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

/* sin and cos of x in (0, pi/4]: Taylor/Horner in x^2. */
static void det_sincos(double x, double* sn, double* cs)
{
    double z = x * x;
    double ps = -1.0 / 355687428096000.0;            /* -1/17! */
    ps = ps * z + 1.0 / 1307674368000.0;              /*  1/15! */
    ps = ps * z - 1.0 / 6227020800.0;                 /* -1/13! */
    ps = ps * z + 1.0 / 39916800.0;                   /*  1/11! */
    ps = ps * z - 1.0 / 362880.0;                     /* -1/9!  */
    ps = ps * z + 1.0 / 5040.0;                       /*  1/7!  */
    ps = ps * z - 1.0 / 120.0;                        /* -1/5!  */
    ps = ps * z - 1.0 / 6.0;                        /* -1/3!  */
    ps = ps * z + 1.0;
    *sn = x * ps;
    double pc = 1.0 / 20922789888000.0;               /*  1/16! */
    pc = pc * z - 1.0 / 87178291200.0;                /* -1/14! */
    pc = pc * z + 1.0 / 479001600.0;                  /*  1/12! */
    pc = pc * z - 1.0 / 3628800.0;                    /* -1/10! */
    pc = pc * z + 1.0 / 40320.0;                      /*  1/8!  */
    pc = pc * z - 1.0 / 720.0;                        /* -1/6!  */
    pc = pc * z + 1.0 / 24.0;                         /*  1/4!  */
    pc = pc * z - 0.5;                                /* -1/2!  */
    pc = pc * z + 1.0;
    *cs = pc;
}

/* Box-Muller on two 64-bit words; the angle is reduced to an octant with integer ops. */
static void normal2_from_bits(uint64_t b0, uint64_t b1, double z[2])
{
    double u1 = rrl_oracle_uniform01(b0);
    double r = sqrt(-2.0 * det_log(u1));
    uint64_t k = b1 >> 12;                         /* 52 bits: 3 octant + 49 fraction */
    uint32_t oct = (uint32_t)(k >> 49);
    uint64_t frac = k & ((1ULL << 49) - 1);
    if (oct & 1u) frac = ((1ULL << 49) - 1) - frac;  /* 1 - f, exactly */
    double phi = ((double)frac + 0.5) * 0x1.0p-49 * 0.7853981633974483;
    double sn, cs, c, s;
    det_sincos(phi, &sn, &cs);
    if (oct & 1u) { c = sn; s = cs; } else { c = cs; s = sn; }
    switch (oct >> 1) {
        case 0: z[0] = r * c;  z[1] = r * s;  break;
        case 1: z[0] = -(r * s); z[1] = r * c;  break;
        case 2: z[0] = -(r * c); z[1] = -(r * s); break;
        default: z[0] = r * s;  z[1] = -(r * c); break;
    }
}

void rrl_oracle_normal2(uint64_t seed, uint32_t idx, uint32_t stream, uint64_t counter,
                        double z[2])
{
    uint64_t b0, b1;
    philox_bits(seed, idx, stream, counter, &b0, &b1);
    normal2_from_bits(b0, b1, z);
}

/* ------------------------------------------------------------------------------------------
 * Obstacles: env/obstacle.py:13-15 (closed intervals), ComplexObstacle :44-45 (max over boxes)
 * Box tables: env/navigation1.py:41-42, env/navigation2.py:41.
 * ---------------------------------------------------------------------------------------- */
static const double NAV1_BOXES[3][4] = {
    {-100.0, 150.0, 5.0, 10.0}, {-100.0, -80.0, -10.0, 10.0}, {-100.0, 150.0, -10.0, -5.0}};
static const double NAV2_BOXES[1][4] = {{-30.0, -20.0, -7.5, 7.5}};

int rrl_oracle_obstacle(int env_kind, double x, double y)
{
    const double (*boxes)[4] = env_kind == RRL_ENV_NAV1 ? NAV1_BOXES : NAV2_BOXES;
    int nb = env_kind == RRL_ENV_NAV1 ? 3 : 1;
    int hit = 0;
    for (int k = 0; k < nb; ++k)
        hit |= (boxes[k][0] <= x && x <= boxes[k][1] && boxes[k][2] <= y && y <= boxes[k][3]);
    return hit;
}

static double clip1(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one transition; returns next state in nx,ny.  env/navigation1.py:71-89,99-110 */
static void nav_transition(int env_kind, double x, double y, double ax, double ay,
                           double ex, double ey, double* nx, double* ny, double* cost)
{
    ax = clip1(ax, -1.0, 1.0);                           /* process_action :50-51 */
    ay = clip1(ay, -1.0, 1.0);
    if (rrl_oracle_obstacle(env_kind, x, y)) {            /* _next_state :100-102: stuck, no noise */
        *nx = x; *ny = y;
    } else {                                              /* :103-104  A s + B a + 0.05 eps */
        *nx = (x + ax) + 0.05 * ex;
        *ny = (y + ay) + 0.05 * ey;
    }
    /* step_cost :106-110 on the OLD state.  np.linalg.norm -> sqrt(ddot(s,s)); OpenBLAS' ddot
     * accumulates with FMA (dot = fma(y,y, x*x)) -- pinned by the golden vectors, where this
     * form reproduces all 5514 reference rewards bit-for-bit and the plain form misses 159. */
    *cost = -sqrt(fma(y, y, x * x));
}

int rrl_oracle_nav_step(int env_kind, int64_t n, double* pos, const float* action,
                        const double* noise, uint64_t seed, uint64_t counter,
                        float* next_obs, float* obs, float* reward, uint8_t* done,
                        uint8_t* constraint, uint8_t* success, uint8_t* ep_done,
                        int32_t* t, int32_t horizon, int auto_reset,
                        double* next_pos64, double* reward64)
{
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return -1;
    for (int64_t i = 0; i < n; ++i) {
        double x = pos[2 * i], y = pos[2 * i + 1];
        double e[2];
        if (noise) { e[0] = noise[2 * i]; e[1] = noise[2 * i + 1]; }
        else rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_STEP, counter, e);
        double nx, ny, cost;
        nav_transition(env_kind, x, y, (double)action[2 * i], (double)action[2 * i + 1],
                       e[0], e[1], &nx, &ny, &cost);
        int cons = rrl_oracle_obstacle(env_kind, nx, ny);
        int succ = cost > -4.0;                           /* :88 */
        int dn = succ || cons;                            /* :80 */
        int32_t ti = t[i] + 1;                            /* :78 */
        int epd = dn || (ti == horizon);                  /* experiment.py:435 */
        if (next_pos64) { next_pos64[2 * i] = nx; next_pos64[2 * i + 1] = ny; }
        if (reward64) reward64[i] = cost;
        next_obs[2 * i] = (float)nx; next_obs[2 * i + 1] = (float)ny;
        reward[i] = (float)cost;
        done[i] = (uint8_t)dn; constraint[i] = (uint8_t)cons; success[i] = (uint8_t)succ;
        if (ep_done) ep_done[i] = (uint8_t)epd;
        if (auto_reset && epd) {                          /* reset :91-97 */
            double z[2];
            rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_RESET, counter, z);
            nx = -50.0 + z[0]; ny = 0.0 + z[1];
            ti = 0;
        }
        pos[2 * i] = nx; pos[2 * i + 1] = ny;
        t[i] = ti;
        if (obs) { obs[2 * i] = (float)nx; obs[2 * i + 1] = (float)ny; }
    }
    return 0;
}

int rrl_oracle_nav_reset(int env_kind, int64_t n, double* pos, float* obs, int32_t* t,
                         const double* noise, uint64_t seed, uint64_t counter)
{
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return -1;
    for (int64_t i = 0; i < n; ++i) {
        double z[2];
        if (noise) { z[0] = noise[2 * i]; z[1] = noise[2 * i + 1]; }
        else rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_RESET, counter, z);
        pos[2 * i] = -50.0 + z[0];                        /* START_STATE + randn(2), :92 */
        pos[2 * i + 1] = 0.0 + z[1];
        if (t) t[i] = 0;
        if (obs) { obs[2 * i] = (float)pos[2 * i]; obs[2 * i + 1] = (float)pos[2 * i + 1]; }
    }
    return 0;
}

int rrl_oracle_nav_rollout(int env_kind, int64_t n, int32_t T, double* pos,
                           const float* actions, uint64_t seed, uint64_t counter,
                           float* obs_seq, float* reward_seq, uint8_t* constraint_seq,
                           uint8_t* done_seq)
{
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return -1;
    for (int64_t i = 0; i < n; ++i) {
        double x = pos[2 * i], y = pos[2 * i + 1];
        for (int32_t k = 0; k < T; ++k) {
            const float* a = actions + ((int64_t)k * n + i) * 2;
            double e[2], nx, ny, cost;
            rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_STEP, counter + (uint64_t)k, e);
            nav_transition(env_kind, x, y, (double)a[0], (double)a[1], e[0], e[1], &nx, &ny, &cost);
            int cons = rrl_oracle_obstacle(env_kind, nx, ny);
            int64_t o = (int64_t)k * n + i;
            obs_seq[2 * o] = (float)nx; obs_seq[2 * o + 1] = (float)ny;
            reward_seq[o] = (float)cost;
            constraint_seq[o] = (uint8_t)cons;
            done_seq[o] = (uint8_t)((cost > -4.0) || cons);
            x = nx; y = ny;
        }
        pos[2 * i] = x; pos[2 * i + 1] = y;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Draw sources.  The generators below take their random numbers either from Philox, keyed by
 * (rollout, slot) so that every rollout is independent (what the HIP kernels do), or -- for the
 * parity tests -- from two flat arrays holding the uniforms and the standard normals in the
 * order the REFERENCE calls np.random.uniform / np.random.randn.  With the reference's own
 * draws injected the same control flow must reproduce the reference's rows bit-for-bit
 * (tests/golden/nav_offline_golden.npz, maze_ref_golden.npz).
 * ---------------------------------------------------------------------------------------- */
static double take_u(rrl_oracle_draws* d)
{
    if (d->i_u >= d->n_u) { d->exhausted = 1; return 0.5; }
    return d->u[d->i_u++];
}

static double take_z(rrl_oracle_draws* d)
{
    if (d->i_z >= d->n_z) { d->exhausted = 1; return 0.0; }
    return d->z[d->i_z++];
}

/* ------------------------------------------------------------------------------------------
 * Offline constraint data (env/navigation1.py:133-164, env/navigation2.py:133-243).
 * Philox mode: rollout i draws from Philox(seed; i, OFFLINE, k, hi): k=0 start uniforms, k=1
 * second start uniforms, per step j: k=2+3j action normals, 3+3j action uniforms, 4+3j noise
 * normals; hi=1+r is the r-th start re-draw of the nav2 phase-0 rejection loop.
 * np.random.uniform(lo, hi) = lo + (hi - lo) * u; every (hi - lo) below is exact in double.
 * ---------------------------------------------------------------------------------------- */
static void offline_uniform2(uint64_t seed, uint32_t i, uint64_t k, uint32_t hi, double u[2])
{
    uint64_t b0, b1;
    philox_bits(seed, i, RRL_STREAM_OFFLINE, k | ((uint64_t)hi << 32), &b0, &b1);
    u[0] = rrl_oracle_uniform01(b0);
    u[1] = rrl_oracle_uniform01(b1);
}

static int nav2_phase(int64_t i, int64_t n0, int64_t n1)
{
    if (i < n0) return 0;
    return 1 + (int)((i - n0) / (n1 > 0 ? n1 : 1));
}

static int64_t nav_offline_core(int env_kind, int64_t num_transitions, uint64_t seed,
                                rrl_oracle_draws* d, float* s, float* a, float* c, float* s2,
                                float* m, double* s64, double* a64, double* s2_64, int64_t capacity)
{
    int64_t n_roll, n0 = 0, n1 = 0;
    if (env_kind == RRL_ENV_NAV1) n_roll = num_transitions / 10;
    else if (env_kind == RRL_ENV_NAV2) {
        n0 = num_transitions / 10 / 3;                    /* navigation2.py:138 */
        n1 = num_transitions / 10 / 4;                    /* :160,180,200,220 */
        n_roll = n0 + 4 * n1;
    } else return -1;
    int64_t w = 0;
    for (int64_t i = 0; i < n_roll; ++i) {
        double u[2], v[2], x, y;
        int phase = 0;
        if (env_kind == RRL_ENV_NAV1) {                   /* navigation1.py:140-147 */
            double coin, ux, uy;
            if (d) { coin = take_u(d); ux = take_u(d); uy = take_u(d); }
            else {
                offline_uniform2(seed, (uint32_t)i, 0, 0, u);
                offline_uniform2(seed, (uint32_t)i, 1, 0, v);
                coin = u[0]; ux = u[1]; uy = v[0];
            }
            x = -80.0 + 130.0 * ux;
            y = (coin < 0.5) ? (-5.0 + 3.0 * uy) : (2.0 + 3.0 * uy);
        } else {
            phase = nav2_phase(i, n0, n1);
            if (d) { u[0] = take_u(d); u[1] = take_u(d); }
            else offline_uniform2(seed, (uint32_t)i, 0, 0, u);
            switch (phase) {
                case 0: {                                  /* navigation2.py:140-146 */
                    x = -40.0 + 50.0 * u[0]; y = -25.0 + 50.0 * u[1];
                    uint32_t r = 0;
                    while (rrl_oracle_obstacle(env_kind, x, y)) {
                        double q[2];
                        if (d) { q[0] = take_u(d); q[1] = take_u(d); if (d->exhausted) return -3; }
                        else offline_uniform2(seed, (uint32_t)i, 0, 1 + r, q);
                        x = -40.0 + 50.0 * q[0]; y = -25.0 + 50.0 * q[1];
                        ++r;
                    }
                    break;
                }
                case 1: x = -35.0 + 5.0 * u[0]; y = -12.0 + 24.0 * u[1]; break;   /* :162-164 */
                case 2: x = -20.0 + 5.0 * u[0]; y = -12.0 + 24.0 * u[1]; break;   /* :182-184 */
                case 3: x = -30.0 + 10.0 * u[0]; y = 10.0 + 5.0 * u[1]; break;    /* :202-204 */
                default: x = -30.0 + 10.0 * u[0]; y = -15.0 + 5.0 * u[1]; break;  /* :222-224 */
            }
        }
        for (int j = 0; j < 10; ++j) {
            double z[2], q[2], e[2], ax, ay;
            if (d) {                                       /* the reference's call order */
                q[0] = 0.0; e[0] = e[1] = 0.0;
                switch (phase) {
                    case 1: case 2: q[0] = take_u(d); z[0] = 0.0; z[1] = take_z(d); break;   /* U(.,.,1), randn(1) */
                    case 3: case 4: z[0] = take_z(d); z[1] = 0.0; q[0] = take_u(d); break;   /* randn(1), U(.,.,1) */
                    default: z[0] = take_z(d); z[1] = take_z(d); break;                      /* randn(2) */
                }
                if (!rrl_oracle_obstacle(env_kind, x, y)) { e[0] = take_z(d); e[1] = take_z(d); }  /* _next_state :100-104 */
            } else {
                rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_OFFLINE, (uint64_t)(2 + 3 * j), z);
                offline_uniform2(seed, (uint32_t)i, (uint64_t)(3 + 3 * j), 0, q);
                rrl_oracle_normal2(seed, (uint32_t)i, RRL_STREAM_OFFLINE, (uint64_t)(4 + 3 * j), e);
            }
            ax = clip1(z[0], -1.0, 1.0); ay = clip1(z[1], -1.0, 1.0);
            switch (phase) {                               /* navigation2.py:166-168,186-188,206-208,226-228 */
                case 1: ax = 0.5 + 0.5 * q[0]; break;
                case 2: ax = -1.0 + 0.5 * q[0]; break;
                case 3: ay = -1.0 + 0.5 * q[0]; break;
                case 4: ay = 0.5 + 0.5 * q[0]; break;
                default: break;
            }
            double nx, ny, cost;                           /* the transition uses the float64 action, as the reference */
            nav_transition(env_kind, x, y, ax, ay, e[0], e[1], &nx, &ny, &cost);
            int cons = rrl_oracle_obstacle(env_kind, nx, ny);
            if (w >= capacity) return -2;
            s[2 * w] = (float)x; s[2 * w + 1] = (float)y;
            a[2 * w] = (float)ax; a[2 * w + 1] = (float)ay;
            c[w] = (float)cons;
            s2[2 * w] = (float)nx; s2[2 * w + 1] = (float)ny;
            m[w] = (float)(!cons);                         /* (state, action, constraint, next_state, not constraint) */
            if (s64) { s64[2 * w] = x; s64[2 * w + 1] = y; }
            if (a64) { a64[2 * w] = ax; a64[2 * w + 1] = ay; }
            if (s2_64) { s2_64[2 * w] = nx; s2_64[2 * w + 1] = ny; }
            ++w;
            x = nx; y = ny;
            if (cons) break;
        }
    }
    if (d && d->exhausted) return -3;
    return w;
}

int64_t rrl_oracle_nav_offline(int env_kind, int64_t num_transitions, uint64_t seed,
                               float* s, float* a, float* c, float* s2, float* m,
                               int64_t capacity)
{
    return nav_offline_core(env_kind, num_transitions, seed, NULL, s, a, c, s2, m, NULL, NULL, NULL, capacity);
}

int64_t rrl_oracle_nav_offline_explicit(int env_kind, int64_t num_transitions, rrl_oracle_draws* draws,
                                        float* s, float* a, float* c, float* s2, float* m,
                                        double* s64, double* a64, double* s2_64, int64_t capacity)
{
    return nav_offline_core(env_kind, num_transitions, 0, draws, s, a, c, s2, m, s64, a64, s2_64, capacity);
}

/* ------------------------------------------------------------------------------------------
 * Maze (env/maze.py:34-232, env/assets/simple_maze.xml).  The reference steps MuJoCo 1.50
 * (mujoco_py==1.50.1.68, install.sh:13), a third-party binary absent from the reference tree
 * and from this image, so the PHYSICS is a documented kinematic surrogate:
 *   - point cylinder r = 0.025 on two slide joints; motor gear 0.05, joint damping 0.01,
 *     mass 1000 * pi r^2 h = 0.09817 kg, 500 semi-implicit Euler steps of dt = 0.002 from rest
 *     => straight-line displacement MAZE_GAIN * a per env step (a in [-0.1, 0.1]);
 *   - contact (ncon > 3) <=> the disc touches a wall rectangle or an arena plane;
 *   - the disc stops at the first of 64 equal sub-steps that is in contact.
 * Everything AROUND the physics -- step / reset / expert / distance / offline-data control flow,
 * reward, termination, reset ranges, wall moves -- is PINNED to env/maze.py itself: the
 * reference module is imported over a stand-in MjSim implementing exactly this surrogate
 * (tests/golden/gen_maze_ref_golden.py) and its outputs are the golden vectors.
 * ---------------------------------------------------------------------------------------- */
#define MAZE_GAIN 0.24667750873451577   /* metres per unit control per env step */
#define MAZE_R 0.025                    /* toolgeom size, simple_maze.xml:28 */
#define MAZE_LIM 0.3                    /* planes / joint range, simple_maze.xml:16-19,29-30 */
#define MAZE_MAX_FORCE 0.1              /* env/maze.py:17 */
#define MAZE_GOAL_X 0.25                /* env/maze.py:135-137 */
#define MAZE_GOAL_Y 0.0
#define MAZE_GOAL_THRESH 0.03           /* env/maze.py:19 */
#define MAZE_SUBSTEPS 64

/* wall rectangles {cx, cy, half x, half y} after reset() moves them (env/maze.py:199-206):
 * geoms 5..8 = wall1A, wall2A, wall1B, wall2B; y centres 0.5-0.08, 0.4+0.08, -0.25-0.08, -0.25+0.08 */
static const double MAZE_WALLS[4][4] = {
    {-0.1, 0.5 + -0.08, 0.005, 0.2}, {0.1, 0.4 + 0.08, 0.005, 0.2},
    {-0.1, -0.25 + -0.08, 0.005, 0.2}, {0.1, -0.25 + 0.08, 0.005, 0.2}};

int rrl_oracle_maze_contact(double x, double y)
{
    if (MAZE_LIM - x <= MAZE_R || x + MAZE_LIM <= MAZE_R) return 1;
    if (MAZE_LIM - y <= MAZE_R || y + MAZE_LIM <= MAZE_R) return 1;
    for (int k = 0; k < 4; ++k) {
        double dx = fabs(x - MAZE_WALLS[k][0]) - MAZE_WALLS[k][2];
        double dy = fabs(y - MAZE_WALLS[k][1]) - MAZE_WALLS[k][3];
        if (dx < 0.0) dx = 0.0;
        if (dy < 0.0) dy = 0.0;
        if (dx * dx + dy * dy <= MAZE_R * MAZE_R) return 1;
    }
    return 0;
}

static double maze_dist(double x, double y)
{   /* get_distance_score, env/maze.py:215-220: sqrt(mean((goal - qpos)^2)) */
    double ex = MAZE_GOAL_X - x, ey = MAZE_GOAL_Y - y;
    return sqrt((ex * ex + ey * ey) / 2.0);
}

double rrl_oracle_maze_distance(double x, double y) { return maze_dist(x, y); }

static void maze_move(double* x, double* y, double ax, double ay)
{   /* env/maze.py:141-147: no motion when already in contact, else 500 sim steps */
    ax = clip1(ax, -MAZE_MAX_FORCE, MAZE_MAX_FORCE);
    ay = clip1(ay, -MAZE_MAX_FORCE, MAZE_MAX_FORCE);
    if (rrl_oracle_maze_contact(*x, *y)) return;
    double dx = MAZE_GAIN * ax, dy = MAZE_GAIN * ay, qx = *x, qy = *y;
    for (int k = 1; k <= MAZE_SUBSTEPS; ++k) {
        double f = (double)k * (1.0 / MAZE_SUBSTEPS);
        qx = clip1(*x + dx * f, -MAZE_LIM, MAZE_LIM);
        qy = clip1(*y + dy * f, -MAZE_LIM, MAZE_LIM);
        if (rrl_oracle_maze_contact(qx, qy)) break;
    }
    *x = qx; *y = qy;
}

/* reset ranges (env/maze.py:188-196), np.random.uniform(lo, hi) = lo + (hi - lo) * u with the
 * difference taken in double, as numpy does (0.22 - 0.14 is not the double nearest to 0.08) */
static void maze_reset_xy(int mode, double u0, double u1, double* x, double* y)
{
    switch (mode) {
        case 1: *x = 0.14 + (0.22 - 0.14) * u0; break;       /* 'e' :191 */
        case 2: *x = -0.04 + (0.04 - -0.04) * u0; break;     /* 'm' :193 */
        case 3: *x = -0.27 + (0.27 - -0.27) * u0; break;     /* None :189 */
        default: *x = -0.22 + (-0.13 - -0.22) * u0; break;   /* 'h' :195 */
    }
    *y = -0.22 + (0.22 - -0.22) * u1;                        /* :196 */
}

static void maze_reset_one(uint64_t seed, uint32_t i, uint64_t counter, int mode, int check,
                           rrl_oracle_draws* d, double* x, double* y)
{   /* env/maze.py:184-213: redraw while in contact (recursion at :209-211, always with the contact check) */
    for (uint32_t r = 0;; ++r) {
        double u0, u1;
        if (d) { u0 = take_u(d); u1 = take_u(d); }
        else {
            uint64_t b0, b1;
            philox_bits(seed, i, RRL_STREAM_RESET, counter | ((uint64_t)r << 48), &b0, &b1);
            u0 = rrl_oracle_uniform01(b0); u1 = rrl_oracle_uniform01(b1);
        }
        maze_reset_xy(mode, u0, u1, x, y);
        if (!check || !rrl_oracle_maze_contact(*x, *y) || r >= 1000 || (d && d->exhausted)) return;
    }
}

int rrl_oracle_maze_step(int64_t n, double* pos, const float* action, uint64_t seed, uint64_t counter,
                         float* next_obs, float* obs, float* reward, uint8_t* done,
                         uint8_t* constraint, uint8_t* success, uint8_t* ep_done, int32_t* t,
                         int32_t horizon, int auto_reset, double* next_pos64, double* reward64)
{
    for (int64_t i = 0; i < n; ++i) {
        double x = pos[2 * i], y = pos[2 * i + 1];
        maze_move(&x, &y, (double)action[2 * i], (double)action[2 * i + 1]);
        int32_t ti = t[i] + 1;                                /* self.steps += 1, :151 */
        int cons = rrl_oracle_maze_contact(x, y);             /* :152 */
        double d = maze_dist(x, y);
        int dn = (ti >= horizon) || cons || (d < MAZE_GOAL_THRESH);   /* :153-154 */
        double rew = -d;                                      /* dense reward :158 */
        int succ = rew > -0.03;                               /* :166 */
        int epd = dn || (ti == horizon);
        if (next_pos64) { next_pos64[2 * i] = x; next_pos64[2 * i + 1] = y; }
        if (reward64) reward64[i] = rew;
        next_obs[2 * i] = (float)x; next_obs[2 * i + 1] = (float)y;
        reward[i] = (float)rew;
        done[i] = (uint8_t)dn; constraint[i] = (uint8_t)cons; success[i] = (uint8_t)succ;
        if (ep_done) ep_done[i] = (uint8_t)epd;
        if (auto_reset && epd) {
            maze_reset_one(seed, (uint32_t)i, counter, 0, 1, NULL, &x, &y);
            ti = 0;
        }
        pos[2 * i] = x; pos[2 * i + 1] = y;
        t[i] = ti;
        if (obs) { obs[2 * i] = (float)x; obs[2 * i + 1] = (float)y; }
    }
    return 0;
}

/* one env step with a float64 action (the reference passes float64 arrays to step(), env/maze.py:139-168) */
int rrl_oracle_maze_step64(double* x, double* y, double ax, double ay, int32_t* steps, int32_t horizon,
                           double* reward, int* done, int* constraint, int* success)
{
    maze_move(x, y, ax, ay);
    *steps += 1;
    *constraint = rrl_oracle_maze_contact(*x, *y);
    double d = maze_dist(*x, *y);
    *done = (*steps >= horizon) || *constraint || (d < MAZE_GOAL_THRESH);
    *reward = -d;
    *success = *reward > -0.03;
    return 0;
}

int rrl_oracle_maze_reset(int64_t n, double* pos, float* obs, int32_t* t, int mode,
                          int check_constraint, uint64_t seed, uint64_t counter)
{
    for (int64_t i = 0; i < n; ++i) {
        double x, y;
        maze_reset_one(seed, (uint32_t)i, counter, mode, check_constraint, NULL, &x, &y);
        pos[2 * i] = x; pos[2 * i + 1] = y;
        if (t) t[i] = 0;
        if (obs) { obs[2 * i] = (float)x; obs[2 * i + 1] = (float)y; }
    }
    return 0;
}

/* one reset from explicit uniforms (two per attempt, env/maze.py:188-196 order: x then y) */
int rrl_oracle_maze_reset_explicit(int mode, int check_constraint, rrl_oracle_draws* draws, double* x, double* y)
{
    maze_reset_one(0, 0, 0, mode, check_constraint, draws, x, y);
    return draws->exhausted ? -3 : 0;
}

void rrl_oracle_maze_expert_action(double x, double y, double act[2])
{   /* env/maze.py:222-232, gain 1.05 (:134) */
    double tx, ty;
    if (x <= -0.151) { tx = -0.15; ty = -0.125; }
    else if (x <= 0.149) { tx = 0.15; ty = 0.125; }
    else { tx = MAZE_GOAL_X; ty = MAZE_GOAL_Y; }
    act[0] = 1.05 * (tx - x);
    act[1] = 1.05 * (ty - y);
}

/* env/maze.py:34-107: half random, half expert actions; reset (no contact check) every 20
 * steps with mode e/m/h drawn 30/30/40 %; rows (state, raw action, constraint, next, not done).
 * The random actions are env.action_space.sample() (float32, gym's own generator): explicit mode
 * takes them from `rand_actions` [half, 2]; the expert action stays float64 into step(). */
static int64_t maze_offline_core(int64_t num_transitions, uint64_t seed, rrl_oracle_draws* d,
                                 const float* rand_actions, float* s, float* a, float* c, float* s2,
                                 float* m, double* s64, double* a64, double* s2_64, int64_t capacity)
{
    int64_t half = num_transitions / 2;
    if (2 * half > capacity) return -2;
    int64_t w = 0;
    for (int part = 0; part < 2; ++part) {
        int64_t n_seg = (half + 19) / 20;
        for (int64_t g = 0; g < n_seg; ++g) {
            uint32_t row = (uint32_t)(part * n_seg + g);
            uint64_t b0, b1;
            double sample;
            if (d) sample = take_u(d);
            else {
                philox_bits(seed, row, RRL_STREAM_OFFLINE, 0, &b0, &b1);
                sample = rrl_oracle_uniform01(b0);
            }
            int mode = sample < 0.3 ? 1 : (sample < 0.6 ? 2 : 0);      /* :45-51 */
            double x, y;
            maze_reset_one(seed, row, 1ULL << 40, mode, 0, d, &x, &y);
            int32_t steps = 0;
            int64_t len = (g == n_seg - 1) ? half - 20 * g : 20;
            for (int64_t j = 0; j < len; ++j) {
                double act[2];
                if (part == 0) {                                       /* action_space.sample() :58 */
                    if (rand_actions) {
                        act[0] = (double)rand_actions[2 * (20 * g + j)];
                        act[1] = (double)rand_actions[2 * (20 * g + j) + 1];
                    } else {
                        philox_bits(seed, row, RRL_STREAM_OFFLINE, (uint64_t)(1 + j), &b0, &b1);
                        act[0] = (double)(float)(-0.1 + 0.2 * rrl_oracle_uniform01(b0));
                        act[1] = (double)(float)(-0.1 + 0.2 * rrl_oracle_uniform01(b1));
                    }
                } else rrl_oracle_maze_expert_action(x, y, act);       /* :88 */
                double nx = x, ny = y, rew;
                int dn, cons, succ;
                rrl_oracle_maze_step64(&nx, &ny, act[0], act[1], &steps, 100, &rew, &dn, &cons, &succ);
                s[2 * w] = (float)x; s[2 * w + 1] = (float)y;
                a[2 * w] = (float)act[0]; a[2 * w + 1] = (float)act[1];
                c[w] = (float)cons;
                s2[2 * w] = (float)nx; s2[2 * w + 1] = (float)ny;
                m[w] = (float)(!dn);
                if (s64) { s64[2 * w] = x; s64[2 * w + 1] = y; }
                if (a64) { a64[2 * w] = act[0]; a64[2 * w + 1] = act[1]; }
                if (s2_64) { s2_64[2 * w] = nx; s2_64[2 * w + 1] = ny; }
                ++w;
                x = nx; y = ny;
            }
        }
    }
    if (d && d->exhausted) return -3;
    return w;
}

int64_t rrl_oracle_maze_offline(int64_t num_transitions, uint64_t seed, float* s, float* a, float* c,
                                float* s2, float* m, int64_t capacity)
{
    return maze_offline_core(num_transitions, seed, NULL, NULL, s, a, c, s2, m, NULL, NULL, NULL, capacity);
}

int64_t rrl_oracle_maze_offline_explicit(int64_t num_transitions, rrl_oracle_draws* draws, const float* rand_actions,
                                         float* s, float* a, float* c, float* s2, float* m, double* s64,
                                         double* a64, double* s2_64, int64_t capacity)
{
    return maze_offline_core(num_transitions, 0, draws, rand_actions, s, a, c, s2, m, s64, a64, s2_64, capacity);
}

/* ------------------------------------------------------------------------------------------
 * Replay (recovery_rl/replay_memory.py)
 * ---------------------------------------------------------------------------------------- */
int rrl_oracle_replay_push(rrl_oracle_replay* rb, int64_t n, const float* s, const float* a,
                           const float* r, const float* s2, const float* m,
                           const uint8_t* valid)
{
    for (int64_t i = 0; i < n; ++i) {                     /* push :21-25 / :47-52, row by row */
        if (valid && !valid[i]) continue;
        int64_t p = rb->pos;
        rb->s[2 * p] = s[2 * i]; rb->s[2 * p + 1] = s[2 * i + 1];
        rb->a[2 * p] = a[2 * i]; rb->a[2 * p + 1] = a[2 * i + 1];
        rb->r[p] = r[i];
        rb->s2[2 * p] = s2[2 * i]; rb->s2[2 * p + 1] = s2[2 * i + 1];
        rb->m[p] = m[i];
        rb->pos = p + 1 < rb->cap ? p + 1 : rb->pinned;   /* pinned = 0: (p + 1) % capacity, replay_memory.py:25 */
        if (rb->size < rb->cap) rb->size += 1;
    }
    return 0;
}

static uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* B distinct draws from [0,size): every slot draws independently; a slot loses a round when an
 * already-accepted slot, or a lower-numbered slot of the same round, holds the same value;
 * losers redraw with the round number bumped.  By symmetry under value relabelling the result
 * is uniform over ordered B-subsets == random.sample (replay_memory.py:28). */
int rrl_oracle_sample_indices(int64_t size, int32_t B, uint64_t seed, uint64_t counter,
                              uint32_t stream, int64_t* idx)
{
    if (B > size || B <= 0) return -1;                    /* random.sample raises ValueError */
    uint8_t* acc = (uint8_t*)calloc((size_t)B, 1);
    uint8_t* lose = (uint8_t*)calloc((size_t)B, 1);
    int pending = B;
    for (uint32_t round = 0; pending > 0; ++round) {
        if (round > 4096) { free(acc); free(lose); return -2; }
        for (int32_t i = 0; i < B; ++i) {
            if (acc[i]) continue;
            uint64_t b0, b1;
            philox_bits(seed, (uint32_t)i, stream, (counter << 12) | round, &b0, &b1);
            idx[i] = (int64_t)mulhi64(b0, (uint64_t)size);
        }
        for (int32_t i = 0; i < B; ++i) {
            lose[i] = 0;
            if (acc[i]) continue;
            for (int32_t j = 0; j < B; ++j)
                if (j != i && idx[j] == idx[i] && (acc[j] || j < i)) { lose[i] = 1; break; }
        }
        for (int32_t i = 0; i < B; ++i)
            if (!acc[i] && !lose[i]) { acc[i] = 1; --pending; }
    }
    free(acc); free(lose);
    return 0;
}

/* rank -> slot of the k-th positive (r != 0) / negative (r == 0) among filled slots, slot order
 * (np.argwhere is sorted, replay_memory.py:58-66) */
static int64_t kth_slot(const rrl_oracle_replay* rb, int64_t k, int want_pos)
{
    for (int64_t p = 0; p < rb->size; ++p) {
        int is_pos = rb->r[p] != 0.0f;
        if (is_pos == want_pos) { if (k == 0) return p; --k; }
    }
    return -1;
}

int rrl_oracle_sample_stratified(const rrl_oracle_replay* rb, int32_t n_pos, int32_t n_neg,
                                 uint64_t seed, uint64_t counter, int64_t* idx)
{
    return rrl_oracle_sample_stratified_clamped(rb, n_pos, n_neg, 0, seed, counter, idx, NULL);
}

/* clamp != 0: a class with too few rows gives all it has and the other class fills the batch (the lock-step loop's
 * rule, include/rrl_hip.h RRL_REPLAY_CLAMP_STRATIFIED); clamp == 0: the reference's ValueError (replay_memory.py:61-66) */
int rrl_oracle_sample_stratified_clamped(const rrl_oracle_replay* rb, int32_t n_pos, int32_t n_neg, int clamp,
                                         uint64_t seed, uint64_t counter, int64_t* idx, int32_t* n_pos_used)
{
    int64_t npos_total = 0;
    for (int64_t p = 0; p < rb->size; ++p) npos_total += rb->r[p] != 0.0f;
    int64_t nneg_total = rb->size - npos_total;
    if (n_pos > npos_total || n_neg > nneg_total) {
        int32_t B = n_pos + n_neg;
        if (!clamp || B > rb->size) return -1;
        if (n_pos > npos_total) n_pos = (int32_t)npos_total;
        else n_pos = B - (int32_t)nneg_total;
        n_neg = B - n_pos;
    }
    if (n_pos_used) *n_pos_used = n_pos;
    /* a class taken whole is listed in slot order (rank i for its i-th batch row): no draw */
    if (n_pos > 0) {
        if (n_pos == npos_total) { for (int32_t i = 0; i < n_pos; ++i) idx[i] = i; }
        else if (rrl_oracle_sample_indices(npos_total, n_pos, seed, counter, RRL_STREAM_SAMPLE, idx)) return -2;
        for (int32_t i = 0; i < n_pos; ++i) idx[i] = kth_slot(rb, idx[i], 1);
    }
    if (n_neg > 0) {
        if (n_neg == nneg_total) { for (int32_t i = 0; i < n_neg; ++i) idx[n_pos + i] = i; }
        else if (rrl_oracle_sample_indices(nneg_total, n_neg, seed, counter, RRL_STREAM_SAMPLE_NEG, idx + n_pos)) return -2;
        for (int32_t i = 0; i < n_neg; ++i) idx[n_pos + i] = kth_slot(rb, idx[n_pos + i], 0);
    }
    return 0;
}

/* Demonstration-share draw (the build's vectorisation rule for the safety critic's batch; include/rrl_hip.h
 * rrl_replay_sample_gather_split): n_demo distinct rows of the pinned range [0, pinned), then n_online distinct rows of
 * [pinned, size).  What it restores: in a one-env run of the reference the demonstrations pushed at
 * experiment.py:278-286 stay about half of recovery_memory (uniform draw, replay_memory.py:54-72; batch clamp
 * qrisk.py:100-105).  A range with too few rows gives all of them, the other fills the batch. */
int rrl_oracle_sample_split(const rrl_oracle_replay* rb, int32_t n_demo, int32_t n_online, uint64_t seed,
                            uint64_t counter, int64_t* idx, int32_t* n_demo_used)
{
    int32_t B = n_demo + n_online;
    if (B <= 0 || B > rb->size) return -1;
    int64_t demo_total = rb->pinned < rb->size ? rb->pinned : rb->size, online_total = rb->size - demo_total;
    if (n_online > online_total) { n_online = (int32_t)online_total; n_demo = B - n_online; }
    else if (n_demo > demo_total) { n_demo = (int32_t)demo_total; n_online = B - n_demo; }
    if (n_demo_used) *n_demo_used = n_demo;
    if (n_demo > 0) {
        if (n_demo == demo_total) { for (int32_t i = 0; i < n_demo; ++i) idx[i] = i; }
        else if (rrl_oracle_sample_indices(demo_total, n_demo, seed, counter, RRL_STREAM_SAMPLE, idx)) return -2;
    }
    if (n_online > 0) {
        if (n_online == online_total) { for (int32_t i = 0; i < n_online; ++i) idx[n_demo + i] = i; }
        else if (rrl_oracle_sample_indices(online_total, n_online, seed, counter, RRL_STREAM_SAMPLE_NEG, idx + n_demo)) return -2;
        for (int32_t i = 0; i < n_online; ++i) idx[n_demo + i] += demo_total;
    }
    return 0;
}

int rrl_oracle_gather(const rrl_oracle_replay* rb, int32_t B, const int64_t* idx, float* s,
                      float* a, float* r, float* s2, float* m)
{
    for (int32_t i = 0; i < B; ++i) {                     /* np.stack over the batch :29 */
        int64_t p = idx[i];
        if (p < 0 || p >= rb->size) return -1;
        s[2 * i] = rb->s[2 * p]; s[2 * i + 1] = rb->s[2 * p + 1];
        a[2 * i] = rb->a[2 * p]; a[2 * i + 1] = rb->a[2 * p + 1];
        r[i] = rb->r[p];
        s2[2 * i] = rb->s2[2 * p]; s2[2 * i + 1] = rb->s2[2 * p + 1];
        m[i] = rb->m[p];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CEM bookkeeping (recovery_rl/optimizers.py:73-124), M independent problems.
 * Truncated normal: Philox normal redrawn until |z| <= 2 (scipy.stats.truncnorm(-2,2).rvs in the
 * reference, :86-89,100 -- same distribution, different generator).
 * ---------------------------------------------------------------------------------------- */
static double truncnorm2(uint64_t seed, uint32_t row, uint64_t counter, uint32_t d)
{
    for (uint32_t attempt = 0;; ++attempt) {
        double z[2];
        rrl_oracle_normal2(seed, row, RRL_STREAM_CEM, (counter << 20) | ((uint64_t)d << 8) | attempt, z);
        if ((z[0] >= -2.0 && z[0] <= 2.0) || attempt >= 255) return clip1(z[0], -2.0, 2.0);
    }
}

int rrl_oracle_cem_sample(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                          const double* lb, const double* ub, double epsilon, int sticky,
                          uint8_t* active, uint64_t seed, uint64_t counter, float* samples)
{
    for (int64_t m = 0; m < M; ++m) {
        double vmax = var[m * dim];
        for (int d = 1; d < dim; ++d) if (var[m * dim + d] > vmax) vmax = var[m * dim + d];
        int act = vmax > epsilon;                                     /* optimizers.py:94 */
        if (sticky && !active[m]) act = 0;
        active[m] = (uint8_t)act;
        if (!act) continue;
        for (int32_t i = 0; i < pop; ++i)
            for (int d = 0; d < dim; ++d) {
                double mu = mean[m * dim + d];
                double lo = (mu - lb[d]) / 2.0, hi = (ub[d] - mu) / 2.0;
                double cv = lo * lo < hi * hi ? lo * lo : hi * hi;    /* :95-99 */
                if (var[m * dim + d] < cv) cv = var[m * dim + d];
                double z = truncnorm2(seed, (uint32_t)(m * pop + i), counter, (uint32_t)d);
                samples[(m * pop + i) * dim + d] = (float)(z * sqrt(cv) + mu);   /* :100-102 */
            }
    }
    return 0;
}

typedef struct { float cost; int32_t idx; } cem_key;
static int cem_cmp(const void* a, const void* b)
{
    const cem_key* x = (const cem_key*)a; const cem_key* y = (const cem_key*)b;
    if (x->cost < y->cost) return -1;
    if (x->cost > y->cost) return 1;
    return x->idx - y->idx;
}

int rrl_oracle_cem_update(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                          const float* samples, const float* costs, double* mean, double* var,
                          const uint8_t* active)
{
    if (num_elites > pop) return -1;                                  /* optimizers.py:66-68 */
    cem_key* keys = (cem_key*)malloc(sizeof(cem_key) * (size_t)pop);
    for (int64_t m = 0; m < M; ++m) {
        if (active && !active[m]) continue;
        for (int32_t i = 0; i < pop; ++i) {
            float c = costs[m * pop + i];
            keys[i].cost = (c != c) ? 1e6f : c;                       /* NaN -> 1e6, MPC.py:415 */
            keys[i].idx = i;
        }
        qsort(keys, (size_t)pop, sizeof(cem_key), cem_cmp);           /* argsort(costs)[:num_elites], :111 */
        for (int d = 0; d < dim; ++d) {
            double sum = 0.0;
            for (int32_t e = 0; e < num_elites; ++e)
                sum += (double)samples[(m * pop + keys[e].idx) * dim + d];
            double em = sum / (double)num_elites;                     /* np.mean(elites, 0) :113 */
            double sq = 0.0;
            for (int32_t e = 0; e < num_elites; ++e) {
                double dv = (double)samples[(m * pop + keys[e].idx) * dim + d] - em;
                sq += dv * dv;
            }
            double ev = sq / (double)num_elites;                      /* np.var(elites, 0) :114 */
            mean[m * dim + d] = alpha * mean[m * dim + d] + (1.0 - alpha) * em;   /* :116 */
            var[m * dim + d] = alpha * var[m * dim + d] + (1.0 - alpha) * ev;     /* :117 */
        }
    }
    free(keys);
    return 0;
}
