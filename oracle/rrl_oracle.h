/*
 * rrl_oracle.h -- CPU ORACLE for the Recovery-RL hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithms (abalakrishna123/recovery-rl)
 * used as the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (recovery_rl_amd)
 * never links, imports or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   navigation1 / navigation2 step, reset, offline data : PINNED against golden vectors
 *       captured by importing the reference (tests/golden/nav_step_golden.npz,
 *       nav_offline_golden.npz; generator tests/golden/gen_env_golden.py).
 *   replay push / sample / stratified sample           : PINNED on composition + ring
 *       semantics against tests/golden/replay_golden.npz (index streams cannot match:
 *       the reference uses python `random`).
 *   maze                                                : control flow PINNED to env/maze.py:34-232 (the
 *       reference module imported over a stand-in MjSim, tests/golden/gen_maze_ref_golden.py ->
 *       maze_ref_golden.npz); the PHYSICS is a documented surrogate (MuJoCo 1.50 is a third-party
 *       dependency absent from the reference tree and this image): trajectories are not MuJoCo's.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef RRL_ORACLE_H
#define RRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RRL_ENV_NAV1 = 0, RRL_ENV_NAV2 = 1, RRL_ENV_MAZE = 2 };

/* Philox stream ids (word 1 of the counter) -- same numbering as include/rrl_hip.h */
enum {
    RRL_STREAM_STEP = 0,
    RRL_STREAM_RESET = 1,
    RRL_STREAM_OFFLINE = 2,
    RRL_STREAM_SAMPLE = 3,
    RRL_STREAM_SAMPLE_NEG = 4,
    RRL_STREAM_CEM = 5,
    RRL_STREAM_ACTION = 6
};

/* ---- RNG primitives (exposed for tests) ---- */
void rrl_oracle_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4]);
void rrl_oracle_normal2(uint64_t seed, uint32_t idx, uint32_t stream, uint64_t counter,
                        double z[2]);
double rrl_oracle_uniform01(uint64_t bits);

/* explicit draw source for the parity tests: uniforms in [0,1) and standard normals in the order the reference
 * calls np.random.uniform / np.random.randn (np.random.uniform(lo, hi) = lo + (hi - lo) * u) */
typedef struct {
    const double* u; int64_t n_u; int64_t i_u;
    const double* z; int64_t n_z; int64_t i_z;
    int exhausted;
} rrl_oracle_draws;

/* ---- navigation1 / navigation2 (env/navigation1.py, env/navigation2.py) ---- */
int rrl_oracle_obstacle(int env_kind, double x, double y);

int rrl_oracle_nav_step(int env_kind, int64_t n, double* pos, const float* action,
                        const double* noise, uint64_t seed, uint64_t counter,
                        float* next_obs, float* obs, float* reward, uint8_t* done,
                        uint8_t* constraint, uint8_t* success, uint8_t* ep_done,
                        int32_t* t, int32_t horizon, int auto_reset,
                        double* next_pos64, double* reward64);

int rrl_oracle_nav_reset(int env_kind, int64_t n, double* pos, float* obs, int32_t* t,
                         const double* noise, uint64_t seed, uint64_t counter);

/* T-step open-loop rollout: actions [T,n,2]; outputs per-step arrays [T,n,...] */
int rrl_oracle_nav_rollout(int env_kind, int64_t n, int32_t T, double* pos,
                           const float* actions, uint64_t seed, uint64_t counter,
                           float* obs_seq, float* reward_seq, uint8_t* constraint_seq,
                           uint8_t* done_seq);

/* offline constraint data, one Philox-driven rollout per index (env/navigation1.py:133-164,
 * env/navigation2.py:133-243).  Outputs are stream-compacted in rollout order.
 * Returns the number of transitions written (<= 10 * n_rollouts), negative on error. */
int64_t rrl_oracle_nav_offline(int env_kind, int64_t num_transitions, uint64_t seed,
                               float* s, float* a, float* c, float* s2, float* m,
                               int64_t capacity);

/* the same generator fed the reference's own draws; also returns the float64 rows the reference holds */
int64_t rrl_oracle_nav_offline_explicit(int env_kind, int64_t num_transitions, rrl_oracle_draws* draws,
                                        float* s, float* a, float* c, float* s2, float* m,
                                        double* s64, double* a64, double* s2_64, int64_t capacity);

/* ---- maze (env/maze.py).  Control flow pinned to env/maze.py:34-232 (reference imported over a stand-in MjSim);
 * the physics is the documented kinematic surrogate of the MuJoCo model ---- */
int rrl_oracle_maze_contact(double x, double y);
double rrl_oracle_maze_distance(double x, double y);
int rrl_oracle_maze_step64(double* x, double* y, double ax, double ay, int32_t* steps, int32_t horizon,
                           double* reward, int* done, int* constraint, int* success);
int rrl_oracle_maze_reset_explicit(int mode, int check_constraint, rrl_oracle_draws* draws, double* x, double* y);
int64_t rrl_oracle_maze_offline_explicit(int64_t num_transitions, rrl_oracle_draws* draws, const float* rand_actions,
                                         float* s, float* a, float* c, float* s2, float* m, double* s64,
                                         double* a64, double* s2_64, int64_t capacity);
int rrl_oracle_maze_step(int64_t n, double* pos, const float* action, uint64_t seed, uint64_t counter,
                         float* next_obs, float* obs, float* reward, uint8_t* done,
                         uint8_t* constraint, uint8_t* success, uint8_t* ep_done, int32_t* t,
                         int32_t horizon, int auto_reset, double* next_pos64, double* reward64);
/* mode: 0 'h' (default), 1 'e', 2 'm', 3 None (env/maze.py:184-197) */
int rrl_oracle_maze_reset(int64_t n, double* pos, float* obs, int32_t* t, int mode,
                          int check_constraint, uint64_t seed, uint64_t counter);
void rrl_oracle_maze_expert_action(double x, double y, double act[2]);
/* env/maze.py:34-107; writes exactly 2*(num_transitions/2) rows, returns that count */
int64_t rrl_oracle_maze_offline(int64_t num_transitions, uint64_t seed, float* s, float* a, float* c,
                                float* s2, float* m, int64_t capacity);

/* ---- replay (recovery_rl/replay_memory.py) ---- */
typedef struct {
    float* s;       /* [cap,2] */
    float* a;       /* [cap,2] */
    float* r;       /* [cap]   */
    float* s2;      /* [cap,2] */
    float* m;       /* [cap]   */
    int64_t cap;
    int64_t pos;    /* next write slot */
    int64_t size;   /* filled rows */
    int64_t pinned; /* rows [0, pinned) are never overwritten (0: the reference's ring); the build's vectorisation rule */
} rrl_oracle_replay;

int rrl_oracle_replay_push(rrl_oracle_replay* rb, int64_t n, const float* s, const float* a,
                           const float* r, const float* s2, const float* m,
                           const uint8_t* valid);
/* B distinct uniform indices in [0,size) (random.sample semantics, replay_memory.py:27-30) */
int rrl_oracle_sample_indices(int64_t size, int32_t B, uint64_t seed, uint64_t counter,
                              uint32_t stream, int64_t* idx);
/* stratified: first n_pos indices are slots with r!=0 ("positives"), then n_neg negatives
 * (replay_memory.py:54-72) */
int rrl_oracle_sample_stratified(const rrl_oracle_replay* rb, int32_t n_pos, int32_t n_neg,
                                 uint64_t seed, uint64_t counter, int64_t* idx);
int rrl_oracle_sample_stratified_clamped(const rrl_oracle_replay* rb, int32_t n_pos, int32_t n_neg, int clamp,
                                         uint64_t seed, uint64_t counter, int64_t* idx, int32_t* n_pos_used);
/* demonstration-share draw: n_demo rows of [0, pinned), then n_online rows of [pinned, size) (vectorisation rule) */
int rrl_oracle_sample_split(const rrl_oracle_replay* rb, int32_t n_demo, int32_t n_online, uint64_t seed,
                            uint64_t counter, int64_t* idx, int32_t* n_demo_used);
int rrl_oracle_gather(const rrl_oracle_replay* rb, int32_t B, const int64_t* idx, float* s,
                      float* a, float* r, float* s2, float* m);

/* ---- CEM bookkeeping (recovery_rl/optimizers.py:73-124) ---- */
int rrl_oracle_cem_sample(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                          const double* lb, const double* ub, double epsilon, int sticky,
                          uint8_t* active, uint64_t seed, uint64_t counter, float* samples);
int rrl_oracle_cem_update(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                          const float* samples, const float* costs, double* mean, double* var,
                          const uint8_t* active);

#ifdef __cplusplus
}
#endif
#endif
