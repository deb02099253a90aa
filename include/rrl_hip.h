/*
 * rrl_hip.h -- C ABI of librrl_hip.so, the MI355X (gfx950) hot path of Recovery RL.
 *
 * Drop-in boundary.  The reference has no FFI layer; its operator
 * boundary for this path is the gym env protocol (env/navigation1.py:55-97), the replay
 * protocol (recovery_rl/replay_memory.py:11-75) and the CEM optimiser
 * (recovery_rl/optimizers.py:73-124).  Every entry point below names the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr());
 *     the library allocates nothing and keeps no global state; all calls are re-entrant;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it
 *     (NULL = the default stream) and is safe to capture in a hipGraph;
 *   - return value: 0 on success, a negative RRL_E* code on error; nothing throws;
 *   - random draws come from Philox4x32-10 keyed by `seed`, counter words
 *     (row index, stream id, counter lo, counter hi).  `counter_dev` (nullable) points to
 *     device uint64[2] = {tick, ticket(internal, keep 0)}: tick is ADDED to `counter`, and
 *     where a function takes `counter_inc` the last workgroup to finish does
 *     tick += counter_inc, so a captured hipGraph advances its own RNG counter on replay;
 *   - env kinds: 0 navigation1, 1 navigation2, 2 maze.
 */
#ifndef RRL_HIP_H
#define RRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRL_OK 0
#define RRL_EINVAL (-1)   /* bad argument (unknown env kind, negative size, null pointer) */
#define RRL_ELAUNCH (-2)  /* hipLaunchKernel failed; see rrl_last_hip_error() */
#define RRL_ECAPTURE (-5) /* a packed launch (rrl_*_packed) met an argument block it has not seen before while the stream was
                           * capturing: building its device copy (hipMalloc + copy) inside the capture would invalidate the
                           * graph.  Launch the same arguments once before the capture (the warm-up iterations do). */
#define RRL_EPLANS (-6)   /* more than 8192 distinct argument blocks of packed launches alive: rrl_pack_clear() */
#define RRL_ERANGE (-3)   /* size outside what the kernel supports (e.g. batch > 1024) */

enum { RRL_ENV_NAV1 = 0, RRL_ENV_NAV2 = 1, RRL_ENV_MAZE = 2 };

enum {
    RRL_STREAM_STEP = 0,       /* env transition noise           */
    RRL_STREAM_RESET = 1,      /* env reset noise                */
    RRL_STREAM_OFFLINE = 2,    /* offline constraint data        */
    RRL_STREAM_SAMPLE = 3,     /* replay sampling (positives)    */
    RRL_STREAM_SAMPLE_NEG = 4, /* replay sampling (negatives)    */
    RRL_STREAM_CEM = 5,        /* CEM truncated-normal samples   */
    RRL_STREAM_ACTION = 6,     /* uniform random actions         */
    RRL_STREAM_PLAN = 7,       /* planner particle noise         */
    RRL_STREAM_NOISE = 8       /* policy noise (rrl_normal_fill) */
};

/* ABI version, bumped on any signature change. */
int rrl_abi_version(void);
/* hipGetLastError() of the calling thread's last failed launch, as an int (0 = none). */
int rrl_last_hip_error(void);
/* *ctr += inc on the stream (one thread).  Lets a captured graph advance its RNG counter. */
int rrl_counter_add(uint64_t* ctr, uint64_t inc, void* stream);

/* --------------------------------------------------------------------------------------------
 * Environments.  Replaces Navigation1.step / Navigation2.step (env/navigation1.py:71-89,
 * env/navigation2.py:70-88) + the horizon rule of the driver (recovery_rl/experiment.py:434-435)
 * for n independent envs in lock-step.
 *   pos        [n,2] f64  in/out  env state (the reference keeps float64 state)
 *   action     [n,2] f32  in      raw action; clipped to [-1,1] inside (process_action :50-51)
 *   noise      [n,2] f64  in      N(0,1) draws to use instead of Philox, or NULL
 *   next_obs   [n,2] f32  out     s' BEFORE any auto-reset (what replay / info["next_state"] get)
 *   obs        [n,2] f32  out     observation for the next policy call (post-reset), nullable
 *   reward     [n]   f32  out     -||s|| of the OLD state (step_cost :106-110)
 *   done       [n]   u8   out     reward > -4 or obstacle(s')            (:80)
 *   constraint [n]   u8   out     obstacle(s')                           (:82)
 *   success    [n]   u8   out     reward > -4                            (:88)
 *   ep_done    [n]   u8   out     done or t == horizon (experiment.py:435), nullable
 *   t          [n]   i32  in/out  per-env step count (self.time :78)
 *   auto_reset != 0: where ep_done, pos <- [-50,0] + N(0,I) (reset :91-97) and t <- 0.
 * ------------------------------------------------------------------------------------------ */
int rrl_nav_step(int env_kind, int64_t n, double* pos, const float* action, const double* noise,
                 uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                 float* next_obs, float* obs, float* reward, uint8_t* done, uint8_t* constraint,
                 uint8_t* success, uint8_t* ep_done, int32_t* t, int32_t horizon, int auto_reset,
                 void* stream);

/* The same step in the compact layout of the bandwidth regime (56 B moved per env-step instead of 72; same
 * arithmetic, same bits).  The step count and the four flags of rrl_nav_step share ONE u16 status word per env:
 *   status     [n]   u16  in/out  bits 0-11 steps taken in the running episode (in: before, out: after; 0 after an
 *                                 auto-reset), bit 12 done, 13 constraint, 14 success, 15 ep_done (out; ignored on input)
 *   reset_obs  [n,2] f32  out     written ONLY at rows with ep_done when auto_reset != 0: the observation after the
 *                                 reset (= float(pos) of that row).  Everywhere else the next policy input is next_obs
 *                                 itself.  Nullable, and NULL is the fast form in the bandwidth regime: ~200 k
 *                                 scattered 8-byte stores cost as much as a dense 8 B/env array (+45 us at 2^24 envs).
 * horizon <= 4095 (RRL_ERANGE otherwise).  pos / action / noise / next_obs / reward 16-byte aligned, status and
 * reset_obs 8-byte aligned (RRL_EINVAL otherwise). */
#define RRL_STATUS_STEPS 0x0fffu
#define RRL_STATUS_DONE 0x1000u
#define RRL_STATUS_CONSTRAINT 0x2000u
#define RRL_STATUS_SUCCESS 0x4000u
#define RRL_STATUS_EP_DONE 0x8000u
int rrl_nav_step_compact(int env_kind, int64_t n, double* pos, const float* action, const double* noise,
                         uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                         float* next_obs, float* reset_obs, float* reward, uint16_t* status, int32_t horizon,
                         int auto_reset, void* stream);

/* Replaces Navigation*.reset (env/navigation1.py:91-97) for n envs. `mask` (nullable, u8[n])
 * restricts the reset to rows with mask != 0.  obs / t nullable. */
int rrl_nav_reset(int env_kind, int64_t n, double* pos, float* obs, int32_t* t,
                  const uint8_t* mask, const double* noise, uint64_t seed, uint64_t counter,
                  const uint64_t* counter_dev, void* stream);

/* T open-loop steps with the state held in registers (no auto-reset, no horizon):
 * actions [T,n,2]; outputs [T,n,...] (each nullable).  The CEM ground-truth-dynamics mode
 * and the roofline sweep use it. Step k uses counter + k. */
int rrl_nav_rollout(int env_kind, int64_t n, int32_t T, double* pos, const float* actions,
                    uint64_t seed, uint64_t counter, const uint64_t* counter_dev,
                    float* obs_seq, float* reward_seq, uint8_t* constraint_seq, uint8_t* done_seq,
                    void* stream);

/* Replaces get_offline_data (env/navigation1.py:133-164, env/navigation2.py:133-243): one
 * scripted <=10-step rollout per thread, stream-compacted in rollout order into replay-row
 * arrays (s,a,constraint,s',mask) of `capacity` rows.  *count_dev (device int64) receives the
 * number of rows written.  scratch: device int32[n_rollouts + 1]. Use rrl_nav_offline_rollouts()
 * for n_rollouts. */
int64_t rrl_nav_offline_rollouts(int env_kind, int64_t num_transitions);
int rrl_nav_offline(int env_kind, int64_t num_transitions, uint64_t seed, float* s, float* a,
                    float* c, float* s2, float* m, int64_t capacity, int64_t* count_dev,
                    int32_t* scratch, void* stream);

/* --------------------------------------------------------------------------------------------
 * Maze.  Replaces MazeNavigation.step / reset / get_offline_data (env/maze.py:139-213, 34-107).
 * The reference steps MuJoCo 1.50 (third-party, absent); these kernels run the kinematic
 * surrogate of DESIGN.md section 6 -- same control flow, reward, termination and geometry.
 * Buffers as rrl_nav_step; `done` already contains the env's own horizon (env/maze.py:153).
 * Reset modes: 0 'h' (default), 1 'e', 2 'm', 3 None (env/maze.py:187-196).
 * ------------------------------------------------------------------------------------------ */
int rrl_maze_step(int64_t n, double* pos, const float* action, uint64_t seed, uint64_t counter,
                  uint64_t* counter_dev, uint64_t counter_inc, float* next_obs, float* obs,
                  float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                  uint8_t* ep_done, int32_t* t, int32_t horizon, int auto_reset, void* stream);
int rrl_maze_reset(int64_t n, double* pos, float* obs, int32_t* t, const uint8_t* mask, int mode,
                   int check_constraint, uint64_t seed, uint64_t counter,
                   const uint64_t* counter_dev, void* stream);
/* writes exactly 2 * (num_transitions / 2) rows (half random, half expert actions) */
int rrl_maze_offline(int64_t num_transitions, uint64_t seed, float* s, float* a, float* c,
                     float* s2, float* m, int64_t capacity, void* stream);

/* --------------------------------------------------------------------------------------------
 * Replay.  Replaces ReplayMemory / ConstraintReplayMemory (recovery_rl/replay_memory.py).
 * Layout: structure-of-arrays ring, f32: s[cap,2] a[cap,2] r[cap] s2[cap,2] m[cap] = 32 B/row.
 *   state   device int64[4]: {position, size, ticket(internal, keep 0), error flag}
 *   pos_cnt device int32[RRL_POS_CNT_LEN(cap)] (zero-initialised) or NULL: number of rows with r != 0 per 64-slot chunk,
 *           then (from the next multiple of 4) per 1024-slot super-chunk, then (from the next multiple of 2) one
 *           64-bit mask per chunk (bit b = slot 64 c + b holds such a row; 8-byte aligned: keep pos_cnt 8-byte
 *           aligned); maintained by push, consumed by the stratified sampler (replay_memory.py:50,58-66), which scans
 *           the second level only (cap / 1024 entries), one super-chunk's 16 first-level counts and one mask per
 *           drawn row.
 * ------------------------------------------------------------------------------------------ */
#define RRL_POS_CNT_LEN(cap) \
    ((((((((cap) + 63) / 64 + 3) / 4) * 4 + ((cap) + 1023) / 1024) + 1) / 2) * 2 + 2 * (((cap) + 63) / 64))
typedef struct {
    float* s;
    float* a;
    float* r;
    float* s2;
    float* m;
    int64_t cap;
    int64_t* state;
    int32_t* pos_cnt;
    int32_t flags;       /* RRL_REPLAY_* bits */
    int64_t pinned;      /* rows [0, pinned) are never overwritten: after slot cap - 1 the ring continues at slot `pinned`
                          * (0 = the reference's plain ring, replay_memory.py:21-25).  The lock-step loop pins the offline
                          * constraint demonstrations: N envs fill a 1e6-row ring in 1e6 / N iterations, whereas the one-env
                          * reference never wraps within a run (4e4 env-steps), i.e. never loses them. */
} rrl_replay_t;

/* Stratified draws (rrl_creplay_sample_gather) that ask for more positives (or negatives) than the ring holds: the
 * reference aborts (random.sample raises ValueError, replay_memory.py:61-66) and so does the default here (error flag
 * state[3] = 1, outputs untouched).  With this bit the draw takes every row of the short class and fills the batch
 * from the other one: n_pos' = min(n_pos, positives), n_neg' = B - n_pos' (and the other way round).  The lock-step
 * loop sets it: thousands of envs overwrite a 1e6-row ring in a few hundred iterations, so a policy that has learned
 * to avoid violations starves the positive class -- a state the one-env reference cannot reach within its runs. */
#define RRL_REPLAY_CLAMP_STRATIFIED 1

/* push (replay_memory.py:21-25,47-52) of n rows in row order; `valid` (nullable u8[n]) drops
 * rows with valid == 0 (used for add_both_transitions, experiment.py:446-448).
 * scratch: device int32[ceil(n/1024) + 1], only read when valid != NULL. */
int rrl_replay_push(const rrl_replay_t* rb, int64_t n, const float* s, const float* a,
                    const float* r, const float* s2, const float* m, const uint8_t* valid,
                    int32_t* scratch, void* stream);

/* sample (replay_memory.py:27-30): B distinct uniform rows gathered into 5 batch tensors.
 * idx_out (nullable, int64[B]) receives the chosen slots.  xu / x2u / xpu (nullable, f32 [B,4]) receive the
 * rows pre-assembled for the networks: xu = (s, a), x2u = (s', -, -), xpu = (s, -, -) (columns 2..3 of
 * the latter two are written later by the policy-head kernels).  B <= 1024, cap < 2^31.  If B > size the error
 * flag state[3] is set to 1 and the outputs are left untouched (the reference raises
 * ValueError; callers guard, experiment.py:397,403). */
int rrl_replay_sample_gather(const rrl_replay_t* rb, int32_t B, uint64_t seed, uint64_t counter,
                             uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a, float* r, float* s2,
                             float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu, void* stream);

/* stratified sample (replay_memory.py:54-72): first n_pos rows uniform among slots with r != 0,
 * then n_neg rows uniform among filled slots with r == 0.  Needs rb->pos_cnt. cap <= 2^21.
 * Too few rows of a class: see RRL_REPLAY_CLAMP_STRATIFIED. */
int rrl_creplay_sample_gather(const rrl_replay_t* rb, int32_t n_pos, int32_t n_neg, uint64_t seed,
                              uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a,
                              float* r, float* s2, float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu,
                              void* stream);

/* Demonstration-share sample -- a vectorisation rule, not a reference function: first n_demo distinct uniform rows of the
 * pinned range [0, rb->pinned) (the offline constraint demonstrations, experiment.py:278-286), then n_online distinct
 * uniform rows of the online range [rb->pinned, size).  In a one-env reference run the 20 000 demonstrations stay about
 * half of recovery_memory from the first to the last episode (uniform draw, replay_memory.py:54-72, qrisk.py:100-105);
 * N lock-step envs push N rows per iteration, so a uniform draw over the ring would show the safety critic the
 * demonstrations -- the only violations a safe policy ever produces -- in 2 % of its batch rows.  This draw keeps their
 * share fixed.  A range with fewer rows than asked gives every row it has and the other range fills the batch
 * (n_online' = min(n_online, size - pinned), n_demo' = B - n_online', and the other way round); B > size sets the
 * error flag state[3] = 1.  Same outputs and the same Philox streams as rrl_creplay_sample_gather (demo group = its
 * positive group, online group = its negative group).  B <= 1024, cap < 2^31, 0 <= pinned < cap. */
int rrl_replay_sample_gather_split(const rrl_replay_t* rb, int32_t n_demo, int32_t n_online, uint64_t seed,
                                   uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a,
                                   float* r, float* s2, float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu,
                                   void* stream);

/* Description of one policy-head evaluation (used by rrl_policy_heads_fwd_multi, by the input head of rrl_stack_t and
 * by the recovery action of rrl_*_step_push_select). */
/* rrl_gauss_head_fwd / rrl_stoch_head_fwd calls that do not depend on each other in ONE launch (n <= 4): a' = pi(s')
 * and pi(s) of one SAC step (sac.py:192-218), the task action and the recovery action of the acting pass
 * (experiment.py:546-577).  kind RRL_HEAD_GAUSS: fields of rrl_gauss_head_fwd (mean_out = mean_action);
 * RRL_HEAD_STOCH: fields of rrl_stoch_head_fwd (head = raw).  Results equal the stand-alone launches'. */
enum { RRL_HEAD_GAUSS = 0, RRL_HEAD_STOCH = 1 };
typedef struct {
    int kind, B;
    const float* head;
    int n_part;
    long long part_stride;
    const float *eps, *scale, *bias;
    float* action;
    int ld_action;
    float *logp, *mean_out;
    const float* obs_in;
    float* obs_out;
    const float* log_std;
    float min_log_std;
} rrl_policy_head_t;

/* The two draws of one lock-step iteration (task buffer -> SAC update, safety buffer -> Q_risk update,
 * experiment.py:397-416) and the iteration's policy noise (rrl_normal_fill) in ONE launch: they do not depend on
 * each other.  A member is a rrl_replay_sample_gather call (stratified = RRL_DRAW_UNIFORM, B = n_pos + n_neg), a
 * rrl_creplay_sample_gather call (RRL_DRAW_STRATIFIED) or a rrl_replay_sample_gather_split call (RRL_DRAW_DEMO_SHARE:
 * n_pos = n_demo, n_neg = n_online); `second` and the noise part (noise_pairs = 0) are optional.
 * Rows, indices and normals equal the stand-alone launches'. */
enum { RRL_DRAW_UNIFORM = 0, RRL_DRAW_STRATIFIED = 1, RRL_DRAW_DEMO_SHARE = 2 };
typedef struct {
    const rrl_replay_t* rb;
    int stratified;
    int32_t n_pos, n_neg;
    uint64_t seed, counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    float *s, *a, *r, *s2, *m;
    int64_t* idx_out;
    float *xu, *x2u, *xpu;
} rrl_draw_t;
int rrl_sample_multi(const rrl_draw_t* first, const rrl_draw_t* second, long long noise_pairs, uint64_t noise_seed,
                     uint64_t noise_counter, uint64_t* noise_counter_dev, uint64_t noise_counter_inc, float* noise_out,
                     void* stream);

/* Fused lock-step iteration tail: env step + reward penalty + bootstrap mask + memory.push +
 * recovery_memory.push + episode counters in ONE launch (the body of recovery_rl/experiment.py:420-461
 * for n navigation envs).  `obs` holds the current observation on entry (it is the stored `state`) and
 * the next observation on return.  Rows stored: memory <- (obs, task_action or real_action if
 * push_real_action, reward - penalty*constraint, next_obs, 1-done); recovery_memory (nullable) <- (obs,
 * real_action, constraint, next_obs, 1-done).  stats = uint64[8] {env_steps, episodes, num_viols,
 * viol_and_recovery, viol_and_no_recovery, num_successes, recovery_steps, constraint_steps};
 * reward_sums = double[2] {sum of rewards, sum of finished-episode returns}; ep_reward = float[n].
 * next_obs, reward, done, constraint, success, ep_done are per-env outputs of the step for callers that read them
 * (episode log, online model re-fit): each may be NULL (17 of the 171 B the kernel moves per env-step are theirs). */
int rrl_nav_step_push(int env_kind, int64_t n, double* pos, int32_t* t, float* obs,
                      const float* task_action, const float* real_action, const uint8_t* recovery,
                      uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                      int32_t horizon, int auto_reset, float reward_penalty, int push_real_action,
                      const rrl_replay_t* memory, const rrl_replay_t* recovery_memory, float* next_obs,
                      float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success, uint8_t* ep_done,
                      uint64_t* stats, double* reward_sums, float* ep_reward, void* stream);
/* the same fused tail for the Maze env (env/maze.py:139-213 + experiment.py:420-461) */
int rrl_maze_step_push(int64_t n, double* pos, int32_t* t, float* obs,
                      const float* task_action, const float* real_action, const uint8_t* recovery,
                      uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                      int32_t horizon, int auto_reset, float reward_penalty, int push_real_action,
                      const rrl_replay_t* memory, const rrl_replay_t* recovery_memory, float* next_obs,
                      float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success, uint8_t* ep_done,
                      uint64_t* stats, double* reward_sums, float* ep_reward, void* stream);

/* The same fused tails with the recovery gate of Experiment.get_action (experiment.py:546-577) evaluated inside:
 * recovery[i] = max(sigmoid(z[i]), sigmoid(z[n + i])) > eps_safe (z = pre-sigmoid twin Q_risk(s, a_task), [2,n], given
 * as z_n_part partial sums z_part_stride floats apart like every stack output: rrl_mlp3_forward with scratch);
 * executed action = recovery ? rec_action[i] : task_action[i]; with rec_action = NULL the recovery action is evaluated in
 * the kernel from rec_head (an RRL_HEAD_STOCH description, rrl_stoch_head_fwd's formula on the recovery policy's stack
 * output).  real_action [n,2] and recovery [n] are OUTPUTS here
 * (what rrl_recovery_select would have written); task_action rows are ld_task floats apart (the [s | a] input
 * of the safety critic, ld_task = 4, can be passed as it is).  One launch less per lock-step iteration. */
int rrl_nav_step_push_select(int env_kind, int64_t n, double* pos, int32_t* t, float* obs, const float* task_action,
                             int ld_task, const float* z, int z_n_part, long long z_part_stride,
                             float eps_safe, const float* rec_action, const rrl_policy_head_t* rec_head,
                             float* real_action,
                             uint8_t* recovery, uint64_t seed, uint64_t counter, uint64_t* counter_dev,
                             uint64_t counter_inc, int32_t horizon, int auto_reset, float reward_penalty,
                             int push_real_action, const rrl_replay_t* memory, const rrl_replay_t* recovery_memory,
                             float* next_obs, float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                             uint8_t* ep_done, uint64_t* stats, double* reward_sums, float* ep_reward, void* stream);
int rrl_maze_step_push_select(int64_t n, double* pos, int32_t* t, float* obs, const float* task_action, int ld_task,
                              const float* z, int z_n_part, long long z_part_stride, float eps_safe, const float* rec_action, const rrl_policy_head_t* rec_head,
                             float* real_action,
                              uint8_t* recovery, uint64_t seed, uint64_t counter, uint64_t* counter_dev,
                              uint64_t counter_inc, int32_t horizon, int auto_reset, float reward_penalty,
                              int push_real_action, const rrl_replay_t* memory, const rrl_replay_t* recovery_memory,
                              float* next_obs, float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                              uint8_t* ep_done, uint64_t* stats, double* reward_sums, float* ep_reward, void* stream);

/* The same kernel through ONE argument struct, which also carries the COMPACT per-env state: `status` (nullable) is
 * one u16 word per env -- step count in bits 0-11 (horizon <= 4095), done / constraint / success / ep_done of the last
 * step in bits 12-15 -- and replaces `t` (then nullable) and the four u8 flag arrays; with it the stored `state` of the
 * replay rows is float(pos), so `obs` is written only (8 B less read, 8 + 4 B less written per env-step than t + flags).
 * next_obs / reward / done / constraint / success / ep_done stay optional outputs (NULL: not written).  The recovery gate
 * (rrl_*_step_push_select) is selected by sel_z != NULL; otherwise real_action (+ recovery, nullable) are read.
 * Replay rows, counters and env state equal the entries above bit for bit.
 * log_state != NULL: the per-episode log (rrl_episode_log_append, the fields of rrl_episode_log_t + its four per-env
 * accumulators) is advanced by this launch as well, from the values the step holds in registers -- the same records, the
 * same accumulator values as the stand-alone launch fed with this step's per-env outputs, without writing those outputs. */
typedef struct {
    int64_t n;
    double* pos;
    int32_t* t;
    uint16_t* status;
    float* obs;
    const float* task_action;
    int32_t ld_task;
    const float* real_action;
    const uint8_t* recovery;
    const float* sel_z;
    int32_t sel_n_part;
    long long sel_part_stride;
    float sel_eps_safe;
    const float* sel_rec_action;
    const rrl_policy_head_t* sel_rec_head;
    float* real_action_out;
    uint8_t* recovery_out;
    uint64_t seed, counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    int32_t horizon, auto_reset;
    float reward_penalty;
    int32_t push_real_action;
    const rrl_replay_t *memory, *recovery_memory;
    float *next_obs, *reward;
    uint8_t *done, *constraint, *success, *ep_done;
    uint64_t* stats;
    double* reward_sums;
    float* ep_reward;
    int32_t* log_rec_i32;    /* rrl_episode_log_t.rec_i32 / rec_f64 / cap / state */
    double* log_rec_f64;
    int64_t log_cap;
    int64_t* log_state;
    int32_t* log_len;        /* per-env accumulators [n]: ep_len, ep_ret, ep_viol, ep_rec of rrl_episode_log_append */
    double* log_ret;
    int32_t *log_viol, *log_rec;
} rrl_step_push_t;
int rrl_nav_step_push_x(int env_kind, const rrl_step_push_t* a, void* stream);
int rrl_maze_step_push_x(const rrl_step_push_t* a, void* stream);

/* --------------------------------------------------------------------------------------------
 * CEM.  Replaces the bookkeeping of CEMOptimizer.obtain_solution (recovery_rl/optimizers.py:73-124)
 * for M independent planning problems (one per env that needs a recovery action); the cost
 * function in between stays with the caller (MPC._compile_cost, recovery_rl/MPC.py:374-416).
 *   mean, var [M,dim] f64 in/out; lb, ub [dim] f64; samples [M,pop,dim] f32; costs [M,pop] f32;
 *   active [M] u8.
 * rrl_cem_sample : active[m] = max(var[m]) > epsilon (the while-condition, :94; an env that went
 *                  inactive stays inactive if `sticky` != 0); for active envs
 *                  constrained_var = min(((mean-lb)/2)^2, ((ub-mean)/2)^2, var)   (:95-99)
 *                  samples = truncnorm(-2,2) * sqrt(constrained_var) + mean -> f32 (:100-102)
 * rrl_cem_update : for active envs: elites = the num_elites lowest-cost samples (NaN cost -> 1e6,
 *                  MPC.py:415; ties broken by sample index), mean <- alpha*mean + (1-alpha)*mean(elites),
 *                  var <- alpha*var + (1-alpha)*var(elites)                        (:111-117)
 * pop <= 1024, dim <= 64.
 * ------------------------------------------------------------------------------------------ */
int rrl_cem_sample(int64_t M, int32_t pop, int32_t dim, const double* mean, const double* var,
                   const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                   uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                   float* samples, void* stream);
int rrl_cem_update(int64_t M, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                   const float* samples, const float* costs, double* mean, double* var,
                   const uint8_t* active, void* stream);

/* The same two steps for a planning set whose size is decided ON THE DEVICE (no host round trip in MPC.act,
 * recovery_rl/MPC.py:322-347; the reference plans for its one env only when Q_risk > eps_safe, experiment.py:568-571):
 *   rrl_cem_begin   ONE launch: idx[0..count) = rows with mask != 0 in ascending order, count[0] = their number, and the
 *                   planner's inputs of the compacted problems: mean[j] = prev_sol[idx[j]] (prev_sol [n,dim] f64),
 *                   var[j] = init_var [dim], cur_obs[j] = obs[idx[j]] (f32 [n,2]), active[j] = 1      (MPC.py:336-341)
 *   rrl_cem_sample_n / rrl_cem_update_n
 *                   rrl_cem_sample / rrl_cem_update with M = m_dev[0] read by the kernel; m_max >= m_dev[0] bounds the
 *                   launch (buffers are sized for m_max); same Philox rows, same bits as the host-count entries
 *   rrl_cem_finish  action[i, 0..du) = float(mean[j, 0..du)) for i = idx[j], 0 for rows that did not plan;
 *                   prev_sol[i] = mean[j] shifted left by du, zero-filled                              (MPC.py:342-344) */
int rrl_cem_begin(int64_t n, const uint8_t* mask, int32_t dim, const double* prev_sol, const double* init_var,
                  const float* obs, int32_t* idx, int32_t* count, double* mean, double* var, float* cur_obs,
                  uint8_t* active, void* stream);
int rrl_cem_sample_n(const int32_t* m_dev, int64_t m_max, int32_t pop, int32_t dim, const double* mean,
                     const double* var, const double* lb, const double* ub, double epsilon, int sticky, uint8_t* active,
                     uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* samples,
                     void* stream);
int rrl_cem_update_n(const int32_t* m_dev, int64_t m_max, int32_t pop, int32_t dim, int32_t num_elites, double alpha,
                     const float* samples, const float* costs, double* mean, double* var, const uint8_t* active,
                     void* stream);
int rrl_cem_finish(int64_t n, const uint8_t* mask, int32_t dim, int32_t du, const int32_t* idx, const int32_t* count,
                   const double* mean, double* prev_sol, float* action, void* stream);

/* --------------------------------------------------------------------------------------------
 * MLP building block.  Replaces the nn.Linear forward/backward of the SAC / Q_risk networks
 * (recovery_rl/model.py:49-76,172-199,295-343,489-530) for G heads in one launch; exact f32
 * (v_mfma_f32_32x32x2_f32), one wavefront per 32x32 output tile.  Row-major, leading dimensions
 * in elements, per-head strides s*.
 *   mode 0 (NT): C[g] = A[g] . B[g]^T (+ bias[g][n]) (relu)          A [M,K], B [N,K]
 *   mode 1 (NN): C[g] = A[g] . B[g]   (zeroed where mask[g] <= 0)    A [M,K], B [K,N], mask like C
 *   mode 2 (TN): C[g] = A[g]^T . B[g] ; colsum[g][m] = sum_k A[k][m] A [K,M], B [K,N]
 *   accumulate != 0: C += result.
 * ------------------------------------------------------------------------------------------ */
int rrl_gemm_f32(int mode, int G, int M, int N, int K, const float* A, int lda, long long sA,
                 const float* B, int ldb, long long sB, float* C, int ldc, long long sC,
                 const float* bias, long long sBias, int relu, const float* mask, int ldmask,
                 long long sMask, float* colsum, long long sColsum, int accumulate, void* stream);

/* Whole 2-hidden-layer stack forward in one launch (QNetwork / QNetworkConstraint heads,
 * GaussianPolicy / StochasticPolicy trunks + last linear; model.py:66-76,188-199,317-323,511-515):
 *   out[g] = W3[g] relu(W2[g] relu(W1[g] x + b1[g]) + b2[g]) + b3[g],  x [M,din] (leading dim ldx)
 * shared by the G heads.  W1 [G,H,din], W2 [G,H,H], W3 [G,dout,H]; h1/h2 [G,M,H] receive the hidden
 * activations when non-null (needed by the backward pass).  H % 16 == 0, H <= 256, din, dout <= 4.
 * scratch (nullable, f32 [4*G*M*dout]): when given and M <= 1024, H % 64 == 0 the hidden-2 columns are
 * split over 4 workgroups per row tile (small batches are bound by streaming W2 through one CU) and their
 * partial last-layer sums [4,G,M,dout] are either added in a fixed order by a second tiny kernel
 * (finalize != 0 -> out) or left in scratch for a consumer that sums them itself (finalize == 0). */
int rrl_mlp3_forward(int G, int M, int H, int din, int dout, const float* x, int ldx, const float* W1,
                     const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                     float* h1, float* h2, float* out, float* scratch, int finalize, void* stream);
/* number of partial sums (4) if rrl_mlp3_forward(M, H, scratch != NULL) takes the split path (so finalize == 0
 * leaves that many partials in scratch), 0 otherwise */
int rrl_mlp3_is_split(int M, int H);

/* Thin ends of the stack backward (one side 1..4 wide, so no MFMA tile):
 *   rrl_mlp_head_backward : dW3 = dOut^T h2, db3 = sum_b dOut, dh2 = [h2 > 0] (dOut W3)   (dW3/db3 nullable)
 *   rrl_mlp_input_backward: dW1 = dh1^T x, db1 = sum_b dh1 (nullable pair), dx = dh1 W1    (dx nullable)
 * dOut [G,B,dout], h2/dh2/dh1 [G,B,H], W3 [G,dout,H], W1 [G,H,din], x [B,din] shared by the heads. */
int rrl_mlp_head_backward(int G, int B, int H, int dout, const float* dOut, const float* h2, const float* W3,
                          float* dW3, float* db3, float* dh2, void* stream);
int rrl_mlp_input_backward(int G, int B, int H, int din, const float* dh1, const float* x, int ldx,
                           const float* W1, float* dW1, float* db1, float* dx, void* stream);

/* Hidden layer of the stack backward in ONE launch (both products read dh2 and are independent):
 *   dW2[g] = dh2[g]^T h1[g], db2[g] = column sums of dh2[g]   and   dh1[g] = (dh2[g] W2[g]) * [h1[g] > 0]
 * dh2, h1, dh1 [G,B,H]; W2, dW2 [G,H,H]; db2 [G,H]. */
int rrl_mlp_hidden_backward(int G, int B, int H, const float* dh2, const float* h1, const float* W2, float* dW2,
                            float* db2, float* dh1, void* stream);

/* rrl_mlp_head_backward with dOut produced in the kernel from a loss description instead of read from memory:
 * saves the stand-alone rrl_*_grad / rrl_*_head_bwd launch in front of every stack backward (same formulas,
 * bit-identical dOut).  kind selects the formula and the meaning of the fields:
 *   RRL_LOSS_SAC_CRITIC   (G=2,dout=1) out=q, out_t=qt, v0=logp2, v1=r, v2=m, v3=penalty (nullable), alpha,
 *                         f0=gamma; loss[2] = the two MSEs                              (sac.py:192-214)
 *   RRL_LOSS_SAC_POLICY   (G=2,dout=1) out=qp, v0=logp, alpha; loss[1]                  (sac.py:216-231)
 *   RRL_LOSS_QRISK_CRITIC (G=2,dout=1) out=z, out_t=zt, v0=c, v1=m, f0=gamma_safe; loss[2]   (qrisk.py:118-148)
 *   RRL_LOSS_QRISK_POLICY (G=2,dout=1) out=zp; loss[1]                                  (qrisk.py:150-154)
 *   RRL_LOSS_GAUSS_HEAD   (G=1,dout=4) out=head, v0=eps, v1=scale, f0=dlogp, d_action/ld/n_heads/head_stride
 *                         (the backward of GaussianPolicy.sample, model.py:324-340)
 *   RRL_LOSS_STOCH_HEAD   (G=1,dout=2) out=raw, v0=eps, v1=log_std, v2=scale, f0=min_log_std, d_action...;
 *                         loss[2] = dlog_std                                            (model.py:511-525)
 * out / out_t take (n_part, part_stride) like the stand-alone kernels. */
enum { RRL_LOSS_SAC_CRITIC = 0, RRL_LOSS_SAC_POLICY = 1, RRL_LOSS_QRISK_CRITIC = 2, RRL_LOSS_QRISK_POLICY = 3,
       RRL_LOSS_GAUSS_HEAD = 4, RRL_LOSS_STOCH_HEAD = 5 };
typedef struct {
    int kind;
    int n_part;
    long long part_stride;
    const float *out, *out_t;
    const float *v0, *v1, *v2, *v3;
    const float* alpha;
    float f0;
    int ld, n_heads;
    long long head_stride;
    const float* d_action;
    float* loss;
    int da_parts;                /* 0/1: d_action is a plain tensor; k: the sum of k partials da_part_stride apart */
    long long da_part_stride;    /* (the dx_part of rrl_first_layer_t: k = H/16 <= 16, or H/64 when the producer folded) */
    int da_group;                /* 0/1: the partials are added one after the other; 4: they are column-TILE partials and every
                                  * four consecutive ones are summed first ((p0 + p1) + p2) + p3, then the group sums one
                                  * after the other -- the value a producer that folds (rrl_first_layer_t.dx_fold) stores */
} rrl_loss_t;
int rrl_mlp_head_backward_loss(const rrl_loss_t* loss, int G, int B, int H, int dout, const float* h2,
                               const float* W3, float* dW3, float* db3, float* dh2, void* stream);

/* --------------------------------------------------------------------------------------------
 * Grouped launches.  One SAC / Q_risk update is a chain of ~40 tiny DEPENDENT kernels whose cost is the launch
 * boundary and a few memory round trips each, not their arithmetic; kernels that do not depend on each other
 * (the three critic forwards of sac.py:192-218 once both actions are sampled; the critic's backward for the
 * critic loss and for the policy loss; the task policy and the recovery policy of the acting pass) share ONE
 * launch here, so the chain is as long as its dependency depth.  Every member runs the code of its stand-alone
 * entry point on its own workgroups: results are bit-identical to the separate launches.  n <= 4.
 *   rrl_mlp3_forward_multi        members = rrl_mlp3_forward calls; all members must take the same path (all with
 *                                 scratch on the split path -- partial sums stay in scratch, finalize = 0 -- or all
 *                                 on the same plain tiling), else RRL_EINVAL.  Split-path members of hidden width 256
 *                                 may differ in size (round 6: a 4096-row acting forward riding with an update's 256-row
 *                                 forwards): they then run on a flat grid of exactly the workgroups each member needs,
 *                                 every member on the tiles of its stand-alone launch -- list the large member first
 *   rrl_mlp_head_backward_multi   members = rrl_mlp_head_backward_loss calls (loss.kind = -1: loss.out is a plain
 *                                 dOut tensor as in rrl_mlp_head_backward)
 *   rrl_mlp_hidden_backward_multi members = rrl_mlp_hidden_backward calls; dW2 = db2 = NULL: only dh1
 *   rrl_mlp_input_backward_multi  members = rrl_mlp_input_backward calls
 *   rrl_mlp_backward_pair_multi   = rrl_mlp_head_backward_multi(n, heads) followed by rrl_mlp_hidden_backward_multi(n, hidden),
 *                                 stack k's two stages linked by heads[k].dh2 == hidden[k].dh2.  When every member is a
 *                                 critic-loss kind (RRL_LOSS_SAC_CRITIC .. RRL_LOSS_QRISK_POLICY, one output) with full
 *                                 aligned tiles, both stages go out as ONE launch: the hidden-backward tiles derive dh2
 *                                 from the saved activation h2, the loss description and W3 themselves, and dh2 is then
 *                                 NOT written (it is scratch between the two stages, sac.py:216-239 / qrisk.py:150-182
 *                                 as autograd would hold it).  Anything else: the two launches.  Same gradients, bit for bit.
 * ------------------------------------------------------------------------------------------ */
/* use_in_head != 0 (din = 4, column-split path only): columns 2..3 of the stack's input are not read from x but computed
 * -- the action in_head yields for the same row (in_head.B is ignored: the stack's M rows) -- so the policy head needs no
 * launch of its own between the policy stack and the critic stack that consumes its action.  Columns 0..1 come from
 * in_head.obs_in (rows 2 floats apart) when given, else from x.  in_head.action / logp / obs_out, when non-null,
 * receive what the stand-alone head kernel would have written (same formulas, same bits). */
typedef struct {
    int G, M, H, din, dout, ldx;
    const float *x, *W1, *b1, *W2, *b2, *W3, *b3;
    float *h1, *h2, *out, *scratch;
    rrl_policy_head_t in_head;
    int use_in_head;
    /* nullable (H = 256, column-split path): the same W2 a second time in MFMA fragment order (rrl_w2_pack; kept in step by
     * rrl_adam_step_multi through rrl_adam_seg_t.w2p).  A wave then fetches its 16 x 256 slice as 16 whole-KB loads instead of
     * 16 x 16 half-used 128-byte lines: -15 % on every forward launch (profiles/round5_fwd_packed/).  Same values, same bits. */
    const float* W2p;
} rrl_stack_t;
typedef struct {
    rrl_loss_t loss;
    int G, B, H, dout;
    const float *h2, *W3;
    float *dW3, *db3, *dh2;
} rrl_head_bwd_t;
/* First layer of the stack backward done by the hidden-layer launch itself (instead of rrl_mlp_input_backward as a
 * dependent launch): every 16 x 16 tile of dh1 = (dh2 W2) * [h1 > 0] also emits its share of
 *   dW1 = dh1^T x, db1 = column sums of dh1   -> first_part [B/16][first_stride]: row-tile t's partial of dW1[g][h][d] at
 *                                                 t*first_stride + (g*H + h)*din + d, of db1[g][h] at ... + G*H*din + g*H + h
 *                                                 (the layout of the head of a flat [W1 | b1 | ...] gradient buffer);
 *   dx  = dh1 W1                              -> dx_part [H/16][G][B][din]: column-tile partials; with dx_fold = 1
 *                                                 [H/64][G][B][din]: the sums ((p0 + p1) + p2) + p3 of four consecutive
 *                                                 column tiles, folded inside the workgroup that holds them (the paired
 *                                                 launches of rrl_mlp_backward_pair_multi and the block form of the packed
 *                                                 hidden backward; the one-tile-per-workgroup launches return RRL_ERANGE).
 * Consumers add the partials in a fixed order: rrl_adam_step_multi (g_part fields of the segment) and the policy-head
 * backward (da_parts / da_group of rrl_loss_t: 16 tile partials with da_group = 4 give the bits of 4 folded ones).
 * x = NULL: no first-layer work (then dh1 must be given).  Needs B, H % 128 == 0. */
typedef struct {
    const float *x, *W1;
    int ldx, din;
    float* first_part;
    long long first_stride;
    float* dx_part;
    int dx_fold;
} rrl_first_layer_t;
typedef struct {
    int G, B, H;
    const float *dh2, *h1, *W2;
    float *dW2, *db2, *dh1;       /* dh1 nullable when `first` consumes it */
    rrl_first_layer_t first;
} rrl_hidden_bwd_t;
typedef struct {
    int G, B, H, din, ldx;
    const float *dh1, *x, *W1;
    float *dW1, *db1, *dx;
} rrl_input_bwd_t;
int rrl_mlp3_forward_multi(int n, const rrl_stack_t* stacks, void* stream);
int rrl_mlp_head_backward_multi(int n, const rrl_head_bwd_t* members, void* stream);
int rrl_mlp_hidden_backward_multi(int n, const rrl_hidden_bwd_t* members, void* stream);
int rrl_mlp_input_backward_multi(int n, const rrl_input_bwd_t* members, void* stream);
int rrl_mlp_backward_pair_multi(int n, const rrl_head_bwd_t* heads, const rrl_hidden_bwd_t* hidden, void* stream);

/* --------------------------------------------------------------------------------------------
 * Fused element-wise pieces of the updates (one launch each instead of a chain of PyTorch ops).
 *   rrl_gauss_head_fwd/bwd   GaussianPolicy.sample and its backward (recovery_rl/model.py:324-340);
 *                            head[b] = (mean0, mean1, log_std0, log_std1) raw linear outputs; the backward
 *                            sums d_action over n_heads critic heads (pointer + head_stride, row stride ld)
 *   rrl_sac_critic_grad      target r + m gamma (min Q' - alpha log pi') and d(mse1+mse2)/dq
 *                            (recovery_rl/sac.py:192-214); q, qt are [2,B]; loss[2] = the two MSEs
 *   rrl_sac_policy_grad      d mean(alpha log pi - min Q)/dq (sac.py:216-231); loss[1]
 *   rrl_qrisk_critic_grad    target c + m gamma_safe max sigmoid(z') and d(mse1+mse2)/dz on PRE-sigmoid
 *                            outputs (recovery_rl/qrisk.py:118-148)
 *   rrl_qrisk_policy_grad    d mean(max sigmoid(z))/dz (qrisk.py:150-154)
 *   rrl_stoch_head_fwd/bwd   StochasticPolicy.sample and its backward (model.py:511-525)
 *   rrl_adam_step            torch.optim.Adam step over one flat f32 buffer + optional Polyak update
 *                            of a target buffer (recovery_rl/utils.py:46-49); step_dev = uint64[2]
 *                            {t, ticket}, t is incremented by the kernel
 *   rrl_recovery_select      recovery gate max sigmoid(z) > eps_safe and action select
 *                            (recovery_rl/experiment.py:546-577)
 * Operands that are stack outputs (head, q, qt, qp, z, zt, zp, raw) take (n_part, part_stride): the value of
 * element i is p[i] + p[part_stride + i] + ... (n_part terms, fixed order) -- the partial last-layer sums of
 * rrl_mlp3_forward(scratch, finalize = 0); n_part = 1 for a plain tensor.
 * ------------------------------------------------------------------------------------------ */
/* obs_in (nullable, [B,2]) is copied to obs_out (row stride ld_action): builds the [s | a] critic input in place */
int rrl_gauss_head_fwd(int B, const float* head, int n_part, long long part_stride, const float* eps,
                       const float* scale, const float* bias, float* action, int ld_action, float* logp,
                       float* mean_action, const float* obs_in, float* obs_out, void* stream);
int rrl_gauss_head_bwd(int B, const float* head, int n_part, long long part_stride, const float* eps,
                       const float* scale, const float* d_action, int ld, int n_heads, long long head_stride,
                       float dlogp, float* dhead, void* stream);
int rrl_sac_critic_grad(int B, const float* q, const float* qt, int n_part, long long part_stride,
                        const float* logp2, const float* r, const float* m, float gamma, const float* alpha,
                        const float* penalty, float* dq, float* loss, void* stream);
int rrl_sac_policy_grad(int B, const float* qp, int n_part, long long part_stride, const float* logp,
                        const float* alpha, float* dqp, float* loss, void* stream);
int rrl_qrisk_critic_grad(int B, const float* z, const float* zt, int n_part, long long part_stride,
                          const float* c, const float* m, float gamma_safe, float* dz, float* loss, void* stream);
int rrl_qrisk_policy_grad(int B, const float* zp, int n_part, long long part_stride, float* dzp, float* loss,
                          void* stream);
int rrl_stoch_head_fwd(int B, const float* raw, int n_part, long long part_stride, const float* eps,
                       const float* log_std, float min_log_std, const float* scale, const float* bias,
                       float* action, int ld_action, float* mean_out, void* stream);
int rrl_stoch_head_bwd(int B, const float* raw, int n_part, long long part_stride, const float* eps,
                       const float* log_std, float min_log_std, const float* scale, const float* d_action, int ld,
                       int n_heads, long long head_stride, float* draw, float* dlog_std, void* stream);
int rrl_policy_heads_fwd_multi(int n, const rrl_policy_head_t* heads, void* stream);
int rrl_adam_step(long long n, float* p, const float* g, float* m, float* v, uint64_t* step_dev, float lr,
                  float beta1, float beta2, float eps, float* target, float tau, void* stream);
/* rrl_adam_step for up to RRL_ADAM_MAX_SEGS flat buffers in one launch (e.g. critic + policy of one update);
 * every segment has its own step counter and optional Polyak target. */
#define RRL_ADAM_MAX_SEGS 12
typedef struct {
    long long n;
    float* p;
    const float* g;
    float* m;
    float* v;
    uint64_t* step_dev;
    float* target;
    float tau;
    float weight_decay;   /* g <- g + weight_decay * p before the moment updates (torch.optim.Adam weight_decay) */
    const float* g2;      /* nullable: second partial gradient, g <- g + g2 (rrl_ens_train_grad) */
    const float* g_part;  /* nullable: the first part_elems gradients are the sum of n_part partials part_stride apart */
    int n_part;           /* (first_part of rrl_first_layer_t; added in a fixed order; n_part <= 64, part_elems % 4 == 0) */
    long long part_stride, part_elems;
    /* nullable: fragment-order copies (rrl_w2_pack layout, hidden width 256) of the w2_heads [256, 256] matrices that start at
     * element w2_off of p (w2_off % 4 == 0, 16-byte aligned pointers): every updated parameter of that range is stored there
     * too -- and the Polyak target's into target_w2p -- so that the forward kernels' rrl_stack_t.W2p stays current */
    float* w2p;
    float* target_w2p;
    long long w2_off;
    int w2_heads;
} rrl_adam_seg_t;
int rrl_adam_step_multi(int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2, float eps,
                        void* stream);
/* W2p = the G row-major [H, H] matrices W2 (out, in -- model.py's nn.Linear weights) in the order the forward kernels' MFMA B
 * operands consume them: W2p[g][n][j][q][i][c] = W2[g][16 n + i][16 j + 4 q + c] (n, j < H / 16; q, c < 4; i < 16), i.e. the float4
 * lane (i, q) of the wave that owns output columns 16 n .. 16 n + 15 needs for K chunk j sits at float4 index
 * (n H / 16 + j) 64 + 16 q + i.  H % 16 == 0.  Pure permutation: layout only. */
int rrl_w2_pack(int G, int H, const float* W2, float* W2p, void* stream);
/* out[2i], out[2i+1] = N(0,1) pair i of Philox stream RRL_STREAM_NOISE at counter (+ device tick): replaces
 * torch.randn for the policy noise of recovery_rl/model.py:324-340,511-525 (x_t = mean + std * eps). */
int rrl_normal_fill(long long n_pairs, uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                    float* out, void* stream);
int rrl_recovery_select(int N, const float* z, float eps_safe, const float* task_action, int ld_task,
                        const float* rec_action, float* real_action, uint8_t* recovery, float* task_out,
                        void* stream);

/* --------------------------------------------------------------------------------------------
 * Episode log.  The reference appends one info dict per env-step to run_stats.pkl and rewrites the whole
 * file after every episode (recovery_rl/experiment.py:421,456-461,540-543, dump_logs :540-543); its plotting
 * code reduces them per episode to: length, sum of rewards, last reward, any(constraint)
 * (plotting/plot_runs.py:194-235).  This entry keeps exactly those per-episode quantities on the device:
 * per-env accumulators (ep_len i32[n], ep_ret f64[n] summed in step order, ep_viol i32[n], ep_rec i32[n]) are
 * advanced every step, and where ep_done[i] != 0 one record is appended and the accumulators are cleared.
 *   rec_i32 [cap, RRL_EPLOG_I32] = {env, iteration, length, constraint steps, recovery steps,
 *                                   flags (1 = success, 2 = constraint, 4 = recovery, all of the LAST step)}
 *   rec_f64 [cap, 2]             = {episode return, last reward}
 *   state   int64[3]             = {count, iteration, ticket}; count keeps growing past cap (overflow is
 *                                  visible to the host; records beyond cap are dropped), iteration is
 *                                  incremented by the kernel (hipGraph replay safe).
 * Records land in completion order; (iteration, env) is unique, hosts sort by it.
 * ------------------------------------------------------------------------------------------ */
#define RRL_EPLOG_I32 6
typedef struct {
    int32_t* rec_i32;
    double* rec_f64;
    int64_t cap;
    int64_t* state;
} rrl_episode_log_t;

int rrl_episode_log_append(int64_t n, const float* reward, const uint8_t* constraint, const uint8_t* success,
                           const uint8_t* ep_done, const uint8_t* recovery, int32_t* ep_len, double* ep_ret,
                           int32_t* ep_viol, int32_t* ep_rec, const rrl_episode_log_t* log, void* stream);

/* --------------------------------------------------------------------------------------------
 * Planner candidate evaluation.  Replaces MPC._compile_cost (recovery_rl/MPC.py:374-416) with
 * _predict_next_obs (:421-439), the ensemble forward (config/navigation1.py:71-96) and
 * QRiskWrapper.get_value (recovery_rl/qrisk.py:184-196) for M planning problems at once:
 *   costs[m, c] = mean over npart particles of sum_{t < plan_hor} max(Q_risk1, Q_risk2)(obs_t, ac_seqs[m, c, t]),
 *   obs_{t+1} = obs_t + mean_e(obs_t, ac_t) + z * sqrt(var_e(obs_t, ac_t)),  particle p uses member p / (npart / n_nets),
 * NaN particle costs -> 1e6.  One MFMA kernel; activations never leave the chip.
 *   rrl_plan_pack      re-packs the live weights into MFMA fragment order (call after every change of the
 *                      safety critic or the ensemble).  Q_risk tensors are the stacked twin heads W1 [2,hq,4],
 *                      b1 [2,hq], W2 [2,hq,hq], b2 [2,hq], W3 [2,1,hq], b3 [2,1] (nn.Linear layout, out x in);
 *                      ensemble tensors are lin0_w [E,4,he], lin0_b [E,1,he], lin1_w/lin2_w [E,he,he],
 *                      lin3_w [E,he,4], lin3_b [E,1,4] (in x out), inputs_mu/sigma [4], max/min_logvar [2].
 *   rrl_plan_cost      cur_obs [M,2], ac_seqs [M,pop,plan_hor*2] f32; noise nullable f32 [plan_hor, M*pop*npart, 2]
 *                      (row = (m*pop + c)*npart + p); when NULL the kernel draws Philox normals (stream
 *                      RRL_STREAM_PLAN, row, counter*16 + t).  scratch: f32 [rrl_plan_scratch_floats(n_nets, M, pop)]
 *                      (first-step values per candidate / per (candidate, member) + per-member cost sums); costs [M,pop].
 *                      Two launches + the finish: the first step once per DISTINCT row (the particles of a candidate share
 *                      (cur_obs, ac_0): MPC.py:393-402), then steps 1..plan_hor-1 per particle without the last step's
 *                      unread prediction (MPC.py:406-412) -- bit-identical to the literal loop, 74.9 % of its FLOPs.
 * Supported shape (rrl_plan_supported): hq = 256, he = 200, npart = 4 n_nets, 2-D obs and actions.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int hq, he, n_nets;
    const float *q_w1, *q_b1, *q_w2, *q_b2, *q_w3, *q_b3;
    const float *e_w0, *e_b0, *e_w1, *e_b1, *e_w2, *e_b2, *e_w3, *e_b3;
    const float *inputs_mu, *inputs_sigma, *max_logvar, *min_logvar;
} rrl_plan_weights_t;

int rrl_plan_supported(int hq, int he, int n_nets, int npart, int d_obs, int d_act);
long long rrl_plan_pack_floats(int hq, int he, int n_nets);
long long rrl_plan_scratch_floats(int n_nets, long long M, int pop);     /* M * pop * (5 n_nets + 1) */
int rrl_plan_pack(const rrl_plan_weights_t* w, float* packed, void* stream);
int rrl_plan_cost(const float* packed, int hq, int he, int n_nets, int npart, long long M, int pop, int plan_hor,
                  const float* cur_obs, const float* ac_seqs, const float* noise, uint64_t seed, uint64_t counter,
                  uint64_t* counter_dev, uint64_t counter_inc, float* scratch, float* costs, void* stream);

/* The same evaluation with the three hidden-layer products (Q_risk 256 x 256, ensemble 200 x 200 twice) on the f16 matrix
 * pipe: every f32 activation and weight is split as hi + lo (two f16 carrying 22 bits of the value) and the product is
 * hi*hi + hi*lo + lo*hi with f32 accumulation -- one v_mfma_f32_16x16x16_f16 (16 cycles) three times instead of four
 * v_mfma_f32_16x16x4_f32 (32 cycles each) per 16-wide k chunk.  Input layers, biases, activations, epilogues and the
 * rollout state stay f32.  Same interface; the packed buffer has the same size but is NOT interchangeable (pack with
 * rrl_plan_pack_f16x3).  Opt-in: results agree with rrl_plan_cost to ~1e-6 relative (tests: the same 2e-4 bound as the
 * f32 kernel against the PyTorch path); values beyond +-65504 in a hidden layer saturate. */
int rrl_plan_pack_f16x3(const rrl_plan_weights_t* w, float* packed, void* stream);
int rrl_plan_cost_f16x3(const float* packed, int hq, int he, int n_nets, int npart, long long M, int pop, int plan_hor,
                        const float* cur_obs, const float* ac_seqs, const float* noise, uint64_t seed, uint64_t counter,
                        uint64_t* counter_dev, uint64_t counter_inc, float* scratch, float* costs, void* stream);

/* rrl_plan_cost / rrl_plan_cost_f16x3 (f16x3 != 0) for M = m_dev[0] planning problems, M read by the kernel (see
 * rrl_cem_begin); the grid covers m_max problems and workgroups past the live ones exit at once.  cur_obs, ac_seqs,
 * scratch and costs are sized for m_max.  Results for the live problems equal the host-count entries' bit for bit. */
int rrl_plan_cost_n(int f16x3, const float* packed, int hq, int he, int n_nets, int npart, const int32_t* m_dev,
                    long long m_max, int pop, int plan_hor, const float* cur_obs, const float* ac_seqs, const float* noise,
                    uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* scratch,
                    float* costs, void* stream);

/* --------------------------------------------------------------------------------------------
 * Ensemble fitting.  One optimiser step of MPC.train (recovery_rl/MPC.py:266-292) for the PETS ensemble
 * (PtModel, config/navigation1.py:23-96): gather of the bootstrap rows idx[e, 0..batch), forward, loss
 *   sum_e mean((mean_e - y)^2 exp(-logvar_e) + logvar_e) + 0.01 (sum max_logvar - sum min_logvar) + decays (:52-59)
 * and its gradient w.r.t. every parameter EXCEPT the decay terms (pass them as the segments' weight_decay:
 * 0.00025 / 0.0005 / 0.0005 / 0.00075 for w0..w3) in ONE launch (+ a 4-thread reduction for the shared logvar bounds);
 * the update itself is rrl_adam_step_multi over the same buffers (torch.optim.Adam, lr 1e-3).
 *   parameters  w0 [E,4,H] b0 [E,1,H] w1,w2 [E,H,H] b1,b2 [E,1,H] w3 [E,H,4] b3 [E,1,4] (in x out), max/min_logvar [2],
 *               mu/sigma [4] (input standardisation, not trained); g_* = gradients, same shapes;
 *               g_logvar_part: scratch [2E,4].  A member's 32 rows are processed by two workgroups of 16 rows:
 *               rows 0..15 write g_*, rows 16..31 write g2_* (same shapes); pass g2 as the Adam segment's second
 *               gradient so that the update uses g + g2
 *   idx         int64 [E, >= batch] with row stride idx_stride (elements): rows of train_in [N,4] / train_targ [N,2]
 *   scratch     float [rrl_ens_scratch_floats(E)]; loss_out (nullable) [E] = the per-net NLL term
 * Supported shape (rrl_ens_train_supported): 4 inputs, H = 200, 4 outputs, batch 1..32 (the mean runs over the
 * real rows, as for the shorter last batch of an epoch).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_nets, d_in, hidden, d_out;
    float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3, *max_logvar, *min_logvar;
    const float *mu, *sigma;
    float *g_w0, *g_b0, *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3, *g_max_logvar, *g_min_logvar, *g_logvar_part;
    float *g2_w0, *g2_b0, *g2_w1, *g2_b1, *g2_w2, *g2_b2, *g2_w3, *g2_b3;   /* second half of the batch (see above) */
    float* loss_part;                                                       /* scratch [2 E] */
} rrl_ens_t;
int rrl_ens_train_supported(int d_in, int hidden, int d_out, int batch);
long long rrl_ens_scratch_floats(int n_nets);
/* one epoch = ceil(n_rows / batch) steps {rrl_ens_train_grad on idx[:, lo:lo+batch], rrl_adam_step_multi(segs)} issued
 * from C (the batch loop of MPC.py:266-292); segs = the Adam segments of the ten parameter tensors */
int rrl_ens_train_epoch(const rrl_ens_t* m, int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2,
                        float eps, const float* train_in, const float* train_targ, const int64_t* idx,
                        long long idx_stride, long long n_rows, int batch, float* scratch, float* loss_out,
                        void* stream);
int rrl_ens_train_grad(const rrl_ens_t* m, int batch, const float* train_in, const float* train_targ,
                       const int64_t* idx, long long idx_stride, float* scratch, float* loss_out, void* stream);

/* The same optimiser step at LARGE batch (the lock-step loop's online re-fit trains on 32 x num_envs rows per member per
 * step, experiment.py:464-480 scaled by the number of envs; any batch >= 1 is accepted): same loss, same gradients
 * (to f32 summation order), same rrl_ens_t, gradients land in g_* only (g2_*, g_logvar_part, loss_part are not used: pass
 * Adam segments without g2).  Three launches: forward + backward per 64-row tile with the activations in LDS (f32 MFMA),
 * split-K weight-gradient products, fixed-order reduction of the partials (deterministic).
 *   scratch     float [rrl_ens_big_scratch_floats(E, batch)]  (6 activation-sized buffers [E][ceil64(batch)][200] + partials)
 *   idx         int64, member e's rows at idx[e * idx_stride + 0 .. batch) */
int rrl_ens_train_big_supported(int d_in, int hidden, int d_out);
long long rrl_ens_big_scratch_floats(int n_nets, long long batch);
int rrl_ens_train_grad_big(const rrl_ens_t* m, long long batch, const float* train_in, const float* train_targ,
                           const int64_t* idx, long long idx_stride, float* scratch, float* loss_out, void* stream);
int rrl_ens_train_epoch_big(const rrl_ens_t* m, int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2,
                            float eps, const float* train_in, const float* train_targ, const int64_t* idx,
                            long long idx_stride, long long n_rows, long long batch, float* scratch, float* loss_out,
                            void* stream);

/* --------------------------------------------------------------------------------------------
 * Packed launches: S independent learners ("seeds" -- own envs, replay rings, networks, Philox keys; the reference's unit
 * of parallelism is the seed loop, scripts/navigation1.sh:4-8) share every launch of the lock-step iteration.  Each entry is
 * its stand-alone counterpart for S argument sets at once: seed s runs exactly the stand-alone code on its own workgroups,
 * so every seed's results equal its solo run bit for bit.  The S argument blocks live in device memory (content-addressed
 * cache inside the library: blocks that do not change from call to call are uploaded once).  S <= 16.
 * Launch structure (round 5): as in the solo group launches the member of a seed's group is blockIdx.y, and the seed follows
 * from blockIdx.x by arithmetic (XCD-aware placement), so a workgroup's argument block arrives in one batch of scalar loads.
 *   rrl_sample_multi_packed                rrl_sample_multi            (args[s])
 *   rrl_mlp3_forward_multi_packed          rrl_mlp3_forward_multi      (n[s] stacks members[s][0..n[s]); column-split path)
 *   rrl_mlp_head_backward_multi_packed     rrl_mlp_head_backward_multi
 *   rrl_mlp_hidden_backward_multi_packed   rrl_mlp_hidden_backward_multi
 *   rrl_mlp_backward_pair_multi_packed     rrl_mlp_backward_pair_multi (heads[s], hidden[s]: one launch when every member of
 *                                          every seed qualifies for the paired form, the two packed launches otherwise)
 *   rrl_adam_step_multi_packed             rrl_adam_step_multi         (lr[s])
 *   rrl_nav_step_push_packed / rrl_maze_step_push_packed    rrl_*_step_push_x (a[s]; one env kind, sizes on one side of 16384)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const rrl_draw_t *first, *second;
    long long noise_pairs;
    uint64_t noise_seed, noise_counter;
    uint64_t* noise_counter_dev;
    uint64_t noise_counter_inc;
    float* noise_out;
} rrl_sample_args_t;
int rrl_sample_multi_packed(int S, const rrl_sample_args_t* args, void* stream);
/* Free the cached argument blocks of every packed launch (host and device copies); returns their number.  Call only when no
 * captured graph that contains a packed launch is alive (the graphs hold the blocks' device addresses as kernel arguments). */
int rrl_pack_clear(void);
int rrl_mlp3_forward_multi_packed(int S, const int* n, const rrl_stack_t* const* members, void* stream);
int rrl_mlp_head_backward_multi_packed(int S, const int* n, const rrl_head_bwd_t* const* members, void* stream);
int rrl_mlp_hidden_backward_multi_packed(int S, const int* n, const rrl_hidden_bwd_t* const* members, void* stream);
int rrl_mlp_backward_pair_multi_packed(int S, const int* n, const rrl_head_bwd_t* const* heads,
                                       const rrl_hidden_bwd_t* const* hidden, void* stream);
int rrl_adam_step_multi_packed(int S, const int* n_seg, const rrl_adam_seg_t* const* segs, const float* lr, float beta1,
                               float beta2, float eps, void* stream);
int rrl_nav_step_push_packed(int S, int env_kind, const rrl_step_push_t* a, void* stream);
int rrl_maze_step_push_packed(int S, const rrl_step_push_t* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
