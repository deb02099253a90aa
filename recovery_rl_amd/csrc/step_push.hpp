// step_push.hpp -- the lock-step iteration tail as ONE kernel, shared by the navigation and maze envs:
// env.step + reward penalty + bootstrap mask + memory.push + recovery_memory.push + episode counters
// (recovery_rl/experiment.py:420-461).  ENV supplies the transition and the reset draw:
//   static Outcome ENV::step(const StepArgs&, int64_t i, uint64_t ctr, double2 pos, float2 action, int32_t t_after)
//   static void    ENV::reset(const StepArgs&, int64_t i, uint64_t ctr, double& x, double& y)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pack.hpp"
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
    // compact per-env state (step_push only; nullable): ONE u16 word instead of the i32 step count + four u8 flags --
    // count in bits 0-11, done / constraint / success / ep_done of the LAST step in bits 12-15; with it the stored state is
    // float(pos) (the observation array is written, never read)
    uint16_t* status;
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
    const float* task_action; // row i at task_action + i * ld_task
    int ld_task;
    const uint8_t* recovery;  // nullable
    // recovery gate evaluated here instead of in a kernel of its own (experiment.py:546-577, rrl_recovery_select):
    // recovery = max(sigmoid(z[i]), sigmoid(z[n + i])) > eps_safe; executed action = recovery ? rec : task
    const float* sel_z;       // nullable: [2, n] pre-sigmoid Q_risk(s, a_task), as sel_np partial sums sel_ps apart
    int sel_np;
    long long sel_ps;
    float sel_eps;
    const float2* sel_rec_action;   // the recovery action, or (null) computed here from the recovery policy's head:
    rrl_policy_head_t sel_rec_head; // rrl_stoch_head_fwd evaluated per env (same formulas, same bits)
    float2* sel_real_out;     // the executed action and the flag are written for the consumers downstream
    uint8_t* sel_recovery_out;
    float reward_penalty;
    int push_real_action;     // disable_action_relabeling (experiment.py:437-441)
    rrl_replay_t memory;
    rrl_replay_t recovery_memory;
    int use_recovery_memory;
    unsigned long long* stats;   // env_steps, episodes, num_viols, viol_and_recovery, viol_and_no_recovery,
                                 // num_successes, recovery_steps, constraint_steps
    double* reward_sums;         // {sum of rewards, sum of finished-episode returns}
    float* ep_reward;            // [n] running episode return
    // per-episode log advanced here (log_kernels.hip's episode_log_kernel, lane for lane): nullable log_state = off
    int32_t* log_rec_i32;
    double* log_rec_f64;
    int64_t log_cap;
    int64_t* log_state;          // {count, iteration, ticket (unused here: the cursors' ticket serves)}
    int32_t* log_len;
    double* log_ret;
    int32_t *log_viol, *log_rec;
};

// Workgroup size of the two variants.  Latency regime (SPECULATE, n <= 16384): 256, up to 64 workgroups.  Bandwidth regime: 1024
// -- every workgroup ends with up to ten atomics on the SAME few addresses (seven counters, two reward sums, the episode
// table's slot counter), and same-address atomics retire one per ~9 ns whatever their number in flight: with 256-thread
// workgroups a 2^20-env launch spent 37 us of its 65 on the cursor ticket alone (round4_step_push_isa.txt).
// Three launch shapes, by size (regime_of): 0 = latency variant (SPECULATE, 256 threads, ticket); 1 = bandwidth variant with
// 256 threads (a 65536-env launch has one wave per SIMD: as 64 workgroups of 1024 it would leave three quarters of the CUs
// idle, 11 -> 16 us); 2 = bandwidth variant with 1024 threads.
constexpr int64_t kSmall = 16384;     // up to here the latency variant: a quarter of the SIMDs busy at most, the reset draw runs beside the step draw (at 65536 envs it costs 13 -> 17 us)
constexpr int64_t kMid = 262144;      // up to here one 256-thread workgroup per CU and SIMD slot: 1024 workgroups
inline int regime_of(int64_t n) { return n <= kSmall ? 0 : (n <= kMid ? 1 : 2); }
__host__ __device__ constexpr int block_of(int regime) { return regime == 2 ? 1024 : rrl_host::kBlock; }

// Episode counters: counted per wave (one ballot + popcount each), added up per workgroup in LDS, and ONE atomic per
// workgroup and non-zero counter reaches memory.
// (One atomic per wave and iteration serialised 115 k atomics on 7 addresses at 2^20 envs: 719 us.)
constexpr int kCounters = 7;

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over each 16-lane row, every lane gets the row's total (xor-butterfly order 8, 4, 2, 1 by row rotations)
__device__ __forceinline__ double row16_sum_f64(double v) {
    v += dpp_move_f64<0x128>(v);
    v += dpp_move_f64<0x124>(v);
    v += dpp_move_f64<0x122>(v);
    v += dpp_move_f64<0x121>(v);
    return v;
}

// {position, size} of both rings, the RNG tick and the episode table's iteration count after this launch's n rows: by the
// thread that knows every workgroup has read the old values
__device__ __forceinline__ void advance_cursors(const StepPushArgs& p, int64_t mpos, int64_t msize, int64_t rpos, int64_t rsize,
                                                uint64_t ctr, int64_t log_iteration) {
    const StepArgs& a = p.step;
    rrl_replay::set_ring(p.memory, mpos, msize, a.n);
    if (p.use_recovery_memory) rrl_replay::set_ring(p.recovery_memory, rpos, rsize, a.n);
    if (a.counter_dev && a.counter_inc) a.counter_dev[0] = ctr - a.counter + a.counter_inc;
    if (p.log_state) p.log_state[1] = log_iteration + 1;
}

// SPECULATE (latency regime, one pass per thread and one wave per SIMD): the reset draw does not depend on the step, so it
// is evaluated next to the step's own draw -- two independent Philox + Box-Muller chains interleaved by the scheduler --
// instead of after it for the lanes whose episode ended (at the bench's termination rate that is every wave: a second
// ~1 us dependent chain).  Same function, same arguments, same bits; in the bandwidth regime it would be +70 % VALU work.
// !SPECULATE (bandwidth regime): the second level of the safety buffer's positive counts is summed per workgroup and pass
// in LDS (one atomic per workgroup and super-chunk instead of one per wave: 64 waves share a super-chunk's counter).
template <class ENV, bool SPECULATE, int BLOCK>
__device__ __forceinline__ void step_push_body(const StepPushArgs& p, const unsigned blk, const unsigned n_blk) {
    constexpr int kBlock = BLOCK;
    constexpr bool kBlockSuper = !SPECULATE;
    __shared__ int super_acc[2];
    __shared__ int log_cnt[kBlock / 64];
    __shared__ long long log_base;
    const bool counts = p.use_recovery_memory && p.recovery_memory.pos_cnt != nullptr;
    if (kBlockSuper) {
        if (threadIdx.x < 2) super_acc[threadIdx.x] = 0;
        __syncthreads();
    }
    const StepArgs& a = p.step;
    unsigned long long ticket = ~0ULL;
    int64_t mpos = 0, msize = 0, rpos = 0, rsize = 0, log_iteration = 0;      // the cursors (read below, behind the per-env requests)
    uint64_t ctr = 0;
    double rsum = 0.0, retsum = 0.0;
    unsigned cnt[kCounters] = {0, 0, 0, 0, 0, 0, 0};
    // ONE pass: the launch covers its envs (grid_cover), workgroup blk steps envs [256 blk, 256 blk + 256).  No grid-stride loop:
    // around a loop the compiler keeps every loop-invariant address expression of the ~35 arrays, plus an induction
    // pointer per array, in scalar registers -- 340 SGPRs spilled into vector lanes, a fifth of the kernel's instructions
    // v_readlane / v_writelane, 113 VGPRs; without it 43-90 spills and 63-82 VGPRs (profiles/round4_step_push_isa.txt).
    {
        const int64_t off = int64_t(blk) * kBlock;
        const int64_t i = off + threadIdx.x;
        const bool live = i < a.n;
        bool cons = false, succ = false, epd = false, rec = false;
        float log_rew = 0.f;
        unsigned long long log_bal = 0;     // latency variant: finished-episode votes of this wave and the leader's reservation
        long long log_base_raw = 0;
        // EVERY per-env input is requested here, before anything waits for the cursors: the per-env accumulators (running
        // return; the episode table's four), position, step count, the task action, the gate's partial sums and the recovery
        // head's.  (Behind the ticket they queued up behind the cursor round trip in wave 0 of every workgroup; behind the
        // replay stores the compiler has to keep them there: it cannot prove they do not alias.)
        float ep_rew_in = 0.f;
        double lg_ret = 0.0;
        int lg_len = 0, lg_viol = 0, lg_rec = 0;
        double2 pp = make_double2(0.0, 0.0);
        float2 task = make_float2(0.f, 0.f), act_in = task, ra_in = task, obs_in = task;
        float zu[4] = {0.f, 0.f, 0.f, 0.f}, zw[4] = {0.f, 0.f, 0.f, 0.f}, hh[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float he[2] = {0.f, 0.f};
        int32_t t_in = 0;
        bool rec_in = false;
        // what the requests below return, untouched until the cursors have been requested too (the first instruction that
        // looks at one of them waits for all of them)
        int lg_len_ = 0, lg_viol_ = 0, lg_rec_ = 0;
        double lg_ret_ = 0.0;
        float zu_[4] = {0.f, 0.f, 0.f, 0.f}, zw_[4] = {0.f, 0.f, 0.f, 0.f}, hh_[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float he_[2] = {0.f, 0.f};
        float2 ra_ = task, act_ = task, obs_ = task;
        uint8_t rec_ = 0;
        uint16_t st_ = 0;
        int32_t t_ = 0;
        if (live) {
            // BRANCH-FREE: an optional array is read through a selected address (its own element, or this env's position --
            // 16 valid bytes -- when the array is absent) and the value dropped afterwards.  A load under a branch is waited
            // for where the branch ends (the merged register must hold the value there): the optional groups below were five
            // round trips in a row instead of one.
            const void* const safe = a.pos + i;
            // (bandwidth variants: under the branch after all -- with thousands of waves in flight nobody waits for one wave's
            // round trips, and the stand-in requests are traffic: 50.7 -> 58.8 us at 2^20 envs)
            const auto rd = [&](const auto* arr, long long idx) {
                using T = std::remove_cv_t<std::remove_pointer_t<decltype(arr)>>;
                if constexpr (SPECULATE) return *(arr ? arr + idx : reinterpret_cast<const T*>(safe));
                else return arr ? arr[idx] : T{};
            };
            ep_rew_in = p.ep_reward[i];
            const bool lg = p.log_state != nullptr;
            lg_len_ = rd(lg ? p.log_len : nullptr, i);
            lg_ret_ = rd(lg ? p.log_ret : nullptr, i);
            lg_viol_ = rd(lg ? p.log_viol : nullptr, i);
            lg_rec_ = rd(lg ? p.log_rec : nullptr, i);
            pp = a.pos[i];
            task = *reinterpret_cast<const float2*>(p.task_action + i * p.ld_task);
            // the gate's partial sums (up to four, all loads issued together) and the recovery action or the head it comes from
            const bool sel = p.sel_z != nullptr;
            const int np = p.sel_np;
            const long long ps = p.sel_ps;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                zu_[k] = rd(p.sel_z, i + (np > k ? k * ps : 0));
                zw_[k] = rd(p.sel_z, i + (np > k ? k * ps : 0) + a.n);
            }
            ra_ = rd(sel ? p.sel_rec_action : nullptr, i);
            const rrl_policy_head_t& hd = p.sel_rec_head;
            const bool from_head = sel && !p.sel_rec_action;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    hh_[jj][k] = rd(from_head ? hd.head : nullptr, 2 * i + jj + (hd.n_part > k ? k * hd.part_stride : 0));
                he_[jj] = rd(from_head ? hd.eps : nullptr, 2 * i + jj);
            }
            act_ = rd(sel ? nullptr : a.action, i);
            rec_ = rd(sel ? nullptr : p.recovery, i);
            // compact layout: the observation IS float(pos) (that is what this kernel and the resets store), so the 8-byte
            // read is dropped, and the step count comes out of the status word
            st_ = rd(a.status, i);
            t_ = rd(a.status ? nullptr : a.t, i);
            obs_ = rd(a.status ? nullptr : a.obs, i);
        }
        // plain (wave-uniform: scalar) loads of the cursors; the ticket below is issued only after they have RETURNED (s_waitcnt)
        // and the compiler may not move them past it ("memory" clobber), so a workgroup's read of a cursor cannot slip behind its
        // ticket.  (Atomic loads here are per-lane vector loads of ONE address: 4 M of them at 2^20 envs, 50 -> 120 us.)
        // The four cursors (task ring, safety ring, RNG tick, table iteration) are requested TOGETHER: the optional ones through
        // a selected address (the task ring's state when absent) instead of under a branch -- a load under a branch is waited for
        // at the branch's end, and four such round trips in a row opened this kernel (~0.5 us each on cold lines).  Behind the
        // per-env requests above: both sets are in flight at once.
        const int64_t* rstate = p.use_recovery_memory ? p.recovery_memory.state : p.memory.state;
        const int64_t* lstate = p.log_state ? p.log_state : p.memory.state;
        const uint64_t* cdev = a.counter_dev ? a.counter_dev : reinterpret_cast<const uint64_t*>(p.memory.state);
        const int64_t mpos_ = p.memory.state[0], msize_ = p.memory.state[1];
        const int64_t rpos_ = rstate[0], rsize_ = rstate[1], liter_ = lstate[1];
        const uint64_t tick_ = cdev[0];
        asm volatile("" ::"s"(mpos_), "s"(msize_), "s"(rpos_), "s"(rsize_), "s"(liter_), "s"(tick_));     // one batch, one wait
        mpos = mpos_;
        msize = msize_;
        rpos = p.use_recovery_memory ? rpos_ : 0;
        rsize = p.use_recovery_memory ? rsize_ : 0;
        log_iteration = p.log_state ? liter_ : 0;      // read before the ticket, like the cursors
        ctr = a.counter_dev ? a.counter + tick_ : a.counter;      // rrl::effective_counter
        // Latency regime: ONE ticket for the three device-side cursors (both replay rings and the RNG tick): a returning
        // device-scope atomic is a ~0.7 us round trip, three in a row were a sixth of this kernel.  The workgroup that draws
        // the last ticket knows that every workgroup has read the cursors, which is all their update has to wait for.  It is
        // drawn as soon as the cursors have arrived and its value is looked at when the workgroup is done, so the round trip
        // runs under the env step.  Bandwidth regime: no ticket (thousands of returning atomics on one address: 9 ns each,
        // serialised) -- the cursors are advanced by advance_cursors_kernel, a one-thread launch behind this one.
        if constexpr (SPECULATE) {
            __syncthreads();            // every wave of this workgroup has requested its copies of the cursors
            if (threadIdx.x == 0) {
                // the cursor loads have RETURNED before the ticket is issued (s_waitcnt; "memory": the compiler keeps the
                // order too).  Not a release operation: at agent scope that is an L2 write-back per workgroup, and nothing
                // written here has to be visible before the kernel ends.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                ticket = __hip_atomic_fetch_add((unsigned long long*)&p.memory.state[2], 1ULL, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        {
            // what was read through a fallback address is dropped (lanes past n keep the zeros).  (The integers pass through an
            // empty asm: the compiler would otherwise test / mask / increment them inside the request block above, i.e. wait
            // for the requests before the cursors are asked for.)
            unsigned rec_u = rec_, st_u = st_;
            asm volatile("" : "+v"(rec_u), "+v"(st_u), "+v"(lg_len_), "+v"(lg_viol_), "+v"(lg_rec_), "+v"(t_));
            const bool lg = p.log_state != nullptr, sel = p.sel_z != nullptr, from_head = sel && !p.sel_rec_action;
            const rrl_policy_head_t& hd = p.sel_rec_head;
            if (lg) { lg_len = lg_len_; lg_ret = lg_ret_; lg_viol = lg_viol_; lg_rec = lg_rec_; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { zu[k] = zu_[k]; zw[k] = zw_[k]; hh[0][k] = hh_[0][k]; hh[1][k] = hh_[1][k]; }
            ra_in = ra_;
            he[0] = (from_head && hd.eps) ? he_[0] : 0.f;
            he[1] = (from_head && hd.eps) ? he_[1] : 0.f;
            act_in = act_;
            rec_in = (!sel && p.recovery) ? rec_u != 0 : false;
            t_in = a.status ? int32_t(st_u & 0xfffu) : t_;
            obs_in = obs_;
        }
        int s0 = 0, s1 = 0;       // super-chunks of the workgroup's first and last safety-buffer slot of this pass
        // the two workgroup sums cover a pass whose consecutive slots touch at most two super-chunks: always, unless
        // the pass wraps around a ring whose capacity is not a multiple of the super-chunk (then it can touch the last
        // two super-chunks AND super-chunk 0): that pass sends its waves' nets straight to memory instead
        bool block_sums = false;
        if (kBlockSuper && counts) {
            const int64_t cap = p.recovery_memory.cap;
            const int64_t first = rrl_replay::ring_slot(p.recovery_memory, rpos, off);
            s0 = int(first / rrl_replay::kSuper);
            s1 = int(rrl_replay::ring_slot(p.recovery_memory, rpos, off + kBlock - 1) / rrl_replay::kSuper);
            block_sums = first + kBlock <= cap ||
                         (cap % rrl_replay::kSuper == 0 && p.recovery_memory.pinned % rrl_replay::kSuper == 0);
        }
        if (live) {
            // the rows this env overwrites: what the positive counts lose (replay_device.hpp), requested before the step
            const int64_t mslot = rrl_replay::ring_slot(p.memory, mpos, i);
            const int64_t rslot = p.use_recovery_memory ? rrl_replay::ring_slot(p.recovery_memory, rpos, i) : 0;
            const rrl_replay::WasRow mrow = rrl_replay::was_positive_request<SPECULATE>(p.memory, mslot, msize, a.pos + i);
            const rrl_replay::WasRow rrow = rrl_replay::was_positive_request<SPECULATE>(
                p.use_recovery_memory ? p.recovery_memory : p.memory, rslot, p.use_recovery_memory ? rsize : 0, a.pos + i);
            float2 act;
            if (p.sel_z) {
                // the partial sums added in the fixed order of the sum kernel
                const int np = p.sel_np;
                float z0 = zu[0], z1 = zw[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    z0 = np > k ? z0 + zu[k] : z0;
                    z1 = np > k ? z1 + zw[k] : z1;
                }
                const float q0 = 1.f / (1.f + expf(-z0)), q1 = 1.f / (1.f + expf(-z1));
                rec = fmaxf(q0, q1) > p.sel_eps;
                float2 ra = ra_in;
                if (!p.sel_rec_action) {
                    const rrl_policy_head_t& hd = p.sel_rec_head;
                    float v[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int hn = hd.n_part;
                        float raw = hh[j][0];
#pragma unroll
                        for (int k = 1; k < 4; ++k) raw = hn > k ? raw + hh[j][k] : raw;
                        const float mean = tanhf(raw) * hd.scale[j] + hd.bias[j];
                        v[j] = mean + expf(fmaxf(hd.log_std[j], hd.min_log_std)) * he[j];
                    }
                    ra = make_float2(v[0], v[1]);
                }
                act = rec ? ra : task;
                p.sel_real_out[i] = act;
                p.sel_recovery_out[i] = uint8_t(rec);
            } else {
                act = act_in;
                rec = rec_in;
            }
            const float2 prev = a.status ? make_float2(float(pp.x), float(pp.y)) : obs_in;
            int32_t ti = t_in;
            ti += 1;
            double rx = 0.0, ry = 0.0;
            if constexpr (SPECULATE) {
                if (a.auto_reset) ENV::reset(a, i, ctr, rx, ry);
            }
            const Outcome out = ENV::step(a, i, ctr, pp, act, ti);
            double nx = out.x, ny = out.y;
            cons = out.constraint;
            succ = out.success;
            const bool dn = out.done;
            epd = dn | (ti == a.horizon);
            if constexpr (SPECULATE) {
                // the episode table's slots for this wave's finished episodes: ONE returning atomic per wave, requested as
                // soon as the outcomes are known and looked at after the replay stores and the count maintenance (its ~0.7 us
                // round trip used to sit at the end of the kernel).  Inside the live branch: lanes past n would vote 0 anyway.
                if (p.log_state) {
                    log_bal = __ballot(epd);
                    if (log_bal && (threadIdx.x & 63) == __ffsll((long long)log_bal) - 1)
                        log_base_raw = (long long)atomicAdd((unsigned long long*)&p.log_state[0],
                                                            (unsigned long long)__popcll(log_bal));
                }
            }
            const float2 nobs = make_float2(float(nx), float(ny));
            const float rew = out.reward;
            // per-env outputs of the step for callers that read them (the episode log, the online ensemble re-fit): the
            // replay rows and the counters below do not need them, and 17 of the kernel's 171 B per env-step are theirs
            if (a.next_obs) a.next_obs[i] = nobs;
            if (a.reward) a.reward[i] = rew;
            if (a.done) a.done[i] = uint8_t(dn);
            if (a.constraint) a.constraint[i] = uint8_t(cons);
            if (a.success) a.success[i] = uint8_t(succ);
            if (a.ep_done) a.ep_done[i] = uint8_t(epd);
            // replay rows (experiment.py:431-448)
            const float mask = dn ? 0.0f : 1.0f;
            const float prew = rew - (cons ? p.reward_penalty : 0.0f);
            const float2 stored = p.push_real_action ? act : task;
            rrl_replay::store_values(p.memory, mslot, rrl_replay::was_positive_of(mrow), prev, stored, prew, nobs, mask);
            if (p.use_recovery_memory)
                rrl_replay::store_values(p.recovery_memory, rslot, rrl_replay::was_positive_of(rrow), prev, act,
                                         cons ? 1.0f : 0.0f, nobs, mask,
                                         block_sums ? super_acc : nullptr, s0);
            // episode accounting
            log_rew = rew;
            const float er = ep_rew_in + rew;
            rsum += double(rew);
            if (epd) retsum += double(er);
            p.ep_reward[i] = epd ? 0.0f : er;
            if (a.auto_reset && epd) {
                if constexpr (SPECULATE) {
                    nx = rx;
                    ny = ry;
                } else {
                    ENV::reset(a, i, ctr, nx, ny);
                }
                ti = 0;
            }
            a.pos[i] = make_double2(nx, ny);
            if (a.status) a.status[i] = uint16_t(unsigned(ti) | (unsigned(dn) << 12) | (unsigned(cons) << 13) |
                                                 (unsigned(succ) << 14) | (unsigned(epd) << 15));
            else a.t[i] = ti;
            a.obs[i] = make_float2(float(nx), float(ny));
        }
        if (p.log_state) {
            // episode_log_kernel (log_kernels.hip) for this lane, fed from registers
            double ret = 0.0;
            int log_len = 0, viol = 0, recs = 0;
            if (live) {
                log_len = lg_len + 1;
                ret = lg_ret + double(log_rew);
                viol = lg_viol + int(cons);
                recs = lg_rec + int(rec);
            }
            const unsigned long long bal = SPECULATE ? log_bal : __ballot(epd);
            const int lane = threadIdx.x & 63;
            long long base = 0;
            if constexpr (SPECULATE) {
                if (bal) base = __shfl(log_base_raw, __ffsll((long long)bal) - 1, 64);
            } else {
                // ONE returning atomic per workgroup (per wave: 16 384 of them on one address at 2^20 envs, 146 us): the
                // waves' counts meet in LDS, thread 0 reserves the workgroup's slots, every wave takes its share in wave order
                const int w = threadIdx.x >> 6;
                if (lane == 0) log_cnt[w] = __popcll(bal);
                __syncthreads();
                if (threadIdx.x == 0) {
                    int tot = 0;
#pragma unroll
                    for (int k = 0; k < kBlock / 64; ++k) tot += log_cnt[k];
                    log_base = tot ? (long long)atomicAdd((unsigned long long*)&p.log_state[0], (unsigned long long)tot) : 0;
                }
                __syncthreads();
                base = log_base;
                for (int k = 0; k < w; ++k) base += log_cnt[k];
            }
            if (bal) {
                if (epd) {
                    const long long slot = base + __popcll(bal & ((1ULL << lane) - 1ULL));
                    if (slot < p.log_cap) {
                        int32_t* ri = p.log_rec_i32 + slot * RRL_EPLOG_I32;
                        ri[0] = int32_t(i);
                        ri[1] = int32_t(log_iteration);
                        ri[2] = log_len;
                        ri[3] = viol;
                        ri[4] = recs;
                        ri[5] = (succ ? 1 : 0) | (cons ? 2 : 0) | (rec ? 4 : 0);
                        p.log_rec_f64[slot * 2 + 0] = ret;
                        p.log_rec_f64[slot * 2 + 1] = double(log_rew);
                    }
                }
            }
            if (live) {
                p.log_len[i] = epd ? 0 : log_len;
                p.log_ret[i] = epd ? 0.0 : ret;
                p.log_viol[i] = epd ? 0 : viol;
                p.log_rec[i] = epd ? 0 : recs;
            }
        }
        if (kBlockSuper && counts) {
            __syncthreads();
            if (threadIdx.x < 2 && super_acc[threadIdx.x] != 0) {
                int32_t* super_cnt = p.recovery_memory.pos_cnt + rrl_replay::super_base(p.recovery_memory.cap);
                atomicAdd(&super_cnt[threadIdx.x ? s1 : s0], super_acc[threadIdx.x]);
                super_acc[threadIdx.x] = 0;
            }
            __syncthreads();
        }
        // episode counters: one ballot + popcount per counter and wave (wave-uniform scalars, no cross-lane shuffles)
        const bool end_viol = epd & cons;
        cnt[0] += __popcll(__ballot(epd));
        cnt[1] += __popcll(__ballot(end_viol));
        cnt[2] += __popcll(__ballot(end_viol & rec));
        cnt[3] += __popcll(__ballot(end_viol & !rec));
        cnt[4] += __popcll(__ballot(epd & succ));
        cnt[5] += __popcll(__ballot(live & rec));
        cnt[6] += __popcll(__ballot(cons));
    }
    __shared__ unsigned block_cnt[kCounters];
    __shared__ double block_sum[2][kBlock / 16];
    if (threadIdx.x < kCounters) block_cnt[threadIdx.x] = 0;
    __syncthreads();
    // reward sums: DPP row rotations inside each 16-lane row (no LDS crossbar round trips), then the 16 row sums of the
    // workgroup in a fixed order
    rsum = row16_sum_f64(rsum);
    retsum = row16_sum_f64(retsum);
    if ((threadIdx.x & 15) == 0) {
        block_sum[0][threadIdx.x >> 4] = rsum;
        block_sum[1][threadIdx.x >> 4] = retsum;
    }
    if ((threadIdx.x & 63) == 0) {
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
        for (int k = 0; k < kBlock / 16; ++k) sum += block_sum[w][k];       // fixed order inside the workgroup
        if (sum != 0.0) atomicAdd(p.reward_sums + w, sum);
    } else if (blk == 0 && threadIdx.x == kCounters + 2) {
        atomicAdd(p.stats, (unsigned long long)a.n);
    }
    if constexpr (SPECULATE) {
        if (threadIdx.x == 0 && ticket == n_blk - 1) {
            p.memory.state[2] = 0;
            advance_cursors(p, mpos, msize, rpos, rsize, ctr, log_iteration);
        }
    }
}

template <class ENV, int REGIME>
__global__ __launch_bounds__(block_of(REGIME)) void step_push_kernel(StepPushArgs p) {
    step_push_body<ENV, REGIME == 0, block_of(REGIME)>(p, blockIdx.x, gridDim.x);
}

// bandwidth regime: launched behind step_push_kernel<ENV, false> on the same stream, one thread
template <class ENV>
__global__ void advance_cursors_kernel(StepPushArgs p) {
    advance_cursors(p, p.memory.state[0], p.memory.state[1], p.use_recovery_memory ? p.recovery_memory.state[0] : 0,
                    p.use_recovery_memory ? p.recovery_memory.state[1] : 0,
                    rrl::effective_counter(p.step.counter, p.step.counter_dev), p.log_state ? p.log_state[1] : 0);
}

// the same launch for S seeds (pack.hpp): seed s steps its envs on workgroups [first[s], first[s + 1]) -- its own grid as
// far as the kernel body can tell (cursor ticket)
__device__ __forceinline__ void globalize(StepPushArgs& p) {
    // copied out of device memory: the pointers are passed through the global address space (rrl_pack::to_global)
    StepArgs& e = p.step;
    rrl_pack::to_global_all(e.pos, e.action, e.noise, e.counter_dev, e.next_obs, e.obs, e.reward, e.done, e.constraint, e.success,
                            e.ep_done, e.t, e.status, p.task_action, p.recovery, p.sel_z, p.sel_rec_action, p.sel_real_out,
                            p.sel_recovery_out, p.stats, p.reward_sums, p.ep_reward, p.log_rec_i32, p.log_rec_f64, p.log_state,
                            p.log_len, p.log_ret, p.log_viol, p.log_rec);
    rrl_pack::globalize(p.sel_rec_head);
    rrl_pack::globalize(p.memory);
    rrl_pack::globalize(p.recovery_memory);
}

template <class ENV, int REGIME>
__global__ __launch_bounds__(block_of(REGIME)) void step_push_pack_kernel(const StepPushArgs* __restrict__ ps,
                                                                          rrl_pack::Idx ix) {
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    StepPushArgs p = ps[s];
    globalize(p);
    step_push_body<ENV, REGIME == 0, block_of(REGIME)>(p, local, ix.first[s + 1] - ix.first[s]);
}

template <class ENV>
__global__ void advance_cursors_pack_kernel(const StepPushArgs* __restrict__ ps) {
    StepPushArgs p = ps[blockIdx.x];      // one single-thread workgroup per seed: the argument block is wave-uniform
    globalize(p);
    advance_cursors(p, p.memory.state[0], p.memory.state[1], p.use_recovery_memory ? p.recovery_memory.state[0] : 0,
                    p.use_recovery_memory ? p.recovery_memory.state[1] : 0,
                    rrl::effective_counter(p.step.counter, p.step.counter_dev), p.log_state ? p.log_state[1] : 0);
}

// host side: the launch covers its envs, one workgroup per 256 / 1024 (n <= 2^32 - 1: at most 2^24 workgroups)
inline int grid_cover(int64_t n) {
    const int b = block_of(regime_of(n));
    const int64_t g = (n + b - 1) / b;
    return int(g < 1 ? 1 : g);
}

// launch of one seed's step: latency variant, or bandwidth variant + the one-thread cursor launch behind it
template <class ENV>
inline void launch(const StepPushArgs& p, int64_t n, hipStream_t st) {
    const dim3 grid(grid_cover(n));
    const int regime = regime_of(n);
    if (regime == 0) {
        hipLaunchKernelGGL((step_push_kernel<ENV, 0>), grid, dim3(block_of(0)), 0, st, p);
        return;
    }
    if (regime == 1) hipLaunchKernelGGL((step_push_kernel<ENV, 1>), grid, dim3(block_of(1)), 0, st, p);
    else hipLaunchKernelGGL((step_push_kernel<ENV, 2>), grid, dim3(block_of(2)), 0, st, p);
    hipLaunchKernelGGL((advance_cursors_kernel<ENV>), dim3(1), dim3(1), 0, st, p);
}
template <class ENV>
inline void launch_pack(const StepPushArgs* dev, const rrl_pack::Idx& ix, int grid, int regime, hipStream_t st) {
    if (regime == 0) {
        hipLaunchKernelGGL((step_push_pack_kernel<ENV, 0>), dim3(grid), dim3(block_of(0)), 0, st, dev, ix);
        return;
    }
    if (regime == 1) hipLaunchKernelGGL((step_push_pack_kernel<ENV, 1>), dim3(grid), dim3(block_of(1)), 0, st, dev, ix);
    else hipLaunchKernelGGL((step_push_pack_kernel<ENV, 2>), dim3(grid), dim3(block_of(2)), 0, st, dev, ix);
    hipLaunchKernelGGL((advance_cursors_pack_kernel<ENV>), dim3(ix.S), dim3(1), 0, st, dev);
}

// host side: argument block shared by the navigation and maze entry points
struct SelectIn {
    const float* z;
    int n_part;
    long long part_stride;
    float eps_safe;
    const float* rec_action;
    const rrl_policy_head_t* rec_head;
    float* real_out;
    uint8_t* recovery_out;
};

inline int fill_args(StepPushArgs& p, int64_t n, double* pos, int32_t* t, float* obs, const float* task_action,
                     int ld_task, const float* real_action, const uint8_t* recovery, const SelectIn* sel, uint64_t seed,
                     uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, int32_t horizon, int auto_reset,
                     float reward_penalty, int push_real_action, const rrl_replay_t* memory,
                     const rrl_replay_t* recovery_memory, float* next_obs, float* reward, uint8_t* done,
                     uint8_t* constraint, uint8_t* success, uint8_t* ep_done, uint64_t* stats, double* reward_sums,
                     float* ep_reward, uint16_t* status = nullptr) {
    if (n < 0 || n > 0xffffffffLL) return RRL_ERANGE;
    if (status && (horizon < 1 || horizon > 4095)) return RRL_ERANGE;      // 12 bits of step count
    if (!pos || !(t || status) || !obs || !task_action || !memory || !stats || !reward_sums || !ep_reward || ld_task < 2 ||
        (ld_task & 1))
        return RRL_EINVAL;
    if (sel ? (!sel->z || (!sel->rec_action && !sel->rec_head) || !sel->real_out || !sel->recovery_out ||
               sel->n_part <= 0 || sel->n_part > 4) : !real_action)
        return RRL_EINVAL;
    if (sel && !sel->rec_action) {
        const rrl_policy_head_t& h = *sel->rec_head;
        if (h.kind != RRL_HEAD_STOCH || !h.head || !h.scale || !h.bias || !h.log_std || h.n_part <= 0 || h.n_part > 4)
            return RRL_EINVAL;
    }
    if (memory->pinned < 0 || n > memory->cap - memory->pinned ||
        (recovery_memory && (recovery_memory->pinned < 0 || n > recovery_memory->cap - recovery_memory->pinned)))
        return RRL_ERANGE;
    p.step = StepArgs{n, (double2*)pos, (const float2*)real_action, nullptr, seed, counter, counter_dev,
                      counter_inc, (float2*)next_obs, (float2*)obs, reward, done, constraint, success, ep_done,
                      t, horizon, auto_reset, status};
    p.task_action = task_action;
    p.ld_task = ld_task;
    p.recovery = recovery;
    p.sel_z = sel ? sel->z : nullptr;
    p.sel_np = sel ? sel->n_part : 1;
    p.sel_ps = sel ? sel->part_stride : 0;
    p.sel_eps = sel ? sel->eps_safe : 0.f;
    p.sel_rec_action = sel ? (const float2*)sel->rec_action : nullptr;
    p.sel_rec_head = (sel && !sel->rec_action) ? *sel->rec_head : rrl_policy_head_t{};
    p.sel_real_out = sel ? (float2*)sel->real_out : nullptr;
    p.sel_recovery_out = sel ? sel->recovery_out : nullptr;
    p.reward_penalty = reward_penalty;
    p.push_real_action = push_real_action;
    p.memory = *memory;
    p.use_recovery_memory = recovery_memory != nullptr;
    p.recovery_memory = recovery_memory ? *recovery_memory : *memory;
    p.stats = (unsigned long long*)stats;
    p.reward_sums = reward_sums;
    p.ep_reward = ep_reward;
    p.log_rec_i32 = nullptr; p.log_rec_f64 = nullptr; p.log_cap = 0; p.log_state = nullptr;
    p.log_len = nullptr; p.log_ret = nullptr; p.log_viol = p.log_rec = nullptr;
    return RRL_OK;
}

// rrl_step_push_t (the struct entry points rrl_nav_step_push_x / rrl_maze_step_push_x) -> kernel arguments
inline int fill_step(StepPushArgs& p, const rrl_step_push_t* a);
inline int fill_log(StepPushArgs& p, const rrl_step_push_t* a, int rc) {
    if (rc != RRL_OK || !a->log_state) return rc;
    if (!a->log_rec_i32 || !a->log_rec_f64 || a->log_cap <= 0 || !a->log_len || !a->log_ret || !a->log_viol || !a->log_rec)
        return RRL_EINVAL;
    p.log_rec_i32 = a->log_rec_i32; p.log_rec_f64 = a->log_rec_f64; p.log_cap = a->log_cap; p.log_state = a->log_state;
    p.log_len = a->log_len; p.log_ret = a->log_ret; p.log_viol = a->log_viol; p.log_rec = a->log_rec;
    return RRL_OK;
}

inline int fill_args(StepPushArgs& p, const rrl_step_push_t* a) {
    if (!a) return RRL_EINVAL;
    return fill_log(p, a, fill_step(p, a));
}

inline int fill_step(StepPushArgs& p, const rrl_step_push_t* a) {
    if (a->sel_z) {
        const SelectIn sel{a->sel_z, a->sel_n_part, a->sel_part_stride, a->sel_eps_safe, a->sel_rec_action, a->sel_rec_head,
                           a->real_action_out, a->recovery_out};
        return fill_args(p, a->n, a->pos, a->t, a->obs, a->task_action, a->ld_task, nullptr, nullptr, &sel, a->seed,
                         a->counter, a->counter_dev, a->counter_inc, a->horizon, a->auto_reset, a->reward_penalty,
                         a->push_real_action, a->memory, a->recovery_memory, a->next_obs, a->reward, a->done, a->constraint,
                         a->success, a->ep_done, a->stats, a->reward_sums, a->ep_reward, a->status);
    }
    return fill_args(p, a->n, a->pos, a->t, a->obs, a->task_action, a->ld_task, a->real_action, a->recovery, nullptr, a->seed,
                     a->counter, a->counter_dev, a->counter_inc, a->horizon, a->auto_reset, a->reward_penalty,
                     a->push_real_action, a->memory, a->recovery_memory, a->next_obs, a->reward, a->done, a->constraint,
                     a->success, a->ep_done, a->stats, a->reward_sums, a->ep_reward, a->status);
}

}  // namespace rrl_step
