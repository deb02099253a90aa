"""Experiment driver (reference: recovery_rl/experiment.py:42-577).

`Experiment(exp_cfg).run()` keeps the reference's surface: log-dir naming, args.pkl,
run_stats.pkl, the printed lines, the update -> act -> step -> push order of one loop
iteration and the counters.  Two loop bodies share the same building blocks:

  * num_envs == 1: `get_train_rollout` / `get_test_rollout`, episode by episode, exactly the
    reference's event order (experiment.py:379-491, 493-538);
  * num_envs  > 1: `VectorLoop.vector_step`, all envs in lock-step with device-side counters
    and no host synchronisation; once in steady state the whole iteration (replay sample ->
    SAC update -> Q_risk update -> policy/Q_risk forward -> env step -> replay push ->
    counters) is captured in ONE hipGraph and replayed.

Vectorisation semantics (SURVEY.md section 8a "Vectorisation semantics"): every env has its own
step count and Philox sub-stream and auto-resets where done or t == horizon; `start_steps`
counts env-steps over all envs; one iteration performs `updates_per_step` updates, so the
update-to-data ratio is updates_per_step / num_envs; episode-keyed events use the global
completed-episode counter.
"""
import datetime
import itertools
import os
import os.path as osp
import pickle
import time

import numpy as np
import torch

from . import distributed as dist_utils
from .env import make_env, make_vec_env, register_env
from .replay_memory import ConstraintReplayMemory, ReplayMemory
from .sac import SAC
from .fused import mlp3_supported
from .utils import linear_schedule, trace_range

# order of the device-side counter vector (also the RCCL-aggregated metric vector)
STAT_KEYS = ("env_steps", "episodes", "num_viols", "viol_and_recovery", "viol_and_no_recovery",
             "num_successes", "recovery_steps", "constraint_steps", "sac_updates", "qrisk_updates")


def uses_constraint_buffer(cfg):
    return cfg.use_recovery or cfg.DGD_constraints or cfg.RCPO          # experiment.py:442


def uses_mb_recovery(cfg):
    return cfg.use_recovery and not (cfg.MF_recovery or cfg.Q_sampling_recovery)   # :92-94


class VectorLoop:
    """One lock-step iteration over all envs, device-resident (counterpart of the body of
    experiment.py:396-452 for N envs)."""

    def __init__(self, cfg, env, agent, memory, recovery_memory, recovery_policy=None,
                 nu_schedule=None):
        self.cfg, self.env, self.agent = cfg, env, agent
        self.memory, self.recovery_memory = memory, recovery_memory
        self.recovery_policy = recovery_policy
        self.nu_schedule = nu_schedule or (lambda ep: cfg.nu)
        self.n = env.num_envs
        dev = env.device
        self.device = dev
        self.obs = None
        self.stats = torch.zeros(len(STAT_KEYS), dtype=torch.int64, device=dev)
        self.reward_sums = torch.zeros(2, dtype=torch.float64, device=dev)   # rewards, finished-episode returns
        self.ep_reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.total_numsteps = 0           # host mirror: every iteration adds num_envs
        self.updates = 0
        self.num_constraint_violations = 0  # offline violations pushed during pre-training
        self.graph = None
        self.host_updates = [0, 0]        # SAC / Q_risk update counts are known on the host
        self._graph_updates = (0, 0)
        self._actor = None
        self.carry_actor = os.environ.get("RRL_CARRY_ACTOR", "1") != "0"
        # vectorisation rule 5: at N > 1 an env's CEM warm start does not survive its episode (MPC.forget_plans)
        self.forget_plans = bool(recovery_policy is not None and self.n > 1 and hasattr(recovery_policy, "forget_plans")
                                 and not getattr(cfg, "keep_plan_warm_start", False))
        self._one = torch.ones((), dtype=torch.int64, device=dev)
        # the random-action phase (experiment.py:559-560) draws from the loop's OWN generator: a seed's trajectory does not
        # depend on what else shares the process (S seeds per GPU, packed.py), only on its seed
        self.action_rng = torch.Generator(device=dev)
        self.action_rng.manual_seed(int(cfg.seed) + 0x5EED)
        self.episode_log = None           # optional EpisodeLog (per-episode records for run_stats)
        # the fused step also writes env.next_obs / reward / done / constraint / success / ep_done (off: nobody reads them
        # in the steady-state graph; the episode log and the online ensemble re-fit switch them on by themselves)
        self.step_outputs = False
        if self.n > 1 and hasattr(recovery_memory, "clamp_stratified"):
            # N envs overwrite the ring in capacity / N iterations: once the policy avoids violations the positive
            # class can drop below int(B * pos_fraction) rows, where the one-env reference would abort
            recovery_memory.clamp_stratified = True

    # -- pieces --------------------------------------------------------------------------------
    def start(self):
        self.obs = self.env.reset()
        return self.obs

    def do_updates(self, i_episode=1, online_qrisk=True, rider=None):
        """experiment.py:397-416.  The caller has checked len(memory) > batch_size.  rider: see actor_rider()."""
        cfg = self.cfg
        fast = getattr(self.agent, "fast", None)
        qr = self.agent.safety_critic
        grouped = (fast is not None and fast.grouped and fast.sync_world == 1 and cfg.batch_size == fast.B
                   and hasattr(self.memory, "draw_desc")
                   and (not online_qrisk or qr.clamp_batch_size(cfg.batch_size, len(self.recovery_memory)) == fast.B))
        for u in range(cfg.updates_per_step):
            if grouped:
                # both replay draws + the policy noise in one launch, independent kernels of the two updates grouped
                # (fast_update.FastUpdater.update_pair): same results as the two calls below, ~30 % fewer launches
                with trace_range("sample+sac_update+qrisk_update"):
                    fast.update_pair(self.memory, self.recovery_memory if online_qrisk else None,
                                     rider=rider if u == cfg.updates_per_step - 1 else None)
                self.host_updates[0] += 1
                if online_qrisk:
                    qr.updates += 1
                    qr.last_losses = (fast.losses[4], fast.losses[5], fast.losses[6] if qr.MF_recovery else None)
                    self.host_updates[1] += 1
                self.updates += 1
                continue
            assert rider is None
            with trace_range("sample+sac_update"):
                self.agent.update_parameters(self.memory, cfg.batch_size, self.updates,
                                             safety_critic=self.agent.safety_critic,
                                             nu=self.nu_schedule(i_episode))
            self.host_updates[0] += 1
            if online_qrisk:
                with trace_range("sample+qrisk_update"):
                    self.agent.safety_critic.update_parameters(memory=self.recovery_memory,
                                                               policy=self.agent.policy,
                                                               batch_size=cfg.batch_size, plot=0)
                self.host_updates[1] += 1
            self.updates += 1

    def act(self, obs, random_actions=False, train=True):
        """Batched get_action (experiment.py:546-577): (task action, executed action, recovery)."""
        cfg = self.cfg
        fast = getattr(self.agent, "fast", None)
        if (fast is not None and train and not random_actions and obs.shape[0] == self.n
                and (not cfg.use_recovery or cfg.MF_recovery)):
            if self._actor is None:
                from .fast_update import FastActor
                self._actor = FastActor(fast, self.n)
            # recovery stays uint8 (what the step kernel reads): no dtype round trip in the captured graph
            return self._actor.act(obs, cfg.eps_safe, cfg.use_recovery, cfg.MF_recovery,
                                   defer_select=fast.grouped and obs is self.env.obs and self._can_fuse_step())
        if random_actions:
            # one env: action_space.sample() semantics on the global generator, as the reference (experiment.py:560)
            action = self.env.sample_actions() if self.n == 1 else self.env.sample_actions(generator=self.action_rng)
        elif (fast is not None and train and obs.shape[0] == self.n and uses_mb_recovery(cfg)
              and mlp3_supported(fast.policy.H, 2, 4) and mlp3_supported(fast.qrisk.H, 4, 1)):
            # model-based recovery: task action and the Q_risk gate on the fused kernels (rows the planner does not need
            # stay where they are); the planner acts for the gated rows only
            if self._actor is None:
                from .fast_update import FastActor
                self._actor = FastActor(fast, self.n)
            action, recovery = self._actor.act_gate(obs, cfg.eps_safe)
            rec_action = self.recovery_policy.act(obs, 0, mask=recovery)
            real_action = torch.where(recovery.bool().unsqueeze(1), rec_action, action)
            return action, real_action, recovery
        else:
            eps = self._policy_eps(self.agent.policy, obs) if self.n > 1 else None
            action = self.agent.select_action(obs, eval=not train) if eps is None else \
                self.agent.select_action(obs, eval=not train, eps=eps)
        if not cfg.use_recovery:
            return action, action, None
        risk = self.agent.safety_critic.get_value(obs, action).squeeze(1)
        recovery = risk > cfg.eps_safe                                      # :568
        if self.n == 1 and not bool(recovery[0].item()):
            # one env: the recovery policy is queried only when the gate fires, as in the reference
            return action, action.clone(), recovery
        if cfg.MF_recovery or cfg.Q_sampling_recovery:
            qr = self.agent.safety_critic
            eps = self._policy_eps(qr.policy, obs) if (cfg.MF_recovery and self.n > 1) else None
            rec_action = qr.select_action(obs) if eps is None else qr.select_action(obs, eps=eps)
        else:
            rec_action = self.recovery_policy.act(obs, 0, mask=recovery)
        real_action = torch.where(recovery.unsqueeze(1), rec_action, action)
        return action, real_action, recovery

    def _policy_eps(self, policy, obs):
        """N(0,1) draws for a policy sampled through its torch module in the lock-step loop (the random-action phase, the
        configurations outside the fused path): from the loop's own generator, so that a seed's trajectory does not depend on
        what else shares the process (seed packing covers the fused path only).  One env, or no fused path: None = torch's
        global generator, as the reference."""
        if self.n == 1 or getattr(self.agent, "fast", None) is None:
            return None      # configurations whose steady state samples through the modules are captured in a hipGraph,
                             # where torch advances its GLOBAL generator's Philox offset per replay (a private one is not)
        from .model import DeterministicPolicy
        shape = (policy.num_actions,) if isinstance(policy, DeterministicPolicy) else (obs.shape[0], 2)
        return torch.randn(*shape, device=self.device, generator=self.action_rng)

    def step_and_store(self, action, real_action, recovery):
        """env.step + reward penalty + mask + pushes + counters (experiment.py:420-461)."""
        cfg = self.cfg
        real_action = real_action.contiguous()
        if self._can_fuse_step():
            pending = self._actor is not None and getattr(self._actor, "pending_select", None) is not None
            return self._fused_step(action if pending else action.contiguous(), real_action, recovery)
        obs, reward, done, info = self.env.step(real_action)
        state, next_state = info["state"], info["next_state"]
        constraint_f = info["constraint"].to(torch.float32)
        push_reward = reward - cfg.constraint_reward_penalty * constraint_f if \
            cfg.constraint_reward_penalty else reward                       # :431-432
        mask = 1.0 - done.to(torch.float32)                                 # :434 (before the horizon)
        task_action = real_action if cfg.disable_action_relabeling else action.contiguous()
        self.memory.push(state, task_action, push_reward, next_state, mask)
        if uses_constraint_buffer(cfg):
            self.recovery_memory.push(state, real_action, constraint_f, next_state, mask)
            if cfg.add_both_transitions and recovery is not None:           # :446-448
                self.memory.push(state, real_action, push_reward, next_state, mask,
                                 valid=recovery.to(torch.uint8).contiguous())
        # counters (experiment.py:66-74,455-461), all on the device
        ep_done = info["ep_done"].bool()
        cons = info["constraint"].bool()
        end_viol = ep_done & cons
        rec = recovery.bool() if recovery is not None else torch.zeros_like(cons)
        st = self.stats
        st[0] += self.n
        st[1] += ep_done.sum()
        st[2] += end_viol.sum()
        st[3] += (end_viol & rec).sum()
        st[4] += (end_viol & ~rec).sum()
        st[5] += (ep_done & info["success"].bool()).sum()
        st[6] += rec.sum()
        st[7] += cons.sum()
        self.reward_sums[0] += reward.sum(dtype=torch.float64)
        self.ep_reward += reward
        self.reward_sums[1] += torch.where(ep_done, self.ep_reward, torch.zeros_like(reward)).sum(dtype=torch.float64)
        self.ep_reward *= (~ep_done).to(torch.float32)
        if self.episode_log is not None:
            self.episode_log.append(reward, info["constraint"], info["success"], info["ep_done"], recovery)
        if self.forget_plans:
            self.recovery_policy.forget_plans(ep_done)
        self.obs = obs
        self.total_numsteps += self.n
        return obs

    def _can_fuse_step(self):
        from .env.maze import MazeVecEnv
        from .env.navigation import NavigationVecEnv
        return (isinstance(self.env, (NavigationVecEnv, MazeVecEnv)) and self.env.auto_reset
                and not self.cfg.add_both_transitions and not getattr(self.cfg, "no_fused_step", False))

    def _fused_step(self, action, real_action, recovery):
        """env step + both replay pushes + counters in ONE launch (rrl_nav_step_push_x / rrl_maze_step_push_x).

        Per-env outputs of the step (env.next_obs, env.reward, the four flags) are written only when something reads them:
        `step_outputs` (callers that look at the env's arrays after a step) and the online ensemble re-fit; the episode log
        is advanced by the same launch from the step's registers (rrl_step_push_t.log_*).
        Without a reader the loop runs on the COMPACT env state: one u16 status word per env instead of the i32 step count
        and four u8 flags, and the stored state taken from `pos` instead of the observation array."""
        import ctypes as C
        from . import _lib
        cfg, env, mem, rmem = self.cfg, self.env, self.memory, self.recovery_memory
        rec_u8 = None
        if recovery is not None:
            rec_u8 = recovery if recovery.dtype == torch.uint8 else recovery.to(torch.uint8)
        use_rmem = uses_constraint_buffer(cfg)
        keep = self.step_outputs or self.recovery_policy is not None
        p = _lib.ptr
        out = (lambda t: p(t)) if keep else (lambda t: None)
        a = _lib.rrl_step_push_t()
        a.n, a.pos, a.obs = self.n, p(env.pos), p(env.obs)
        if keep:
            env.use_arrays()
            a.t, a.status = p(env.t), None
        else:
            a.t, a.status = None, p(env.use_status())
        a.seed, a.counter, a.counter_dev, a.counter_inc = env.seed_value, 0, p(env.tick), 1
        a.horizon, a.auto_reset = env.horizon, 1
        a.reward_penalty = float(cfg.constraint_reward_penalty)
        a.push_real_action = int(bool(cfg.disable_action_relabeling))
        a.memory = C.pointer(mem._desc)
        a.recovery_memory = C.pointer(rmem._desc) if use_rmem else None
        a.next_obs, a.reward = out(env.next_obs), out(env.reward)
        a.done, a.constraint, a.success, a.ep_done = out(env.done), out(env.constraint), out(env.success), out(env.ep_done)
        a.stats, a.reward_sums, a.ep_reward = p(self.stats), p(self.reward_sums), p(self.ep_reward)
        if self.episode_log is not None:
            self.episode_log.attach(a)
        select = getattr(self._actor, "pending_select", None) if self._actor is not None else None
        if select is not None:
            # the recovery gate runs inside the step kernel: `action` is the strided task action, `real_action` and
            # `rec_u8` are written by this launch
            self._actor.pending_select = None
            zq, z_parts, z_stride, eps_safe, rec_action, rec_head = select
            assert action.stride(1) == 1 and rec_u8 is not None and rec_u8.dtype == torch.uint8
            a.task_action, a.ld_task = p(action), action.stride(0)
            a.sel_z, a.sel_n_part, a.sel_part_stride, a.sel_eps_safe = p(zq), z_parts, z_stride, eps_safe
            a.sel_rec_action = p(rec_action)
            a.sel_rec_head = C.pointer(rec_head) if rec_head is not None else None
            a.real_action_out, a.recovery_out = p(real_action), p(rec_u8)
        else:
            if self.recovery_policy is not None:
                # the online ensemble re-fit reads (state, clipped action, next state) of this step (experiment.py:464-480
                # collects them per episode): env.step() writes these buffers, the fused kernel does not
                env.prev_obs.copy_(env.obs)
                hi = float(env.action_space.high[0])
                torch.clamp(real_action, -hi, hi, out=env.action_clipped)
            a.task_action, a.ld_task = p(action), 2
            a.real_action, a.recovery = p(real_action), p(rec_u8)
        self._step_args = a          # keeps the ctypes pointers alive until the launch has been issued
        from .fast_update import record
        record("step", env.env_name, getattr(env, "kind", -1), a)
        if env.env_name == "maze":
            rc = env.lib.rrl_maze_step_push_x(C.byref(a), _lib.current_stream())
        else:
            rc = env.lib.rrl_nav_step_push_x(env.kind, C.byref(a), _lib.current_stream())
        _lib.check(rc, "rrl_step_push_x")
        mem._len = min(mem._len + self.n, mem.capacity)
        if use_rmem:
            rmem._len = min(rmem._len + self.n, rmem.capacity)
        if self.forget_plans:
            self.recovery_policy.forget_plans(env.ep_done)        # vectorisation rule 5 (the step wrote the episode-end flags)
        self.obs = env.obs
        self.total_numsteps += self.n
        return env.obs

    # -- whole iteration -----------------------------------------------------------------------
    def actor_rider(self, do_update, random_actions, online_qrisk):
        """(FastActor, obs) when the acting pass that follows this iteration's updates can take its task-policy and Q_risk
        forwards along in the Q_risk update's launches (FastUpdater.qrisk_update_grouped): the grouped fused updates with an
        online Q_risk update, the fused acting pass with the gate in the step kernel, on the loop's own observations.
        `carry_actor = False` (or RRL_CARRY_ACTOR=0) keeps the acting pass's own two launches (A/B and the bit-identity test)."""
        cfg = self.cfg
        fast = getattr(self.agent, "fast", None)
        if not (self.carry_actor and do_update and online_qrisk and not random_actions and fast is not None
                and cfg.use_recovery and cfg.MF_recovery and fast.can_carry_actor() and fast.sync_world == 1
                and cfg.batch_size == fast.B and hasattr(self.memory, "draw_desc")
                and self.agent.safety_critic.clamp_batch_size(cfg.batch_size, len(self.recovery_memory)) == fast.B
                and self.obs is self.env.obs and self.obs.shape[0] == self.n and self._can_fuse_step()):
            return None
        if self._actor is None:
            from .fast_update import FastActor
            self._actor = FastActor(fast, self.n)
        return (self._actor, self.obs) if self._actor.qr.split else None

    def vector_step(self, do_update=True, random_actions=False, online_qrisk=True, i_episode=1):
        if do_update:
            self.do_updates(i_episode, online_qrisk, rider=self.actor_rider(do_update, random_actions, online_qrisk))
        with trace_range("act"):
            action, real_action, recovery = self.act(self.obs, random_actions)
        self._last_recovery, self._last_real_action = recovery, real_action
        with trace_range("env_step+push"):
            return self.step_and_store(action, real_action, recovery)

    def capture(self, online_qrisk=True, warmup=3, iters=None):
        """Capture the steady-state iteration (updates + act + step + push + counters) into one
        hipGraph.  Host-side bookkeeping (total_numsteps, updates) is advanced by replay().  The `warmup`
        iterations before the capture are real iterations; their number is returned."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.vector_step(True, False, online_qrisk)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        saved = (self.total_numsteps, self.updates, list(self.host_updates))
        qr_updates = self.agent.safety_critic.updates
        lens = (self.memory._len, self.recovery_memory._len)
        def restore():
            self.total_numsteps, self.updates, self.host_updates = saved[0], saved[1], list(saved[2])
            self.agent.safety_critic.updates = qr_updates      # the captured call only recorded launches
            self.memory._len, self.recovery_memory._len = lens

        def record(iterations):
            g = torch.cuda.CUDAGraph()
            # configurations that sample a policy through its torch module (e.g. --Q_sampling_recovery with N > 1) draw from
            # the loop's own generator inside the graph: registered, its Philox offset advances per replay like the global one's
            g.register_generator_state(self.action_rng)
            with torch.cuda.graph(g):
                for _ in range(iterations):
                    self.vector_step(True, False, online_qrisk)
            return g
        g = record(1)
        # which of the env's two state representations (status word / t + flag arrays) the captured kernels read and write
        self._graph_status_live = getattr(self.env, "_status_live", None)
        self._graph_updates = (self.host_updates[0] - saved[2][0], self.host_updates[1] - saved[2][1])
        # rows one replay appends to each ring (host mirrors of the device-side sizes)
        self._graph_rows = (self.n, self.n if uses_constraint_buffer(self.cfg) else 0)
        restore()
        self.graph = g
        # what the last captured iteration left for the host to look at (graph-owned tensors: observation, executed action,
        # recovery mask), per graph: replay() / advance() point the loop's attributes at the ones of the graph that ran last
        self._graph_out = (self.obs, self._last_recovery, self._last_real_action)
        # ... and `graph_iterations` iterations as ONE graph: between two graphs the queue idles ~2.7 us (0.1638 -> 0.1613 ms per
        # iteration with four per graph, 0.1611 with eight); advance() uses it wherever that many iterations fit before the
        # caller's next host-side decision (log point, end of a timed block), so nothing observable moves
        self.graph_many, self.graph_many_iters = None, max(1, int(iters if iters is not None else
                                                                   getattr(self.cfg, "graph_iterations", 4)))
        if self.graph_many_iters > 1:
            self.graph_many = record(self.graph_many_iters)
            restore()
            self._graph_many_out = (self.obs, self._last_recovery, self._last_real_action)
            self.obs, self._last_recovery, self._last_real_action = self._graph_out
        return warmup

    def replay(self):
        if getattr(self.env, "_status_live", None) != self._graph_status_live:
            # an eager env.reset() / env.step() / checkpoint load since the capture switched the env to the other
            # representation: the graph would keep stepping the stale one
            raise RuntimeError("the env's live state representation changed since the graph was captured "
                               "(status word live: %r at capture, %r now); capture again"
                               % (self._graph_status_live, getattr(self.env, "_status_live", None)))
        self.graph.replay()
        self.obs, self._last_recovery, self._last_real_action = self._graph_out
        self._advance_mirrors(1)
        return self.obs

    def _advance_mirrors(self, iterations):
        self.total_numsteps += self.n * iterations
        self.host_updates[0] += self._graph_updates[0] * iterations
        self.host_updates[1] += self._graph_updates[1] * iterations
        self.updates += self.cfg.updates_per_step * iterations
        self.agent.safety_critic.updates += self._graph_updates[1] * iterations
        for mem, rows in zip((self.memory, self.recovery_memory), self._graph_rows):
            mem._len = min(mem._len + rows * iterations, mem.capacity)
        if self.cfg.add_both_transitions:
            self.memory._len_exact = False                 # masked pushes: the size is known on the device

    def advance(self, iterations):
        """`iterations` steady-state iterations from the captured graphs: the many-iteration graph while that many remain, single
        iterations for the rest.  The same launches in the same order as `iterations` calls of replay()."""
        k = self.graph_many_iters if self.graph_many is not None else 0
        while k > 1 and iterations >= k:
            if getattr(self.env, "_status_live", None) != self._graph_status_live:
                break                                      # replay() raises with the explanation
            self.graph_many.replay()
            self.obs, self._last_recovery, self._last_real_action = self._graph_many_out
            self._advance_mirrors(k)
            iterations -= k
        for _ in range(iterations):
            self.replay()
        return self.obs

    def read_stats(self):
        """One device->host copy of the counter vector (+ the samplers' error flags: a draw the reference would
        abort with ValueError -- random.sample on too small a population, replay_memory.py:28,61-66 -- leaves the
        batch unwritten on the device, so the run must stop here instead of training on stale rows)."""
        mems = [m for m in (self.memory, self.recovery_memory) if hasattr(m, "state") and torch.is_tensor(m.state)]
        if len(mems) == 2 and self.stats.is_cuda:
            # ONE device->host copy for the counters, the two error flags and the two f64 sums (four synchronising copies
            # per log point before: each one is a pipeline bubble of the replayed graph)
            packed = torch.cat([self.stats.to(torch.int64), mems[0].state[3:4], mems[1].state[3:4],
                                self.reward_sums.view(torch.int64)]).cpu()
            n = self.stats.numel()
            for m, code in zip(mems, packed[n:n + 2].tolist()):
                if code:
                    m.check_error()              # raises with the reference's message
            vals = packed[:n].tolist()
            sums = packed[n + 2:].view(torch.float64).tolist()
        else:
            self.memory.check_error()
            self.recovery_memory.check_error()
            vals = self.stats.cpu().tolist()
            sums = self.reward_sums.cpu().tolist()
        vals[8], vals[9] = self.host_updates
        out = dict(zip(STAT_KEYS, vals))
        out["reward_sum"], out["episode_return_sum"] = float(sums[0]), float(sums[1])
        return out


MAX_COVER_ROWS = 1 << 28          # 8.6 GB per buffer at 32 B per row: far inside 288 GB of HBM
ROW_BYTES = 32                    # s, a, s2 f32[2] + r, m f32


def cover_rows_limit(device=None, seeds_per_gpu=1):
    """Rows per buffer the automatic growth of rule 4 may ask for: MAX_COVER_ROWS, and never more than an eighth of the
    device's TOTAL memory for each of the two buffers of each seed packed on it (a property of the device and the command
    line, not of what happens to be free: the same run gets the same capacities on every rank, for every packed seed and
    on a rerun)."""
    limit = MAX_COVER_ROWS
    if device is not None and torch.cuda.is_available() and torch.device(device).type == "cuda":
        total = torch.cuda.get_device_properties(torch.device(device)).total_memory
        limit = min(limit, int(total // 8 // max(int(seeds_per_gpu), 1) // ROW_BYTES))
    return limit


def replay_capacities(cfg, device=None):
    """(task buffer, safety buffer) capacities of a run.  Vectorisation rule 4: in the reference a buffer never wraps within a
    run -- its defaults are replay_size = safe_replay_size = num_steps = 1e6 and a run stops at num_steps env-steps
    (arg_utils.py, experiment.py:375) -- so both buffers hold a run's WHOLE history.  A lock-step run of N envs is given a step
    budget N times larger; with the default rings it would keep only the last capacity / N iterations (245 at 4096 envs) and its
    critics forget everything older: measured on config 4, the learning seeds then show violation bursts every ~250 iterations
    from iteration ~1 000 on, and none with buffers that cover the run (profiles/round4_learning_vec4096_config4_*.json).  So
    at N > 1 a capacity below the run's step budget is raised to it (+ the demonstrations, + one iteration of head-room), the
    relation the reference's defaults have.  The stratified sampler's count tables bound the safety buffer at 2^21 rows with
    --pos_fraction; `--keep_replay_size` keeps the rings as given."""
    cap, safe_cap = int(cfg.replay_size), int(cfg.safe_replay_size)
    n = int(getattr(cfg, "num_envs", 1))
    if n <= 1 or getattr(cfg, "keep_replay_size", False) or cfg.num_steps <= min(cap, safe_cap):
        return cap, safe_cap                 # (the reference's defaults, num_steps == both capacities, stay as they are)
    limit = cover_rows_limit(device, getattr(cfg, "seeds_per_gpu", 1))
    if limit < cfg.num_steps:
        # the whole-history relation is NOT met: say so where it is decided (and in vector_rules: Experiment.__init__)
        print("WARNING: vectorisation rule 4 not met: --num_steps %d exceeds the %d rows per buffer this device allows; the "
              "buffers will wrap and the critics forget the oldest %d env-steps" % (cfg.num_steps, limit, cfg.num_steps - limit))
    steps = int(min(cfg.num_steps, limit)) + 2 * n
    cap = max(cap, steps)
    safe = steps + int(cfg.num_unsafe_transitions)
    if cfg.pos_fraction >= 0:
        safe = min(safe, 1 << 21)
    safe_cap = max(safe_cap, safe)
    return cap, safe_cap


class Experiment:
    def __init__(self, exp_cfg, rank=0, world_size=1):
        self.exp_cfg = exp_cfg
        self.rank, self.world_size = rank, world_size
        if not hasattr(exp_cfg, "num_envs"):
            exp_cfg.num_envs = 1
        self.env_shard = world_size > 1 and getattr(exp_cfg, "dp_mode", "replicas") == "env_shard"
        if getattr(exp_cfg, "dp_mode", "replicas") not in ("replicas", "env_shard"):
            raise ValueError("--dp_mode must be 'replicas' or 'env_shard'")
        if self.env_shard:
            from .fast_update import fast_path_supported
            if exp_cfg.num_envs < 2 or not fast_path_supported(exp_cfg) or getattr(exp_cfg, "no_fast_path", False) \
                    or exp_cfg.batch_size % world_size or uses_mb_recovery(exp_cfg):
                raise ValueError("--dp_mode env_shard needs the lock-step loop (--num_envs > 1), a configuration of "
                                 "the fused update path, and batch_size divisible by the number of ranks")
            # ONE learner with the reference's batch: every rank contributes batch_size / world rows
            exp_cfg.global_batch_size = exp_cfg.batch_size
            exp_cfg.batch_size = exp_cfg.batch_size // world_size
        # logging setup (experiment.py:46-55)
        self.logdir = os.path.join(
            exp_cfg.logdir, '{}_SAC_{}_{}_{}'.format(
                datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"), exp_cfg.env_name,
                exp_cfg.policy, exp_cfg.logdir_suffix))
        if world_size > 1:
            self.logdir += "_seed%d" % exp_cfg.seed      # one directory per rank (= per seed), as the reference's runs
        if not os.path.exists(self.logdir):
            os.makedirs(self.logdir)
        print("LOGDIR: ", self.logdir)
        pickle.dump(exp_cfg, open(os.path.join(self.logdir, "args.pkl"), "wb"))

        self.experiment_setup()

        dev = self.device
        cap, safe_cap = replay_capacities(exp_cfg, device=dev)
        if getattr(exp_cfg, "resume", ""):
            # a resumed run keeps the buffers it was written with (a larger --num_steps, the usual way to continue, or a
            # checkpoint from before rule 4 would otherwise ask for other capacities and be refused); they only GROW, and only
            # while the checkpoint's ring has not wrapped (checkpoint.load_replay_state)
            from . import checkpoint
            have = checkpoint.peek_capacities(exp_cfg.resume)
            cap = cap if (cap > have["memory"][0] and not have["memory"][1]) else have["memory"][0]
            safe_cap = safe_cap if (safe_cap > have["recovery_memory"][0] and not have["recovery_memory"][1]) \
                else have["recovery_memory"][0]
        if (cap, safe_cap) != (exp_cfg.replay_size, exp_cfg.safe_replay_size):
            print("Replay capacities set to cover the run: %d / %d rows = %.0f + %.0f MB of device memory (--replay_size %d, "
                  "--safe_replay_size %d, --num_steps %d; --keep_replay_size keeps the rings)"
                  % (cap, safe_cap, cap * ROW_BYTES / 1e6, safe_cap * ROW_BYTES / 1e6, exp_cfg.replay_size,
                     exp_cfg.safe_replay_size, exp_cfg.num_steps))
        self.memory = ReplayMemory(cap, exp_cfg.seed, device=dev)
        self.recovery_memory = ConstraintReplayMemory(safe_cap, exp_cfg.seed, device=dev)
        self.all_ep_data = []
        self.vector_rules = {"demo_share": 0.0, "pinned_demonstrations": 0, "replay_capacities": (cap, safe_cap),
                             "cover_rows_limit": cover_rows_limit(dev, getattr(exp_cfg, "seeds_per_gpu", 1)),
                             "buffers_cover_the_run": bool(exp_cfg.num_envs <= 1 or min(cap, safe_cap) >= exp_cfg.num_steps
                                                           or exp_cfg.pos_fraction >= 0),
                             "plan_warm_start": "kept across episodes (the reference)" if (
                                 exp_cfg.num_envs <= 1 or getattr(exp_cfg, "keep_plan_warm_start", False))
                             else "per episode (rule 5)"}

        self.total_numsteps = 0
        self.updates = 0
        self.num_constraint_violations = 0
        self.num_unsafe_transitions = 0
        self.num_viols = 0
        self.num_successes = 0
        self.viol_and_recovery = 0
        self.viol_and_no_recovery = 0
        self.task_demos = exp_cfg.task_demos
        self.constraint_demo_data, self.task_demo_data = self.get_offline_data()

        if exp_cfg.nu_schedule:                                             # :81-87
            self.nu_schedule = linear_schedule(exp_cfg.nu_start, exp_cfg.nu_end, exp_cfg.num_eps)
        else:
            self.nu_schedule = linear_schedule(exp_cfg.nu, exp_cfg.nu, 0)
        from .fast_update import fast_path_supported
        if fast_path_supported(exp_cfg) and not getattr(exp_cfg, "no_fast_path", False):
            self.agent.enable_fast_path(exp_cfg.batch_size)
        if self.env_shard:
            self.agent.fast.enable_grad_sync(world_size)
        self.loop = VectorLoop(exp_cfg, self.env, self.agent, self.memory, self.recovery_memory,
                               self.recovery_policy, self.nu_schedule)

    # -- setup -----------------------------------------------------------------------------------
    def experiment_setup(self):
        cfg = self.exp_cfg
        if not cfg.cuda:
            raise RuntimeError("recovery_rl_amd runs on the GPU only: pass --cuda (no CPU fallback)")
        torch.manual_seed(cfg.seed)
        np.random.seed(cfg.seed)
        self.device = torch.device("cuda", torch.cuda.current_device())
        register_env(cfg.env_name)
        self.env = make_vec_env(cfg.env_name, cfg.num_envs, device=self.device, seed=cfg.seed)
        self.env.seed(cfg.seed)
        self.env.action_space.seed(cfg.seed)
        self.agent = self.agent_setup(self.env)
        self.recovery_policy = None
        if uses_mb_recovery(cfg):
            from .MPC import MPC
            from .config import create_config
            mpc_cfg = create_config(cfg.env_name, "MPC", dict(cfg.ctrl_arg), cfg.override, self.logdir,
                                    env=self.env)
            self.recovery_policy = MPC(mpc_cfg.ctrl_cfg, mb_dynamics=getattr(cfg, "mb_dynamics", "model"),
                                       seed=int(getattr(cfg, "plan_seed", 0)),
                                       plan_precision=getattr(cfg, "plan_precision", "") or None)
            self.recovery_policy.update_value_func(self.agent.safety_critic)

    def agent_setup(self, env):
        return SAC(env.observation_space, env.action_space, self.exp_cfg, self.logdir, tmp_env=None)

    def get_offline_data(self):
        """Constraint demonstrations (experiment.py:177-249) as device tensors."""
        cfg = self.exp_cfg
        if cfg.task_demos:
            raise NotImplementedError("task demos exist only for the extraction envs (out of scope)")
        data = self.env.transition_function(cfg.num_unsafe_transitions)
        return data, None

    def _apply_demo_share(self):
        """Vectorisation rule for the safety critic's training mix (arg_utils --demo_share): with N > 1 envs and pinned
        demonstrations, int(B * share) rows of every Q_risk batch come from the demonstrations and the rest from the
        online rows.  The one-env reference draws uniformly from a buffer the 20 000 demonstrations fill from 100 % (first
        episode) to about half (400th episode; experiment.py:278-286,438-448, replay_memory.py:54-72): at 4096 envs a
        uniform draw would show them in 2 % of the rows from the 245th iteration on."""
        cfg = self.exp_cfg
        share = float(getattr(cfg, "demo_share", -1.0))
        pinned = getattr(self.recovery_memory, "pinned", 0)
        if share < 0:
            share = 0.5 if (cfg.num_envs > 1 and pinned > 0) else 0.0
        if share > 0 and pinned <= 0:
            raise ValueError("--demo_share needs pinned demonstrations (lock-step loop without --no_pin_demos)")
        self.agent.safety_critic.demo_share = share if share > 0 else None
        # the rules that change what the critics train on, next to the results they produced: printed, written into
        # run_stats.pkl ("vector_rules") and the checkpoint
        self.vector_rules = {"demo_share": share if share > 0 else 0.0, "pinned_demonstrations": int(pinned),
                             **{k: self.vector_rules[k] for k in ("replay_capacities", "cover_rows_limit",
                                                                   "buffers_cover_the_run", "plan_warm_start")}}
        if cfg.num_envs > 1:
            print("Q_risk batch: %s (--demo_share; 0 = the reference's single uniform draw, replay_memory.py:54-72)"
                  % ("%d of %d rows from the %d pinned demonstrations, the rest from the online rows"
                     % (int(cfg.batch_size * share), cfg.batch_size, pinned) if share > 0 else
                     "one uniform draw over the safety buffer" if cfg.pos_fraction < 0 else
                     "stratified draw, --pos_fraction %g" % cfg.pos_fraction))

    # -- pre-training ----------------------------------------------------------------------------
    def pretrain_critic_recovery(self):
        """experiment.py:261-305."""
        cfg = self.exp_cfg
        s, a, c, s2, m = (x[:cfg.num_unsafe_transitions].contiguous() for x in self.constraint_demo_data)
        n_demo = int(c.shape[0])
        if n_demo:
            self.recovery_memory.push(s, a, c, s2, m)
            if cfg.num_envs > 1 and not getattr(cfg, "no_pin_demos", False):
                # vectorisation rule: N envs fill the 1e6-row ring in 1e6 / N iterations and would overwrite the
                # demonstrations -- the only violations a safe policy ever shows the safety critic; the one-env reference
                # (4e4 env-steps per run) never wraps its ring, i.e. keeps them for the whole run
                self.recovery_memory.pin()
            self._apply_demo_share()
        self.num_unsafe_transitions = n_demo
        self.num_constraint_violations += int(c.sum().item())
        self.loop.num_constraint_violations = self.num_constraint_violations
        print("Number of Constraint Transitions: ", self.num_unsafe_transitions)
        print("Number of Constraint Violations: ", self.num_constraint_violations)
        batch = min(cfg.batch_size, int(self.constraint_demo_data[2].shape[0]))
        for i in range(cfg.critic_safe_pretraining_steps):
            if i % 100 == 0:
                print("CRITIC SAFE UPDATE STEP: ", i)
            self.agent.safety_critic.update_parameters(memory=self.recovery_memory,
                                                       policy=self.agent.policy, batch_size=batch)
        self.recovery_memory.check_error()
        if not (cfg.MF_recovery or cfg.Q_sampling_recovery or cfg.DGD_constraints or cfg.RCPO):
            self.train_MB_recovery(s, a, s2, epochs=50)

    def train_MB_recovery(self, states, actions, next_states=None, epochs=50):
        if next_states is not None:
            self.recovery_policy.train(states, actions, random=True, next_obs=next_states, epochs=epochs)
        else:
            self.recovery_policy.train(states, actions)

    # -- main loop -------------------------------------------------------------------------------
    def online_qrisk_enabled(self, num_viols=None):
        """Gate of experiment.py:407-410."""
        cfg = self.exp_cfg
        nv = self.num_viols if num_viols is None else num_viols
        if self.env_shard:
            # every rank must take the same branch (the updates contain collectives): decide on the all-reduced
            # counters (identical everywhere) and the learner's global batch
            nv = self._global_viols
            return (not cfg.disable_online_updates
                    and self._global_rmem_len > cfg.global_batch_size                # len(recovery_memory), :408
                    and (nv + self._global_offline_viols) / cfg.global_batch_size > cfg.pos_fraction)
        return (not cfg.disable_online_updates
                and len(self.recovery_memory) > cfg.batch_size
                and (nv + self.num_constraint_violations) / cfg.batch_size > cfg.pos_fraction)

    def run(self):
        cfg = self.exp_cfg
        resume = getattr(cfg, "resume", "")
        if resume and cfg.num_envs == 1:
            raise NotImplementedError("--resume continues the lock-step loop (--num_envs > 1)")
        if not resume and not cfg.disable_offline_updates and uses_constraint_buffer(cfg):
            self.pretrain_critic_recovery()
        if cfg.num_envs > 1:
            return self.run_vectorized()
        train_rollouts, test_rollouts = [], []
        for i_episode in itertools.count(1):
            train_rollouts.append(self.get_train_rollout(i_episode))
            if i_episode % 10 == 0 and cfg.eval:
                test_rollouts.append(self.get_test_rollout(i_episode))
            if self.total_numsteps > cfg.num_steps or i_episode > cfg.num_eps:
                break
            self.dump_logs(train_rollouts, test_rollouts)

    def _single_env(self):
        if not hasattr(self, "_env1"):
            self._env1 = SingleEnvView(self.env)
        return self._env1

    def get_train_rollout(self, i_episode):
        """One training episode at num_envs == 1 in the reference's event order
        (experiment.py:379-491)."""
        cfg, loop = self.exp_cfg, self.loop
        env = self._single_env()
        episode_reward, episode_steps, done = 0, 0, False
        state = env.reset()
        train_rollout_info = []
        ep_states, ep_actions = [state], []
        if i_episode % 10 == 0:
            print("SEED: ", cfg.seed)
            print("LOGDIR: ", self.logdir)
        while not done:
            if len(self.memory) > cfg.batch_size:
                for _ in range(cfg.updates_per_step):
                    self.agent.update_parameters(self.memory, min(cfg.batch_size, len(self.memory)),
                                                 self.updates, safety_critic=self.agent.safety_critic,
                                                 nu=self.nu_schedule(i_episode))
                    if self.online_qrisk_enabled():
                        self.agent.safety_critic.update_parameters(
                            memory=self.recovery_memory, policy=self.agent.policy,
                            batch_size=cfg.batch_size, plot=0)
                    self.updates += 1
            action, real_action, recovery = loop.act(
                env.obs_tensor, random_actions=cfg.start_steps > self.total_numsteps)
            recovery_used = bool(recovery[0].item()) if recovery is not None else False
            next_state, reward, done, info = env.step(real_action)
            info['recovery'] = recovery_used
            train_rollout_info.append(info)
            episode_steps += 1
            episode_reward += reward
            self.total_numsteps += 1
            push_reward = reward - cfg.constraint_reward_penalty if info['constraint'] else reward
            mask = float(not done)
            done = done or episode_steps == env._max_episode_steps
            env.push_transition(self.memory, real_action if cfg.disable_action_relabeling else action,
                                push_reward, mask)
            if uses_constraint_buffer(cfg):
                env.push_transition(self.recovery_memory, real_action, float(info['constraint']), mask)
                if recovery_used and cfg.add_both_transitions:
                    env.push_transition(self.memory, real_action, push_reward, mask)
            state = next_state
            ep_states.append(state)
            ep_actions.append(info['action'])
        if info['constraint']:
            self.num_viols += 1
            if info['recovery']:
                self.viol_and_recovery += 1
            else:
                self.viol_and_no_recovery += 1
        self.num_successes += int(info['success'])
        if cfg.use_recovery and not cfg.disable_online_updates:
            self.all_ep_data.append({'obs': np.array(ep_states), 'ac': np.array(ep_actions)})
            if i_episode % cfg.recovery_policy_update_freq == 0 and uses_mb_recovery(cfg) \
                    and not cfg.DGD_constraints:
                dev = self.device
                self.train_MB_recovery(
                    [torch.as_tensor(d['obs'], dtype=torch.float32, device=dev) for d in self.all_ep_data],
                    [torch.as_tensor(d['ac'], dtype=torch.float32, device=dev) for d in self.all_ep_data])
                self.all_ep_data = []
        print("Episode: {}, total numsteps: {}, episode steps: {}, reward: {}".format(
            i_episode, self.total_numsteps, episode_steps, round(episode_reward, 2)))
        print("Num Violations So Far: %d" % self.num_viols)
        print("Violations with Recovery: %d" % self.viol_and_recovery)
        print("Violations with No Recovery: %d" % self.viol_and_no_recovery)
        print("Num Successes So Far: %d" % self.num_successes)
        return train_rollout_info

    def get_test_rollout(self, i_episode):
        """experiment.py:493-538 (deterministic task actions; images/gifs are out of scope)."""
        env = self._single_env()
        test_rollout_info = []
        env.reset()
        episode_reward, episode_steps, done = 0, 0, False
        while not done:
            action, real_action, recovery = self.loop.act(env.obs_tensor, train=False)
            _, reward, done, info = env.step(real_action)
            info['recovery'] = bool(recovery[0].item()) if recovery is not None else False
            done = done or episode_steps == env._max_episode_steps        # :515 (checked before ++)
            test_rollout_info.append(info)
            episode_reward += reward
            episode_steps += 1
        print("----------------------------------------")
        print("Avg. Reward: {}".format(round(episode_reward, 2)))
        print("----------------------------------------")
        return test_rollout_info

    def dump_logs(self, train_rollouts, test_rollouts):
        data = {"test_stats": test_rollouts, "train_stats": train_rollouts}
        with open(osp.join(self.logdir, "run_stats.pkl"), "wb") as f:
            pickle.dump(data, f)

    # -- vectorised loop -------------------------------------------------------------------------
    def run_vectorized(self):
        """num_envs > 1: lock-step loop; the steady state is replayed from one hipGraph.  Stops
        when env-steps > num_steps or completed episodes > num_eps (experiment.py:375)."""
        cfg, loop = self.exp_cfg, self.loop
        n = cfg.num_envs
        self._global_viols = 0
        loop.start()
        log_every = cfg.log_every if getattr(cfg, "log_every", 0) else max(1, 100)
        if self.world_size > 1 and log_every < 4:
            # a graph (re)capture adds up to 3 iterations on ONE rank; with a shorter cadence ranks could cross a
            # different number of log boundaries per pass and issue mismatched metric all-reduces
            raise ValueError("--log_every must be >= 4 with more than one rank")
        from . import checkpoint
        from .episode_log import EpisodeLog, EPISODE_DTYPE, InfoRing
        loop.episode_log = EpisodeLog(n, n * (log_every + 4), self.device)   # a capture adds <= 3 iterations
        info_k = min(int(getattr(cfg, "info_envs", 0) or 0), n)
        info = InfoRing(info_k, log_every + 4, self.device, self.env.action_space.high[0],
                        mid_episode=bool(getattr(cfg, "resume", ""))) if info_k else None
        if info is not None:
            loop.step_outputs = True        # the per-step info stream reads the env's per-env outputs after every step
        train_stats = []
        episodes = [np.zeros(0, dtype=EPISODE_DTYPE)]
        history = []
        evals = []
        next_eval = 10 * n                  # eval every 10 episodes per env (experiment.py:372)
        it = 0
        mb_resume = []
        if getattr(cfg, "resume", ""):
            extra = checkpoint.load(self, cfg.resume)
            self._apply_demo_share()
            it, next_eval = extra["iteration"], extra["next_eval"]
            history, evals, episodes = extra["history"], extra["evals"], [extra["episodes"]]
            mb_resume = [tuple(x.to(self.device) for x in row) for row in extra["mb_new"]]
            self._global_viols = extra.get("global_viols", 0)
            was = extra.get("vector_rules", {}).get("demo_share")
            if was is not None and was != self.vector_rules["demo_share"]:
                print("WARNING: the checkpoint was written with --demo_share %g, this run continues with %g"
                      % (was, self.vector_rules["demo_share"]))
            print("Resumed from %s at iteration %d (%d env-steps)" % (cfg.resume, it, loop.total_numsteps))
        # after a resume the offline count comes from the checkpoint's counters (pre-training is skipped)
        start = dist_utils.aggregate_stats(
            {k: {"num_viols": self.num_constraint_violations, "env_steps": len(self.recovery_memory)}.get(k, 0)
             for k in dist_utils.METRIC_KEYS}, self.world_size, self.device)
        self._global_offline_viols, self._global_rmem_len = start["num_viols"], start["env_steps"]
        logged = it // log_every
        t_loop = time.time()
        self.log_wall = []                  # (iteration, seconds since here) per log point: this process's clock, kept out of
                                            # the history (a resumed run's history equals the uninterrupted run's)
        ckpt_every = getattr(cfg, "checkpoint_every", 0)
        ckpt_path = osp.join(self.logdir, "checkpoint.pt")

        def write_checkpoint():
            checkpoint.save(self, ckpt_path, {"iteration": it, "next_eval": next_eval, "history": history,
                                              "evals": evals, "episodes": np.concatenate(episodes), "mb_new": mb_new,
                                              "global_viols": self._global_viols, "vector_rules": self.vector_rules})
        ep_file = open(osp.join(self.logdir, "episode_stats.bin"), "wb")   # append-only, O(new) per log
        ep_file.write(episodes[0].tobytes())
        try:
            captured_gate, warm = None, 0
            mb = uses_mb_recovery(cfg)
            # model-based recovery: the ensemble is re-fitted on the transitions gathered since the last
            # fit every recovery_policy_update_freq * horizon iterations (the reference re-fits every
            # recovery_policy_update_freq episodes, experiment.py:464-480); the batch size scales with
            # num_envs so that an epoch keeps the reference's number of optimiser steps per env-step
            mb_new = mb_resume
            mb_every = cfg.recovery_policy_update_freq * self.env._max_episode_steps
            # env_shard: the updates contain RCCL all-reduces, launched eagerly (not captured)
            # model-based recovery: capturable when the controller counts its planning set on the device (no host read in
            # MPC.act: padded rrl_*_n launches) -- the re-fit and the per-iteration copies for it stay outside the graph
            rp = self.recovery_policy
            mb_graph = (not mb) or (rp is not None and rp.fused is not None and rp.device_count and rp.mb_dynamics == "model"
                                    and rp.has_been_trained and rp.prev_sol.shape[0] == n
                                    and getattr(self.agent, "fast", None) is not None)
            graph_ok = (cfg.target_update_interval == 1 and not cfg.nu_schedule and mb_graph and not self.env_shard)
            while True:
                have_batch = len(self.memory) > cfg.batch_size
                random_actions = cfg.start_steps > loop.total_numsteps
                gate = self.online_qrisk_enabled() if uses_constraint_buffer(cfg) else False
                steady = have_batch and not random_actions and graph_ok
                replay = steady
                if steady and (loop.graph is None or captured_gate != gate):
                    if info is not None or mb:
                        # what an iteration leaves behind is collected per iteration (the per-step info stream; the
                        # transitions of the online re-fit): the capture's warm-up iterations run through this loop's own
                        # body -- three eager steady iterations -- and the capture itself executes nothing
                        if captured_gate != gate:
                            loop.graph, captured_gate, warm = None, gate, 0
                        if warm < 3:
                            warm += 1
                            replay = False
                        else:
                            loop.capture(online_qrisk=gate, warmup=0, iters=1)
                    else:
                        it += loop.capture(online_qrisk=gate)
                        captured_gate = gate
                if info is not None:
                    info.before_step(loop.obs)
                done = 1
                if replay:
                    # nothing on the host looks at the run before the next log point (gate, evaluation, checkpoint and the end
                    # of the run are all decided there): the iterations up to it go out as many-iteration graphs
                    if info is None and not mb and loop.graph_many is not None:
                        done = max(1, min(loop.graph_many_iters, (logged + 1) * log_every - it))
                    loop.advance(done)
                else:
                    loop.vector_step(do_update=have_batch, random_actions=random_actions, online_qrisk=gate)
                if info is not None:
                    info.after_step(self.env, loop._last_real_action, loop._last_recovery)
                it += done
                if mb and not cfg.disable_online_updates:
                    info_s, info_a, info_s2 = self.env.prev_obs, self.env.action_clipped, self.env.next_obs
                    mb_new.append((info_s.clone(), info_a.clone(), info_s2.clone()))
                    if it % mb_every == 0:
                        S, A, S2 = (torch.cat(x) for x in zip(*mb_new))
                        self.recovery_policy.train(S, A, random=True, next_obs=S2, batch_size=32 * n)
                        mb_new = []
                if it // log_every > logged:
                    logged = it // log_every
                    stats = loop.read_stats()
                    self._absorb(stats)
                    new = loop.episode_log.drain()
                    if info is not None:
                        train_stats.extend(info.drain())
                    episodes.append(new)
                    ep_file.write(new.tobytes())
                    ep_file.flush()
                    agg = dist_utils.aggregate_stats(stats, self.world_size, self.device)
                    self._global_viols = agg["num_viols"]
                    if self.env_shard and self._global_rmem_len <= cfg.global_batch_size:
                        self._global_rmem_len = dist_utils.aggregate_stats(
                            {k: (len(self.recovery_memory) if k == "env_steps" else 0)
                             for k in dist_utils.METRIC_KEYS}, self.world_size, self.device)["env_steps"]
                    history.append(dict(stats, iteration=it))
                    self.log_wall.append((it, time.time() - t_loop))
                    if self.rank == 0:
                        print("Iter: {}, total numsteps: {}, episodes: {}, mean episode reward: {}".format(
                            it, agg["env_steps"], agg["episodes"],
                            round(agg["episode_return_sum"] / max(agg["episodes"], 1), 2)))
                        print("Num Violations So Far: %d" % agg["num_viols"])
                        print("Violations with Recovery: %d" % agg["viol_and_recovery"])
                        print("Violations with No Recovery: %d" % agg["viol_and_no_recovery"])
                        print("Num Successes So Far: %d" % agg["num_successes"])
                    if cfg.eval and stats["episodes"] >= next_eval:
                        evals.append(self.get_test_rollout_vectorized(stats["episodes"]))
                        next_eval += 10 * n
                    with open(osp.join(self.logdir, "run_stats.pkl"), "wb") as f:
                        pickle.dump({"vector_stats": history, "eval_stats": evals, "num_envs": n,
                                     "vector_rules": self.vector_rules,
                                     **({"train_stats": train_stats, "test_stats": []} if info is not None else {})}, f)
                    if ckpt_every and logged % ckpt_every == 0:
                        write_checkpoint()
                    # multi-rank: all ranks leave at the same log point (the next aggregate would hang otherwise);
                    # the thresholds apply to the per-rank mean
                    w = max(self.world_size, 1)
                    if agg["env_steps"] > cfg.num_steps * w or agg["episodes"] > cfg.num_eps * w:
                        break
        finally:
            ep_file.close()
        write_checkpoint()
        with open(osp.join(self.logdir, "run_stats.pkl"), "wb") as f:
            pickle.dump({"vector_stats": history, "eval_stats": evals, "num_envs": n, "vector_rules": self.vector_rules,
                         "episode_stats": np.concatenate(episodes),
                         **({"train_stats": train_stats, "test_stats": [], "info_envs": info_k} if info is not None else {})}, f)
        return history

    def get_test_rollout_vectorized(self, label):
        """Deterministic-policy evaluation (experiment.py:493-538) for all envs at once on a separate
        env instance, so the training episodes are not disturbed: every env runs ONE episode; returns
        the mean episode return, success rate and violation rate."""
        cfg = self.exp_cfg
        env = self.eval_env()
        obs = env.reset()
        n = cfg.num_envs
        alive = torch.ones(n, dtype=torch.bool, device=self.device)
        ret = torch.zeros(n, device=self.device)
        succ = torch.zeros(n, dtype=torch.bool, device=self.device)
        viol = torch.zeros(n, dtype=torch.bool, device=self.device)
        for _ in range(env._max_episode_steps + 1):                  # the reference runs horizon + 1 steps (:515)
            action, real_action, _ = self.loop.act(obs, train=False)
            obs, reward, done, info = env.step(real_action.contiguous())
            ret += torch.where(alive, reward, torch.zeros_like(reward))
            succ |= alive & info["success"].bool()
            viol |= alive & info["constraint"].bool()
            alive &= ~done.bool()
            obs = obs.clone()
        out = {"label": label, "avg_reward": float(ret.mean().item()), "success_rate": float(succ.float().mean().item()),
               "violation_rate": float(viol.float().mean().item())}
        if self.rank == 0:
            print("----------------------------------------")
            print("Avg. Reward: {}".format(round(out["avg_reward"], 2)))
            print("----------------------------------------")
        return out

    def eval_env(self):
        if getattr(self, "_eval_env", None) is None:
            cfg = self.exp_cfg
            self._eval_env = make_vec_env(cfg.env_name, cfg.num_envs, device=self.device,
                                          seed=cfg.seed + 7919, auto_reset=False)
        return self._eval_env

    def _absorb(self, stats):
        self.total_numsteps = stats["env_steps"]
        self.num_viols = stats["num_viols"]
        self.viol_and_recovery = stats["viol_and_recovery"]
        self.viol_and_no_recovery = stats["viol_and_no_recovery"]
        self.num_successes = stats["num_successes"]
        self.updates = self.loop.updates


class SingleEnvView:
    """num_envs == 1 adapter giving the driver the reference's numpy step protocol while the
    state, action and transition rows stay on the device."""

    def __init__(self, vec_env):
        assert vec_env.num_envs == 1
        self.vec = vec_env
        self.vec.auto_reset = False
        self._max_episode_steps = vec_env._max_episode_steps
        self.obs_tensor = None

    def reset(self):
        self.obs_tensor = self.vec.reset()
        return self.vec.pos[0].cpu().numpy().copy()

    def step(self, real_action):
        real_action = real_action.contiguous()
        old_state = self.vec.pos[0].cpu().numpy().copy()
        obs, reward, done, info = self.vec.step(real_action)
        self.obs_tensor = obs
        self._last = (info["state"].clone(), info["next_state"].clone())
        state = self.vec.pos[0].cpu().numpy().copy()
        cost = float(reward[0].item())
        return state, cost, bool(done[0].item()), {
            "constraint": int(info["constraint"][0].item()), "reward": cost, "state": old_state,
            "next_state": state, "action": info["action"][0].cpu().numpy(),
            "success": bool(info["success"][0].item())}

    def push_transition(self, memory, action, reward, mask):
        dev = self.vec.device
        memory.push(self._last[0], action.contiguous(),
                    torch.full((1,), float(reward), dtype=torch.float32, device=dev), self._last[1],
                    torch.full((1,), float(mask), dtype=torch.float32, device=dev))


def run_packed(exp_cfg, rank=0, world_size=1):
    """`--seeds_per_gpu S` (S > 1): S independent experiments -- seeds seed, seed + 1, ..., each with its own log directory,
    envs, replay rings, networks and pre-training, exactly what S runs of the reference's seed loop
    (scripts/navigation1.sh:4-8) would create -- advanced together on one GPU: once every seed is in its steady state the
    iteration of all of them is ONE hipGraph whose launches are shared (packed.PackedLoop); every seed's trajectory is the one
    its solo run produces.  Logging per seed at the `--log_every` cadence: counters, the per-episode table (`episode_stats`,
    `episode_stats.bin`) and, with `--info_envs K`, the per-step `train_stats` of the first K envs -- the files the solo
    lock-step run of that seed writes.  Returns the list of per-seed histories."""
    import copy
    from .fast_update import fast_path_supported
    from .packed import PackedLoop
    S = int(exp_cfg.seeds_per_gpu)
    if exp_cfg.num_envs < 2 or not fast_path_supported(exp_cfg) or uses_mb_recovery(exp_cfg) or \
            not (exp_cfg.use_recovery and exp_cfg.MF_recovery) or getattr(exp_cfg, "dp_mode", "replicas") != "replicas":
        raise ValueError("--seeds_per_gpu needs the lock-step loop (--num_envs > 1) on the fused update path with model-free "
                         "recovery (every rank of a multi-GPU launch packs its own seeds: replicas, no exchange)")
    if getattr(exp_cfg, "resume", "") or getattr(exp_cfg, "checkpoint_every", 0):
        raise ValueError("--seeds_per_gpu: checkpoints are written and resumed by the solo lock-step run (--seeds_per_gpu 1)")
    from .episode_log import EPISODE_DTYPE, EpisodeLog, InfoRing
    n = exp_cfg.num_envs
    log_every = exp_cfg.log_every if getattr(exp_cfg, "log_every", 0) else 100
    info_k = min(int(getattr(exp_cfg, "info_envs", 0) or 0), n)
    exps, infos, tables, train_stats, ep_files = [], [], [], [], []
    for k in range(S):
        cfg = copy.deepcopy(exp_cfg)
        cfg.seed = exp_cfg.seed + k
        cfg.logdir_suffix = "%s_seed%d" % (exp_cfg.logdir_suffix, cfg.seed)
        exp = Experiment(cfg)
        if not cfg.disable_offline_updates:
            exp.pretrain_critic_recovery()
        exp.loop.start()
        # per-episode table and per-step info stream from the first iteration on, as in the solo lock-step run
        exp.loop.episode_log = EpisodeLog(n, n * (log_every + 8), exp.device)
        infos.append(InfoRing(info_k, log_every + 8, exp.device, exp.env.action_space.high[0]) if info_k else None)
        if info_k:
            exp.loop.step_outputs = True    # the per-step info stream reads the env's per-env outputs after every step
        tables.append([np.zeros(0, dtype=EPISODE_DTYPE)])
        train_stats.append([])
        ep_files.append(open(osp.join(exp.logdir, "episode_stats.bin"), "wb"))
        exps.append(exp)
    cfg = exp_cfg
    histories = [[] for _ in exps]

    def before():
        for e, info in zip(exps, infos):
            if info is not None:
                info.before_step(e.loop.obs)

    def after():
        for e, info in zip(exps, infos):
            if info is not None:
                info.after_step(e.env, e.loop._last_real_action, e.loop._last_recovery)

    def log_point(it):
        """Per-seed counters, table and info stream at the logging cadence; True when every seed has reached its budget."""
        done = True
        for k, (e, hist) in enumerate(zip(exps, histories)):
            stats = e.loop.read_stats()
            e._absorb(stats)
            hist.append(dict(stats, iteration=it))
            print("Seed: {}, Iter: {}, total numsteps: {}, episodes: {}, mean episode reward: {}".format(
                e.exp_cfg.seed, it, stats["env_steps"], stats["episodes"],
                round(stats["episode_return_sum"] / max(stats["episodes"], 1), 2)))
            print("Num Violations So Far: %d" % stats["num_viols"])
            print("Num Successes So Far: %d" % stats["num_successes"])
            new = e.loop.episode_log.drain()
            tables[k].append(new)
            ep_files[k].write(new.tobytes())
            ep_files[k].flush()
            if infos[k] is not None:
                train_stats[k].extend(infos[k].drain())
            done = done and (stats["env_steps"] > cfg.num_steps or stats["episodes"] > cfg.num_eps)
            with open(osp.join(e.logdir, "run_stats.pkl"), "wb") as f:
                pickle.dump({"vector_stats": hist, "eval_stats": [], "num_envs": n, "seeds_per_gpu": S,
                             "vector_rules": e.vector_rules, "episode_stats": np.concatenate(tables[k]),
                             **({"train_stats": train_stats[k], "test_stats": [], "info_envs": info_k}
                                if infos[k] is not None else {})}, f)
        return done

    # eager until every seed has a batch, has left the random-action phase and trains Q_risk online
    it, logged, finished = 0, 0, False
    while True:
        ready = [len(e.memory) > cfg.batch_size and e.loop.total_numsteps >= cfg.start_steps and e.online_qrisk_enabled()
                 for e in exps]
        if all(ready) or finished:
            break
        if it > 20 * log_every:
            raise RuntimeError("--seeds_per_gpu: the seeds did not all reach the steady state (online Q_risk gate)")
        before()
        for e in exps:
            e.loop.vector_step(do_update=len(e.memory) > cfg.batch_size,
                               random_actions=cfg.start_steps > e.loop.total_numsteps,
                               online_qrisk=e.online_qrisk_enabled())
        after()
        it += 1
        if it // log_every > logged:
            logged = it // log_every
            finished = log_point(it)
    if not finished:
        packed = PackedLoop([e.loop for e in exps], online_qrisk=True)
        # (a capture advances every seed by <= 5 real iterations: the tables and rings have 8 iterations of head-room)
        many = 1 if info_k else max(1, int(getattr(exp_cfg, "graph_iterations", 4)))
        it += packed.capture(around=(before, after) if info_k else None, iters=many)
        while True:
            before()
            done = max(1, min(many, (logged + 1) * log_every - it))     # (as in run_vectorized: whole graphs up to the log point)
            packed.advance(done)
            after()
            it += done
            if it // log_every > logged:
                logged = it // log_every
                if log_point(it):
                    break
    for f in ep_files:
        f.close()
    return histories
