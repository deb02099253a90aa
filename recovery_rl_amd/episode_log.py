"""Per-episode training statistics kept on the device and drained at logging cadence.

The reference appends one info dict per env-step to `train_stats` and re-pickles the whole history after
every episode (recovery_rl/experiment.py:421,456-461, dump_logs :540-543).  Its plotting code only uses, per
episode: the length, sum(reward), the last reward and any(constraint) (plotting/plot_runs.py:194-235).  For
N lock-step envs `EpisodeLog` keeps those quantities in per-env accumulators and appends one record per
finished episode (`rrl_episode_log_append`), so the cost is O(1) per episode and nothing crosses PCIe per step.
`episode_metrics` restates the reference's metric code for both schemas.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

EPISODE_DTYPE = np.dtype([("env", "<i4"), ("iteration", "<i4"), ("length", "<i4"), ("constraint_steps", "<i4"),
                          ("recovery_steps", "<i4"), ("flags", "<i4"), ("ret", "<f8"), ("last_reward", "<f8")])
FLAG_SUCCESS, FLAG_CONSTRAINT, FLAG_RECOVERY = 1, 2, 4


class EpisodeLog:
    """Device accumulators + record table.  `capacity` must cover the episodes that can finish between two
    drains (at most num_envs per iteration); `drain()` raises if records were dropped."""

    def __init__(self, num_envs, capacity, device):
        dev = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.n, self.capacity, self.device = int(num_envs), int(capacity), dev
        self.ep_len = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(self.n, dtype=torch.float64, device=dev)
        self.ep_viol = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.ep_rec = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.rec_i32 = torch.zeros(self.capacity, _lib.EPLOG_I32, dtype=torch.int32, device=dev)
        self.rec_f64 = torch.zeros(self.capacity, 2, dtype=torch.float64, device=dev)
        self.state = torch.zeros(3, dtype=torch.int64, device=dev)
        self._desc = _lib.rrl_episode_log_t(self.rec_i32.data_ptr(), self.rec_f64.data_ptr(), self.capacity,
                                            self.state.data_ptr())

    def append(self, reward, constraint, success, ep_done, recovery=None):
        """Advance the accumulators with one lock-step transition (u8 masks, f32 reward, all [N])."""
        if recovery is not None and recovery.dtype != torch.uint8:
            recovery = recovery.to(torch.uint8)
        args = (self.n, _lib.ptr(reward), _lib.ptr(constraint), _lib.ptr(success), _lib.ptr(ep_done),
                _lib.ptr(recovery), _lib.ptr(self.ep_len), _lib.ptr(self.ep_ret), _lib.ptr(self.ep_viol),
                _lib.ptr(self.ep_rec), C.byref(self._desc))
        # seed packing re-issues this launch as it is (one per seed; the table is not on the iteration's critical path)
        from .fast_update import record
        record("call", self.lib.rrl_episode_log_append, args, (reward, constraint, success, ep_done, recovery, self))
        _lib.check(self.lib.rrl_episode_log_append(*args, _lib.current_stream()), "rrl_episode_log_append")

    def attach(self, a):
        """Let the fused env-step launch advance this log (rrl_step_push_t.log_*): same records and accumulator values as
        `append` fed with that step's per-env outputs."""
        p = _lib.ptr
        a.log_rec_i32, a.log_rec_f64, a.log_cap, a.log_state = p(self.rec_i32), p(self.rec_f64), self.capacity, p(self.state)
        a.log_len, a.log_ret, a.log_viol, a.log_rec = p(self.ep_len), p(self.ep_ret), p(self.ep_viol), p(self.ep_rec)

    def drain(self):
        """Copy the finished-episode records to the host (sorted by iteration, env) and clear the table."""
        count = int(self.state[0].item())
        if count > self.capacity:
            raise _lib.RRLError("episode log overflow: %d records for capacity %d (drain more often)"
                                % (count, self.capacity))
        out = np.zeros(count, dtype=EPISODE_DTYPE)
        if count:
            # records arrive in completion order (one atomic slot per finished episode): sorted by (iteration, env) on the
            # device -- a key sort of a few thousand 64-bit words -- so that the host only copies (a numpy lexsort of the
            # 16 000 records 4096 envs finish between two log points cost more than the hundred iterations' launches)
            ri, rf = self.rec_i32[:count], self.rec_f64[:count]
            order = torch.argsort(ri[:, 1].to(torch.int64) * (1 << 32) + ri[:, 0].to(torch.int64))
            ri, rf = ri[order].cpu().numpy(), rf[order].cpu().numpy()
            for k, name in enumerate(("env", "iteration", "length", "constraint_steps", "recovery_steps", "flags")):
                out[name] = ri[:, k]
            out["ret"], out["last_reward"] = rf[:, 0], rf[:, 1]
        self.state[0].zero_()
        return out


class InfoRing:
    """Per-STEP `info` stream of the reference (env/navigation1.py:82-89 + `recovery`, experiment.py:421,427) for the first K
    envs of a lock-step run, so that `run_stats.pkl` carries `train_stats` in the reference's own schema -- a list of episodes,
    each a list of step dicts -- and plotting/plot_runs.py:147-235 reads the run unchanged.  Opt-in (`--info_envs K`): two small
    device copies per iteration next to the captured graph (state before the step, one packed row after it), drained at
    logging cadence; episodes are emitted in completion order (iteration, then env)."""
    FIELDS = 11      # state 2 | action 2 | next_state 2 | reward | constraint | success | ep_done | recovery

    def __init__(self, k, steps, device, action_high, mid_episode=False):
        """mid_episode: the stream starts inside running episodes (--resume): each env's first, partial, episode is dropped."""
        self.k, self.steps, self.device = int(k), int(steps), device
        self.partial = [bool(mid_episode)] * int(k)
        self.rows = torch.zeros(self.steps, self.k, self.FIELDS, dtype=torch.float32, device=device)
        self.state = torch.zeros(self.k, 2, dtype=torch.float32, device=device)
        self.filled = 0
        self.hi = float(action_high)
        self.open = [[] for _ in range(self.k)]          # unfinished episodes per env (host side)

    def before_step(self, obs):
        self.state.copy_(obs[:self.k])

    def after_step(self, env, real_action, recovery):
        if self.filled >= self.steps:
            raise _lib.RRLError("info ring overflow (drain more often)")
        k, row = self.k, self.rows[self.filled]
        row[:, 0:2] = self.state
        row[:, 2:4] = real_action[:k].clamp(-self.hi, self.hi)          # info['action'] = the clipped executed action
        row[:, 4:6] = env.next_obs[:k]
        row[:, 6] = env.reward[:k]
        row[:, 7] = env.constraint[:k]
        row[:, 8] = env.success[:k]
        row[:, 9] = env.ep_done[:k]
        row[:, 10] = 0 if recovery is None else recovery[:k]
        self.filled += 1

    def drain(self):
        """Finished episodes since the last drain, in the reference's schema."""
        rows = self.rows[:self.filled].cpu().numpy()
        self.filled = 0
        done = []
        for t in range(rows.shape[0]):
            for e in range(self.k):
                r = rows[t, e]
                self.open[e].append({"constraint": int(r[7]), "reward": float(r[6]),
                                     "state": r[0:2].astype(np.float64), "next_state": r[4:6].astype(np.float64),
                                     "action": r[2:4].copy(), "success": bool(r[8]), "recovery": bool(r[10])})
                if r[9]:
                    if not self.partial[e]:
                        done.append(self.open[e])
                    self.partial[e] = False
                    self.open[e] = []
        return done


def records_from_train_stats(train_stats):
    """The same table from the reference's per-step schema (list of episodes of info dicts)."""
    out = np.zeros(len(train_stats), dtype=EPISODE_DTYPE)
    for k, traj in enumerate(train_stats):
        ret = 0
        for step in traj:
            ret += step["reward"]
        last = traj[-1]
        out[k] = (0, k, len(traj), int(sum(int(s["constraint"]) for s in traj)),
                  int(sum(int(bool(s.get("recovery", False))) for s in traj)),
                  (FLAG_SUCCESS if last.get("success") else 0) | (FLAG_CONSTRAINT if last["constraint"] else 0)
                  | (FLAG_RECOVERY if last.get("recovery") else 0), ret, last["reward"])
    return out


def episode_metrics(run_stats, experiment="navigation1", max_eps=None):
    """Restatement of the per-run metric code of plotting/plot_runs.py:194-235.  `run_stats` is either the
    reference's dict ({"train_stats": [[info, ...], ...]}) or this stack's ({"episode_stats": records}).
    Returns cumulative violations, cumulative task successes, the violation-masked returns and lengths."""
    if "episode_stats" in run_stats:
        rec = run_stats["episode_stats"]
    else:
        rec = records_from_train_stats(run_stats["train_stats"])
    rec = rec[:max_eps]
    ep_lengths = rec["length"].astype(np.int64)
    violations = np.cumsum(rec["constraint_steps"] > 0)                  # :213-217
    rewards_safe = rec["ret"].astype(np.float64).copy()
    rewards_safe[violations > 0] = np.nan                                # :219-221
    last = rec["last_reward"]
    if "maze" in experiment:                                            # :225-232
        successes = (-last < 0.03).astype(int)
    elif "extraction" in experiment:
        successes = (last == 0).astype(int)
    else:
        successes = (last > -4).astype(int)
    return {"ep_lengths": ep_lengths, "train_violations": violations, "task_successes": np.cumsum(successes),
            "train_rewards_safe": rewards_safe}


def moving_average(x, N):
    """NaN-aware window mean of plotting/plot_runs.py:22-34 (vectorised)."""
    x = np.asarray(x, dtype=np.float64)
    if len(x) < N:
        return np.zeros(0)
    win = np.lib.stride_tricks.sliding_window_view(x, N)
    cnt = N - np.isnan(win).sum(axis=1)
    out = np.full(len(win), np.nan)
    ok = cnt > 0
    out[ok] = np.array([np.nansum(w) for w in win[ok]]) / cnt[ok]
    return out


def plot_curves(run_stats, experiment="navigation1", max_eps=None):
    """The four curves plot_runs.py draws for ONE run (PLOT_TYPE ratio / success / violation / reward,
    plotting/plot_runs.py:237-312)."""
    m = episode_metrics(run_stats, experiment, max_eps)
    s, v = m["task_successes"], m["train_violations"]
    return {"success": s.astype(np.float64), "violation": v.astype(np.float64), "ratio": (s + 1) / (v + 1),
            "reward": moving_average(m["train_rewards_safe"], 100)}
