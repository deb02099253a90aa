"""Per-episode diagnostics of a model-based recovery run (scripts/navigation2.sh:14), the same on both stacks: the
reference's classes and this stack's have the same names, so the probes are installed from outside on either
(run_reference_mb_diag.py imports the reference here in the container; profiles/mb_diag.py runs this stack on the GPU).

Per episode: recovery steps, the gate's input Q_risk(s, a_task) over the episode (mean / max / share above eps_safe), the
planner's calls; per ensemble re-fit (recovery_rl/MPC.py:213-309, called after every episode: experiment.py:464-480): rows
in the training set, and on the NEW episode's transitions the ensemble's one-step error and predicted standard deviation
before and after the re-fit."""
import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def behaviour(info):
    """Where an episode went and what the two controllers did, from its per-step info dicts (keys of env.step's info on both
    stacks: state, next_state, action = the executed action, recovery)."""
    s = np.array([_np(x["state"]).reshape(-1) for x in info], dtype=np.float64)
    ns = np.array([_np(x["next_state"]).reshape(-1) for x in info], dtype=np.float64)
    a = np.array([_np(x["action"]).reshape(-1) for x in info], dtype=np.float64)
    rec = np.array([bool(x.get("recovery", False)) for x in info])
    out = {"final": ns[-1].round(2).tolist(), "max_x": float(s[:, 0].max()), "mean_abs_y": float(np.abs(s[:, 1]).mean()),
           "task_action_mean": a[~rec].mean(0).round(3).tolist() if (~rec).any() else None}
    if rec.any():
        out.update(rec_action_mean=a[rec].mean(0).round(3).tolist(), rec_action_norm=float(np.sqrt((a[rec] ** 2).sum(1)).mean()),
                   rec_x_range=[float(s[rec, 0].min()), float(s[rec, 0].max())], rec_abs_y=float(np.abs(s[rec, 1]).mean()),
                   rec_first_step=int(np.argmax(rec)))
    return out


class Probe:
    def __init__(self, eps_safe):
        self.eps_safe = float(eps_safe)
        self.episodes, self.refits = [], []
        self._gate, self._plans = [], 0

    # -- the gate ----------------------------------------------------------------------------------------------------
    def gate(self, value):
        self._gate.append(float(value))

    def plan(self):
        self._plans += 1

    def end_episode(self, steps, success, violation, recovery_steps, info=None):
        g = np.asarray(self._gate if self._gate else [0.0])
        self.episodes.append({"steps": int(steps), "success": int(success), "violation": int(violation),
                              "recovery_steps": int(recovery_steps), "planner_calls": self._plans,
                              "gate_mean": float(g.mean()), "gate_max": float(g.max()),
                              "gate_on": float((g > self.eps_safe).mean())})
        if info is not None:
            self.episodes[-1].update(behaviour(info))
        self._gate, self._plans = [], 0

    # -- the ensemble ------------------------------------------------------------------------------------------------
    def model_error(self, mpc, obs_trajs, acs_trajs):
        """One-step error of the ensemble on trajectories (lists of [T+1, 2] observations and [T, 2] actions): mean over
        members and rows of |mean prediction - (s' - s)|^2 per dimension summed, and of the predicted sd."""
        ins, tgs = [], []
        for obs, acs in zip(obs_trajs, acs_trajs):
            o, a = _np(obs).astype(np.float32), _np(acs).astype(np.float32)
            ins.append(np.concatenate([o[:-1], a], axis=-1))
            tgs.append(o[1:] - o[:-1])
        x, y = np.concatenate(ins), np.concatenate(tgs)
        model = mpc.model
        dev = model.lin0_w.device
        xt = torch.as_tensor(x, device=dev)[None].expand(model.num_nets, -1, -1).contiguous()
        with torch.no_grad():
            mean, var = model(xt)
        mean, var = _np(mean), _np(var)
        err = ((mean - y[None]) ** 2).sum(-1)
        return {"rows": int(x.shape[0]), "mse": float(err.mean()), "mse_worst_member": float(err.mean(1).max()),
                "pred_sd": float(np.sqrt(var).mean()), "target_norm": float(np.sqrt((y ** 2).sum(-1)).mean())}

    def wrap_train(self, mpc):
        real = mpc.train

        def train(obs_trajs, acs_trajs, *a, **k):
            online = not k.get("random", False) and not (len(a) > 0 and a[0])
            rec = None
            if online and mpc.has_been_trained:
                rec = {"before": self.model_error(mpc, obs_trajs, acs_trajs)}
            out = real(obs_trajs, acs_trajs, *a, **k)
            if rec is not None:
                rec["after"] = self.model_error(mpc, obs_trajs, acs_trajs)
                rec["train_rows"] = int(mpc.train_in.shape[0])
                rec["episode"] = len(self.episodes)
                self.refits.append(rec)
            return out
        mpc.train = train

    def wrap_planner(self, mpc):
        real = mpc.act

        def act(*a, **k):
            if mpc.has_been_trained:
                self.plan()
            return real(*a, **k)
        mpc.act = act

    def result(self, **extra):
        return dict(extra, eps_safe=self.eps_safe, episodes=self.episodes, refits=self.refits)
