"""What does the REFERENCE's pre-trained safety critic say at the start of a Navigation2 episode?

The env starts at [-50, 0] + N(0, I) (env/navigation2.py:90-96) while the offline constraint data of
`get_offline_data` covers x in [-40, 10] only (env/navigation2.py:133-243): Q_risk at the start region is an
EXTRAPOLATION of the network fitted by `pretrain_critic_recovery` (recovery_rl/experiment.py:261-297: 10 000
QRiskWrapper.update_parameters steps on 20 000 offline transitions).  Where that extrapolation exceeds eps_safe = 0.2
(scripts/navigation2.sh:14) the recovery gate (experiment.py:566-571) is closed from the first step of the run.

This runs exactly that pre-training with the reference's code (model-based controller training skipped: it does not touch
Q_risk) for one seed and records Q_risk(s, a) on the y = 0 line and at the start state, for the task policy's mean action
and for eight compass actions.  Output: tests/golden/ref_qrisk_gate_seed<seed>.json (numbers only).

Run: python tests/golden/ref_qrisk_gate_probe.py <seed>
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    seed = int(sys.argv[1])
    import arg_utils
    import recovery_rl.experiment as rexp
    import recovery_rl.sac as rsac
    rexp.torchify = lambda x: torch.FloatTensor(x)                                  # harness patch (a)
    orig_init = rsac.SAC.__init__

    def patched_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.safety_critic.policy.log_std.data = self.safety_critic.policy.log_std.data.float()   # (c)
    rsac.SAC.__init__ = patched_init
    sys.argv = ["rrl_main", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                "--logdir", tempfile.mkdtemp(), "--logdir_suffix", "RRL_MB", "--num_eps", "400",
                "--num_unsafe_transitions", "20000", "--seed", str(seed), "--eval", ""]
    cfg = arg_utils.get_args()
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = rexp.Experiment(cfg)
        exp.train_MB_recovery = lambda *a, **k: None        # the PETS fit (experiment.py:299-305) does not touch Q_risk
        exp.pretrain_critic_recovery()
    qr, pol = exp.agent.safety_critic, exp.agent.policy
    xs = np.arange(-70.0, 12.5, 2.5)
    ang = np.arange(8) * np.pi / 4
    acts = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)

    def line(y):
        s = torch.FloatTensor(np.stack([xs, np.full_like(xs, y)], 1))
        with torch.no_grad():
            _, _, mean = pol.sample(s)
            q_pi = qr.get_value(s, mean).reshape(-1).numpy()
            q_dir = np.stack([qr.get_value(s, torch.FloatTensor(a).expand_as(s)).reshape(-1).numpy() for a in acts])
        return {"q_pi": q_pi.round(4).tolist(), "q_min_dir": q_dir.min(0).round(4).tolist(),
                "q_east": q_dir[0].round(4).tolist()}
    rng = np.random.RandomState(0)
    starts = torch.FloatTensor(np.array([-50.0, 0.0]) + rng.randn(256, 2))
    with torch.no_grad():
        _, _, mean = pol.sample(starts)
        q_start = qr.get_value(starts, mean).reshape(-1).numpy()
    out = {"seed": seed, "eps_safe": cfg.eps_safe, "gamma_safe": cfg.gamma_safe, "grid_x": xs.tolist(),
           "y0": line(0.0), "y10": line(10.0), "ym10": line(-10.0),
           "q_start_mean": float(q_start.mean()), "q_start_share_above_eps": float((q_start > cfg.eps_safe).mean()),
           "offline_transitions": exp.num_unsafe_transitions, "offline_violations": exp.num_constraint_violations,
           "pretraining_steps": cfg.critic_safe_pretraining_steps, "wall_seconds": time.time() - t0}
    json.dump(out, open(os.path.join(HERE, "ref_qrisk_gate_seed%d.json" % seed), "w"))
    print({k: v for k, v in out.items() if not isinstance(v, (list, dict))})


if __name__ == "__main__":
    main()
