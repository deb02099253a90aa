"""This stack's counterpart of tests/golden/ref_qrisk_gate_probe.py: Q_risk after `pretrain_critic_recovery` (10 000 steps
on 20 000 offline Navigation2 transitions, scripts/navigation2.sh:14) at the start region x = -50 -- outside the offline
data's x range [-40, 10] -- for seeds 1..K.  Prints one JSON list.  Usage: python profiles/qrisk_gate_probe.py [K=16]"""
import contextlib
import io
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
xs = np.arange(-70.0, 12.5, 2.5)
out = []
for seed in range(1, K + 1):
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe",
                              "0.2", "--num_unsafe_transitions", "20000", "--logdir", tempfile.mkdtemp(), "--seed",
                              str(seed), "--num_envs", "64"])
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        exp.train_MB_recovery = lambda *a, **k: None
        exp.pretrain_critic_recovery()
    qr, pol, dev = exp.agent.safety_critic, exp.agent.policy, exp.device
    s = torch.as_tensor(np.stack([xs, np.zeros_like(xs)], 1), dtype=torch.float32, device=dev)
    starts = torch.as_tensor(np.array([-50.0, 0.0]) + np.random.RandomState(0).randn(256, 2), dtype=torch.float32, device=dev)
    with torch.no_grad():
        _, _, mean = pol.sample(s)
        q_pi = qr.get_value(s, mean).reshape(-1)
        _, _, m2 = pol.sample(starts)
        q_start = qr.get_value(starts, m2).reshape(-1)
    out.append({"seed": seed, "grid_x": xs.tolist(), "y0_q_pi": q_pi.cpu().numpy().round(4).tolist(),
                "q_start_mean": float(q_start.mean()), "q_start_share_above_eps": float((q_start > 0.2).float().mean()),
                "offline_violations": exp.num_constraint_violations})
    print(seed, round(out[-1]["q_start_mean"], 3), out[-1]["q_start_share_above_eps"], file=sys.stderr)
print(json.dumps(out))
