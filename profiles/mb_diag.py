"""This stack's model-based recovery line (scripts/navigation2.sh:14, one env, reference-order loop) with the per-episode
probes of tests/golden/mb_diag_common.py -- the twin of tests/golden/run_reference_mb_diag.py (which runs the imported
reference on the CPU): same probes, installed from outside on the same class and method names.

    python profiles/mb_diag.py [seeds=1,3,4] [episodes=45] [variant] [extra flags ...]   ->  one JSON line per seed
variant (which of this stack's hand-written paths run; "-" = all of them, the default):
    updates=autograd   SAC / Q_risk updates through the torch modules + autograd (--no_fast_path)
    planner=torch      MPC._compile_cost through the torch modules instead of rrl_plan_cost
    fit=torch          the ensemble re-fit through torch autograd instead of rrl_ens_train_*
    all=torch          the three together: the loop then runs the reference's mathematics line by line on torch kernels
    math=torch         updates + planner (what the gate and the recovery action are computed by; the re-fit stays on its kernels)
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import arg_utils  # noqa: E402
from mb_diag_common import Probe  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402


def run(seed, num_eps, extra=(), variant="-"):
    torch_updates = variant in ("updates=autograd", "all=torch", "math=torch")
    extra = list(extra) + (["--no_fast_path"] if torch_updates else [])
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                              "--logdir", tempfile.mkdtemp(), "--logdir_suffix", "RRL_MB", "--num_eps", str(num_eps),
                              "--num_unsafe_transitions", "20000", "--seed", str(seed), "--eval", ""] + list(extra))
    t0 = time.time()
    probe = Probe(cfg.eps_safe)
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        if variant in ("planner=torch", "all=torch", "math=torch"):
            exp.recovery_policy.use_fused_planner = False
            exp.recovery_policy.fused = None
            exp.recovery_policy.device_count = False
        if variant in ("fit=torch", "all=torch"):
            exp.recovery_policy.fused_train = False
            exp.recovery_policy.graph_train = False
        demos = os.environ.get("RRL_MB_DIAG_DEMOS")        # an .npz of the REFERENCE's offline demonstrations (seed<k>.{s,a,c,n,m}):
        if demos:                                          # what get_offline_data drew on the other stack for this seed
            import numpy as np
            import torch
            z = np.load(demos)
            exp.constraint_demo_data = tuple(torch.as_tensor(z["seed%d.%s" % (seed, k)], device=exp.device) for k in "sacnm")
        probe.wrap_train(exp.recovery_policy)
        probe.wrap_planner(exp.recovery_policy)
        real_act = exp.loop.act

        def act(obs, random_actions=False, train=True):
            action, real_action, recovery = real_act(obs, random_actions=random_actions, train=train)
            if recovery is not None:               # the gate's input: Q_risk(s, a_task) (experiment.py:548-556)
                probe.gate(float(exp.agent.safety_critic.get_value(obs, action).reshape(-1)[0]))
            return action, real_action, recovery
        exp.loop.act = act
        real_rollout = exp.get_train_rollout

        def rollout(i_episode):
            info = real_rollout(i_episode)
            probe.end_episode(len(info), info[-1]["reward"] > -4, any(s["constraint"] for s in info),
                              sum(int(bool(s.get("recovery", False))) for s in info), info=info)
            return info
        exp.get_train_rollout = rollout
        exp.run()
    return probe.result(stack="recovery_rl_amd", seed=seed, wall_seconds=time.time() - t0, extra_flags=list(extra), variant=variant,
                        planner_fused=exp.recovery_policy.fused is not None, updates_fused=getattr(exp.agent, "fast", None) is not None)


if __name__ == "__main__":
    seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1,3,4").split(",")]
    eps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    variant = sys.argv[3] if len(sys.argv) > 3 else "-"
    for s in seeds:
        print(json.dumps(run(s, eps, sys.argv[4:], variant)), flush=True)
