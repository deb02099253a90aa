"""The REFERENCE's model-based recovery line (scripts/navigation2.sh:14) on the CPU in this container with the per-episode
probes of mb_diag_common.py installed from outside (harness patches as run_reference_training.py: torchify -> CPU, critic
step deferred behind policy_loss.backward(), float32 log_std).

Run: python tests/golden/run_reference_mb_diag.py [seed=1] [episodes=40]  ->  tests/golden/ref_mb_diag_seed<seed>.json
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
import torch  # noqa: E402

from mb_diag_common import Probe  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    num_eps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    import arg_utils
    import recovery_rl.experiment as rexp
    import recovery_rl.sac as rsac
    rexp.torchify = lambda x: torch.FloatTensor(x)
    orig_init = rsac.SAC.__init__

    def patched_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.safety_critic.policy.log_std.data = self.safety_critic.policy.log_std.data.float()
        real_c, real_p, snap = self.critic_optim.step, self.policy_optim.step, {}

        def deferred():
            snap["g"] = [p.grad.clone() for p in self.critic.parameters()]

        def both():
            real_p()
            for p, g in zip(self.critic.parameters(), snap["g"]):
                p.grad = g
            real_c()
        self.critic_optim.step, self.policy_optim.step = deferred, both
    rsac.SAC.__init__ = patched_init
    tmp = tempfile.mkdtemp()
    sys.argv = ["rrl_main", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                "--logdir", tmp, "--logdir_suffix", "RRL_MB", "--num_eps", str(num_eps), "--num_unsafe_transitions", "20000",
                "--seed", str(seed), "--eval", ""]
    cfg = arg_utils.get_args()
    t0 = time.time()
    probe = Probe(cfg.eps_safe)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        exp = rexp.Experiment(cfg)
        probe.wrap_train(exp.recovery_policy)
        probe.wrap_planner(exp.recovery_policy)
        real_value = exp.agent.safety_critic.get_value

        def get_value(states, actions, **k):
            v = real_value(states, actions, **k)
            if v.numel() == 1:                       # the gate's query (experiment.py:548-556); the planner asks for 8000 rows
                probe.gate(float(v))
            return v
        exp.agent.safety_critic.get_value = get_value
        real_rollout = exp.get_train_rollout

        def rollout(i_episode):
            info = real_rollout(i_episode)
            probe.end_episode(len(info), info[-1]["reward"] > -4, any(s["constraint"] for s in info),
                              sum(int(bool(s.get("recovery", False))) for s in info), info=info)
            return info
        exp.get_train_rollout = rollout
        exp.run()
    out = probe.result(stack="reference", seed=seed, wall_seconds=time.time() - t0)
    json.dump(out, open(os.path.join(HERE, "ref_mb_diag_seed%d.json" % seed), "w"))
    print("episodes", len(out["episodes"]), "wall", round(out["wall_seconds"]))


if __name__ == "__main__":
    main()
