"""Run the REFERENCE's training loop (rrl_main configuration of scripts/navigation1.sh:7) on the CPU in
this container and record the learning-level statistics that plotting/plot_runs.py:214-235 derives from
run_stats.pkl (cumulative task successes / constraint violations, episode lengths).  Used as the
learning-level anchor of SURVEY.md section 8d ("parity gates"); RNG streams differ, so only the statistics
are comparable.

Harness patches, applied from outside: (a) torchify -> CPU, (b) critic step deferred until the
policy loss has been back-propagated (torch >= 1.5 rejects the reference's order), (c) float32 log_std.

Run: python tests/golden/run_reference_training.py [seed] [num_eps] [nav1|nav2|nav2_mb] [logdir]
  -> tests/golden/ref_learning_<nav1|nav2|nav2_mb>_seed<seed>.json
     (nav2 = scripts/navigation2.sh:7, model-free recovery; nav2_mb = scripts/navigation2.sh:14, PETS/CEM
     recovery through MPC.py:213-347 with the ensemble re-fit of experiment.py:464-480 after every episode)
     python tests/golden/run_reference_training.py summarize <run_stats.pkl> <out.json> [seed]
     turns the run_stats.pkl the reference re-writes after every episode (experiment.py:540-543) into the
     same record, so a run that is still going (the model-based line takes hours on CPU) can be read.
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import torch  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    num_eps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    which = sys.argv[3] if len(sys.argv) > 3 else "nav1"
    env_name, gamma_safe, eps_safe = {"nav1": ("navigation1", "0.8", "0.3"), "nav2": ("navigation2", "0.65", "0.2"),
                                      "nav2_mb": ("navigation2", "0.65", "0.2")}[which]
    import arg_utils
    import recovery_rl.experiment as rexp
    import recovery_rl.sac as rsac
    rexp.torchify = lambda x: torch.FloatTensor(x)                                  # (a)
    orig_init = rsac.SAC.__init__

    def patched_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.safety_critic.policy.log_std.data = self.safety_critic.policy.log_std.data.float()   # (c)
        real_c, real_p, snap = self.critic_optim.step, self.policy_optim.step, {}

        def deferred():
            snap["g"] = [p.grad.clone() for p in self.critic.parameters()]

        def both():
            real_p()
            for p, g in zip(self.critic.parameters(), snap["g"]):
                p.grad = g
            real_c()
        self.critic_optim.step, self.policy_optim.step = deferred, both              # (b)
    rsac.SAC.__init__ = patched_init
    tmp = sys.argv[4] if len(sys.argv) > 4 else tempfile.mkdtemp()
    recovery = ["--use_recovery"] if which == "nav2_mb" else ["--use_recovery", "--MF_recovery"]
    sys.argv = ["rrl_main", "--env-name", env_name] + recovery + ["--gamma_safe", gamma_safe,
                "--eps_safe", eps_safe, "--logdir", tmp, "--logdir_suffix", "RRL_MB" if which == "nav2_mb" else "RRL_MF",
                "--num_eps", str(num_eps), "--num_unsafe_transitions", "20000", "--seed", str(seed), "--eval", ""]
    cfg = arg_utils.get_args()
    t0 = time.time()
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        exp = rexp.Experiment(cfg)
        exp.run()
    res = summarize(os.path.join(exp.logdir, "run_stats.pkl"), seed)
    res.update({"argv": sys.argv[1:], "num_constraint_transitions": exp.num_unsafe_transitions,
                "num_constraint_violations_offline": exp.num_constraint_violations, "wall_seconds": time.time() - t0,
                "env_steps": exp.total_numsteps})
    json.dump(res, open(os.path.join(HERE, "ref_learning_%s_seed%d.json" % (which, seed)), "w"))
    print({k: v for k, v in res.items() if not isinstance(v, list)})


def summarize(path, seed):
    """plotting/plot_runs.py:214-235 on a run_stats.pkl of the reference (one list of step dicts per episode)."""
    stats = pickle.load(open(path, "rb"))["train_stats"]
    viol = [int(any(s["constraint"] for s in ep)) for ep in stats]
    succ = [int(ep[-1]["reward"] > -4) for ep in stats]
    rec = [sum(int(bool(s.get("recovery", False))) for s in ep) for ep in stats]
    return {"seed": seed, "episodes": len(stats), "episode_lengths": [len(ep) for ep in stats],
            "violations": viol, "successes": succ, "recovery_steps": rec, "total_violations": sum(viol),
            "total_successes": sum(succ), "env_steps": sum(len(ep) for ep in stats)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        r = summarize(sys.argv[2], int(sys.argv[4]) if len(sys.argv) > 4 else -1)
        r["partial"] = True
        json.dump(r, open(sys.argv[3], "w"))
        print({k: v for k, v in r.items() if not isinstance(v, list)})
    else:
        main()
