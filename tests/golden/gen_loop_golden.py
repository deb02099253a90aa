"""Generate golden fixture G9 (driver loop semantics at one env) by running the REFERENCE's
recovery_rl.experiment.Experiment in this container with a scripted env and a scripted agent
(tests/golden/loop_scenario.py): the recorded event stream -- every replay push of both buffers, every
update call with the buffer lengths it saw, the counters and the per-step info of run_stats.pkl --
pins the update -> act -> step -> push order, the mask-before-horizon rule (experiment.py:434-435),
action relabelling and the add_both_transitions / reward-penalty / disable_online_updates variants.

Run: python tests/golden/gen_loop_golden.py -> tests/golden/loop_golden.json (data only).
Harness patch (a) of SURVEY 8c: `torchify` -> CPU.
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import torch  # noqa: E402

import loop_scenario as sc  # noqa: E402


def tolist(x):
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.astype(np.float64).tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    return x


def run_variant(name, extra):
    import recovery_rl.experiment as rexp
    import arg_utils
    script = sc.Script()
    log = {"sac_updates": [], "qrisk_updates": []}

    class Space:
        shape = (2,)
        low, high = -np.ones(2), np.ones(2)

        def seed(self, s=None):
            pass

        def sample(self):
            return script.random_action()

    class Env:
        _max_episode_steps = sc.HORIZON
        action_space, observation_space = Space(), Space()

        def seed(self, s=None):
            pass

        def reset(self):
            self.state = script.start_episode()
            return self.state

        def step(self, a):
            old = self.state
            nxt, r, done, cons, succ = script.transition(old, a)
            self.state = nxt
            return nxt, r, done, {"constraint": cons, "reward": r, "state": old, "next_state": nxt,
                                  "action": np.asarray(a), "success": succ}

        def transition_function(self, num, task_demos=False):
            return script.offline_data(num)

    env = Env()

    class SafetyCritic:
        def update_parameters(self, memory=None, policy=None, batch_size=None, plot=False):
            log["qrisk_updates"].append([len(memory), batch_size])

        def get_value(self, s, a):
            return torch.tensor([[script.risk()]])

        def select_action(self, state, eval=False):
            return script.recovery_action()

    class Agent:
        policy = object()
        safety_critic = SafetyCritic()

        def select_action(self, state, eval=False):
            return script.task_action()

        def update_parameters(self, memory, batch_size, updates, nu=None, safety_critic=None):
            log["sac_updates"].append([len(memory), batch_size, updates])
            return 0.0, 0.0, 0.0, 0.0, 0.0

    rexp.torchify = lambda x: torch.FloatTensor(x)                       # patch (a)
    rexp.register_env = lambda name: None
    rexp.make_env = lambda name: env
    rexp.Experiment.agent_setup = lambda self, e: Agent()
    tmp = tempfile.mkdtemp()
    argv = sys.argv
    sys.argv = ["rrl_main"] + sc.BASE_ARGV + ["--logdir", tmp] + extra
    try:
        cfg = arg_utils.get_args()
    finally:
        sys.argv = argv
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        exp = rexp.Experiment(cfg)
        exp.run()
    stats = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    res = {
        "argv": sc.BASE_ARGV + extra,
        "memory": [tolist(t) for t in exp.memory.buffer],
        "recovery_memory": [tolist(t) for t in exp.recovery_memory.buffer],
        "sac_updates": log["sac_updates"], "qrisk_updates": log["qrisk_updates"],
        "counters": {k: int(getattr(exp, k)) for k in ("total_numsteps", "updates", "num_viols", "viol_and_recovery",
                                                        "viol_and_no_recovery", "num_successes",
                                                        "num_constraint_violations", "num_unsafe_transitions")},
        "train_stats": [[{k: tolist(v) for k, v in step.items()} for step in ep] for ep in stats["train_stats"]],
        "n_test_rollouts": len(stats["test_stats"]),
        "episode_lines": [l for l in out.getvalue().splitlines() if l.startswith(("Episode:", "Num ", "Violations "))],
    }
    return res


def main():
    out = {name: run_variant(name, extra) for name, extra in sc.VARIANTS.items()}
    json.dump(out, open(os.path.join(HERE, "loop_golden.json"), "w"), indent=0)
    for k, v in out.items():
        print(k, v["counters"], "pushes", len(v["memory"]), len(v["recovery_memory"]))


if __name__ == "__main__":
    main()
