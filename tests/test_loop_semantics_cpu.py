"""CPU suite: G9 -- the single-env driver loop reproduces the REFERENCE's event stream.
tests/golden/loop_golden.json was recorded by running the reference's Experiment with the scripted
env / agent of tests/golden/loop_scenario.py (generator: tests/golden/gen_loop_golden.py); here the same
script drives this repo's Experiment and every replay push, update call, counter and run_stats entry
must agree."""
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

import arg_utils

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import loop_scenario as sc  # noqa: E402

import recovery_rl_amd.experiment as rexp  # noqa: E402
from recovery_rl_amd.spaces import Box  # noqa: E402


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "loop_golden.json")))


class StubMem:
    def __init__(self, capacity, seed, device="cpu"):
        self.rows = []

    def push(self, state, action, reward, next_state, done, valid=None):
        n = reward.shape[0]
        for i in range(n):
            if valid is not None and not bool(valid[i]):
                continue
            self.rows.append([state[i].double().tolist(), action[i].double().tolist(), float(reward[i]),
                              next_state[i].double().tolist(), float(done[i])])

    def __len__(self):
        return len(self.rows)

    def check_error(self):
        pass


def run_mine(name, extra, tmp_path, monkeypatch):
    script = sc.Script()
    log = {"sac_updates": [], "qrisk_updates": []}
    t32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).reshape(1, 2)

    class Env:
        num_envs, auto_reset, device = 1, False, torch.device("cpu")
        _max_episode_steps = horizon = sc.HORIZON
        action_space, observation_space = Box(-np.ones(2), np.ones(2)), Box(-np.ones(2), np.ones(2))

        def __init__(self):
            self.pos = torch.zeros(1, 2, dtype=torch.float64)
            self.transition_function = self.offline

        def seed(self, s=None):
            pass

        def sample_actions(self):
            return t32(script.random_action())

        def reset(self):
            self.pos[0] = torch.as_tensor(script.start_episode())
            self.obs = self.pos.float()
            return self.obs

        def step(self, action):
            prev = self.obs.clone()
            nxt, r, done, cons, succ = script.transition(self.pos[0].numpy(), action[0].numpy())
            self.pos[0] = torch.as_tensor(nxt)
            self.obs = self.pos.float()
            u8 = lambda v: torch.tensor([int(v)], dtype=torch.uint8)
            info = {"constraint": u8(cons), "reward": torch.tensor([r], dtype=torch.float32), "state": prev,
                    "next_state": self.obs.clone(), "action": action.clone(), "success": u8(succ),
                    "ep_done": u8(done)}
            return self.obs, info["reward"], u8(done), info

        def offline(self, num, task_demos=False):
            rows = script.offline_data(num)
            col = lambda i, w: torch.as_tensor(np.array([np.asarray(r[i], dtype=np.float32) for r in rows]).reshape(len(rows), *w))
            return col(0, (2,)), col(1, (2,)), col(2, ()), col(3, (2,)), col(4, ())

    class SafetyCritic:
        def update_parameters(self, memory=None, policy=None, batch_size=None, plot=False):
            log["qrisk_updates"].append([len(memory), batch_size])

        def get_value(self, s, a):
            return torch.tensor([[script.risk()]])

        def select_action(self, state, eval=False):
            return t32(script.recovery_action())

    class Agent:
        policy, safety_critic, fast = object(), SafetyCritic(), None

        def select_action(self, state, eval=False):
            return t32(script.task_action())

        def update_parameters(self, memory, batch_size, updates, nu=None, safety_critic=None):
            log["sac_updates"].append([len(memory), batch_size, updates])

    def setup(self):
        self.device = torch.device("cpu")
        self.env, self.agent, self.recovery_policy = Env(), Agent(), None

    monkeypatch.setattr(rexp.Experiment, "experiment_setup", setup)
    monkeypatch.setattr(rexp, "ReplayMemory", StubMem)
    monkeypatch.setattr(rexp, "ConstraintReplayMemory", StubMem)
    cfg = arg_utils.get_args(sc.BASE_ARGV + ["--logdir", str(tmp_path)] + extra)
    cfg.no_fast_path = True
    exp = rexp.Experiment(cfg)
    exp.run()
    stats = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    return exp, log, stats


def close(a, b):
    return np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(sc.VARIANTS))
def test_single_env_loop_reproduces_the_reference_event_stream(name, G, tmp_path, monkeypatch, capsys):
    g = G[name]
    exp, log, stats = run_mine(name, sc.VARIANTS[name], tmp_path, monkeypatch)
    out = capsys.readouterr().out
    # every push of both buffers, in order: (state, action, reward, next_state, mask)
    for mine, ref, what in ((exp.memory.rows, g["memory"], "memory"),
                            (exp.recovery_memory.rows, g["recovery_memory"], "recovery_memory")):
        assert len(mine) == len(ref), what
        for i, (m, r) in enumerate(zip(mine, ref)):
            for a, b in zip(m, r):
                assert close(a, b), (what, i, m, r)
    # every update call with the buffer length it saw, its batch size and the running update index
    assert log["sac_updates"] == g["sac_updates"]
    assert log["qrisk_updates"] == g["qrisk_updates"]
    for k, v in g["counters"].items():
        assert int(getattr(exp, k)) == v, k
    # run_stats.pkl: per-episode, per-step info incl. the recovery flag
    assert len(stats["train_stats"]) == len(g["train_stats"]) and len(stats["test_stats"]) == g["n_test_rollouts"]
    for ep_m, ep_r in zip(stats["train_stats"], g["train_stats"]):
        assert len(ep_m) == len(ep_r)
        for sm, sr in zip(ep_m, ep_r):
            assert set(sm) == set(sr)
            assert int(sm["constraint"]) == int(sr["constraint"]) and bool(sm["success"]) == bool(sr["success"])
            assert bool(sm["recovery"]) == bool(sr["recovery"])
            assert close(sm["reward"], sr["reward"]) and close(sm["state"], sr["state"])
            assert close(sm["next_state"], sr["next_state"]) and close(sm["action"], sr["action"])
    # the printed episode lines are identical
    mine_lines = [l for l in out.splitlines() if l.startswith(("Episode:", "Num ", "Violations "))]
    assert mine_lines == g["episode_lines"]


def test_scenario_exercises_the_interesting_cases(G):
    g = G["mf"]
    masks = [row[4] for row in g["memory"]]
    lens = [len(ep) for ep in g["train_stats"]]
    assert sc.HORIZON in lens and min(lens) == 1
    # time-out at the horizon keeps mask = 1 (experiment.py:434-435), constraint / success ends give mask = 0
    assert masks[sc.HORIZON - 1] == 1.0 and 0.0 in masks
    assert any(s["recovery"] for ep in g["train_stats"] for s in ep)
    assert len(G["mf_both"]["memory"]) > len(g["memory"])           # add_both_transitions pushes extra rows
    assert G["mf_norelabel"]["memory"] != g["memory"]               # relabelling changes the stored action
    assert G["penalty"]["recovery_memory"] == [] and G["online_off"]["qrisk_updates"] != g["qrisk_updates"]
