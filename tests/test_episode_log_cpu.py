"""Episode-record metrics: the restatement of plotting/plot_runs.py:194-312 against curves
recorded from the reference's own plotting code (G11, tests/golden/gen_plot_golden.py), for both the
reference's per-step schema and the per-episode table this stack writes."""
import os

import numpy as np

from oracle.log_oracle import EpisodeLogOracle
from recovery_rl_amd.episode_log import EPISODE_DTYPE, plot_curves, records_from_train_stats

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "plot_golden.npz"))


def _train_stats():
    out, k = [], 0
    for L in GOLD["lengths"]:
        out.append([{"constraint": int(GOLD["constraint"][k + j]), "reward": float(GOLD["reward"][k + j])}
                    for j in range(L)])
        k += L
    return out


def _check(curves):
    for kind in ("ratio", "success", "violation", "reward"):
        np.testing.assert_array_equal(curves[kind], GOLD["curve_" + kind], err_msg=kind)


def test_metrics_from_reference_schema_equal_the_reference_plot_curves():
    _check(plot_curves({"train_stats": _train_stats()}, "navigation1", max_eps=300))


def test_metrics_from_episode_table_equal_the_reference_plot_curves():
    """One env, the golden steps fed through the episode-log checker: the per-episode table carries
    everything the plots need."""
    log = EpisodeLogOracle(1)
    k = 0
    for L in GOLD["lengths"]:
        for j in range(L):
            r = np.float32(GOLD["reward"][k + j])
            log.append([r], [GOLD["constraint"][k + j]], [r > -4], [j == L - 1])
        k += L
    rec = np.array(log.records, dtype=EPISODE_DTYPE)
    assert len(rec) == len(GOLD["lengths"])
    np.testing.assert_array_equal(rec["length"], GOLD["lengths"])
    _check(plot_curves({"episode_stats": rec}, "navigation1", max_eps=300))
    ref = records_from_train_stats(_train_stats())
    for name in ("length", "constraint_steps", "ret", "last_reward"):
        np.testing.assert_array_equal(rec[name], ref[name], err_msg=name)


def test_info_ring_splits_steps_into_reference_episodes():
    """InfoRing (the `--info_envs K` stream) on host tensors: K envs' step rows, drained in two parts, come back as the
    reference's list of episodes of step dicts in completion order, with an unfinished episode carried across the drains."""
    import types

    import torch

    from recovery_rl_amd.episode_log import InfoRing
    K, n = 2, 5
    ring = InfoRing(K, 8, "cpu", action_high=1.0)
    rng = np.random.default_rng(0)
    ends = {0: {2, 5}, 1: {3}}                                   # env -> steps at which its episode ends
    expected = {0: [[]], 1: [[]]}
    drained = []
    obs = torch.tensor(rng.normal(size=(n, 2)), dtype=torch.float32)
    for t in range(7):
        ring.before_step(obs)
        act = torch.tensor(rng.normal(size=(n, 2)) * 2, dtype=torch.float32)
        nxt = torch.tensor(rng.normal(size=(n, 2)), dtype=torch.float32)
        env = types.SimpleNamespace(
            next_obs=nxt, reward=torch.tensor(rng.normal(size=n), dtype=torch.float32),
            constraint=torch.tensor([t in ends.get(e, ()) and e == 0 for e in range(n)], dtype=torch.uint8),
            success=torch.tensor([t == 3 and e == 1 for e in range(n)], dtype=torch.uint8),
            ep_done=torch.tensor([t in ends.get(e, ()) for e in range(n)], dtype=torch.uint8))
        rec = torch.tensor([(t + e) % 2 for e in range(n)], dtype=torch.uint8)
        ring.after_step(env, act, rec)
        for e in range(K):
            expected[e][-1].append((obs[e].numpy().copy(), act[e].clamp(-1, 1).numpy(), nxt[e].numpy(),
                                    float(env.reward[e]), bool(rec[e])))
            if t in ends[e]:
                expected[e].append([])
        obs = nxt
        if t == 3:
            drained += ring.drain()
    drained += ring.drain()
    # completion order: (t=2, env 0), (t=3, env 1), (t=5, env 0)
    assert [len(ep) for ep in drained] == [3, 4, 3]
    for ep, (e, k) in zip(drained, ((0, 0), (1, 0), (0, 1))):
        for step, (s, a, s2, r, rec) in zip(ep, expected[e][k]):
            np.testing.assert_array_equal(step["state"], s)
            np.testing.assert_array_equal(step["action"], a)
            np.testing.assert_array_equal(step["next_state"], s2)
            assert step["reward"] == r and step["recovery"] == rec
        assert ep[-1]["constraint"] == (1 if e == 0 else 0) and ep[-1]["success"] == (e == 1)
    assert len(ring.open[0]) == 1 and len(ring.open[1]) == 3     # unfinished episodes stay open
    rec = records_from_train_stats(drained)
    np.testing.assert_array_equal(rec["length"], [3, 4, 3])


def test_info_ring_started_mid_episode_drops_the_partial_first_episodes():
    import types

    import torch

    from recovery_rl_amd.episode_log import InfoRing
    ring = InfoRing(1, 8, "cpu", action_high=1.0, mid_episode=True)
    z2 = torch.zeros(1, 2)
    for t in range(6):
        ring.before_step(z2)
        env = types.SimpleNamespace(next_obs=z2, reward=torch.full((1,), float(t)), constraint=torch.zeros(1, dtype=torch.uint8),
                                    success=torch.zeros(1, dtype=torch.uint8),
                                    ep_done=torch.tensor([t in (1, 4)], dtype=torch.uint8))
        ring.after_step(env, z2, None)
    eps = ring.drain()
    assert [[s["reward"] for s in ep] for ep in eps] == [[2.0, 3.0, 4.0]]      # steps 0-1 belonged to an episode begun earlier
    assert len(ring.open[0]) == 1
