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
