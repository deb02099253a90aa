"""G11: per-run metrics of the reference's plotting code (plotting/plot_runs.py:140-312) on a synthetic
run_stats.pkl.  The reference's `plot_experiment` is run unmodified for each PLOT_TYPE with a recording
stand-in for the matplotlib axes; the curve it hands to `axs.plot` (mean over ONE run = the run's own metric)
is the expected output.  Fixture: tests/golden/plot_golden.npz (inputs: episode lengths, per-step reward and
constraint; outputs: the four curves).

Run: python tests/golden/gen_plot_golden.py
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
sys.path.insert(0, os.path.join(_ref_shims.REFERENCE_ROOT, "plotting"))


class _Axes:
    def __init__(self):
        self.curves = []

    def plot(self, y, **k):
        self.curves.append(np.asarray(y, dtype=np.float64))

    def __getattr__(self, name):
        return lambda *a, **k: None


def main():
    rng = np.random.RandomState(11)
    E = 340                                     # > eps["navigation1"] = 300: the [:max_eps] cut is exercised
    lengths = rng.randint(1, 40, size=E)
    total = int(lengths.sum())
    reward = -rng.uniform(0.0, 60.0, size=total).astype(np.float32).astype(np.float64)
    # make a third of the episodes end inside the goal radius
    ends = np.cumsum(lengths) - 1
    reward[ends[rng.rand(E) < 0.35]] = -rng.uniform(0, 3.9)
    reward = reward.astype(np.float32).astype(np.float64)     # f32-representable, as this stack's rewards are
    constraint = (rng.rand(total) < 0.004).astype(np.int64)
    constraint[: ends[60]] = 0                  # violations start late, so the reward curve has real values
    train_stats, k = [], 0
    for L in lengths:
        train_stats.append([{"constraint": int(constraint[k + j]), "reward": float(reward[k + j])} for j in range(L)])
        k += L
    root = tempfile.mkdtemp()
    run_dir = os.path.join(root, "2021-01-01_00-00-00_SAC_navigation1_Gaussian_RRL_MF")
    os.makedirs(run_dir)
    pickle.dump({"train_stats": train_stats, "test_stats": []}, open(os.path.join(run_dir, "run_stats.pkl"), "wb"))
    os.chdir(tempfile.mkdtemp())                # logdir must hold run directories only
    import plot_runs
    out = {}
    for kind in ("ratio", "success", "violation", "reward"):
        ax = _Axes()
        plot_runs.PLOT_TYPE = kind
        plot_runs.plt.subplots = lambda *a, **k: (None, ax)
        plot_runs.plt.subplots_adjust = lambda *a, **k: None
        plot_runs.plt.savefig = lambda *a, **k: None
        plot_runs.plt.show = lambda *a, **k: None
        plot_runs.plot_experiment("navigation1", root)
        assert len(ax.curves) == 1
        out[kind] = ax.curves[0]
        print(kind, ax.curves[0].shape, ax.curves[0][-3:])
    np.savez_compressed(os.path.join(HERE, "plot_golden.npz"), lengths=lengths, reward=reward,
                        constraint=constraint, **{"curve_" + k: v for k, v in out.items()})


if __name__ == "__main__":
    main()
