"""The offline constraint demonstrations the REFERENCE draws for the model-based line (scripts/navigation2.sh:14):
`Experiment.constraint_demo_data` of the imported reference (recovery_rl/experiment.py:186-250 -> env/navigation2.py:133-243,
the global MT19937 stream after the reference's own seeding) for seeds 1, 3, 4.

Why a fixture: a seed fixes the same initial networks on both stacks (torch.manual_seed) but NOT the same demonstrations
(this stack draws them from Philox streams).  Round 6 found that how long a run stays in the stalemate between task policy and
recovery controller follows the demonstration set; running this stack on the reference's set of the same
seed is the experiment that shows it (profiles/mb_diag.py with RRL_MB_DIAG_DEMOS; tests/test_learning_level_gpu.py).

Run: python tests/golden/gen_ref_demos.py  ->  tests/golden/ref_demos_nav2_seed1.npz (committed: seed 1, the decisive one)
                                               profiles/_ab_ref_demos.npz (seeds 1, 3, 4; not committed: 1.2 MB)
"""
import contextlib
import io
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
import numpy as np  # noqa: E402


def main():
    import arg_utils
    import recovery_rl.experiment as rexp
    out = {}
    for seed in (1, 3, 4):
        sys.argv = ["rrl_main", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                    "--logdir", tempfile.mkdtemp(), "--logdir_suffix", "RRL_MB", "--num_eps", "5", "--num_unsafe_transitions",
                    "20000", "--seed", str(seed), "--eval", ""]
        cfg = arg_utils.get_args()
        with contextlib.redirect_stdout(io.StringIO()):
            exp = rexp.Experiment(cfg)
        d = exp.constraint_demo_data
        cols = [np.array([t[i] for t in d], dtype=np.float32) for i in range(5)]
        print("seed", seed, "rows", cols[0].shape[0], "violations", int(cols[2].sum()))
        for k, v in zip("sacnm", cols):
            out["seed%d.%s" % (seed, k)] = v
    profiles = os.path.join(HERE, "..", "..", "profiles")
    if os.path.isdir(profiles):                      # (a scratch copy of this directory alone regenerates the fixture only)
        np.savez_compressed(os.path.join(profiles, "_ab_ref_demos.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "ref_demos_nav2_seed1.npz"), **{k: v for k, v in out.items() if k.startswith("seed1.")})


if __name__ == "__main__":
    main()
