"""Inputs of the production-shape known-answer tests (hidden 256, batch 256, 4096 acting rows -- arg_utils.py:77,89 defaults):
everything here is re-created from seeded numpy generators, so `model_golden_256.npz` holds only what the REFERENCE
produced from these inputs (gen_model_golden_256.py), not the 1.6 MB of weights that went in.

Used by the fixture generator (imports the reference) and by tests/test_kat256_gpu.py (imports this stack): both build the
same weights, batch, noise and observations from this file.  np.random.RandomState streams are stable across numpy versions.
"""
import zlib

import numpy as np

H, B, N_ACT = 256, 256, 4096
SAMPLES = 192                     # entries of every tensor the fixture records
ARGV = ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3"]     # scripts/navigation1.sh:7


def _rs(*key):
    return np.random.RandomState(zlib.crc32(".".join(str(k) for k in key).encode()))


def weights(state_dict, tag):
    """{key: float32 array} for the linear layers of `state_dict` (a module's state_dict(), this stack's or the
    reference's: the key names and shapes are the same): Xavier-uniform weights (what model.py's weights_init_ draws) and
    biases in [-0.2, 0.2] (non-trivial biases, as the H = 16 KATs have), each tensor from its own stream keyed by
    (tag, key).  Batch-norm buffers and StochasticPolicy.log_std keep the module's values."""
    out = {}
    for key, v in state_dict.items():
        shape = tuple(v.shape)
        if not (key.startswith(("linear", "mean", "log_std_linear")) and key.endswith((".weight", ".bias"))):
            continue
        rs = _rs(tag, key)
        if key.endswith(".weight"):
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            out[key] = rs.uniform(-bound, bound, shape).astype(np.float32)
        else:
            out[key] = rs.uniform(-0.2, 0.2, shape).astype(np.float32)
    return out


def batch():
    """(s, a, r, s2, m), constraint column, eps_next, eps_pi -- float32, B rows, Navigation1-like scales."""
    rs = _rs("batch")
    s = (rs.randn(B, 2) * [20, 3] + [-30, 0]).astype(np.float32)
    a = rs.uniform(-1, 1, (B, 2)).astype(np.float32)
    r = (-np.abs(rs.randn(B)) * 30).astype(np.float32)
    s2 = (s + a + 0.05 * rs.randn(B, 2)).astype(np.float32)
    m = (rs.uniform(size=B) < 0.8).astype(np.float32)
    c = (rs.uniform(size=B) < 0.3).astype(np.float32)
    eps_next = rs.randn(B, 2).astype(np.float32)
    eps_pi = rs.randn(B, 2).astype(np.float32)
    return (s, a, r, s2, m), c, eps_next, eps_pi


def acting():
    """obs [N_ACT, 2], noise [2, N_ACT, 2] (task policy, recovery policy) of one acting pass (experiment.py:546-577)."""
    rs = _rs("acting")
    obs = (rs.randn(N_ACT, 2) * [20, 4] + [-30, 0]).astype(np.float32)
    noise = rs.randn(2, N_ACT, 2).astype(np.float32)
    return obs, noise


def sample_index(key, numel):
    """The entries of tensor `key` the fixture records (all of them for small tensors)."""
    if numel <= SAMPLES:
        return np.arange(numel)
    return np.sort(_rs("idx", key).choice(numel, SAMPLES, replace=False))
