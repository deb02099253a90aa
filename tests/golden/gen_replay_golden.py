"""Generate golden fixture G6 (replay semantics) by IMPORTING the reference's
recovery_rl/replay_memory.py in this container.

Run: python tests/golden/gen_replay_golden.py  ->  tests/golden/replay_golden.npz (data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()
from recovery_rl.replay_memory import ConstraintReplayMemory, ReplayMemory  # noqa: E402


def main():
    rng = np.random.RandomState(7)
    out = {}
    # (a) stratified composition: positives first, then negatives
    n, B, pf = 3000, 256, 0.3
    constraint = (rng.uniform(size=n) < 0.1).astype(np.float64)
    mem = ConstraintReplayMemory(4096, 11)
    for i in range(n):
        mem.push(np.array([i, 0.0]), np.array([0.0, i]), constraint[i], np.array([i, 1.0]), 1.0)
    s, a, c, s2, m = mem.sample(B, pos_fraction=pf)
    out.update(constraint=constraint.astype(np.uint8), B=B, pos_fraction=pf,
               n_pos_ref=int(c.sum()), ref_batch_constraint=c.astype(np.uint8),
               ref_batch_slots=s[:, 0].astype(np.int64))
    assert len(set(s[:, 0])) == B
    # (b) uniform sampling is without replacement
    s, a, c, s2, m = mem.sample(B)
    out["ref_uniform_distinct"] = int(len(set(s[:, 0])))
    # (c) ring semantics: capacity 10, 13 pushes
    rm = ReplayMemory(10, 3)
    for i in range(13):
        rm.push(np.array([i, i]), np.array([-i, -i]), float(100 + i), np.array([i + 0.5, i + 0.5]), float(i % 2))
    out["ring_len"] = len(rm)
    out["ring_position"] = rm.position
    out["ring_rewards_by_slot"] = np.array([t[2] for t in rm.buffer])
    out["ring_masks_by_slot"] = np.array([t[4] for t in rm.buffer])
    # (d) sample(B) with B > len raises ValueError
    try:
        rm.sample(11)
        out["oversample_raises"] = 0
    except ValueError:
        out["oversample_raises"] = 1
    np.savez_compressed(os.path.join(HERE, "replay_golden.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
