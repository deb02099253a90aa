"""Self-golden for the Maze surrogate.  PARITY UNPINNED: the reference's maze needs MuJoCo 1.50
(mujoco_py), which is neither in the reference tree nor in this image, so these vectors come from
the build's OWN C oracle (oracle/rrl_oracle.c) and only guard it against regressions.

Run: python tests/golden/gen_maze_golden.py -> tests/golden/maze_oracle_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import c_oracle as co  # noqa: E402


def main():
    rng = np.random.RandomState(11)
    n = 3000
    pos = np.c_[rng.uniform(-0.29, 0.29, n), rng.uniform(-0.29, 0.29, n)]
    # rows hugging every wall face and the arena planes
    edge = []
    for cx, cy in ((-0.1, 0.42), (0.1, 0.48), (-0.1, -0.33), (0.1, -0.17)):
        for d in (0.0249, 0.025, 0.0251, 0.03):
            for side in (-1, 1):
                edge.append([cx + side * (0.005 + d), cy - 0.1])
            edge.append([cx, cy - 0.2 - d])
            edge.append([cx, cy + 0.2 + d])
    for d in (0.2749, 0.275, 0.2751):
        edge += [[d, 0.0], [-d, 0.0], [0.0, d], [0.0, -d]]
    pos = np.vstack([pos, np.array(edge)])
    act = rng.uniform(-0.15, 0.15, (len(pos), 2)).astype(np.float32)
    t = rng.randint(0, 100, len(pos)).astype(np.int32)
    o = co.maze_step(pos, act, t, seed=5, counter=3, auto_reset=True)
    out = dict(pos=pos, act=act, t=t)
    for k in ("next_pos64", "reward64", "done", "constraint", "success", "ep_done", "pos", "t", "obs"):
        out["out_" + k] = o[k]
    out["contact"] = np.array([co.maze_contact(x, y) for x, y in pos], dtype=np.uint8)
    s, a, c, s2, m = co.maze_offline(2000, 9)
    out.update(off_s=s, off_a=a, off_c=c, off_s2=s2, off_m=m)
    np.savez_compressed(os.path.join(HERE, "maze_oracle_golden.npz"), **out)
    print("rows", len(pos), "contacts", int(out["contact"].sum()), "done", int(o["done"].sum()))


if __name__ == "__main__":
    main()
