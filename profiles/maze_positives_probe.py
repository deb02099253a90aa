"""How many constraint-positive rows the Maze safety buffer holds while 4096 lock-step envs overwrite the ring
(capacity 1e6 = 244 iterations): the stratified sampler needs int(256 * 0.3) = 76 of them per batch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import arg_utils  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = arg_utils.get_args(bench.config_argv("maze", 1, n))
loop = bench.build_loop(cfg, torch.device("cuda:0"))
rm = loop.recovery_memory
rows = []
for it in range(1500):
    loop.vector_step(True, False, True)
    if it % 100 == 0:
        st = loop.stats.cpu().tolist()
        rows.append({"iteration": it, "positives_in_ring": int(rm.pos_cnt.sum().item()), "ring_size": int(rm.state[1].item()),
                     "error_flag": int(rm.state[3].item()), "episodes": st[1], "violations": st[2],
                     "constraint_steps": st[7], "recovery_steps": st[6]})
        print(json.dumps(rows[-1]))
