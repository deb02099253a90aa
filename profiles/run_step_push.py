"""Launch the step + push kernel R times at N envs (profiling target for rocprofv3 --pmc / --kernel-trace): the env step +
two replay pushes + episode counters kernel of the timed iteration, through the launcher bench.py times.
    python profiles/run_step_push.py [N=4096] [R=50] [compact_log|compact|arrays]
compact_log (round 4: what the timed graph launches): compact + the per-episode table advanced by the launch;
compact: u16 status word, stored state from pos, no per-env output arrays;
arrays: the reference-shaped i32 step count + four u8 flags + every optional output (the round-2 measurement)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
layout = sys.argv[3] if len(sys.argv) > 3 else "compact"
dev = torch.device("cuda:0")
launch = bench.step_push_launcher(dev, "navigation1", n, compact=layout.startswith("compact"), log=layout == "compact_log")
for _ in range(reps):
    rc = launch()
    assert rc == 0, rc
torch.cuda.synchronize()
print("ok", n, reps, layout)
