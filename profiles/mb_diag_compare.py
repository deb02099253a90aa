"""Reference (tests/golden/ref_mb_diag_seed*.json, run_reference_mb_diag.py) beside this stack (profiles/mb_diag.py output) on the
model-based recovery line, episode windows of 5: recovery steps per episode, the gate's input Q_risk(s, a_task) (mean / share
above eps_safe), successes, and the ensemble's one-step error on each new episode before its re-fit.
    python profiles/mb_diag_compare.py <ours.jsonl> [window=5]"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ours = {}
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        ours[d["seed"]] = d
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_mb_diag_seed*.json"))):
    ref = json.load(open(path))
    seed = ref["seed"]
    if seed not in ours:
        continue
    n = min(len(ref["episodes"]), len(ours[seed]["episodes"]))
    print("seed %d, %d episodes; per window of %d: recovery steps/episode | gate mean | gate on | successes | pre-fit mse" % (seed, n, W))
    for lo in range(0, n - W + 1, W):
        cells = []
        for d in (ref, ours[seed]):
            ep = d["episodes"][lo:lo + W]
            rf = [r for r in d["refits"] if lo <= r["episode"] - (0 if d is ref else 0) < lo + W]
            cells.append("%5.1f | %.3f | %.2f | %d | %.4f" % (np.mean([e["recovery_steps"] for e in ep]), np.mean([e["gate_mean"] for e in ep]),
                                                           np.mean([e["gate_on"] for e in ep]), sum(e["success"] for e in ep),
                                                           np.mean([r["before"]["mse"] for r in rf]) if rf else float("nan")))
        print("  ep %2d-%2d   REF %s    OURS %s" % (lo, lo + W - 1, cells[0], cells[1]))
