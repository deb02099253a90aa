"""A/B of two builds of librrl_hip.so on the SAME box (boxes of the pool differ by ~2 %): the headline leg of bench.py with the
in-tree library and with an alternative one (RRL_HIP_LIB), alternating.
    python profiles/ab_lib.py <alternative.so> [rounds=3] [extra bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
alt = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
out = {"in_tree": [], "alternative": []}
for _ in range(rounds):
    for name, lib in (("in_tree", ""), ("alternative", alt)):
        env = dict(os.environ)
        if lib:
            env["RRL_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no_legs", "--no_cpu_baseline", "--no_planner",
                            "--min_seconds", "1.5"] + extra, env=env, capture_output=True, text=True)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        out[name].append(line["ms_per_step"])
        print(name, line["ms_per_step"], file=sys.stderr)
out["alternative_so"] = alt
print(json.dumps(out))
