"""A/B of one environment switch on the SAME box: the headline leg of bench.py with and without `NAME=VALUE`, alternating.
    python profiles/ab_env.py NAME=VALUE [rounds=3] [extra bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, value = sys.argv[1].split("=", 1)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
out = {"default": [], sys.argv[1]: []}
for _ in range(rounds):
    for key in out:
        env = dict(os.environ)
        if key != "default":
            env[name] = value
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no_legs", "--no_cpu_baseline", "--no_planner",
                            "--min_seconds", "1.5"] + extra, env=env, capture_output=True, text=True)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        out[key].append(line["ms_per_step"])
        print(key, line["ms_per_step"], file=sys.stderr)
print(json.dumps(out))
