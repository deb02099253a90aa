"""A/B of the iteration with non-temporal stores for the stage-to-stage intermediates (saved activations, dh2, weight
gradients): builds a second library with -DRRL_NT_STORES next to the product one and runs the headline leg on each.
    python profiles/nt_store_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recovery_rl_amd import _lib  # noqa: E402

alt = "/tmp/librrl_hip_nt.so"
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + _lib.HIPCC_FLAGS + ["-DRRL_NT_STORES", "-I", _lib.INCLUDE,
                      "-o", alt] + _lib._sources())
out = {}
for name, lib in (("default", ""), ("nt_stores", alt), ("default_again", ""), ("nt_stores_again", alt)):
    env = dict(os.environ)
    if lib:
        env["RRL_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no_legs", "--no_cpu_baseline", "--no_planner",
                        "--min_seconds", "1.5"], env=env, capture_output=True, text=True)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    out[name] = {"ms_per_step": line["ms_per_step"], "env_steps_per_s": line["value"]}
    print(name, out[name], file=sys.stderr)
print(json.dumps(out))
