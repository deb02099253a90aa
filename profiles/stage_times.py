"""Per-stage times of the timed iteration (bench.roofline_stages) for the library in RRL_HIP_LIB / the in-tree one:
    python profiles/stage_times.py [tag]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

a = types.SimpleNamespace(env="navigation1", num_envs=bench.NUM_ENVS)
r = bench.roofline_stages(a, torch.device("cuda:0"), 0.19)
st = r["stages"] if isinstance(r, dict) else r
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
print(tag, "sum", round(sum(s["us"] for s in st), 1), " ".join("%.2f" % s["us"] for s in st))
