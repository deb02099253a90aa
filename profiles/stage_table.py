"""The stage table of bench.roofline_stages alone (every recorded launch of one config-2 iteration re-issued 50x in its own
graph): one row per launch.    python profiles/stage_table.py [num_envs=4096]"""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with contextlib.redirect_stdout(sys.stderr):
    st = bench.roofline_stages(argparse.Namespace(env="navigation1", num_envs=n), torch.device("cuda:0"), 0.0)
for r in st["stages"]:
    print("%-70s %-30s %7.2f us  frac %.3f" % (r["stage"], r["kernel"], r["us"], r["frac"]))
print("sum %.1f us, %d launches" % (st["stand_alone_sum_us"], st["launches"]))
print(json.dumps(st["by_kernel"]))
