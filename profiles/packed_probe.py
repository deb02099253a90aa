"""Seed packing (recovery_rl_amd/packed.py) on one MI355X: aggregate rates for S seeds sharing every launch.
    python profiles/packed_probe.py [updates_per_step=1] [S list = 1,2,4,8] [steps = 300]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (also applies the bench's runtime setting before HIP starts)
import torch  # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seeds = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
a = argparse.Namespace(env="navigation1", num_envs=4096, steps=int(sys.argv[3]) if len(sys.argv) > 3 else 300, warmup=20)
bench.run_seed_pack_leg(argparse.Namespace(env="navigation1", num_envs=4096, steps=3000, warmup=20), torch.device("cuda:0"), seeds=(1,),
                        updates_per_step=1)       # untimed: the first process on a fresh box runs slow for a second or two
out = bench.run_seed_pack_leg(a, torch.device("cuda:0"), seeds=seeds, updates_per_step=U)
for r in out:
    print(r, file=sys.stderr)
print(json.dumps(out))
