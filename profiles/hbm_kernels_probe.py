"""Achieved bandwidth / rate of the other streaming kernels at sizes where they are not launch-latency bound
(DESIGN.md section 5 lists their algorithmic bytes): replay push, env reset, maze step, fused step + push, CEM
sample / elite update, replay sampling (latency).  Prints one JSON object; HIP events around 20 launches each.

    python profiles/hbm_kernels_probe.py > gpurun_out/hbm_kernels.json
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402
from recovery_rl_amd.optimizers import CEMOptimizer  # noqa: E402
from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory  # noqa: E402

dev = "cuda:0"
PEAK = 8000.0


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def entry(name, seconds, units, bytes_per_unit, unit_name):
    gbs = units * bytes_per_unit / seconds / 1e9
    return {"kernel": name, "launch_us": seconds * 1e6, "units": units, "unit": unit_name,
            "algorithmic_bytes_per_unit": bytes_per_unit, "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / PEAK}


out = []
r = lambda *s: torch.randn(*s, device=dev)
# replay push: 32 B read + 32 B written per row
n = 1 << 22
mem = ReplayMemory(1 << 23, 1, device=dev)
rows = (r(n, 2), r(n, 2), r(n), r(n, 2), (torch.rand(n, device=dev) < 0.9).float())
out.append(entry("push_kernel (rrl_replay_push), 2^22 rows", timed(lambda: mem.push(*rows)), n, 64, "row"))
cmem = ConstraintReplayMemory(1 << 21, 1, device=dev)
n2 = 1 << 20
crow = (r(n2, 2), r(n2, 2), (torch.rand(n2, device=dev) < 0.1).float(), r(n2, 2), torch.ones(n2, device=dev))
out.append(entry("push_kernel with positive counts (constraint buffer), 2^20 rows", timed(lambda: cmem.push(*crow)), n2, 64, "row"))
# sampling: latency-bound single workgroup
t = timed(lambda: mem.sample(256), 100)
out.append(entry("sample_gather_kernel B=256 (latency-bound, one workgroup)", t, 256, 64, "row"))
t = timed(lambda: cmem.sample(256, pos_fraction=0.3), 100)
out.append(entry("creplay_sample_gather_kernel B=256, capacity 2^21 (latency-bound)", t, 256, 64, "row"))
# env reset and maze step
for name, envname, nenv in (("nav_reset_kernel", "navigation1", 1 << 22), ("maze_reset_kernel", "maze", 1 << 22)):
    env = make_vec_env(envname, nenv, device=dev, seed=1)
    out.append(entry("%s (%s), 2^22 envs" % (name, envname), timed(lambda: env.reset()), nenv, 28, "env"))
env = make_vec_env("maze", 1 << 22, device=dev, seed=1)
env.reset()
act = torch.rand(1 << 22, 2, device=dev) * 0.2 - 0.1
out.append(entry("maze_step_kernel (rrl_maze_step), 2^22 envs, 64 collision sub-steps", timed(lambda: env.step(act)), 1 << 22, 39,
                 "env-step"))
# fused step + push at the benchmark size and at 2^20
for nenv in (4096, 1 << 20):
    env = make_vec_env("navigation1", nenv, device=dev, seed=1)
    env.reset()
    m1, m2 = ReplayMemory(1 << 22, 1, device=dev), ConstraintReplayMemory(1 << 21, 1, device=dev)
    stats = torch.zeros(10, dtype=torch.int64, device=dev)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    epr = torch.zeros(nenv, device=dev)
    a = torch.rand(nenv, 2, device=dev) * 2 - 1
    rec = torch.zeros(nenv, dtype=torch.uint8, device=dev)
    lib = _lib.load()

    def fused():
        return lib.rrl_nav_step_push(0, nenv, _lib.ptr(env.pos), _lib.ptr(env.t), _lib.ptr(env.obs), _lib.ptr(a), _lib.ptr(a),
                                     _lib.ptr(rec), 1, 0, _lib.ptr(env.tick), 1, 100, 1, 0.0, 0, C.byref(m1._desc),
                                     C.byref(m2._desc), _lib.ptr(env.next_obs), _lib.ptr(env.reward), _lib.ptr(env.done),
                                     _lib.ptr(env.constraint), _lib.ptr(env.success), _lib.ptr(env.ep_done),
                                     _lib.ptr(stats), _lib.ptr(sums), _lib.ptr(epr), _lib.current_stream())
    out.append(entry("step_push_kernel<Nav1> (rrl_nav_step_push), %d envs" % nenv, timed(fused), nenv, 39 + 64, "env-step"))
# CEM bookkeeping: M problems x 400 candidates x 10 dims
M = 4096
opt = CEMOptimizer(10, 1, 400, 40, lambda s: (s ** 2).sum(-1), np.ones(10), -np.ones(10), alpha=0.1, device=dev, seed=1)
mean = torch.zeros(M, 10, dtype=torch.float64, device=dev)
var = torch.full((M, 10), 0.25, dtype=torch.float64, device=dev)
t = timed(lambda: opt.obtain_solution(mean, var, iters=1), 10)
out.append(entry("cem_sample + cost + cem_update, 4096 problems x 400 x 10 (one CEM iteration)", t, M * 400, 10 * 4 * 2 + 4,
                 "candidate"))
print(json.dumps(out, indent=1))
