"""Latency of the fused step + push kernel at 4096 envs, Maze next to Navigation 1, with and without auto-reset and with
zero actions (Maze: no move -> no collision search).  HIP events around a captured graph of 200 launches."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402
from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def time_case(env_name, auto_reset, zero_action, with_rmem=True):
    env = make_vec_env(env_name, n, device=dev, seed=1)
    env.reset()
    hi = float(env.action_space.high[0])
    act = (torch.rand(n, 2, device=dev) * 2 - 1) * hi
    if zero_action:
        act.zero_()
    real = act.clone()
    rec = torch.zeros(n, dtype=torch.uint8, device=dev)
    mem, rmem = ReplayMemory(1000000, 1, device=dev), ConstraintReplayMemory(1000000, 1, device=dev)
    stats = torch.zeros(10, dtype=torch.int64, device=dev)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    ep_reward = torch.zeros(n, device=dev)
    if env_name == "maze":
        entry, head = lib.rrl_maze_step_push, ()
    else:
        entry, head = lib.rrl_nav_step_push, (env.kind,)

    def launch():
        return entry(*head, n, _lib.ptr(env.pos), _lib.ptr(env.t), _lib.ptr(env.obs), _lib.ptr(act), _lib.ptr(real),
                     _lib.ptr(rec), env.seed_value, 0, _lib.ptr(env.tick), 1, env.horizon, int(auto_reset), 0.0, 0,
                     C.byref(mem._desc), C.byref(rmem._desc) if with_rmem else None, _lib.ptr(env.next_obs),
                     _lib.ptr(env.reward), _lib.ptr(env.done), _lib.ptr(env.constraint), _lib.ptr(env.success),
                     _lib.ptr(env.ep_done), _lib.ptr(stats), _lib.ptr(sums), _lib.ptr(ep_reward), _lib.current_stream())
    for _ in range(10):
        assert launch() == 0
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            launch()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / 200, 2)


out = {"n": n}
for env_name in ("navigation1", "maze"):
    for auto in (1, 0):
        for zero in (0, 1):
            out["%s_auto%d_zero%d_us" % (env_name, auto, zero)] = time_case(env_name, auto, zero)
    out["%s_no_safety_buffer_us" % env_name] = time_case(env_name, 1, 0, with_rmem=False)
print(json.dumps(out, indent=1))
