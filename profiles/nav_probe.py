"""nav_step at N = 2^24: how much of the launch is RNG math vs memory? (external noise / no auto-reset variants)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402

dev = "cuda:0"
lib = _lib.load()
for logn in (20, 24):
    n = 1 << logn
    env = make_vec_env("navigation1", n, device=dev, seed=1)
    env.reset()
    act = torch.rand(n, 2, device=dev) * 2 - 1
    noise = torch.randn(n, 2, dtype=torch.float64, device=dev)

    def run(ext, auto):
        def f():
            lib.rrl_nav_step(0, n, _lib.ptr(env.pos), _lib.ptr(act), _lib.ptr(noise) if ext else None, 1, 0,
                             _lib.ptr(env.tick), 1, _lib.ptr(env.next_obs), _lib.ptr(env.obs), _lib.ptr(env.reward),
                             _lib.ptr(env.done), _lib.ptr(env.constraint), _lib.ptr(env.success), _lib.ptr(env.ep_done),
                             _lib.ptr(env.t), 100, auto, _lib.current_stream())
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e6
    print("N=2^%d  philox+reset %.1f us | philox, no reset %.1f us | external noise, no reset %.1f us" %
          (logn, run(False, 1), run(False, 0), run(True, 0)))
