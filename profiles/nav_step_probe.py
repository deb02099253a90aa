"""Where the env step's time goes in the bandwidth regime: rrl_nav_step / rrl_nav_step_compact at N envs, all variants
timed in ONE process, interleaved, best of 3 rounds (box-to-box and run-to-run differences are ~8 %, larger than most of
the effects).  Each timing: reset, 30 untimed steps (termination rate settles at ~1.2 % per step), 20 timed steps.
    python profiles/nav_step_probe.py [log2 N = 24]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
dev = torch.device("cuda:0")
lib = _lib.load()
env = make_vec_env("navigation1", n, device=dev, seed=1)
act = torch.rand(n, 2, device=dev) * 2 - 1
noise = torch.randn(n, 2, dtype=torch.float64, device=dev)
status = torch.zeros(n, dtype=torch.int16, device=dev)


def general(nz=None, auto=1):
    return lambda: lib.rrl_nav_step(0, n, _lib.ptr(env.pos), _lib.ptr(act), _lib.ptr(nz), 1, 0, _lib.ptr(env.tick), 1,
                                    _lib.ptr(env.next_obs), _lib.ptr(env.obs), _lib.ptr(env.reward), _lib.ptr(env.done),
                                    _lib.ptr(env.constraint), _lib.ptr(env.success), _lib.ptr(env.ep_done),
                                    _lib.ptr(env.t), 100, auto, _lib.current_stream())


def compact(nz=None, auto=1, reset_obs=True):
    return lambda: lib.rrl_nav_step_compact(0, n, _lib.ptr(env.pos), _lib.ptr(act), _lib.ptr(nz), 1, 0,
                                            _lib.ptr(env.tick), 1, _lib.ptr(env.next_obs),
                                            _lib.ptr(env.obs) if reset_obs else None,
                                            _lib.ptr(env.reward), _lib.ptr(status), 100, auto, _lib.current_stream())


def timeit(f, reps=20):
    env.reset()
    status.zero_()
    for _ in range(30):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cases = [("general", general()), ("general_no_auto_reset", general(auto=0)), ("general_injected_noise", general(noise)),
         ("compact", compact(reset_obs=False)), ("compact_with_reset_obs", compact()),
         ("compact_no_auto_reset", compact(auto=0, reset_obs=False)),
         ("compact_injected_noise", compact(noise, reset_obs=False))]
out = {"n": n}
for rnd in range(3):
    for name, f in cases:
        out[name + "_us"] = round(min(timeit(f), out.get(name + "_us", 1e9)), 1)
a = torch.empty(n * 7, dtype=torch.float64, device=dev)   # a copy of the compact layout's footprint: 56 B per env each way
b = torch.empty_like(a)
out["copy_56B_per_env_each_way_us"] = round(timeit(lambda: b.copy_(a) is None and 0), 1)
print(json.dumps(out, indent=1))
