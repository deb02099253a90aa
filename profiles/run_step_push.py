"""Launch rrl_nav_step_push R times at N envs (profiling target for rocprofv3 --pmc / --kernel-trace): the env step +
two replay pushes + episode counters kernel of the timed iteration."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402
from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
env = make_vec_env("navigation1", n, device=dev, seed=1)
env.reset()
act = torch.rand(n, 2, device=dev) * 2 - 1
real = torch.rand(n, 2, device=dev) * 2 - 1
rec = (torch.rand(n, device=dev) < 0.2).to(torch.uint8)
cap = min(max(1000000, 2 * n), 1 << 21)
mem, rmem = ReplayMemory(max(cap, n), 1, device=dev), ConstraintReplayMemory(cap if cap >= n else 1 << 21, 1, device=dev)
use_rmem = n <= rmem.capacity
stats = torch.zeros(10, dtype=torch.int64, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
ep_reward = torch.zeros(n, device=dev)
lib = _lib.load()
for _ in range(reps):
    rc = lib.rrl_nav_step_push(0, n, _lib.ptr(env.pos), _lib.ptr(env.t), _lib.ptr(env.obs), _lib.ptr(act), _lib.ptr(real),
                               _lib.ptr(rec), 1, 0, _lib.ptr(env.tick), 1, 100, 1, 0.0, 0, C.byref(mem._desc),
                               C.byref(rmem._desc) if use_rmem else None, _lib.ptr(env.next_obs), _lib.ptr(env.reward),
                               _lib.ptr(env.done), _lib.ptr(env.constraint), _lib.ptr(env.success), _lib.ptr(env.ep_done),
                               _lib.ptr(stats), _lib.ptr(sums), _lib.ptr(ep_reward), _lib.current_stream())
    assert rc == 0, rc
torch.cuda.synchronize()
print("ok", n, reps, "recovery buffer" if use_rmem else "task buffer only")
