"""Launch rrl_nav_step_compact R times at N envs (profiling target for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = make_vec_env("navigation1", n, device="cuda:0", seed=1)
env.reset()
act = torch.rand(n, 2, device="cuda:0") * 2 - 1
status = torch.zeros(n, dtype=torch.int16, device="cuda:0")
lib = _lib.load()
for _ in range(reps):
    rc = lib.rrl_nav_step_compact(0, n, _lib.ptr(env.pos), _lib.ptr(act), None, 1, 0, _lib.ptr(env.tick), 1,
                                  _lib.ptr(env.next_obs), None, _lib.ptr(env.reward), _lib.ptr(status),
                                  100, 1, _lib.current_stream())
    assert rc == 0
torch.cuda.synchronize()
print("ok", n, reps)
