"""Phase stamps (s_memtime, wave 0 of every workgroup) of the column-split stack forward: builds a second library with
-DRRL_FWD_TIMING next to the product one and prints, per phase, the mean cycles over the workgroups.
    python profiles/mlp_fwd_timing.py [M] [G]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/librrl_hip_timing.so"
os.environ["RRL_HIP_LIB"] = so
from recovery_rl_amd import _lib  # noqa: E402

# RRL_TIMING_FLAGS: extra compile flags for the timing build, e.g. "-DRRL_COALESCE_W2=1 -DRRL_SPLIT_PAD=4"
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + _lib.HIPCC_FLAGS + ["-DRRL_FWD_TIMING"] +
                      os.environ.get("RRL_TIMING_FLAGS", "").split() + ["-I", _lib.INCLUDE,
                      "-o", so] + _lib._sources())
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recovery_rl_amd.fast_update import FlatNet, Stack  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
din = 4 if G == 2 else 2
f = FlatNet([("W1", (G, 256, din)), ("b1", (G, 256)), ("W2", (G, 256, 256)), ("b2", (G, 256)), ("W3", (G, 1, 256)),
             ("b3", (G, 1))], dev)
f.flat.normal_(0, 0.05)
f.G, f.H, f.din, f.dout = G, 256, din, 1
st = Stack(f, M)
x = torch.randn(M, din, device=dev)
for _ in range(3):
    st.forward(x, save=False)
torch.cuda.synchronize()
lib = _lib.load()
lib.rrl_debug_fwd_stamps.argtypes = [C.c_void_p, C.c_int]
rows = 32 if M > 1024 else 16
n_blocks = min(8192, ((M + rows - 1) // rows) * G * 4)
buf = np.zeros((n_blocks, 8), np.uint64)
assert lib.rrl_debug_fwd_stamps(buf.ctypes.data, n_blocks) == 0
t = buf.astype(np.int64)
names = ["args->loads issued", "loads issued->layer 1 + h1 stores", "sync->layer 2 MFMAs + h2 stores", "sync", "layer 3 + output"]
print("M=%d G=%d workgroups=%d (rows per workgroup %d); s_memtime ticks = shader cycles" % (M, G, n_blocks, rows))
for k, nme in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print("  %-40s mean %8.1f  min %6d  max %6d ticks" % (nme, d.mean(), d.min(), d.max()))
tot = t[:, 5] - t[:, 0]
print("  workgroup total mean %.1f ticks; first start -> last end %d ticks" % (tot.mean(), t[:, 5].max() - t[:, 0].min()))
# the cycle counters are not comparable across workgroups (per-CU / per-XCD counters); slots 6 / 7 hold the global 100 MHz
# real-time counter at the workgroup's start and end (10 ns per tick)
rs, re = t[:, 6], t[:, 7]
print("  real time (10 ns ticks): starts spread over %d ticks, first start -> last end %d ticks, mean workgroup %.1f ticks"
      % (rs.max() - rs.min(), re.max() - rs.min(), (re - rs).mean()))
q = np.percentile(rs - rs.min(), [10, 50, 90, 99])
print("  start time of the workgroups after the first one: 10 %% %d, 50 %% %d, 90 %% %d, 99 %% %d ticks" % tuple(q))
order = np.argsort(t[:, 0])
print("  start spread: %d ticks between the first and the last workgroup start" % (t[:, 0].max() - t[:, 0].min()))
