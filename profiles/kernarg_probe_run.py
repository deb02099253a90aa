"""Runner of profiles/kernarg_probe.hip: per-launch cost of do-nothing kernels inside a replayed graph by size of the
by-value argument block, in both graph replay modes.
    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o profiles/_ab_kernarg.so profiles/kernarg_probe.hip
    python profiles/kernarg_probe_run.py [default]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "default":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
import recovery_rl_amd  # noqa: E402,F401
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab_kernarg.so"))
for f in (lib.probe_small, lib.probe_mid, lib.probe_big):
    f.argtypes = [C.c_void_p, C.c_void_p]
lib.probe_ptr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(4, dtype=torch.int64, device=dev)
blob = torch.zeros(256, dtype=torch.int64, device=dev)
K = 20


def timed(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / reps / K, 2)


st = _lib.current_stream
print({"graph_packet_capture": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"),
       "us_per_launch_16B": timed(lambda: lib.probe_small(out.data_ptr(), st())),
       "us_per_launch_512B": timed(lambda: lib.probe_mid(out.data_ptr(), st())),
       "us_per_launch_2KB": timed(lambda: lib.probe_big(out.data_ptr(), st())),
       "us_per_launch_2KB_behind_a_pointer": timed(lambda: lib.probe_ptr(out.data_ptr(), blob.data_ptr(), st()))})
