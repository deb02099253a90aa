"""What does a kernel boundary inside a replayed hipGraph cost, and does the size of the by-value argument block matter?
Chains of 20 dependent launches of (a) rrl_counter_add (24 B of kernel arguments) and (b) rrl_adam_step_multi over 8
four-element segments (a ~1 KB argument block, a few hundred bytes of traffic), in both graph replay modes.
    python profiles/kernarg_probe.py [default]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "default":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
import recovery_rl_amd  # noqa: E402,F401
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
ctr = torch.zeros(2, dtype=torch.int64, device=dev)
K = 20
bufs = [torch.zeros(4, device=dev) for _ in range(8 * 4)]
steps = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(8)]
segs = (_lib.rrl_adam_seg_t * 8)()
for k in range(8):
    p, g, m, v = bufs[4 * k:4 * k + 4]
    segs[k] = _lib.rrl_adam_seg_t(4, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), steps[k].data_ptr(), None,
                                  0.0, 0.0, None, None, 0, 0, 0)


def small():
    for _ in range(K):
        lib.rrl_counter_add(_lib.ptr(ctr), 1, _lib.current_stream())


def big():
    for _ in range(K):
        assert lib.rrl_adam_step_multi(8, segs, 1e-3, 0.9, 0.999, 1e-8, _lib.current_stream()) == 0


def timed(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / K


print({"graph_packet_capture": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"),
       "us_per_launch_small_args": round(timed(small), 2), "us_per_launch_1KB_args": round(timed(big), 2)})
