"""Do two independent branches of a captured hipGraph overlap?  Two chains of K small dependent kernels (torch add_ on
separate buffers), captured (a) on one stream, (b) forked onto two streams and joined; time per replay with the graph
replay mode of the package (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) and with the runtime default."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "default":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
import recovery_rl_amd  # noqa: E402,F401
import torch  # noqa: E402

dev = torch.device("cuda:0")
K, n = 20, 1 << 16
a, b = torch.zeros(n, device=dev), torch.zeros(n, device=dev)


def chain(x):
    for _ in range(K):
        x.add_(1.0)


def timed(g, reps=50):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for _ in range(3):
    chain(a); chain(b)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    chain(a); chain(b)
side = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain(b)
    chain(a)
    main.wait_stream(side)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    chain(a)
print({"mode": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "one_chain_us": round(timed(g3), 1),
       "two_chains_one_stream_us": round(timed(g1), 1), "two_chains_forked_us": round(timed(g2), 1)})
