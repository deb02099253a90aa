"""Child process of tests/test_w2_permute_gpu.py (and runnable by hand): with whatever library RRL_HIP_LIB names, run

* the fused stack forward (`rrl_mlp3_forward`, split path) on seeded inputs of several shapes -- 4096 rows, ragged row
  counts either side of the 1024-row switch to two row tiles per workgroup, the B = 256 update batch, one / two heads;
* the headline iteration (Navigation1, 4096 envs, SAC + Q_risk + model-free recovery, hidden 256, batch 256) from its
  hipGraph for a fixed number of replays: acting forwards with the policy head, grouped update launches, env step, pushes;

and write every result (activations, outputs, network parameters, env state, counters) to <out>.pt plus launch timings of the
4096-row forward and of the iteration to <out>.json.      python tests/w2_permute_probe.py <out-prefix> [replays]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arg_utils  # noqa: E402
import bench  # noqa: E402
from recovery_rl_amd import _lib, fused  # noqa: E402

DEV = "cuda:0"
SHAPES = [(4096, 4, 1, 2), (2061, 2, 4, 1), (1040, 4, 1, 2), (256, 4, 1, 2), (256, 2, 4, 1)]   # (the harness covers eight)


def forward_cases():
    out = {}
    for M, din, dout, G in SHAPES:
        g = torch.Generator(device=DEV).manual_seed(1000 * M + 10 * din + G)
        r = lambda *s: torch.randn(*s, device=DEV, generator=g)
        H = 256
        x = r(M, din) * 3
        W1, b1, W2, b2, W3, b3 = r(G, H, din), r(G, H), r(G, H, H) / 16, r(G, H), r(G, dout, H) / 16, r(G, dout)
        h1, h2 = torch.empty(G, M, H, device=DEV), torch.empty(G, M, H, device=DEV)
        o = fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, h1=h1, h2=h2, scratch=torch.empty(4, G, M, dout, device=DEV))
        key = "fwd_%d_%d_%d_%d" % (M, din, dout, G)
        out[key + "_h1"], out[key + "_h2"], out[key + "_out"] = h1.cpu(), h2.cpu(), o.cpu()
    return out


def time_forward(M=4096, din=4, dout=1, G=2, reps=300):
    g = torch.Generator(device=DEV).manual_seed(3)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    H = 256
    x, W1, b1, W2, b2, W3, b3 = r(M, din), r(G, H, din), r(G, H), r(G, H, H) / 16, r(G, H), r(G, dout, H) / 16, r(G, dout)
    scratch, o = torch.empty(4, G, M, dout, device=DEV), torch.empty(G, M, dout, device=DEV)
    run = lambda: fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=o, scratch=scratch)
    for _ in range(20):
        run()
    graph = torch.cuda.CUDAGraph()                 # back to back on the device: launch overhead of the host is not in it
    with torch.cuda.graph(graph):
        for _ in range(reps):
            run()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best            # us per forward (stack kernel + the fixed-order sum of the four partials)


def iteration(replays):
    cfg = arg_utils.get_args(bench.config2_argv(seed=11))
    device = torch.device("cuda", 0)            # as bench.main builds it
    torch.cuda.set_device(device)
    loop = bench.build_loop(cfg, device)
    loop.capture(online_qrisk=True)
    for _ in range(replays):
        loop.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        loop.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    fast = loop.agent.fast
    out = {"it_" + name: getattr(fast, name).flat.cpu() for name in ("critic", "policy", "qrisk", "recpolicy")}
    out["it_pos"], out["it_stats"] = loop.env.pos.cpu(), loop.stats.cpu()
    out["it_mem_state"], out["it_rec_state"] = loop.memory.state.cpu(), loop.recovery_memory.state.cpu()
    rows = int(loop.memory.state[1].item())                      # live rows of the replay ring
    out["it_mem_s2"], out["it_mem_a"] = loop.memory.s2[:rows].cpu(), loop.memory.a[:rows].cpu()
    return out, ms


def main():
    prefix = sys.argv[1]
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    lib = _lib.load()
    assert lib.rrl_abi_version() >= 1
    res = forward_cases()
    it, ms = iteration(replays)
    res.update(it)
    torch.save(res, prefix + ".pt")
    info = {"library": _lib.SO_PATH, "forward_4096x2_us": time_forward(), "forward_4096x1_us": time_forward(G=1, din=2, dout=4),
            "forward_256x2_us": time_forward(M=256), "ms_per_iteration": ms, "replays": replays + 500}
    with open(prefix + ".json", "w") as f:
        json.dump(info, f)
    print(json.dumps(info))


if __name__ == "__main__":
    main()
