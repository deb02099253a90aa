"""Per-launch time of the small kernels of the lock-step iteration when the SAME kernel is replayed 200 times
from one hipGraph (warm instruction cache) -- compare with their in-situ averages in round1_bench_kernel_stats.csv."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recovery_rl_amd import _lib  # noqa: E402

dev = "cuda:0"
lib = _lib.load()
st = lambda: _lib.current_stream()
p = _lib.ptr


def bench(fn, n=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


r = lambda *s: torch.randn(*s, device=dev)
N, B = 4096, 256
z, ta, ra, real, tout = r(2, N), r(N, 4), r(N, 2), r(N, 2), r(N, 2)
rec = torch.zeros(N, dtype=torch.uint8, device=dev)
print("recovery_select N=4096: %.2f us" % bench(lambda: lib.rrl_recovery_select(N, p(z), 0.3, p(ta), 4, p(ra), p(real), p(rec), p(tout), st())))
out = r(9216, 2)
tick = torch.zeros(2, dtype=torch.int64, device=dev)
print("normal_fill 9216 pairs (ticket): %.2f us" % bench(lambda: lib.rrl_normal_fill(9216, 1, 0, p(tick), 1, p(out), st())))
print("normal_fill 9216 pairs (no ticket): %.2f us" % bench(lambda: lib.rrl_normal_fill(9216, 1, 0, None, 0, p(out), st())))
print("normal_fill 64 pairs (no ticket): %.2f us" % bench(lambda: lib.rrl_normal_fill(64, 1, 0, None, 0, p(out), st())))
for n in (256, 4096):
    head, eps, scale, bias, act, logp = r(n, 4), r(n, 2), torch.ones(2, device=dev), torch.zeros(2, device=dev), r(n, 4), r(n)
    print("gauss_head_fwd n=%d: %.2f us" % (n, bench(lambda: lib.rrl_gauss_head_fwd(n, p(head), 1, 0, p(eps), p(scale), p(bias), p(act), 4, p(logp), None, None, None, st()))))
    parts = r(4, n, 4)
    print("gauss_head_fwd n=%d 4 partials: %.2f us" % (n, bench(lambda: lib.rrl_gauss_head_fwd(n, p(parts), 4, n * 4, p(eps), p(scale), p(bias), p(act), 4, p(logp), None, None, None, st()))))
ctr = torch.zeros(2, dtype=torch.int64, device=dev)
print("counter_add: %.2f us" % bench(lambda: lib.rrl_counter_add(p(ctr), 1, st())))
a, b = r(N, 2), r(N, 2)
print("torch add [4096,2]: %.2f us" % bench(lambda: torch.add(a, b, out=a)))
a1 = r(64)
print("torch add [64]: %.2f us" % bench(lambda: torch.add(a1, a1, out=a1)))

# --- the acting pass as a sequence vs the sum of its kernels in isolation -----------------------------------
import arg_utils  # noqa: E402
from recovery_rl_amd.env import make_vec_env, register_env  # noqa: E402
from recovery_rl_amd.fast_update import FastActor  # noqa: E402
from recovery_rl_amd.sac import SAC  # noqa: E402

cfg = arg_utils.get_args(["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8",
                          "--eps_safe", "0.3"])
register_env("navigation1")
env = make_vec_env("navigation1", N, device=dev, seed=1)
agent = SAC(env.observation_space, env.action_space, cfg, "/tmp")
fast = agent.enable_fast_path(256)
actor = FastActor(fast, N)
obs = env.reset()
noise = r(2, N, 2)
print("FastActor.act (7 launches) as a sequence: %.2f us per act" % bench(lambda: actor.act(obs, 0.3, True, True, noise=noise), n=50))
t_pol = bench(lambda: actor.pol.forward(obs, save=False), n=50)
actor.qr.finalize = True
t_qr = bench(lambda: actor.qr.forward(actor.xa, save=False), n=50)
t_rec = bench(lambda: actor.rec.forward(obs, save=False), n=50)
print("isolated: policy fwd %.2f, qrisk fwd (+sum) %.2f, recpolicy fwd %.2f us" % (t_pol, t_qr, t_rec))
