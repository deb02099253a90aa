"""Round 6: the planner without its dead work against round 5's literal rollout, on the same box and the same inputs.
  * bit-identity: costs of rrl_plan_cost / rrl_plan_cost_f16x3 (in-tree: first step once per distinct row + no prediction at
    the last step) against the round-5 kernel (every particle row through both networks at every step), explicit noise and
    in-kernel Philox noise, several shapes including ragged ones;
  * time of both at M planning problems (HIP events around `reps` calls).

    python profiles/plan_dedupe_ab.py <round5 plan_kernels .so> [M=256] [reps=5]
The round-5 library is built from `git show <round-5 commit>:recovery_rl_amd/csrc/plan_kernels.hip` (hipcc --offload-arch=gfx950
-O3 -std=c++17 -fPIC -shared -ffp-contract=off)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402

DEV = "cuda:0"


def main():
    old = C.CDLL(os.path.abspath(sys.argv[1]))
    M_time = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    from test_plan_gpu import build
    vp, ci, ll, u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_uint64
    sig = [vp, ci, ci, ci, ci, ll, ci, ci, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp]
    out = {"identical": [], "time": []}
    for f16x3 in (False, True):
        env, mpc, _ = build(f16x3=f16x3)
        fp = mpc.fused
        entry_old = old.rrl_plan_cost_f16x3 if f16x3 else old.rrl_plan_cost
        entry_old.argtypes, entry_old.restype = sig, ci

        def run_old(acs, obs, noise, tick):
            M, pop = acs.shape[0], acs.shape[1]
            partial = torch.empty(M * pop * fp.n_nets, device=DEV)
            costs = torch.empty(M, pop, device=DEV)
            rc = entry_old(_lib.ptr(fp.packed), fp.hq, fp.he, fp.n_nets, mpc.npart, M, pop, mpc.plan_hor, _lib.ptr(obs),
                           _lib.ptr(acs), _lib.ptr(noise), fp.seed, 0, _lib.ptr(tick), 1, _lib.ptr(partial), _lib.ptr(costs),
                           _lib.current_stream())
            assert rc == 0
            torch.cuda.synchronize()
            return costs

        for M, pop, hor in ((1, 400, 5), (3, 400, 5), (2, 30, 5), (5, 7, 5), (64, 400, 5), (2, 400, 1), (2, 400, 2), (3, 100, 9)):
            mpc.plan_hor = hor
            g = torch.Generator(device=DEV).manual_seed(M * 1000 + pop + hor)
            acs = torch.rand(M, pop, hor * 2, device=DEV, generator=g) * 2 - 1
            obs = torch.randn(M, 2, device=DEV, generator=g) * torch.tensor([1.5, 1.0], device=DEV) + \
                torch.tensor([-0.5, 0.3], device=DEV)
            noise = torch.randn(hor, M * pop * mpc.npart, 2, device=DEV, generator=g)
            for nz in (noise, None):
                fp.tick.zero_()
                tick_old = torch.zeros(2, dtype=torch.int64, device=DEV)
                a = run_old(acs, obs, nz, tick_old)
                b = fp.cost(acs, obs, nz)
                torch.cuda.synchronize()
                same = bool(torch.equal(a, b)) and int(tick_old[0]) == int(fp.tick[0])
                out["identical"].append({"f16x3": f16x3, "M": M, "pop": pop, "plan_hor": hor, "noise": "array" if nz is not None else "philox",
                                         "bit_identical": same, "max_abs_diff": float((a - b).abs().max()), "std": float(b.std())})
        mpc.plan_hor = 5
        g = torch.Generator(device=DEV).manual_seed(1)
        acs = torch.rand(M_time, 400, 10, device=DEV, generator=g) * 2 - 1
        obs = torch.randn(M_time, 2, device=DEV, generator=g)
        tick_old = torch.zeros(2, dtype=torch.int64, device=DEV)
        for name, f in (("round5", lambda: run_old(acs, obs, None, tick_old)), ("round6", lambda: fp.cost(acs, obs))):
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            out["time"].append({"f16x3": f16x3, "kernel": name, "M": M_time, "ms": e0.elapsed_time(e1) / reps})
    out["all_bit_identical"] = all(r["bit_identical"] for r in out["identical"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
