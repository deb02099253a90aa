"""Micro-benchmark of rrl_plan_cost (fused planner kernel): M planning problems x 400 candidates x 20 particles
x 5 steps.  Prints kernel time (HIP events on the launch stream), row-steps/s and TFLOP/s against the f32 MFMA
peak (157.3 TF).  Algorithmic FLOPs per row-step: twin Q_risk 2 x 2 x (4*256 + 256*256 + 256) = 267 264,
ensemble member 2 x (4*200 + 2*200*200 + 200*4) = 163 200.

    python profiles/plan_probe.py [M] [reps] [torch|f16x3]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

FLOPS_PER_ROW_STEP = 267264 + 163200
PEAK_TF = 157.3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    use_torch = len(sys.argv) > 3 and sys.argv[3] == "torch"
    f16x3 = len(sys.argv) > 3 and sys.argv[3] == "f16x3"
    from test_plan_gpu import build, inputs
    env, mpc, _ = build(f16x3=f16x3)
    pop = 400
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acs = torch.rand(M, pop, mpc.plan_hor * 2, device="cuda:0", generator=g) * 2 - 1
    obs = torch.randn(M, 2, device="cuda:0", generator=g)
    f = (lambda: mpc._compile_cost(acs, obs, fused=False)) if use_torch else (lambda: mpc.fused.cost(acs, obs))
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / reps
    row_steps = M * pop * mpc.npart * mpc.plan_hor
    tf = row_steps * FLOPS_PER_ROW_STEP / dt / 1e12
    print({"M": M, "path": "torch" if use_torch else ("rrl_plan_cost_f16x3" if f16x3 else "rrl_plan_cost"), "ms": dt * 1e3, "row_steps_per_s": row_steps / dt,
           "tflops": tf, "frac_of_f32_mfma_peak": tf / PEAK_TF})


if __name__ == "__main__":
    main()
