set +e
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_loop_gpu.py tests/test_mpc_gpu.py tests/test_plan_gpu.py tests/test_checkpoint_gpu.py tests/test_scripts_gpu.py -m gpu -q -p no:cacheprovider --durations=5 2>&1 | tail -25) > gpurun_out/r4_pytest_mbgraph.txt 2>&1; tail -14 gpurun_out/r4_pytest_mbgraph.txt | cut -c1-300
python - <<'P'
import sys, json, contextlib, io
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
for prec in ("f32", "f16x3"):
    for graph in (True, False):
        with contextlib.redirect_stdout(io.StringIO()):
            r = bench.run_config4_leg(dev, prec, graph=graph)
        print(prec, "graph" if graph else "eager", round(r["ms_per_step"], 2), "ms", round(r["env_steps_per_s"]), "env-steps/s planned", r["planned_actions"], "roofline frac", round(r["roofline"]["frac"], 3), r["recovery_set_sizes"][:8])
P
