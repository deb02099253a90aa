set +e
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_nav_gpu.py tests/test_episode_log_gpu.py tests/test_packed_gpu.py tests/test_maze_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4) 2>&1 | cut -c1-200
python bench.py --sweep --no_legs --no_planner --no_cpu_baseline --min_seconds 0.5 > gpurun_out/r4_bench_sweep.json 2> gpurun_out/r4_bench_sweep.err; python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench_sweep.json") if l.startswith("{")][0])
print(d["ms_per_step"], d["value"])
print("sweep step_push", [(r["n_envs"], round(r["launch_us"],1), round(r["frac"],3)) for r in d["roofline_sweep_step_push"]])
P
