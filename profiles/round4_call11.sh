set +e
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r4_pytest_gpu_head.txt 2>&1; tail -5 gpurun_out/r4_pytest_gpu_head.txt | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --sweep --no_legs --no_planner --no_cpu_baseline --min_seconds 0.5 > gpurun_out/r4_bench_sweep.json 2> gpurun_out/r4_bench_sweep.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_sweep.json") if l.startswith("{")][0])
    print("sweep nav_step", [(r["n_envs"], round(r["launch_us"],1), round(r["frac"],3)) for r in d["roofline_sweep"]])
    print("sweep compact", [(r["n_envs"], round(r["launch_us"],1), round(r["frac"],3)) for r in d["roofline_sweep_compact"]["rows"]])
except Exception as e:
    print("sweep parse failed", e)
P
