set +e
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/r4_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r4_pytest_gpu.txt | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["launch_us"], d["roofline"]["frac"], d.get("roofline_mlp",{}).get("frac"))
for k in ("utd_1_256","config4","seed_pack"):
    if k in d: print(k, json.dumps(d[k])[:400])
P
bash profiles/bench_profile.sh --no_legs 2>&1 | tail -30
python bench.py --sweep --no_legs --no_planner --no_cpu_baseline --min_seconds 0.5 > gpurun_out/r4_bench_sweep.json 2> gpurun_out/r4_bench_sweep.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_sweep.json") if l.startswith("{")][0])
    print("sweep step_push", [(r["n_envs"], round(r["launch_us"],1), round(r["frac"],3)) for r in d["roofline_sweep_step_push"]])
except Exception as e:
    print("sweep parse failed", e)
P
