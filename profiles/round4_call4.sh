set +e
mkdir -p gpurun_out
timeout 200 profiles/_ab_persist_probe 100 > gpurun_out/r4_persist_probe.txt 2>&1; cat gpurun_out/r4_persist_probe.txt
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/r4_pytest_gpu.txt 2>&1
tail -6 gpurun_out/r4_pytest_gpu.txt | cut -c1-400
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench.json") if l.startswith("{")][0])
    print("BENCH %.4f ms/iter %.2f M env-steps/s, step_push %.2f us frac %.4f" % (d["ms_per_step"], d["value"]/1e6, d["roofline"]["launch_us"], d["roofline"]["frac"]))
    rs=d.get("roofline_stages",{})
    print("stages:", rs.get("launches"), rs.get("stand_alone_sum_us"), [(g["group"], g["launches"], round(g["us"],1), round(g["frac"],4)) for g in rs.get("by_group",[])] if "by_group" in rs else rs)
    print("utd", d.get("utd_1_256",{}).get("ms_per_step"), "seed_pack", [(x["seeds_per_gpu"], round(x["ms_per_packed_iteration"],4)) for x in d.get("seed_pack",[])] if isinstance(d.get("seed_pack"),list) else d.get("seed_pack"))
    print("config4", {k:(v.get("env_steps_per_s") if isinstance(v,dict) else v) for k,v in d.get("config4",{}).items()})
    print("cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r4_bench.err").read()[-2000:])
P
bash profiles/bench_profile.sh --no_legs --no_planner --min_seconds 1 2>&1 | tail -28
SIZES="4096 1048576" bash profiles/pmc_step_push_r4.sh 2>&1 | tail -12
python bench.py --sweep --no_legs --no_planner --no_cpu_baseline --min_seconds 0.5 > gpurun_out/r4_bench_sweep.json 2> gpurun_out/r4_bench_sweep.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_sweep.json") if l.startswith("{")][0])
    print("sweep step_push", [(r["n_envs"], round(r["launch_us"],1), round(r["frac"],3)) for r in d["roofline_sweep_step_push"]])
except Exception as e:
    print("sweep parse failed", e)
P
