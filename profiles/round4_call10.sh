set +e
mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -25) > gpurun_out/r4_pytest_gpu.txt 2>&1; tail -16 gpurun_out/r4_pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench.json") if l.startswith("{")][0])
    print("BENCH %.4f ms/iter %.2f M env-steps/s, step_push %.2f us frac %.4f traffic %s" % (d["ms_per_step"], d["value"]/1e6, d["roofline"]["launch_us"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    rs=d.get("roofline_stages",{})
    print("stages:", rs.get("launches"), rs.get("stand_alone_sum_us"), [(g["group"], g["launches"], round(g["us"],1), g["bound"], round(g["frac"],4)) for g in rs.get("by_group",[])] if "by_group" in rs else rs)
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r4_bench.err").read()[-2000:])
P
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_cmd.json 2> gpurun_out/r4_bench_driver_cmd.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_driver_cmd.json") if l.startswith("{")][0])
    print("DRIVER-CMD BENCH %.4f ms/iter %.2f M env-steps/s" % (d["ms_per_step"], d["value"]/1e6))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r4_bench_driver_cmd.err").read()[-2000:])
P
bash profiles/bench_profile.sh --no_legs --no_planner --min_seconds 1 2>&1 | tail -3
