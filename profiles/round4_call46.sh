set +e
mkdir -p gpurun_out
python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["launch_us"], d["roofline"]["frac"], d.get("roofline_mlp",{}).get("frac"))
P
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_cmd.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4_bench_driver_cmd.json') if l.startswith('{')][-1]); print('driver cmd', d['ms_per_step'], d['value'])"
bash profiles/bench_profile.sh --no_legs 2>&1 | grep TOTAL
