set +e
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_fast_update_gpu.py tests/test_mlp_gpu.py tests/test_packed_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8) > gpurun_out/r4_pytest_g16.txt 2>&1; tail -4 gpurun_out/r4_pytest_g16.txt | cut -c1-300
python bench.py --no_legs --no_cpu_baseline --steps 3000 --warmup 300 > gpurun_out/r4_bench_g16.json 2> gpurun_out/r4_bench_g16.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r4_bench_g16.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])
for s in d['roofline_stages']['stages'] if isinstance(d['roofline_stages'], dict) else d['roofline_stages']:
    print(round(s['us'],2), s['stage'])
P
