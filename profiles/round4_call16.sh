set +e
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_nav_gpu.py tests/test_episode_log_gpu.py tests/test_packed_gpu.py tests/test_maze_gpu.py tests/test_loop_gpu.py tests/test_fast_update_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r4_pytest_sp.txt 2>&1; tail -8 gpurun_out/r4_pytest_sp.txt | cut -c1-300
python - <<'P'
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
for n in (4096, 16384, 65536, 1 << 20, 1 << 22):
    for log in (True, False):
        t = bench.time_step_push_kernel(dev, "navigation1", n, reps=200 if n < (1 << 22) else 50, compact=True, log=log)
        print("step_push n", n, "log", log, round(t * 1e6, 2), "us", round(103 * n / t / 1e9, 1), "GB/s")
P
python bench.py --steps 2000 --warmup 200 > gpurun_out/r4_bench_sp.json 2> gpurun_out/r4_bench_sp.err; python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_sp.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['launch_us'])"
