set +e
mkdir -p gpurun_out
R=$PWD
cat > /tmp/t_sp.py <<'P'
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
for n in (4096, 16384):
    for log in (True, False):
        t = min(bench.time_step_push_kernel(dev, "navigation1", n, reps=200, compact=True, log=log) for _ in range(3))
        print(sys.argv[1], "step_push n", n, "log", log, round(t * 1e6, 2), "us")
P
for L in prev new; do
  if [ $L = new ]; then unset RRL_HIP_LIB; else export RRL_HIP_LIB=$R/profiles/_ab_$L.so; fi
  python /tmp/t_sp.py $L 2>/dev/null
done
for rep in 1 2; do
for L in prev new; do
  if [ $L = new ]; then unset RRL_HIP_LIB; else export RRL_HIP_LIB=$R/profiles/_ab_$L.so; fi
  python bench.py --no_legs --no_cpu_baseline --steps 4000 --warmup 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],5), round(d['value']/1e6,3))"
done; done
unset RRL_HIP_LIB
(timeout 1200 python -m pytest tests/test_nav_gpu.py tests/test_episode_log_gpu.py tests/test_packed_gpu.py tests/test_maze_gpu.py tests/test_loop_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r4_pytest_sp.txt 2>&1; tail -4 gpurun_out/r4_pytest_sp.txt | cut -c1-300
