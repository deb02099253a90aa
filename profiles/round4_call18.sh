set +e
mkdir -p gpurun_out
cat > /tmp/t_sp.py <<'P'
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
sizes = [int(x) for x in sys.argv[1].split(",")]
for n in sizes:
    for log in (True, False):
        t = min(bench.time_step_push_kernel(dev, "navigation1", n, reps=200 if n < (1 << 22) else 50, compact=True, log=log) for _ in range(3))
        print(sys.argv[2], "step_push n", n, "log", log, round(t * 1e6, 2), "us", round(103 * n / t / 1e9, 1), "GB/s")
P
for B in 64 128; do RRL_HIP_LIB=$PWD/profiles/_ab/librrl_hip_lat$B.so python /tmp/t_sp.py 4096,16384 lat$B; done
python /tmp/t_sp.py 4096,16384,65536,262144,1048576,4194304 default
(timeout 1200 python -m pytest tests/test_nav_gpu.py tests/test_episode_log_gpu.py tests/test_packed_gpu.py tests/test_maze_gpu.py tests/test_loop_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r4_pytest_sp.txt 2>&1; tail -4 gpurun_out/r4_pytest_sp.txt | cut -c1-300
TAG=new SIZES="1048576" bash profiles/pmc_step_push_issue.sh 2>&1 | grep -v "^$" | tail -24
