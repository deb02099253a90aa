set +e
mkdir -p gpurun_out
R=$PWD
for rep in 1 2; do
for L in prev new; do
  if [ $L = new ]; then unset RRL_HIP_LIB; else export RRL_HIP_LIB=$R/profiles/_ab_$L.so; fi
  python bench.py --no_legs --no_cpu_baseline --steps 4000 --warmup 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],5), round(d['value']/1e6,3))"
done; done
unset RRL_HIP_LIB
(timeout 1200 python -m pytest tests/test_fast_update_gpu.py tests/test_packed_gpu.py tests/test_loop_gpu.py tests/test_replay_gpu.py tests/test_demo_share_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r4_pytest_g16.txt 2>&1; tail -4 gpurun_out/r4_pytest_g16.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_new
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_new -o p -- python $R/bench.py --no_legs --no_cpu_baseline --steps 1000 --warmup 100 --min_seconds 0 > /tmp/kt_new.log 2>&1
f=$(find /tmp/kt_new -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n for k in ("backward_pair",'mlp3_fwd_split_group','gemm16_group','head_bwd_group','head_bwd_loss','adam_multi','sample_group','step_push')):
        print(n.replace('(anonymous namespace)::','')[:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
