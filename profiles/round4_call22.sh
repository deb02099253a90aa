set +e
mkdir -p gpurun_out
for rep in 1 2; do
for L in head g16only fwdonly new; do
  if [ $L = new ]; then unset RRL_HIP_LIB; else export RRL_HIP_LIB=$PWD/profiles/_ab_$L.so; fi
  python bench.py --no_legs --no_cpu_baseline --steps 4000 --warmup 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],5), round(d['value']/1e6,3))"
done; done
unset RRL_HIP_LIB
(timeout 1200 python -m pytest tests/test_fast_update_gpu.py tests/test_packed_gpu.py tests/test_loop_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8) > gpurun_out/r4_pytest_g16.txt 2>&1; tail -4 gpurun_out/r4_pytest_g16.txt | cut -c1-300
