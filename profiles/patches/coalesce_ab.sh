mkdir -p gpurun_out
{
echo "== tests"
timeout 500 python -m pytest tests/test_gemm_gpu.py tests/test_fast_update_gpu.py tests/test_packed_gpu.py tests/test_loop_gpu.py -x -q 2>&1 | tail -3
echo "== headline A/B (in_tree = coalesced W2 loads, alternative = -DRRL_COALESCE_W2=0)"
timeout 300 python profiles/ab_lib.py profiles/_ab_nocoalesce.so 3 2>/dev/null
run() { timeout 200 python profiles/packed_probe.py $1 $2 $3 2>/dev/null | python -c "
import json,sys
print(' '.join('S=%d %.4f ms' % (r['seeds_per_gpu'], r['ms_per_packed_iteration']) for r in json.loads(sys.stdin.read())))
"; }
echo "U=16 coalesced:    $(run 16 4,8,16 100)"
echo "U=16 alternative:  $(RRL_HIP_LIB=$PWD/profiles/_ab_nocoalesce.so run 16 4,8,16 100)"
echo "U=16 coalesced:    $(run 16 4,8,16 100)"
echo "== phases"
timeout 200 python profiles/mlp_fwd_timing.py 4096 2 2>&1 | tail -9
} > gpurun_out/coalesce_ab.txt 2>&1
cat gpurun_out/coalesce_ab.txt
