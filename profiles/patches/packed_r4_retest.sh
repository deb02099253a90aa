#!/bin/bash
# Re-test, on the build whose packed kernels read through global instructions, of the forward shapes that had lost before:
# four row tiles per workgroup for the packed forwards (RRL_PACK_R4_MIN_SEEDS), two row tiles from 2 seeds on.
mkdir -p gpurun_out
run() { timeout 300 python profiles/packed_probe.py $1 $2 $3 2>/dev/null | python -c "
import json,sys
print(' '.join('S=%d %.4f ms' % (r['seeds_per_gpu'], r['ms_per_packed_iteration']) for r in json.loads(sys.stdin.read())))
"; }
{
echo "U=16 defaults:          $(run 16 2,4,8,16 100)"
echo "U=16 R4 from 3 seeds:   $(RRL_PACK_R4_MIN_SEEDS=3 run 16 4,8,16 100)"
echo "U=16 R4 from 9 seeds:   $(RRL_PACK_R4_MIN_SEEDS=9 run 16 16 100)"
echo "U=16 small R2 from 2:   $(RRL_PACK_SMALL_R2_MIN_SEEDS=2 run 16 2 100)"
echo "U=16 panel 64 only:     $(RRL_PACK_BLOCK=0 RRL_PACK_PANEL32_MIN_SEEDS=99 run 16 2,4 100)"
echo "U=16 defaults:          $(run 16 2,4,8,16 100)"
} > gpurun_out/packed_r4_retest.txt 2>&1
cat gpurun_out/packed_r4_retest.txt
