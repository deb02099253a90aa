#!/bin/bash
# Block form of the packed hidden-layer backward (mlp_kernels.hip: gemm_block_pack_kernel): parity of the packed seeds with
# their solo runs under each shape, then ms per packed iteration (RRL_PACK_BLOCK = 0: 16 x 16 tiles; 22 / 12 / 11: 64 x 64 /
# 32 x 64 / 32 x 32 blocks per four-wave workgroup; RRL_PACK_BLOCK_MIN_SEEDS: seeds from which the block form is used)
mkdir -p gpurun_out
{
for shape in 12 22 11; do
  echo "== parity RRL_PACK_BLOCK=$shape"
  RRL_PACK_BLOCK=$shape RRL_PACK_BLOCK_MIN_SEEDS=2 timeout 600 python -m pytest tests/test_packed_gpu.py -x -q 2>&1 | tail -1
done
run() { python profiles/packed_probe.py $1 $2 $3 2>/dev/null | python -c "
import json,sys
print(' '.join('S=%d %.4f ms' % (r['seeds_per_gpu'], r['ms_per_packed_iteration']) for r in json.loads(sys.stdin.read())))
"; }
for shape in 0 22 12 11 0 12; do
  echo "U=16 RRL_PACK_BLOCK=$shape: $(RRL_PACK_BLOCK=$shape RRL_PACK_BLOCK_MIN_SEEDS=2 run 16 2,3,4,8,16 100)"
done
for shape in 0 12; do
  echo "U=1 RRL_PACK_BLOCK=$shape: $(RRL_PACK_BLOCK=$shape RRL_PACK_BLOCK_MIN_SEEDS=2 run 1 2,4,8 300)"
done
} > gpurun_out/blk_ab.txt 2>&1
cat gpurun_out/blk_ab.txt
