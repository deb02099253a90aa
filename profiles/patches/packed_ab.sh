#!/bin/bash
# Packed-launch shapes (mlp_kernels.hip: pack_panel, pack_small_r2_min_seeds), each alone and together, ms per packed iteration:
#   RRL_PACK_PANEL64_MIN_SEEDS / RRL_PACK_PANEL32_MIN_SEEDS   K-panel width of the hidden-layer backward tiles from S seeds on
#   RRL_PACK_SMALL_R2_MIN_SEEDS                               two row tiles per workgroup for the B <= 1024 forwards from S seeds on
run() { python profiles/packed_probe.py $1 $2 $3 2>/dev/null | python -c "
import json,sys
print(' '.join('S=%d %.4f ms' % (r['seeds_per_gpu'], r['ms_per_packed_iteration']) for r in json.loads(sys.stdin.read())))
"; }
for U in 16 1; do
for cfg in "99 99 99 solo-shapes" "2 99 99 panel64" "2 3 99 panel64/32" "99 99 3 r2" "2 3 3 defaults"; do set -- $cfg
echo "U=$U $4: $(RRL_PACK_PANEL64_MIN_SEEDS=$1 RRL_PACK_PANEL32_MIN_SEEDS=$2 RRL_PACK_SMALL_R2_MIN_SEEDS=$3 run $U 2,3,4,8 $((U==16?100:300)))"
done; done
