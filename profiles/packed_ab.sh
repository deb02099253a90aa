run() { python profiles/packed_probe.py $1 $2 $3 2>/dev/null | python -c "
import json,sys
print(' '.join('S=%d %.4f ms' % (r['seeds_per_gpu'], r['ms_per_packed_iteration']) for r in json.loads(sys.stdin.read())))
"; }
for U in 16 1; do
for cfg in "99 99 64" "2 99 64" "2 99 32" "99 2 64" "2 2 32"; do set -- $cfg
echo "U=$U panel_min=$1 r2_min=$2 panel=$3: $(RRL_PACK_PANEL64_MIN_SEEDS=$1 RRL_PACK_SMALL_R2_MIN_SEEDS=$2 RRL_PACK_PANEL=$3 run $U 2,3 $((U==16?100:300)))"
done; done
