#!/bin/bash
# L2 hit / miss counts of the packed launches at S seeds (eager packed iterations).
#   bash profiles/pmc_packed_l2.sh [S=16] [U=4]    -> gpurun_out/pmc_packed/S<S>_l2.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-16}; U=${2:-4}
OUT=$R/gpurun_out/pmc_packed
mkdir -p $OUT
rm -rf /tmp/pmc_l2
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/pmc_l2 -o p -- python $R/profiles/packed_eager.py $S $U 6 > /tmp/pmc_l2.log 2>&1
f=$(find /tmp/pmc_l2 -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' | tee $OUT/S${S}_l2.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '')
    if 'pack_kernel' not in name:
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:48]
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
print("kernel | launches | L2 requests | hits | misses | hit rate   (per launch, 128-byte lines)")
for k in sorted(acc, key=lambda k: -sum(acc[k].get('TCC_REQ_sum', [0]))):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    n = len(next(iter(acc[k].values())))
    h, m = c.get('TCC_HIT_sum', 0), c.get('TCC_MISS_sum', 0)
    print("%-48s | %4d | %.3g | %.3g | %.3g | %.3f" % (k, n, c.get('TCC_REQ_sum', 0), h, m, h / max(h + m, 1)))
PY
else
  echo "no counter csv"; tail -20 /tmp/pmc_l2.log
fi
