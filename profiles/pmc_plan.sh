#!/bin/bash
# SQ counters of plan_cost_kernel (one pass, 8 SQ slots; kernel-trace only).   bash profiles/pmc_plan.sh [M] [f16x3]
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_plan
mkdir -p $OUT
rm -rf /tmp/pmc_plan
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d /tmp/pmc_plan -o p -- python $GRAFT_REPO_ROOT/profiles/plan_probe.py ${1:-128} 2 ${2:-} > /tmp/pmc_plan.log 2>&1
f=$(find /tmp/pmc_plan -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' | tee $OUT/plan_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'plan_cost_kernel' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print("%-28s launches=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
else
  echo "no counter csv"; tail -20 /tmp/pmc_plan.log
fi
