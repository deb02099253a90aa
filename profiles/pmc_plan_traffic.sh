#!/bin/bash
# HBM traffic of plan_cost_kernel per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), kernel-trace only, csv output.  M planning envs = $1 (default 128).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_plan
mkdir -p $OUT
: > $OUT/plan_traffic.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_plan_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_plan_$C -o p -- \
    python $GRAFT_REPO_ROOT/profiles/plan_probe.py ${1:-128} 2 > /tmp/pmc_plan_$C.log 2>&1
  f=$(find /tmp/pmc_plan_$C -name "*counter_collection.csv" | head -1)
  python - "$f" $C <<'PY' >> $OUT/plan_traffic.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'plan_cost_kernel' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[2]]
print("%s launches=%d mean_counter=%.1f" % (sys.argv[2], len(vals), sum(vals) / max(len(vals), 1)))
PY
done
cat $OUT/plan_traffic.txt
