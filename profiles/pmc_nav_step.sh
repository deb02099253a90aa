#!/bin/bash
# HBM traffic of nav_step_kernel per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only, csv output.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
for N in 4096 1048576 16777216; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${N}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${N}_$C -o p -- \
      python $GRAFT_REPO_ROOT/profiles/run_nav_step.py $N 20 > /tmp/pmc_${N}_$C.log 2>&1
    f=$(find /tmp/pmc_${N}_$C -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" $N $C <<'PY' >> $OUT/nav_step_pmc.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'nav_step_kernel' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[3]]
if vals:
    print("N=%s %s launches=%d mean_counter=%.3f (counter units per launch)" % (sys.argv[2], sys.argv[3], len(vals), sum(vals) / len(vals)))
else:
    print("N=%s %s no rows; columns=%s" % (sys.argv[2], sys.argv[3], list(rows[0].keys()) if rows else 'none'))
PY
    else
      echo "N=$N $C: no counter csv" >> $OUT/nav_step_pmc.txt; tail -5 /tmp/pmc_${N}_$C.log >> $OUT/nav_step_pmc.txt
    fi
  done
done
cat $OUT/nav_step_pmc.txt
