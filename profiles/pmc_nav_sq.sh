#!/bin/bash
# SQ counters of the env-step kernels at 2^24 envs (one pass per counter group, kernel-trace only).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_nav_sq
mkdir -p $OUT
N=${N:-16777216}
: > $OUT/nav_sq.txt
for PROG in run_nav_step run_nav_step_compact; do
 for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc_sq_$PROG
  rm -rf $D
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $D -o p -- python $R/profiles/$PROG.py $N 12 > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
   python - "$f" $PROG <<'PY' >> $OUT/nav_sq.txt
import csv, sys, collections
acc = collections.defaultdict(list)
meta = None
for r in csv.DictReader(open(sys.argv[1])):
    if 'nav_step' not in r.get('Kernel_Name', ''):
        continue
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
    meta = (r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')), r.get('SGPR_Count', '?'), r.get('LDS_Block_Size', '?'), r.get('Grid_Size', '?'))
print(sys.argv[2], "vgpr/sgpr/lds/grid", meta)
for k, v in acc.items():
    print("   %-24s %.4g  (launches %d)" % (k, sum(v[2:]) / max(len(v[2:]), 1), len(v)))
PY
  else
   echo "$PROG: no counter csv for $GROUP" >> $OUT/nav_sq.txt; tail -5 $D.log >> $OUT/nav_sq.txt
  fi
 done
done
cat $OUT/nav_sq.txt
