#!/bin/bash
# Round 4: is the step + push kernel instruction-issue bound in the large-N regime?  SQ instruction / cycle counters of
# step_push_kernel at N envs, one counter group per pass (kernel-trace only).  Writes gpurun_out/pmc/step_push_issue_<tag>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
TAG=${TAG:-base}
mkdir -p $OUT
: > $OUT/step_push_issue_$TAG.txt
for N in ${SIZES:-1048576}; do
 for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
          "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    D=/tmp/pmc_issue_${N}_$(echo $G | tr ' ' '_' | cut -c1-40)
    rm -rf $D
    timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $D -o p -- python $R/profiles/run_step_push.py $N 12 ${LAYOUT:-compact_log} > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" $N <<'PY' >> $OUT/step_push_issue_$TAG.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'step_push_kernel' in r.get('Kernel_Name', '')]
by = collections.defaultdict(list)
for r in rows: by[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in by.items():
    print(sys.argv[2], k, len(v), sum(v[2:]) / max(len(v[2:]), 1))
PY
    else
      echo "$N [$G] no csv" >> $OUT/step_push_issue_$TAG.txt; tail -3 $D.log >> $OUT/step_push_issue_$TAG.txt
    fi
 done
done
cat $OUT/step_push_issue_$TAG.txt
