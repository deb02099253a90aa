#!/bin/bash
# Round 2: rocprofv3 kernel stats + SQ counters of both planner kernels (f32 MFMA and f16x3) -> gpurun_out/plan2/
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/plan2
mkdir -p $OUT
for P in f32 f16x3; do
  A=""; [ $P = f16x3 ] && A=f16x3
  python $R/profiles/plan_probe.py 512 5 $A 2>/dev/null | tail -1 > $OUT/plan_probe_$P.txt
  rm -rf /tmp/plan_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/plan_prof -o p -- python $R/profiles/plan_probe.py 256 5 $A > /tmp/plan_prof.log 2>&1
  cp $(find /tmp/plan_prof -name "*kernel_stats.csv" | head -1) $OUT/plan_kernel_stats_$P.csv
  bash $R/profiles/pmc_plan.sh 128 $A > $OUT/plan_pmc_$P.txt 2>&1
done
cat $OUT/plan_probe_*.txt $OUT/plan_pmc_*.txt; head -3 $OUT/plan_kernel_stats_f16x3.csv
