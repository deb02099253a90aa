#!/bin/bash
# rocprofv3 kernel stats + SQ counters of the fused planner kernel, and config-4 rates; summaries -> gpurun_out/plan/
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/plan
mkdir -p $OUT
python $R/profiles/plan_probe.py 512 5 2>&1 | tail -1 > $OUT/plan_probe_fused.txt
python $R/profiles/plan_probe.py 512 2 torch 2>&1 | tail -1 > $OUT/plan_probe_torch.txt
rm -rf /tmp/plan_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/plan_prof -o p -- python $R/profiles/plan_probe.py 256 5 > /tmp/plan_prof.log 2>&1
cp $(find /tmp/plan_prof -name "*kernel_stats.csv" | head -1) $OUT/plan_kernel_stats.csv
bash $R/profiles/pmc_plan.sh 128 > $OUT/plan_pmc.txt 2>&1
timeout 900 python $R/profiles/config_rates.py 4 2>&1 | tail -1 > $OUT/config4.json
timeout 300 python $R/profiles/config_rates.py 3 2>&1 | tail -1 > $OUT/config3.json
cat $OUT/plan_probe_fused.txt $OUT/plan_probe_torch.txt $OUT/plan_pmc.txt $OUT/config4.json $OUT/config3.json; head -5 $OUT/plan_kernel_stats.csv
