#!/bin/bash
# rocprofv3 kernel stats of the default bench command -> gpurun_out/bench_prof/
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bench_prof
mkdir -p $OUT
rm -rf /tmp/bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -o p -- python $R/bench.py --no_cpu_baseline "$@" > $OUT/bench_under_rocprof.json 2>/tmp/bench_prof.err
cp $(find /tmp/bench_prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# iterations the run executed = launches of the grouped sampler (one per iteration; the step kernel is also launched by
# bench.py's stand-alone roofline timing), falling back to the step kernel
STEPS=$(python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
n = [int(r["Calls"]) for r in rows if "sample_group_kernel" in r["Name"]]
print(max(n or [int(r["Calls"]) for r in rows if "step_push_kernel" in r["Name"]] or [1]))
PY
)
python $R/profiles/kernel_breakdown.py $OUT/kernel_stats.csv $STEPS > $OUT/breakdown.txt
head -40 $OUT/breakdown.txt
