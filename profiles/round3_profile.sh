#!/bin/bash
# Round 3 rocprofv3 kernel stats -> gpurun_out/r3_prof/:
#   bench_*      `python bench.py --no_cpu_baseline --no_legs --no_planner` (the headline leg alone: per-iteration breakdown)
#   config4_*    profiles/config4_refit_run.py (config 4 through the driver incl. one online ensemble re-fit)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3_prof
mkdir -p $OUT
rm -rf /tmp/p1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o p -- python $R/bench.py --no_cpu_baseline --no_legs --no_planner > $OUT/bench_under_rocprof.json 2>/tmp/p1.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
STEPS=$(python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
n = [int(r["Calls"]) for r in rows if "sample_group_kernel" in r["Name"]]
print(max(n or [1]))
PY
)
python $R/profiles/kernel_breakdown.py $OUT/bench_kernel_stats.csv $STEPS > $OUT/bench_kernel_breakdown.txt
head -14 $OUT/bench_kernel_breakdown.txt
rm -rf /tmp/p2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o p -- python $R/profiles/config4_refit_run.py f16x3 > $OUT/config4_refit_under_rocprof.json 2>/tmp/p2.err
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $OUT/config4_refit_kernel_stats.csv
cat $OUT/config4_refit_under_rocprof.json
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/config4_refit_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("vendor GEMM kernels (Cijk_*):", sum(int(r["Calls"]) for r in rows if r["Name"].startswith("Cijk")))
print("autograd / backward kernels :", sum(int(r["Calls"]) for r in rows if "Backward" in r["Name"] or "backward" in r["Name"].lower() and "rrl" not in r["Name"]))
for r in rows[:14]:
    print("%-90s calls %7s  total ms %9.2f  avg us %9.2f" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
