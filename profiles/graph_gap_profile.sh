#!/bin/bash
# Per kernel of the replayed lock-step graph: mean duration and mean gap to the next kernel's start, in both graph replay
# modes (DEBUG_CLR_GRAPH_PACKET_CAPTURE = 0: the package's setting, 1: the runtime default).  -> gpurun_out/graph_gaps.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/graph_gaps.txt
for MODE in 0 1; do
  rm -rf /tmp/gg
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=$MODE rocprofv3 --kernel-trace --output-format csv -d /tmp/gg -o p -- python $R/bench.py --no_cpu_baseline --no_planner --steps 200 --warmup 20 > /tmp/gg.log 2>&1
  f=$(find /tmp/gg -name "*kernel_trace.csv" | head -1)
  python - "$f" $MODE <<'PY' >> $R/gpurun_out/graph_gaps.txt
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-44:]
# the timed region: the last 60 % of the launches of the step kernel
steps = [i for i, r in enumerate(rows) if "step_push_kernel" in r["Kernel_Name"]]
lo, hi = steps[int(len(steps) * 0.4)], steps[-2]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i in range(lo, hi):
    r, nx = rows[i], rows[i + 1]
    dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gap[name(r)].append((int(nx["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
print("graph packet capture = %s" % sys.argv[2])
tot = 0.0
for k in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
    n = len(dur[k]); d = sum(dur[k]) / n; g = sum(gap[k]) / n
    print("  %-46s n %5d  duration %6.2f us  gap to next %6.2f us" % (k, n, d, g))
per_iter = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3 / (steps.index(hi) - steps.index(lo))
print("  per iteration %.1f us (under the profiler)" % per_iter)
PY
done
cat $R/gpurun_out/graph_gaps.txt
