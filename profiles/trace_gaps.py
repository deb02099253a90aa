"""Per kernel of a replayed graph: mean duration and mean gap to the NEXT kernel's start, from a rocprofv3 --kernel-trace csv.
    python profiles/trace_gaps.py <kernel_trace.csv> [anchor substring = step_push]   (one anchor launch per iteration)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "step_push"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-48:]
steps = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
lo, hi = steps[int(len(steps) * 0.4)], steps[-2]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i in range(lo, hi):
    r, nx = rows[i], rows[i + 1]
    dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gap[name(r)].append((int(nx["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
iters = steps.index(hi) - steps.index(lo)
tot_d = tot_g = 0.0
for k in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
    n = len(dur[k]); d = sum(dur[k]) / n; g = sum(gap[k]) / n
    tot_d += sum(dur[k]) / iters; tot_g += sum(gap[k]) / iters
    print("  %-50s per iter %5.2f  duration %6.2f us  gap to next %6.2f us" % (k, n / iters, d, g))
per_iter = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3 / iters
print("  per iteration %.1f us = %.1f us of kernels + %.1f us of gaps (under the profiler)" % (per_iter, tot_d, tot_g))
