#!/bin/bash
# roctx ranges (RRL_ROCTX=1: sample+sac_update+qrisk_update / act / env_step+push) + kernel trace of the EAGER loop:
# attributes every kernel to its stage without name matching.  -> gpurun_out/marker/
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/marker
mkdir -p $OUT
rm -rf /tmp/marker_prof
RRL_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/marker_prof -o p -- \
  python $R/bench.py --no_graph --steps 60 --warmup 10 --no_cpu_baseline --no_planner > $OUT/bench_eager.json 2>/tmp/marker.err
for f in $(find /tmp/marker_prof -name "*marker_api_trace.csv" -o -name "*marker_api_stats.csv" | head -2); do cp $f $OUT/; done
python - /tmp/marker_prof $OUT <<'PY'
import csv, glob, sys, collections, bisect
root, out = sys.argv[1], sys.argv[2]
mk = glob.glob(root + '/**/*marker_api_trace.csv', recursive=True)
kt = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
if not mk or not kt:
    print("no marker / kernel trace found", mk, kt); sys.exit(0)
ranges = []
for r in csv.DictReader(open(mk[0])):
    name = r.get('Function', r.get('Name', ''))
    try:
        ranges.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
    except (KeyError, ValueError):
        pass
print("marker ranges:", len(ranges), collections.Counter(n for _, _, n in ranges).most_common(8))
PY
ls $OUT
