#!/bin/bash
# Per-kernel averages of the packed iteration at S = 1 and S = 4 (U = 16): which launches stretch when seeds share them.
#   packed_prof.sh [U=16] ["S list"="1 4"] [out dir under gpurun_out = packed_prof]
#   -> gpurun_out/<out>/S{1,4}_kernel_stats.csv + S{1,4}.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${3:-packed_prof}
mkdir -p $OUT
U=${1:-16}
SLIST=${2:-"1 4"}
for S in $SLIST; do
  rm -rf /tmp/pp$S
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$S -o p -- python $R/profiles/packed_probe.py $U $S 100 > $OUT/S$S.json 2>/tmp/pp$S.err
  cp $(find /tmp/pp$S -name "*kernel_stats.csv" | head -1) $OUT/S${S}_kernel_stats.csv
  python - <<PY > $OUT/S$S.txt
import csv
rows = list(csv.DictReader(open("$OUT/S${S}_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-80s calls %7s  total ms %9.2f  avg us %9.2f" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  cat $OUT/S$S.txt
done
