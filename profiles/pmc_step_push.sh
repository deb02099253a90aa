#!/bin/bash
# (round 4 ->) HBM traffic per launch of the step + push kernel (compact_log = what the timed graph launches since round 4:
# compact env state + the per-episode table; compact = round 3's launch;
# arrays = the round-2 layout): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# kernel-trace only.  Writes gpurun_out/pmc/${ROUND}_step_push_pmc.json (compact; bench.py reads its 4096 entry as
# roofline.traffic) and ${ROUND}_step_push_compact_pmc.json: {N: {fetch_bytes, write_bytes}} with the guide's gfx950 correction
# (FETCH_SIZE counts 64 B per 128-B request: doubled; both counters are in KB).
set -u
ROUND=${ROUND:-round5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
: > $OUT/raw_counters_${ROUND}.txt
for LAYOUT in compact_log compact; do
 for N in ${SIZES:-4096 1048576}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_sp_${LAYOUT}_${N}_$C
    rm -rf $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python $R/profiles/run_step_push.py $N 20 $LAYOUT > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" $LAYOUT $N $C <<'PY' >> $OUT/raw_counters_${ROUND}.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'step_push_kernel' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[4]]
print(sys.argv[2], sys.argv[3], sys.argv[4], len(vals), (sum(vals[2:]) / max(len(vals[2:]), 1)) if vals else -1)
PY
    else
      echo "$LAYOUT $N $C 0 -1" >> $OUT/raw_counters_${ROUND}.txt; tail -3 $D.log >> $OUT/raw_counters_${ROUND}.txt
    fi
  done
 done
done
cat $OUT/raw_counters_${ROUND}.txt
ROUND=$ROUND python - $OUT/raw_counters_${ROUND}.txt $OUT <<'PY'
import json, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(dict))
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) == 5 and p[2] in ('FETCH_SIZE', 'WRITE_SIZE') and float(p[4]) >= 0:
        kb = float(p[4])
        d[p[0]][p[1]]['fetch_bytes' if p[2] == 'FETCH_SIZE' else 'write_bytes'] = int(kb * 1024 * (2 if p[2] == 'FETCH_SIZE' else 1))
import os
rnd = os.environ.get('ROUND', 'round5')
for layout, name in (('compact_log', rnd + '_step_push_pmc.json'), ('compact', rnd + '_step_push_compact_pmc.json')):
    rec = {n: dict(v, bytes_per_env_step=round((v['fetch_bytes'] + v['write_bytes']) / int(n), 1)) for n, v in d[layout].items() if len(v) == 2}
    json.dump(rec, open(sys.argv[2] + '/' + name, 'w'), indent=1)
    print(name, rec)
PY
