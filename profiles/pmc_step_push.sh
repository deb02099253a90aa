#!/bin/bash
# HBM traffic per launch of the step + push kernel (the `roofline` kernel of bench.py) and of rrl_nav_step:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
# Writes gpurun_out/pmc/round2_step_push_pmc.json: {N: {fetch_bytes, write_bytes}} with the guide's gfx950 correction
# (FETCH_SIZE counts 64 B per 128-B request: doubled; both counters are in KB).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
: > $OUT/raw_counters.txt
for PROG in ${PROGS:-run_step_push run_nav_step run_nav_step_compact}; do
 for N in ${SIZES:-4096 1048576 ${BIGN:-2097152}}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${PROG}_${N}_$C
    rm -rf $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python $R/profiles/$PROG.py $N 20 > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" $PROG $N $C <<'PY' >> $OUT/raw_counters.txt
import csv, sys
want = 'step_push_kernel' if sys.argv[2] == 'run_step_push' else 'nav_step'
rows = [r for r in csv.DictReader(open(sys.argv[1])) if want in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[4]]
print(sys.argv[2], sys.argv[3], sys.argv[4], len(vals), (sum(vals[2:]) / max(len(vals[2:]), 1)) if vals else -1)
PY
    else
      echo "$PROG $N $C 0 -1" >> $OUT/raw_counters.txt; tail -3 $D.log >> $OUT/raw_counters.txt
    fi
  done
 done
done
cat $OUT/raw_counters.txt
python - $OUT/raw_counters.txt $OUT <<'PY'
import json, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(dict))
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) == 5 and p[2] in ('FETCH_SIZE', 'WRITE_SIZE') and float(p[4]) >= 0:
        kb = float(p[4])
        d[p[0]][p[1]]['fetch_bytes' if p[2] == 'FETCH_SIZE' else 'write_bytes'] = int(kb * 1024 * (2 if p[2] == 'FETCH_SIZE' else 1))
for prog, name in (('run_step_push', 'round2_step_push_pmc.json'), ('run_nav_step', 'round2_nav_step_pmc.json'),
                   ('run_nav_step_compact', 'round2_nav_step_compact_pmc.json')):
    rec = {n: v for n, v in d[prog].items() if len(v) == 2}
    json.dump(rec, open(sys.argv[2] + '/' + name, 'w'), indent=1)
    print(name, rec)
PY
