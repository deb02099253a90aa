#!/bin/bash
# Round 4: where does instruction issue (rather than memory latency or launch overhead) bound the 22 launches of the timed
# iteration?  SQ instruction / cycle counters per kernel of `bench.py --no_legs` (headline loop only), one counter group per
# pass, kernel-trace only.  Writes gpurun_out/pmc/iteration_issue.txt: per kernel name the per-dispatch averages.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
: > $OUT/iteration_issue_raw.txt
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    D=/tmp/pmc_iter_$i
    rm -rf $D
    timeout 400 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $D -o p -- python $R/bench.py --no_legs --no_cpu_baseline --steps 60 --warmup 10 --min_seconds 0 > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python - "$f" <<'PY' >> $OUT/iteration_issue_raw.txt
import csv, sys, collections, re
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '')
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    name = re.sub(r'\(.*', '', name)
    by[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in by.items():
    for c, v in d.items():
        if len(v) >= 40: print("%s\t%s\t%d\t%.1f" % (k[:70], c, len(v), sum(v) / len(v)))
PY
    else
      echo "[$G] no csv" >> $OUT/iteration_issue_raw.txt; tail -3 $D.log >> $OUT/iteration_issue_raw.txt
    fi
done
python - $OUT/iteration_issue_raw.txt > $OUT/iteration_issue.txt <<'PY'
import sys, collections
t = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) == 4: t[p[0]][p[1]] = (int(p[2]), float(p[3]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F32"]
print("kernel\tdispatches\t" + "\t".join(cols))
for k, d in sorted(t.items()):
    n = max(v[0] for v in d.values())
    print(k + "\t" + str(n) + "\t" + "\t".join("%.0f" % d[c][1] if c in d else "-" for c in cols))
PY
cat $OUT/iteration_issue.txt
