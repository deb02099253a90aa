#!/bin/bash
# SQ instruction mix of every kernel of the packed iteration (profiles/packed_probe.py at S seeds, U updates per step): per
# kernel the per-wave instruction counts and the busy / wait cycles.   bash profiles/pmc_packed.sh <S> <U>  ->  gpurun_out/pmc/packed_S<S>_U<U>.txt
set -u
S=${1:-16}; U=${2:-16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
RAW=$OUT/packed_S${S}_U${U}_raw.txt
: > $RAW
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    D=/tmp/pmc_packed_$i
    rm -rf $D
    timeout 600 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $D -o p -- python $R/profiles/packed_probe.py $U $S 40 > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python3 - "$f" <<'PY' >> $RAW
import csv, sys, collections, re
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '').replace('(anonymous namespace)::', '').replace('void ', '')
    name = re.sub(r'\(.*', '', name)
    if 'pack' not in name: continue
    by[name[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in by.items():
    for c, v in d.items():
        if len(v) >= 10: print("%s\t%s\t%d\t%.1f" % (k, c, len(v), sum(v) / len(v)))
PY
    else
      echo "[$G] no csv" >> $RAW; tail -3 $D.log >> $RAW
    fi
done
python3 - $RAW > $OUT/packed_S${S}_U${U}.txt <<'PY'
import sys, collections
t = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) == 4: t[p[0]][p[1]] = float(p[3])
    else: print(line.rstrip())
for k, d in sorted(t.items()):
    w = d.get("SQ_WAVES", 0) or 1
    mf = d.get("SQ_INSTS_MFMA", 0)
    print(k)
    print("   waves %d; per wave: VALU (incl. MFMA) %.0f  MFMA %.0f  SALU %.0f  SMEM %.0f  LDS %.0f  VMEM rd %.0f wr %.0f   non-MFMA VALU per MFMA %.2f" % (
        w, d.get("SQ_INSTS_VALU", 0) / w, mf / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_SMEM", 0) / w,
        d.get("SQ_INSTS_LDS", 0) / w, d.get("SQ_INSTS_VMEM_RD", 0) / w, d.get("SQ_INSTS_VMEM_WR", 0) / w,
        (d.get("SQ_INSTS_VALU", 0) - mf) / mf if mf else 0))
    print("   " + "  ".join("%s %.0f" % (c, v) for c, v in sorted(d.items()) if not c.startswith("SQ_INSTS") and c != "SQ_WAVES"))
PY
cat $OUT/packed_S${S}_U${U}.txt
