#!/bin/bash
# SQ counters of the packed forward launches (profiles/fwd_packed_probe.cpp at PROBE_S seeds): per kernel and grid the per-dispatch
# averages -- instruction counts per wave, busy / wait cycles.  One counter group per pass, kernel trace only.
#   bash profiles/pmc_fwd_packed.sh <S> [library.so]  ->  gpurun_out/pmc/fwd_packed_S<S>.txt
set -u
S=${1:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${2:-$R/recovery_rl_amd/csrc/librrl_hip.so}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
: > $OUT/fwd_packed_S${S}_raw.txt
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    D=/tmp/pmc_fwdp_$i
    rm -rf $D
    PROBE_S=$S PROBE_REPS=20 timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $D -o p -- $R/profiles/_ab_fwd_packed_probe $LIB > $D.log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
      python3 - "$f" <<'PY' >> $OUT/fwd_packed_S${S}_raw.txt
import csv, sys, collections, re
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '').replace('(anonymous namespace)::', '').replace('void ', '')
    name = re.sub(r'\(.*', '', name)
    key = "%s grid %s" % (name[:50], r.get('Grid_Size', r.get('Grid_Size_X', '?')))
    by[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in by.items():
    for c, v in d.items():
        if len(v) >= 10: print("%s\t%s\t%d\t%.1f" % (k, c, len(v), sum(v) / len(v)))
PY
    else
      echo "[$G] no csv" >> $OUT/fwd_packed_S${S}_raw.txt; tail -3 $D.log >> $OUT/fwd_packed_S${S}_raw.txt
    fi
done
python3 - $OUT/fwd_packed_S${S}_raw.txt > $OUT/fwd_packed_S${S}.txt <<'PY'
import sys, collections
t = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) == 4: t[p[0]][p[1]] = float(p[3])
    else: print(line.rstrip())
for k, d in sorted(t.items()):
    w = d.get("SQ_WAVES", 0) or 1
    print(k)
    print("   waves %d; per wave: VALU %.0f  MFMA %.0f  SALU %.0f  SMEM %.0f  LDS %.0f  VMEM rd %.0f wr %.0f" % (
        w, d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_MFMA", 0) / w, d.get("SQ_INSTS_SALU", 0) / w, d.get("SQ_INSTS_SMEM", 0) / w,
        d.get("SQ_INSTS_LDS", 0) / w, d.get("SQ_INSTS_VMEM_RD", 0) / w, d.get("SQ_INSTS_VMEM_WR", 0) / w))
    print("   " + "  ".join("%s %.0f" % (c, v) for c, v in sorted(d.items()) if not c.startswith("SQ_INSTS") and c != "SQ_WAVES"))
PY
cat $OUT/fwd_packed_S${S}.txt
