#!/bin/bash
# SQ counters of the packed launches at S seeds (eager packed iterations): MFMA busy cycles against the kernel's busy cycles.
#   bash profiles/pmc_packed.sh [S=16] [U=4]    -> gpurun_out/pmc_packed/S<S>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-16}; U=${2:-4}
OUT=$R/gpurun_out/pmc_packed
mkdir -p $OUT
rm -rf /tmp/pmc_packed
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d /tmp/pmc_packed -o p -- python $R/profiles/packed_eager.py $S $U 6 > /tmp/pmc_packed.log 2>&1
f=$(find /tmp/pmc_packed -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' | tee $OUT/S$S.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '')
    if 'pack_kernel' not in name:
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:48]
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    meta[short] = (r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')), r.get('LDS_Block_Size', '?'), r.get('Workgroup_Size', '?'),
                   r.get('Grid_Size', '?'))
print("kernel | launches | VGPRs | LDS B/wg | wg size | grid | waves | SQ busy cyc (sum over SEs) | wave cyc | wait_inst/wave_cyc | MFMA busy cyc | MFMA busy / SQ busy | LDS conflict cyc")
for k in sorted(acc, key=lambda k: -sum(acc[k].get('SQ_BUSY_CYCLES', [0]))):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    n = len(next(iter(acc[k].values())))
    wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    busy = max(c.get('SQ_BUSY_CYCLES', 0.0), 1.0)
    print("%-48s | %4d | %s | %s | %s | %s | %.0f | %.3g | %.3g | %.2f | %.3g | %.3f | %.3g" % (
        k, n, *meta[k], c.get('SQ_WAVES', 0), busy, wc, c.get('SQ_WAIT_INST_ANY', 0) / wc,
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy, c.get('SQ_LDS_BANK_CONFLICT', 0)))
PY
else
  echo "no counter csv"; tail -20 /tmp/pmc_packed.log
fi
