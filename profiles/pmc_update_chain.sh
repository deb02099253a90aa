#!/bin/bash
# SQ counters of every kernel of the lock-step iteration (bench.py, eager launches so that each kernel is its own
# dispatch record): one pass, 8 SQ slots, kernel-trace only.  Per kernel: launches, mean of each counter, and the
# derived per-launch figures (waves, busy cycles, share of wave cycles spent waiting, MFMA busy, LDS conflicts).
#   bash profiles/pmc_update_chain.sh            -> gpurun_out/pmc_chain/update_chain_pmc.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_chain
mkdir -p $OUT
rm -rf /tmp/pmc_chain
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d /tmp/pmc_chain -o p -- \
  python $R/bench.py --no_graph --steps 40 --warmup 5 --no_cpu_baseline --no_planner > /tmp/pmc_chain.log 2>&1
f=$(find /tmp/pmc_chain -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' | tee $OUT/update_chain_pmc.txt
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get('Kernel_Name', '')
    if 'anonymous namespace' not in name and 'rrl_step' not in name:
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:60]
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    meta[short] = (r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')), r.get('LDS_Block_Size', '?'), r.get('Workgroup_Size', '?'),
                   r.get('Grid_Size', '?'))
print("kernel | launches | VGPRs | LDS B/wg | wg size | grid | waves | busy cyc | wave cyc | wait_inst/wave_cyc | active_inst/wave_cyc | MFMA busy cyc | LDS conflict cyc | wait LDS")
for k in sorted(acc, key=lambda k: -sum(acc[k].get('SQ_BUSY_CYCLES', [0]))):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    n = len(next(iter(acc[k].values())))
    wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    print("%-60s | %4d | %s | %s | %s | %s | %.0f | %.3g | %.3g | %.2f | %.2f | %.3g | %.3g | %.3g" % (
        k, n, *meta[k], c.get('SQ_WAVES', 0), c.get('SQ_BUSY_CYCLES', 0), wc, c.get('SQ_WAIT_INST_ANY', 0) / wc,
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), c.get('SQ_LDS_BANK_CONFLICT', 0),
        c.get('SQ_WAIT_INST_LDS', 0)))
PY
else
  echo "no counter csv"; tail -20 /tmp/pmc_chain.log
fi
