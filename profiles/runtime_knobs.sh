#!/bin/bash
# Headline leg (22-launch graph, 4096 envs) under HIP runtime knobs that touch kernel boundaries / graph submission.
# One line per setting: ms per iteration.  (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is the launcher's default)
# Every run under its own `timeout`: ROC_SYSTEM_SCOPE_SIGNAL=0 HANGS the bench process (first run of this script, round 3:
# the call sat in that setting until gpurun's limit; round3_runtime_knobs.txt holds the settings before it).
mkdir -p gpurun_out
run() { timeout 120 env "$@" python bench.py --no_cpu_baseline --no_legs --no_planner 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4f ms  %.2f M env-steps/s' % (d['ms_per_step'], d['value']/1e6))
"; }
{
for kv in X=0 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 AMD_OPT_FLUSH=3 ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 \
          DEBUG_CLR_MAX_BATCH_SIZE=1 DEBUG_CLR_MAX_BATCH_SIZE=4096 ROC_USE_FGS_KERNARG=0 ROC_SKIP_KERNEL_ARG_COPY=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 \
          DEBUG_HIP_DYNAMIC_QUEUES=0 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 ROC_SIGNAL_POOL_SIZE=4096 X=1; do
  echo "$kv: $(run $kv)"
done
echo "packet capture on:"
for kv in X=0 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=3 ROC_SYSTEM_SCOPE_SIGNAL=0; do
  echo "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $kv: $(run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $kv)"
done
} > gpurun_out/runtime_knobs.txt 2>&1
cat gpurun_out/runtime_knobs.txt
