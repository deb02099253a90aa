#!/bin/bash
# Round 4: the runtime knobs round 3 did not get to (its call sat in ROC_SYSTEM_SCOPE_SIGNAL=0 until the time limit): headline leg
# (22-launch graph, 4096 envs, production loop) under each setting, one line per setting.  Every run under its own `timeout`.
mkdir -p gpurun_out
run() { timeout 150 env "$@" python bench.py --no_cpu_baseline --no_legs --no_planner --min_seconds 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.4f ms  %.2f M env-steps/s' % (d['ms_per_step'], d['value']/1e6))
"; }
{
for kv in X=0 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=64 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 DEBUG_CLR_MAX_BATCH_SIZE=1 \
          DEBUG_CLR_MAX_BATCH_SIZE=4096 ROC_USE_FGS_KERNARG=0 ROC_SKIP_KERNEL_ARG_COPY=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 \
          DEBUG_HIP_DYNAMIC_QUEUES=0 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 ROC_SIGNAL_POOL_SIZE=4096 \
          ROC_ACTIVE_WAIT_TIMEOUT=1000 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=1 X=1; do
  echo "$kv: $(run $kv)"
done
echo "packet capture on:"
for kv in X=0 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 DEBUG_CLR_MAX_BATCH_SIZE=4096; do
  echo "RRL_GRAPH_PACKET_CAPTURE=1 $kv: $(run RRL_GRAPH_PACKET_CAPTURE=1 $kv)"
done
} > gpurun_out/runtime_knobs_r4.txt 2>&1
cat gpurun_out/runtime_knobs_r4.txt
