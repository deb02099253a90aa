run() { echo "== $*"; env "$@" python bench.py --no_cpu_baseline --no_planner 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=1
run HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run A=1
