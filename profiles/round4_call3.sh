set +e
mkdir -p gpurun_out
timeout 120 profiles/_ab_persist_probe 200 > gpurun_out/r4_persist_probe.txt 2>&1; cat gpurun_out/r4_persist_probe.txt
for mode in "off 100" "chain 100" "branches 100" "chain 0" "off 0"; do
  set -- $mode
  timeout 300 python bench.py --overlap $1 --log_every $2 --no_legs --no_planner --no_cpu_baseline --min_seconds 2 > gpurun_out/r4_bench_ab_$1_$2.json 2> gpurun_out/r4_bench_ab_$1_$2.err
  python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_ab_%s_%s.json"%(sys.argv[1],sys.argv[2])) if l.startswith("{")][0])
    print("BENCH overlap=%s log_every=%s: %.4f ms/iter  %.2f M env-steps/s  step_push %.2f us" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"]/1e6, d["roofline"]["launch_us"]))
except Exception as e:
    print("bench parse failed", sys.argv[1:], e); print(open("gpurun_out/r4_bench_ab_%s_%s.err"%(sys.argv[1],sys.argv[2])).read()[-1500:])
P
done
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r4_pytest_gpu.txt 2>&1
tail -12 gpurun_out/r4_pytest_gpu.txt | cut -c1-400
for s in 3 5 1 7; do timeout 600 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --demo_share 0 > gpurun_out/r4_c4_share0_seed$s.json 2> gpurun_out/r4_c4_share0_seed$s.err; grep "^{" gpurun_out/r4_c4_share0_seed$s.err | cut -c1-520; done
for s in 3 5; do timeout 600 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --demo_share 0.25 > gpurun_out/r4_c4_share025_seed$s.json 2> gpurun_out/r4_c4_share025_seed$s.err; grep "^{" gpurun_out/r4_c4_share025_seed$s.err | cut -c1-520; done
