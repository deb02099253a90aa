set +e
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r4_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r4_pytest_gpu.txt
timeout 120 profiles/_ab_persist_probe 200 > gpurun_out/r4_persist_probe.txt 2>&1; cat gpurun_out/r4_persist_probe.txt
timeout 600 python bench.py > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err; tail -c 600 gpurun_out/r4_bench_a.err; python - <<'P'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_bench_a.json") if l.startswith("{")][0])
    print("BENCH", d["value"], d["ms_per_step"], d["roofline"]["launch_us"], d.get("roofline_stages",{}).get("by_group"))
    print("seed_pack", [(x["seeds_per_gpu"], round(x["ms_per_packed_iteration"],4)) for x in d.get("seed_pack",[])] if isinstance(d.get("seed_pack"),list) else d.get("seed_pack"))
except Exception as e:
    print("bench parse failed", e)
P
timeout 300 python profiles/qrisk_mix_probe.py 12000 3000 > gpurun_out/r4_qrisk_mix_probe.json 2> gpurun_out/r4_qrisk_mix_probe.err; tail -3 gpurun_out/r4_qrisk_mix_probe.err | cut -c1-1500
timeout 300 python profiles/learning_vec4096.py 16 1500 1 8 2 > gpurun_out/r4_learning_vec4096_config2.json 2> gpurun_out/r4_c2.err; grep "^{" gpurun_out/r4_c2.err | cut -c1-400
timeout 200 python profiles/learning_vec4096.py 16 1500 1 2 3 > gpurun_out/r4_learning_vec4096_config3.json 2> gpurun_out/r4_c3.err; grep "^{" gpurun_out/r4_c3.err | cut -c1-400
timeout 400 python profiles/learning_other_configs.py nav2_mf 5,6,7,8 > gpurun_out/r4_learning_nav2_mf_one_env.jsonl 2> gpurun_out/r4_mf1.err; cut -c1-300 gpurun_out/r4_learning_nav2_mf_one_env.jsonl
for s in 1 3 4 5 7 2; do timeout 1200 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 > gpurun_out/r4_c4_seed$s.json 2> gpurun_out/r4_c4_seed$s.err; grep "^{" gpurun_out/r4_c4_seed$s.err | cut -c1-500; done
