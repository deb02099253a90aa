set +e
mkdir -p gpurun_out
timeout 300 profiles/_ab_persist_probe 100 > gpurun_out/r4_persist_probe.txt 2>&1; cat gpurun_out/r4_persist_probe.txt
for s in 3 5; do timeout 600 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --demo_share 0.75 > gpurun_out/r4_c4_share075_seed$s.json 2> gpurun_out/r4_c4_share075_seed$s.err; grep "^{" gpurun_out/r4_c4_share075_seed$s.err | cut -c1-520; done
