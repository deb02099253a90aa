set +e
mkdir -p gpurun_out
for s in 1 3 4 5 7; do timeout 700 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 > gpurun_out/r4_c4_cover_seed$s.json 2> gpurun_out/r4_c4_cover_seed$s.err; grep "^{" gpurun_out/r4_c4_cover_seed$s.err | cut -c1-420; done
timeout 300 python profiles/learning_vec4096.py 16 1500 1 8 2 > gpurun_out/r4_c2_cover.json 2> gpurun_out/r4_c2_cover.err; grep "^{" gpurun_out/r4_c2_cover.err | cut -c1-330
timeout 200 python profiles/learning_vec4096.py 16 1500 1 2 3 > gpurun_out/r4_c3_cover.json 2> gpurun_out/r4_c3_cover.err; grep "^{" gpurun_out/r4_c3_cover.err | cut -c1-330
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/r4_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4_pytest_gpu.txt | cut -c1-300
for s in 8 6 2; do timeout 1300 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 > gpurun_out/r4_c4_cover_seed$s.json 2> gpurun_out/r4_c4_cover_seed$s.err; grep "^{" gpurun_out/r4_c4_cover_seed$s.err | cut -c1-420; done
