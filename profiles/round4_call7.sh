set +e
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_loop_gpu.py -m gpu -q -p no:cacheprovider -k "q_sampling or representation" 2>&1 | tail -8) > gpurun_out/r4_pytest_new2.txt 2>&1; tail -4 gpurun_out/r4_pytest_new2.txt | cut -c1-300
for s in 3 5 7 1; do timeout 600 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --safe_replay_size 7300000 > gpurun_out/r4_c4_bigring_seed$s.json 2> gpurun_out/r4_c4_bigring_seed$s.err; grep "^{" gpurun_out/r4_c4_bigring_seed$s.err | cut -c1-520; done
for s in 5 7; do timeout 600 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --safe_replay_size 7300000 --replay_size 7300000 > gpurun_out/r4_c4_bigboth_seed$s.json 2> gpurun_out/r4_c4_bigboth_seed$s.err; grep "^{" gpurun_out/r4_c4_bigboth_seed$s.err | cut -c1-520; done
