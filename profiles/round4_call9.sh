set +e
mkdir -p gpurun_out
for s in 5 7 3 1 8; do timeout 1300 python profiles/learning_vec4096.py 16 1650 $s $s 4 f16x3 --demo_share 0 > gpurun_out/r4_c4_cover_share0_seed$s.json 2> gpurun_out/r4_c4_cover_share0_seed$s.err; grep "^{" gpurun_out/r4_c4_cover_share0_seed$s.err | cut -c1-420; done
