#!/bin/bash
# First GPU call of round 4: the whole-line W2 loads + ds_bpermute restage of the fused forward (csrc/mlp_kernels.hip,
# -DRRL_COALESCE_W2=2 / 3; side libraries librrl_hip_w2perm.so / _w2perm_all.so built by __graft_entry__.build()) on hardware.
#   1. bit-equality with the default library + timings of the 4096-row / 256-row forward and of the iteration
#      (tests/test_w2_permute_gpu.py; timings in the warnings summary)
#   2. A/B of the headline leg on ONE box (profiles/ab_lib.py: boxes of the pool differ by ~2 %)
#   3. rocprofv3 kernel stats of the headline leg per library: the average of mlp3_fwd_split_* must move, not only the probe
#   4. the packed iteration at S = 16 per library (the forward stack is its largest launch: DESIGN 5b)
# Results -> gpurun_out/r4_w2perm/.   gpurun --timeout 1500 -- 'bash profiles/round4_w2perm.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4_w2perm
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_w2_permute_gpu.py -m gpu -q -rxX -W always 2>&1 | tail -30 > $OUT/pytest.txt
cat $OUT/pytest.txt
for v in w2perm w2perm_all w2perm_bwd; do
    lib=$R/recovery_rl_amd/csrc/librrl_hip_$v.so
    [ -f $lib ] || { echo "missing $lib"; continue; }
    timeout 300 python profiles/ab_lib.py $lib 3 > $OUT/ab_$v.json 2> $OUT/ab_$v.err
    cat $OUT/ab_$v.json
done
cd /tmp && export TMPDIR=/tmp
for v in default w2perm w2perm_all w2perm_bwd; do
    lib=$R/recovery_rl_amd/csrc/librrl_hip_$v.so
    [ $v = default ] && lib=$R/recovery_rl_amd/csrc/librrl_hip.so
    rm -rf /tmp/pw_$v
    RRL_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_$v -o p -- \
        python $R/bench.py --no_cpu_baseline --no_legs --no_planner --min_seconds 1.0 > $OUT/bench_$v.json 2>/tmp/pw_$v.err
    f=$(find /tmp/pw_$v -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/kernel_stats_$v.csv && grep -E "mlp3_fwd|Name" $OUT/kernel_stats_$v.csv | cut -c1-200
done
cd $R
for v in default w2perm w2perm_all w2perm_bwd; do
    lib=$R/recovery_rl_amd/csrc/librrl_hip_$v.so
    [ $v = default ] && lib=$R/recovery_rl_amd/csrc/librrl_hip.so
    RRL_HIP_LIB=$lib timeout 300 python profiles/packed_probe.py 1 4,16 300 > $OUT/packed_$v.json 2> $OUT/packed_$v.err
    cat $OUT/packed_$v.json
done
