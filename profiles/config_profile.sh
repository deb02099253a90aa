#!/bin/bash
# rocprofv3 kernel stats of the Maze leg (bench.py --env maze) and of the config-4 loop with both planner kernels
# (config_rates.py 4 ...) -> gpurun_out/config_prof/
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/config_prof
mkdir -p $OUT
rm -rf /tmp/cp1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp1 -o p -- python $R/bench.py --env maze --no_cpu_baseline --no_planner > $OUT/maze_under_rocprof.json 2>/dev/null
cp $(find /tmp/cp1 -name "*kernel_stats.csv" | head -1) $OUT/maze_kernel_stats.csv
for P in f32 f16x3; do
  rm -rf /tmp/cp2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp2 -o p -- python $R/profiles/config_rates.py 4 4096 10000 $P > $OUT/config4_${P}_under_rocprof.json 2>/dev/null
  cp $(find /tmp/cp2 -name "*kernel_stats.csv" | head -1) $OUT/config4_${P}_kernel_stats.csv
done
head -6 $OUT/maze_kernel_stats.csv | cut -c1-160; head -6 $OUT/config4_f16x3_kernel_stats.csv | cut -c1-160
