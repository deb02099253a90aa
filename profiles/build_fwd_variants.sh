#!/bin/bash
# side libraries of csrc/mlp_fwd_kernels.hip built with extra -D switches (timing ablations / variants of the forward kernels;
# profiles/fwd_packed_probe.cpp and fwd_stream_probe.cpp time them):   bash profiles/build_fwd_variants.sh tag:-DX=1,-DY=2 ...
set -e
cd "$(dirname "$0")/.."
B=recovery_rl_amd/csrc/_build
OBJS=$(ls $B/*.o | grep -v mlp_fwd_kernels)
for spec in "$@"; do
  tag=${spec%%:*}; defs=$(echo "${spec#*:}" | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $defs -I include -I recovery_rl_amd/csrc \
      -c -o /tmp/_fwdv_$tag.o recovery_rl_amd/csrc/mlp_fwd_kernels.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o profiles/_ab_fwd_$tag.so /tmp/_fwdv_$tag.o $OBJS
  echo built profiles/_ab_fwd_$tag.so "($defs)"
done
