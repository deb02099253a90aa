#!/bin/bash
# side libraries of the stream-forward timing ablations (WRONG results by design; profiles/fwd_stream_probe.cpp times them):
#   profiles/_ab_fwd_ablate.hip = csrc/mlp_fwd_kernels.hip + the guards of profiles/patches/stream_ablate.patch
#   ABL bits: 1 matrix waves without MFMAs, 2 without LDS reads and MFMAs, 4 helpers without layer 1 (in the loop),
#             8 helpers without layer 3, 16 no W2 loads
set -e
cd "$(dirname "$0")/.."
B=recovery_rl_amd/csrc/_build
OBJS=$(ls $B/*.o | grep -v mlp_fwd_kernels)
for abl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DABL=$abl -I include -I recovery_rl_amd/csrc \
      -c -o /tmp/_abl_$abl.o -x hip profiles/_ab_fwd_ablate.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o profiles/_ab_stream_$abl.so /tmp/_abl_$abl.o $OBJS
done
/opt/rocm/bin/hipcc -O2 -o profiles/_ab_fwd_stream_probe profiles/fwd_stream_probe.cpp -ldl
