"""recovery_rl_amd -- MI355X-native hot path of Recovery RL (batched env step -> device replay
-> SAC / Q_risk updates -> PETS/CEM recovery), behind the reference's env / replay / agent
interfaces.  HIP kernels + C ABI in csrc/ (include/rrl_hip.h); no CPU fallback."""

__version__ = "0.1.0"

import os as _os

# Runtime setting, read by the HIP runtime at its first call (so importing this package before touching the GPU is enough):
# the lock-step iteration is a hipGraph of ~22 dependent tiny kernels, and on ROCm 7.2 / MI355X replaying it through the
# regular command path is 2.8 % faster than through pre-captured AQL packets (0.1945 vs 0.2003 ms per iteration, measured
# back to back in one session).  An explicit setting in the environment wins.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
