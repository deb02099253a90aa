"""recovery_rl_amd -- MI355X-native hot path of Recovery RL (batched env step -> device replay
-> SAC / Q_risk updates -> PETS/CEM recovery), behind the reference's env / replay / agent
interfaces.  HIP kernels + C ABI in csrc/ (include/rrl_hip.h); no CPU fallback.

Importing the package has no side effects on the process environment; launchers that want the runtime settings of
`recovery_rl_amd.runtime` call `runtime.configure()` before the first HIP call."""

__version__ = "0.1.0"
