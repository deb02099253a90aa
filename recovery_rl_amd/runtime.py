"""HIP runtime settings a LAUNCHER may opt into (bench.py, rrl_main.py) -- never applied by importing the package.

`graph_packet_capture`: ROCm's hipGraph replay either re-submits pre-captured AQL packets (runtime default) or walks
the regular command path (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, a debug variable of the ROCm 7.x runtime).  For the
lock-step iteration -- one graph of ~20 dependent tiny kernels -- the regular path measured 2.8 % faster on ROCm 7.2 /
MI355X.  That is a finding about ONE runtime release, so it is a LAUNCHER's choice gated on that release (`MEASURED_ROCM`; both launchers of this repository, bench.py and
rrl_main.py, make the same one: `configure(LAUNCHER_GRAPH_PACKET_CAPTURE)`; `RRL_GRAPH_PACKET_CAPTURE=1` in the environment
switches it back), it is logged, it never overrides an explicit DEBUG_CLR_GRAPH_PACKET_CAPTURE, and `settings()` reports what is in force so that a bench line says how it ran.
The variable is read by the runtime at its first call: `configure` must run before anything touches the GPU.
"""
import os
import sys

_VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
# ROCm releases on which the launchers' choice was MEASURED (a debug variable's meaning is not a contract across releases):
# on any other runtime the launchers leave the default alone unless the environment asks explicitly
MEASURED_ROCM = ("7.2",)
# what BOTH launchers (bench.py and rrl_main.py) ask for, so that the timed configuration is the one `python -m rrl_main` runs
LAUNCHER_GRAPH_PACKET_CAPTURE = 0
_applied = {}


def rocm_version():
    """'major.minor' of the installed ROCm runtime (ROCM_PATH/.info/version), or None when it cannot be told."""
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        try:
            with open(os.path.join(root, ".info", "version")) as f:
                return ".".join(f.read().strip().split(".")[:2])
        except (OSError, TypeError):
            continue
    return None


def configure(graph_packet_capture=None, log=True):
    """graph_packet_capture: None = leave the runtime default unless RRL_GRAPH_PACKET_CAPTURE is set; 0 / 1 = ask for
    that mode -- honoured only on the ROCm releases it was measured on (MEASURED_ROCM); an explicit
    RRL_GRAPH_PACKET_CAPTURE in the environment is honoured everywhere.  Returns settings()."""
    want = os.environ.get("RRL_GRAPH_PACKET_CAPTURE")
    if want is None and graph_packet_capture is not None:
        if rocm_version() in MEASURED_ROCM:
            want = str(int(graph_packet_capture))
        elif log:
            print("recovery_rl_amd.runtime: ROCm %s is not a release the hipGraph replay mode was measured on (%s): %s left "
                  "at the runtime default" % (rocm_version(), ", ".join(MEASURED_ROCM), _VAR), file=sys.stderr)
    if want is not None and _VAR not in os.environ:
        hip_started = "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized()
        if hip_started:
            if log:
                print("recovery_rl_amd.runtime: HIP is already initialised, %s left at the runtime default" % _VAR,
                      file=sys.stderr)
        else:
            os.environ[_VAR] = want
            _applied[_VAR] = want
            if log:
                print("recovery_rl_amd.runtime: %s=%s (hipGraph replay through the %s path)"
                      % (_VAR, want, "regular command" if want == "0" else "pre-captured packet"), file=sys.stderr)
    return settings()


def settings():
    return {_VAR: os.environ.get(_VAR, "runtime default"), "set_by_launcher": _VAR in _applied, "rocm": rocm_version()}
