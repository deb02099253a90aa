"""Loader for librrl_hip.so (the C-ABI HIP library, include/rrl_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a GPU is not
present, the callers raise.  `build()` compiles the library in-tree with hipcc for gfx950
(cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("RRL_HIP_LIB") or os.path.join(CSRC, "librrl_hip.so")   # override: A/B builds in profiles/
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

HIP_SOURCES = ["nav_kernels.hip", "replay_kernels.hip", "maze_kernels.hip", "cem_kernels.hip",
               "mlp_kernels.hip", "mlp_fwd_kernels.hip", "update_kernels.hip", "log_kernels.hip", "plan_kernels.hip", "ens_train_kernels.hip",
               "ens_train_big_kernels.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wno-unused-function",
               "-Wno-bitwise-instead-of-logical"]

# Per-source additions.  The env translation units switch LLVM's atomic optimiser off: it rewrites every same-address
# atomicAdd as "first active lane adds, v_readfirstlane the result" -- and the readfirstlane puts an s_waitcnt vmcnt(0)
# right behind the atomic, so the cursor ticket and the episode table's slot reservation of step_push_kernel (single-lane
# returning atomics whose ~0.7 us round trip is meant to run under the env step) were waited for on the spot.  Every atomic in
# those sources is issued by one lane per wave already.
SOURCE_FLAGS = {
    "nav_kernels.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
    "maze_kernels.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
    "update_kernels.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],     # the Adam step ticket, same reason
}

EXPORTS = [
    "rrl_abi_version", "rrl_last_hip_error", "rrl_counter_add",
    "rrl_nav_step", "rrl_nav_step_compact", "rrl_nav_reset", "rrl_nav_rollout", "rrl_nav_offline_rollouts",
    "rrl_nav_offline",
    "rrl_maze_step", "rrl_maze_reset", "rrl_maze_offline",
    "rrl_replay_push", "rrl_replay_sample_gather", "rrl_creplay_sample_gather", "rrl_replay_sample_gather_split",
    "rrl_sample_multi",
    "rrl_nav_step_push", "rrl_maze_step_push", "rrl_nav_step_push_select", "rrl_maze_step_push_select",
    "rrl_nav_step_push_x", "rrl_maze_step_push_x",
    "rrl_sample_multi_packed", "rrl_pack_clear", "rrl_mlp3_forward_multi_packed", "rrl_mlp_head_backward_multi_packed",
    "rrl_mlp_hidden_backward_multi_packed", "rrl_mlp_backward_pair_multi_packed", "rrl_adam_step_multi_packed", "rrl_nav_step_push_packed",
    "rrl_maze_step_push_packed",
    "rrl_cem_sample", "rrl_cem_update", "rrl_cem_begin", "rrl_cem_sample_n", "rrl_cem_update_n", "rrl_cem_finish",
    "rrl_gemm_f32", "rrl_mlp3_forward", "rrl_mlp3_is_split", "rrl_mlp_head_backward", "rrl_mlp_head_backward_loss", "rrl_mlp_hidden_backward",
    "rrl_mlp_input_backward", "rrl_mlp3_forward_multi", "rrl_mlp_head_backward_multi", "rrl_mlp_hidden_backward_multi",
    "rrl_mlp_input_backward_multi", "rrl_mlp_backward_pair_multi", "rrl_policy_heads_fwd_multi",
    "rrl_gauss_head_fwd", "rrl_gauss_head_bwd", "rrl_sac_critic_grad", "rrl_sac_policy_grad",
    "rrl_qrisk_critic_grad", "rrl_qrisk_policy_grad", "rrl_stoch_head_fwd", "rrl_stoch_head_bwd",
    "rrl_adam_step", "rrl_adam_step_multi", "rrl_w2_pack", "rrl_normal_fill", "rrl_recovery_select", "rrl_episode_log_append",
    "rrl_plan_supported", "rrl_plan_pack_floats", "rrl_plan_scratch_floats", "rrl_plan_pack", "rrl_plan_cost", "rrl_plan_pack_f16x3",
    "rrl_plan_cost_f16x3", "rrl_plan_cost_n",
    "rrl_ens_train_supported", "rrl_ens_scratch_floats", "rrl_ens_train_grad", "rrl_ens_train_epoch",
    "rrl_ens_train_big_supported", "rrl_ens_big_scratch_floats", "rrl_ens_train_grad_big", "rrl_ens_train_epoch_big",
]

class RRLError(RuntimeError):
    pass


def _sources():
    return [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = _sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(INCLUDE, "rrl_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


_FLAG_PROBES = {}


def flags_supported(hipcc, flags):
    """True when this hipcc accepts `flags` (probed once per flag set by compiling an empty translation unit): the per-source
    additions are latency tuning that only newer LLVM knows ('Unknown command line argument' on older ROCm would fail the
    whole build).  tests/test_isa_cpu.py skips the assertions that depend on them when they are not in force."""
    key = (hipcc,) + tuple(flags)
    if key not in _FLAG_PROBES:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "probe.hip")
            open(src, "w").write("// empty\n")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-c", "-o", os.path.join(tmp, "probe.o"), src] + list(flags),
                               capture_output=True)
            _FLAG_PROBES[key] = r.returncode == 0
    return _FLAG_PROBES[key]


def build(force=False, verbose=False, jobs=None):
    """Compile csrc/*.hip into csrc/librrl_hip.so for gfx950: one object per source (compiled in parallel, re-used
    while neither the source nor any header changed), then one link."""
    if not force and not _stale():
        return SO_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "_build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(INCLUDE, "rrl_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        extra = SOURCE_FLAGS.get(os.path.basename(src), [])
        if extra and not flags_supported(hipcc, extra):
            extra = []              # a tuning flag of newer LLVM: the kernels are correct without it
        cmd = [hipcc] + compile_flags + extra + ["-I", INCLUDE, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(4, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", SO_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


class rrl_replay_t(C.Structure):
    _fields_ = [("s", C.c_void_p), ("a", C.c_void_p), ("r", C.c_void_p), ("s2", C.c_void_p),
                ("m", C.c_void_p), ("cap", C.c_int64), ("state", C.c_void_p),
                ("pos_cnt", C.c_void_p), ("flags", C.c_int32), ("pinned", C.c_int64)]


REPLAY_CLAMP_STRATIFIED = 1
DRAW_UNIFORM, DRAW_STRATIFIED, DRAW_DEMO_SHARE = 0, 1, 2


class rrl_episode_log_t(C.Structure):
    _fields_ = [("rec_i32", C.c_void_p), ("rec_f64", C.c_void_p), ("cap", C.c_int64), ("state", C.c_void_p)]


EPLOG_I32 = 6
ADAM_MAX_SEGS = 12
LOSS_SAC_CRITIC, LOSS_SAC_POLICY, LOSS_QRISK_CRITIC, LOSS_QRISK_POLICY, LOSS_GAUSS_HEAD, LOSS_STOCH_HEAD = range(6)


class rrl_ens_t(C.Structure):
    _fields_ = [("n_nets", C.c_int), ("d_in", C.c_int), ("hidden", C.c_int), ("d_out", C.c_int)] + [
        (name, C.c_void_p) for name in ("w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3", "max_logvar", "min_logvar",
                                        "mu", "sigma", "g_w0", "g_b0", "g_w1", "g_b1", "g_w2", "g_b2", "g_w3", "g_b3",
                                        "g_max_logvar", "g_min_logvar", "g_logvar_part", "g2_w0", "g2_b0", "g2_w1",
                                        "g2_b1", "g2_w2", "g2_b2", "g2_w3", "g2_b3", "loss_part")]


class rrl_loss_t(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_part", C.c_int), ("part_stride", C.c_longlong), ("out", C.c_void_p),
                ("out_t", C.c_void_p), ("v0", C.c_void_p), ("v1", C.c_void_p), ("v2", C.c_void_p),
                ("v3", C.c_void_p), ("alpha", C.c_void_p), ("f0", C.c_float), ("ld", C.c_int),
                ("n_heads", C.c_int), ("head_stride", C.c_longlong), ("d_action", C.c_void_p),
                ("loss", C.c_void_p), ("da_parts", C.c_int), ("da_part_stride", C.c_longlong), ("da_group", C.c_int)]


HEAD_GAUSS, HEAD_STOCH = 0, 1


class rrl_policy_head_t(C.Structure):
    _fields_ = [("kind", C.c_int), ("B", C.c_int), ("head", C.c_void_p), ("n_part", C.c_int),
                ("part_stride", C.c_longlong), ("eps", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
                ("action", C.c_void_p), ("ld_action", C.c_int), ("logp", C.c_void_p), ("mean_out", C.c_void_p),
                ("obs_in", C.c_void_p), ("obs_out", C.c_void_p), ("log_std", C.c_void_p), ("min_log_std", C.c_float)]


class rrl_step_push_t(C.Structure):
    _fields_ = [("n", C.c_int64), ("pos", C.c_void_p), ("t", C.c_void_p), ("status", C.c_void_p), ("obs", C.c_void_p),
                ("task_action", C.c_void_p), ("ld_task", C.c_int32), ("real_action", C.c_void_p), ("recovery", C.c_void_p),
                ("sel_z", C.c_void_p), ("sel_n_part", C.c_int32), ("sel_part_stride", C.c_longlong),
                ("sel_eps_safe", C.c_float), ("sel_rec_action", C.c_void_p),
                ("sel_rec_head", C.POINTER(rrl_policy_head_t)), ("real_action_out", C.c_void_p),
                ("recovery_out", C.c_void_p), ("seed", C.c_uint64), ("counter", C.c_uint64), ("counter_dev", C.c_void_p),
                ("counter_inc", C.c_uint64), ("horizon", C.c_int32), ("auto_reset", C.c_int32),
                ("reward_penalty", C.c_float), ("push_real_action", C.c_int32), ("memory", C.POINTER(rrl_replay_t)),
                ("recovery_memory", C.POINTER(rrl_replay_t)), ("next_obs", C.c_void_p), ("reward", C.c_void_p),
                ("done", C.c_void_p), ("constraint", C.c_void_p), ("success", C.c_void_p), ("ep_done", C.c_void_p),
                ("stats", C.c_void_p), ("reward_sums", C.c_void_p), ("ep_reward", C.c_void_p),
                ("log_rec_i32", C.c_void_p), ("log_rec_f64", C.c_void_p), ("log_cap", C.c_int64), ("log_state", C.c_void_p),
                ("log_len", C.c_void_p), ("log_ret", C.c_void_p), ("log_viol", C.c_void_p), ("log_rec", C.c_void_p)]


class rrl_stack_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("G", "M", "H", "din", "dout", "ldx")] + [
        (n, C.c_void_p) for n in ("x", "W1", "b1", "W2", "b2", "W3", "b3", "h1", "h2", "out", "scratch")] + [
        ("in_head", rrl_policy_head_t), ("use_in_head", C.c_int), ("W2p", C.c_void_p)]


class rrl_head_bwd_t(C.Structure):
    _fields_ = [("loss", rrl_loss_t)] + [(n, C.c_int) for n in ("G", "B", "H", "dout")] + [
        (n, C.c_void_p) for n in ("h2", "W3", "dW3", "db3", "dh2")]


class rrl_first_layer_t(C.Structure):
    _fields_ = [("x", C.c_void_p), ("W1", C.c_void_p), ("ldx", C.c_int), ("din", C.c_int), ("first_part", C.c_void_p),
                ("first_stride", C.c_longlong), ("dx_part", C.c_void_p), ("dx_fold", C.c_int)]


class rrl_hidden_bwd_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("G", "B", "H")] + [
        (n, C.c_void_p) for n in ("dh2", "h1", "W2", "dW2", "db2", "dh1")] + [("first", rrl_first_layer_t)]


class rrl_input_bwd_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("G", "B", "H", "din", "ldx")] + [
        (n, C.c_void_p) for n in ("dh1", "x", "W1", "dW1", "db1", "dx")]


class rrl_draw_t(C.Structure):
    _fields_ = [("rb", C.POINTER(rrl_replay_t)), ("stratified", C.c_int), ("n_pos", C.c_int32), ("n_neg", C.c_int32),
                ("seed", C.c_uint64), ("counter", C.c_uint64), ("counter_dev", C.c_void_p), ("counter_inc", C.c_uint64)] + [
        (n, C.c_void_p) for n in ("s", "a", "r", "s2", "m", "idx_out", "xu", "x2u", "xpu")]


class rrl_sample_args_t(C.Structure):
    _fields_ = [("first", C.POINTER(rrl_draw_t)), ("second", C.POINTER(rrl_draw_t)), ("noise_pairs", C.c_longlong),
                ("noise_seed", C.c_uint64), ("noise_counter", C.c_uint64), ("noise_counter_dev", C.c_void_p),
                ("noise_counter_inc", C.c_uint64), ("noise_out", C.c_void_p)]


class rrl_adam_seg_t(C.Structure):
    _fields_ = [("n", C.c_longlong), ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("step_dev", C.c_void_p), ("target", C.c_void_p), ("tau", C.c_float), ("weight_decay", C.c_float),
                ("g2", C.c_void_p), ("g_part", C.c_void_p), ("n_part", C.c_int), ("part_stride", C.c_longlong),
                ("part_elems", C.c_longlong), ("w2p", C.c_void_p), ("target_w2p", C.c_void_p), ("w2_off", C.c_longlong),
                ("w2_heads", C.c_int)]


class rrl_plan_weights_t(C.Structure):
    _fields_ = [("hq", C.c_int), ("he", C.c_int), ("n_nets", C.c_int)] + [
        (name, C.c_void_p) for name in ("q_w1", "q_b1", "q_w2", "q_b2", "q_w3", "q_b3", "e_w0", "e_b0", "e_w1",
                                        "e_b1", "e_w2", "e_b2", "e_w3", "e_b3", "inputs_mu", "inputs_sigma",
                                        "max_logvar", "min_logvar")]

_lib = None


def _declare(lib):
    vp, i32, i64, u64, ci, f64, f32, ll = (C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_int, C.c_double,
                                          C.c_float, C.c_longlong)
    rp = C.POINTER(rrl_replay_t)
    sig = {
        "rrl_abi_version": (ci, []),
        "rrl_last_hip_error": (ci, []),
        "rrl_counter_add": (ci, [vp, u64, vp]),
        "rrl_nav_step": (ci, [ci, i64, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, vp, vp, vp,
                              vp, i32, ci, vp]),
        "rrl_nav_step_compact": (ci, [ci, i64, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, i32, ci, vp]),
        "rrl_nav_reset": (ci, [ci, i64, vp, vp, vp, vp, vp, u64, u64, vp, vp]),
        "rrl_nav_rollout": (ci, [ci, i64, i32, vp, vp, u64, u64, vp, vp, vp, vp, vp, vp]),
        "rrl_nav_offline_rollouts": (i64, [ci, i64]),
        "rrl_nav_offline": (ci, [ci, i64, u64, vp, vp, vp, vp, vp, i64, vp, vp, vp]),
        "rrl_maze_step": (ci, [i64, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, i32, ci, vp]),
        "rrl_maze_reset": (ci, [i64, vp, vp, vp, vp, ci, ci, u64, u64, vp, vp]),
        "rrl_maze_offline": (ci, [i64, u64, vp, vp, vp, vp, vp, i64, vp]),
        "rrl_replay_push": (ci, [rp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_replay_sample_gather": (ci, [rp, i32, u64, u64, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_creplay_sample_gather": (ci, [rp, i32, i32, u64, u64, vp, u64, vp, vp, vp, vp, vp,
                                           vp, vp, vp, vp, vp]),
        "rrl_replay_sample_gather_split": (ci, [rp, i32, i32, u64, u64, vp, u64, vp, vp, vp, vp, vp,
                                                vp, vp, vp, vp, vp]),
        "rrl_nav_step_push": (ci, [ci, i64, vp, vp, vp, vp, vp, vp, u64, u64, vp, u64, i32, ci, f32, ci, rp, rp,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_maze_step_push": (ci, [i64, vp, vp, vp, vp, vp, vp, u64, u64, vp, u64, i32, ci, f32, ci, rp, rp,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_nav_step_push_select": (ci, [ci, i64, vp, vp, vp, vp, ci, vp, ci, ll, f32, vp, C.POINTER(rrl_policy_head_t), vp, vp,
                                          u64, u64, vp, u64, i32, ci,
                                          f32, ci, rp, rp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_maze_step_push_select": (ci, [i64, vp, vp, vp, vp, ci, vp, ci, ll, f32, vp, C.POINTER(rrl_policy_head_t), vp, vp,
                                           u64, u64, vp, u64, i32, ci,
                                           f32, ci, rp, rp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_sample_multi": (ci, [C.POINTER(rrl_draw_t), C.POINTER(rrl_draw_t), ll, u64, u64, vp, u64, vp, vp]),
        "rrl_mlp3_forward_multi": (ci, [ci, C.POINTER(rrl_stack_t), vp]),
        "rrl_mlp_head_backward_multi": (ci, [ci, C.POINTER(rrl_head_bwd_t), vp]),
        "rrl_mlp_hidden_backward_multi": (ci, [ci, C.POINTER(rrl_hidden_bwd_t), vp]),
        "rrl_mlp_input_backward_multi": (ci, [ci, C.POINTER(rrl_input_bwd_t), vp]),
        "rrl_mlp_backward_pair_multi": (ci, [ci, C.POINTER(rrl_head_bwd_t), C.POINTER(rrl_hidden_bwd_t), vp]),
        "rrl_policy_heads_fwd_multi": (ci, [ci, C.POINTER(rrl_policy_head_t), vp]),
        "rrl_sample_multi_packed": (ci, [ci, C.POINTER(rrl_sample_args_t), vp]),
        "rrl_pack_clear": (ci, []),
        "rrl_mlp3_forward_multi_packed": (ci, [ci, C.POINTER(ci), C.POINTER(C.POINTER(rrl_stack_t)), vp]),
        "rrl_mlp_head_backward_multi_packed": (ci, [ci, C.POINTER(ci), C.POINTER(C.POINTER(rrl_head_bwd_t)), vp]),
        "rrl_mlp_hidden_backward_multi_packed": (ci, [ci, C.POINTER(ci), C.POINTER(C.POINTER(rrl_hidden_bwd_t)), vp]),
        "rrl_mlp_backward_pair_multi_packed": (ci, [ci, C.POINTER(ci), C.POINTER(C.POINTER(rrl_head_bwd_t)),
                                                    C.POINTER(C.POINTER(rrl_hidden_bwd_t)), vp]),
        "rrl_adam_step_multi_packed": (ci, [ci, C.POINTER(ci), C.POINTER(C.POINTER(rrl_adam_seg_t)), C.POINTER(f32), f32,
                                            f32, f32, vp]),
        "rrl_nav_step_push_packed": (ci, [ci, ci, C.POINTER(rrl_step_push_t), vp]),
        "rrl_maze_step_push_packed": (ci, [ci, C.POINTER(rrl_step_push_t), vp]),
        "rrl_nav_step_push_x": (ci, [ci, C.POINTER(rrl_step_push_t), vp]),
        "rrl_maze_step_push_x": (ci, [C.POINTER(rrl_step_push_t), vp]),
        "rrl_cem_sample": (ci, [i64, i32, i32, vp, vp, vp, vp, f64, ci, vp, u64, u64, vp, u64, vp, vp]),
        "rrl_cem_update": (ci, [i64, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp]),
        "rrl_cem_begin": (ci, [i64, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_cem_sample_n": (ci, [vp, i64, i32, i32, vp, vp, vp, vp, f64, ci, vp, u64, u64, vp, u64, vp, vp]),
        "rrl_cem_update_n": (ci, [vp, i64, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp]),
        "rrl_cem_finish": (ci, [i64, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
        "rrl_gemm_f32": (ci, [ci, ci, ci, ci, ci, vp, ci, C.c_longlong, vp, ci, C.c_longlong, vp, ci,
                              C.c_longlong, vp, C.c_longlong, ci, vp, ci, C.c_longlong, vp, C.c_longlong,
                              ci, vp]),
        "rrl_mlp3_forward": (ci, [ci, ci, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
        "rrl_mlp3_is_split": (ci, [ci, ci]),
        "rrl_mlp_head_backward": (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_mlp_head_backward_loss": (ci, [C.POINTER(rrl_loss_t), ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]),
        "rrl_mlp_hidden_backward": (ci, [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
        "rrl_mlp_input_backward": (ci, [ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp]),
        "rrl_gauss_head_fwd": (ci, [ci, vp, ci, ll, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp]),
        "rrl_gauss_head_bwd": (ci, [ci, vp, ci, ll, vp, vp, vp, ci, ci, ll, f32, vp, vp]),
        "rrl_sac_critic_grad": (ci, [ci, vp, vp, ci, ll, vp, vp, vp, f32, vp, vp, vp, vp, vp]),
        "rrl_sac_policy_grad": (ci, [ci, vp, ci, ll, vp, vp, vp, vp, vp]),
        "rrl_qrisk_critic_grad": (ci, [ci, vp, vp, ci, ll, vp, vp, f32, vp, vp, vp]),
        "rrl_qrisk_policy_grad": (ci, [ci, vp, ci, ll, vp, vp, vp]),
        "rrl_stoch_head_fwd": (ci, [ci, vp, ci, ll, vp, vp, f32, vp, vp, vp, ci, vp, vp]),
        "rrl_stoch_head_bwd": (ci, [ci, vp, ci, ll, vp, vp, f32, vp, vp, ci, ci, ll, vp, vp, vp]),
        "rrl_adam_step": (ci, [C.c_longlong, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, f32, vp]),
        "rrl_adam_step_multi": (ci, [ci, C.POINTER(rrl_adam_seg_t), f32, f32, f32, f32, vp]),
        "rrl_w2_pack": (ci, [ci, ci, vp, vp, vp]),
        "rrl_normal_fill": (ci, [ll, u64, u64, vp, u64, vp, vp]),
        "rrl_recovery_select": (ci, [ci, vp, f32, vp, ci, vp, vp, vp, vp, vp]),
        "rrl_plan_supported": (ci, [ci, ci, ci, ci, ci, ci]),
        "rrl_plan_pack_floats": (ll, [ci, ci, ci]),
        "rrl_plan_scratch_floats": (ll, [ci, ll, ci]),
        "rrl_plan_pack": (ci, [C.POINTER(rrl_plan_weights_t), vp, vp]),
        "rrl_plan_cost": (ci, [vp, ci, ci, ci, ci, ll, ci, ci, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp]),
        "rrl_plan_pack_f16x3": (ci, [C.POINTER(rrl_plan_weights_t), vp, vp]),
        "rrl_plan_cost_f16x3": (ci, [vp, ci, ci, ci, ci, ll, ci, ci, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp]),
        "rrl_plan_cost_n": (ci, [ci, vp, ci, ci, ci, ci, vp, ll, ci, ci, vp, vp, vp, u64, u64, vp, u64, vp, vp, vp]),
        "rrl_ens_train_supported": (ci, [ci, ci, ci, ci]),
        "rrl_ens_scratch_floats": (ll, [ci]),
        "rrl_ens_train_epoch": (ci, [C.POINTER(rrl_ens_t), ci, C.POINTER(rrl_adam_seg_t), f32, f32, f32, f32, vp, vp,
                                     vp, ll, ll, ci, vp, vp, vp]),
        "rrl_ens_train_grad": (ci, [C.POINTER(rrl_ens_t), ci, vp, vp, vp, ll, vp, vp, vp]),
        "rrl_ens_train_big_supported": (ci, [ci, ci, ci]),
        "rrl_ens_big_scratch_floats": (ll, [ci, ll]),
        "rrl_ens_train_grad_big": (ci, [C.POINTER(rrl_ens_t), ll, vp, vp, vp, ll, vp, vp, vp]),
        "rrl_ens_train_epoch_big": (ci, [C.POINTER(rrl_ens_t), ci, C.POINTER(rrl_adam_seg_t), f32, f32, f32, f32, vp, vp,
                                         vp, ll, ll, ll, vp, vp, vp]),
        "rrl_episode_log_append": (ci, [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(rrl_episode_log_t), vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name in EXPORTS:
        if name not in sig:
            getattr(lib, name)


def load():
    """Load the library (after torch, so that both share torch's libamdhip64 runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- must come first: same HIP runtime instance as torch
    if not os.path.exists(SO_PATH):
        raise RRLError(
            "librrl_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
            "g.build()'`; there is no CPU fallback for the hot path." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    _declare(lib)
    _lib = lib
    return lib


_RC_TEXT = {-1: "invalid argument", -2: "launch failed", -3: "size out of range",
            -5: "a packed launch met a new argument block while the stream was capturing (launch it once before the capture)",
            -6: "too many distinct argument blocks of packed launches alive (rrl_pack_clear)"}


def check(rc, what):
    if rc != 0:
        lib = load()
        raise RRLError("%s failed: rc=%d (%s) hipError=%d" % (what, rc, _RC_TEXT.get(rc, "?"), lib.rrl_last_hip_error()))


def require_gpu(device):
    import torch
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RRLError("recovery_rl_amd runs its hot path on an MI355X only (device=%r, "
                       "cuda available=%s); there is no CPU fallback." % (device, torch.cuda.is_available()))
    return dev


def ptr(t):
    return None if t is None else t.data_ptr()


def current_stream():
    """Raw hipStream_t of torch's current stream on the current device.  Called once per kernel launch, so it uses
    torch's C accessors when they exist (0.3 us) instead of building a torch.cuda.Stream object (9 us)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    dev = getattr(torch._C, "_cuda_getDevice", None)
    if raw is not None and dev is not None:
        return raw(dev())
    return torch.cuda.current_stream().cuda_stream
