"""bench.py -- headline benchmark (BASELINE.json): env-steps/s + SAC grad-steps/s,
Navigation1, 4096 vectorised envs, SAC + Q_risk safety critic with model-free recovery
(configs[1] = scripts/navigation1.sh:7 + --num_envs 4096), one process per GPU.

A "step" is one lock-step iteration of the hot path over all envs: replay sample -> SAC update
-> Q_risk (+ recovery policy) update -> policy / Q_risk forward + recovery select -> env step
-> two replay pushes -> counters, replayed from ONE hipGraph.  Inputs are synthetic and
already resident in HBM.  Prints one JSON line (rank 0).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import arg_utils  # noqa: E402
from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd import distributed as dist_utils  # noqa: E402

NUM_ENVS = 4096
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
NAV_STEP_ALGO_BYTES = 39         # algorithmic bytes per env-step (f32 contract)


def config2_argv(seed, num_envs=NUM_ENVS):
    return ["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery",
            "--gamma_safe", "0.8", "--eps_safe", "0.3", "--num_unsafe_transitions", "20000",
            "--num_envs", str(num_envs), "--seed", str(seed)]


def build_loop(cfg, device, fast=True):
    from recovery_rl_amd.env import make_vec_env, register_env
    from recovery_rl_amd.experiment import VectorLoop
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    from recovery_rl_amd.sac import SAC
    torch.manual_seed(cfg.seed)
    register_env(cfg.env_name)
    env = make_vec_env(cfg.env_name, cfg.num_envs, device=device, seed=cfg.seed)
    agent = SAC(env.observation_space, env.action_space, cfg, "/tmp")
    if fast:
        agent.enable_fast_path(cfg.batch_size)
    memory = ReplayMemory(cfg.replay_size, cfg.seed, device=device)
    recovery_memory = ConstraintReplayMemory(cfg.safe_replay_size, cfg.seed, device=device)
    # offline constraint demonstrations + a short (untimed) Q_risk pre-training
    s, a, c, s2, m = env.transition_function(cfg.num_unsafe_transitions)
    recovery_memory.push(s.contiguous(), a.contiguous(), c.contiguous(), s2.contiguous(), m.contiguous())
    for _ in range(50):
        agent.safety_critic.update_parameters(memory=recovery_memory, policy=agent.policy,
                                              batch_size=cfg.batch_size)
    loop = VectorLoop(cfg, env, agent, memory, recovery_memory)
    loop.start()
    # leave the random-action / empty-buffer regime (start_steps=100, batch=256) eagerly
    while not (len(memory) > cfg.batch_size and loop.total_numsteps >= cfg.start_steps):
        loop.vector_step(do_update=False, random_actions=True)
    return loop


def time_nav_step_kernel(device, n, reps=200):
    """Average duration of ONE rrl_nav_step launch over `reps` launches, bracketed by events on
    the stream the kernel is launched on (torch's current stream)."""
    from recovery_rl_amd.env import make_vec_env
    env = make_vec_env("navigation1", n, device=device, seed=1)
    env.reset()
    lib = _lib.load()
    act = torch.rand(n, 2, device=device) * 2 - 1

    def launch():
        return lib.rrl_nav_step(0, n, _lib.ptr(env.pos), _lib.ptr(act), None, 1, 0, _lib.ptr(env.tick), 1,
                                _lib.ptr(env.next_obs), _lib.ptr(env.obs), _lib.ptr(env.reward),
                                _lib.ptr(env.done), _lib.ptr(env.constraint), _lib.ptr(env.success),
                                _lib.ptr(env.ep_done), _lib.ptr(env.t), 100, 1, _lib.current_stream())
    for _ in range(10):
        launch()
    torch.cuda.synchronize(device)
    # back-to-back launches from one captured graph: removes host launch gaps from the average
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            launch()
    g.replay()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_nav_rollout_kernel(device, n=1 << 20, T=100):
    """The fused-rollout variant of SURVEY 8d: T scripted steps per env in ONE launch with the state in registers
    (rrl_nav_rollout); per env-step only the 8-byte action is read and the reward + constraint flag written."""
    lib = _lib.load()
    pos = torch.randn(n, 2, dtype=torch.float64, device=device) + torch.tensor([-50.0, 0.0], dtype=torch.float64,
                                                                                device=device)
    acts = torch.rand(T, n, 2, device=device) * 2 - 1
    rew = torch.empty(T, n, device=device)
    cons = torch.empty(T, n, dtype=torch.uint8, device=device)

    def launch():
        return lib.rrl_nav_rollout(0, n, T, _lib.ptr(pos), _lib.ptr(acts), 1, 0, None, None, _lib.ptr(rew),
                                   _lib.ptr(cons), None, _lib.current_stream())
    launch()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e-3 / 3


def pmc_traffic(n):
    """HBM bytes per nav_step launch measured with rocprofv3 PMC counters (committed under profiles/;
    PMC passes cannot run inside this process).  None when no measurement exists for this size."""
    path = os.path.join(ROOT, "profiles", "round1_nav_step_pmc.json")
    try:
        rec = json.load(open(path)).get(str(n))
        return None if rec is None else rec["fetch_bytes"] + rec["write_bytes"]
    except (OSError, ValueError):
        return None


def planner_traffic(n_plans):
    """HBM bytes per plan_cost_kernel launch from the committed PMC passes (profiles/pmc_plan_traffic.sh)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "round1_planner_traffic.json"))).get(str(n_plans))
        return None if rec is None else rec["fetch_bytes"] + rec["write_bytes"]
    except (OSError, ValueError):
        return None


# algorithmic FLOPs of one lock-step iteration: SAC update 0.685 GFLOP, Q_risk + recovery
# update 0.62 GFLOP, acting = per env 2 x (policy 67 072 + twin Q_risk 133 632 + recovery policy 66 560) MAC
def iteration_flops(num_envs):
    return 0.685e9 + 0.62e9 + num_envs * 2.0 * (67072 + 133632 + 66560)


PLAN_FLOPS_PER_ROW_STEP = 267264 + 163200      # twin Q_risk (4-256-256-1 x2) + one ensemble member (4-200-200-200-4)
F32_MFMA_PEAK_TF = 157.3                       # MI355X_MICROARCH.md; 155.4 measured on this pool (profiles/mfma_peak.hip)


def time_planner_kernel(device, n_plans=256, reps=3):
    """rrl_plan_cost (MPC._compile_cost of config 4: 400 candidates x 20 particles x 5 steps per planning
    env): seconds per launch from HIP events on the launch stream."""
    from recovery_rl_amd.MPC import MPC
    from recovery_rl_amd.config import create_config
    from recovery_rl_amd.env import make_vec_env
    from recovery_rl_amd.sac import SAC
    env = make_vec_env("navigation2", 4, device=device, seed=1)
    mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1)
    args = arg_utils.get_args(["--env-name", "navigation2", "--cuda", "--use_recovery", "--gamma_safe", "0.65",
                               "--eps_safe", "0.2"])
    agent = SAC(env.observation_space, env.action_space, args, "/tmp")
    mpc.model.fit_input_stats(torch.randn(500, 4, device=device))
    mpc.has_been_trained = True
    mpc.update_value_func(agent.safety_critic)
    if mpc.fused is None:
        raise _lib.RRLError("rrl_plan_cost is not available for the config-4 planner shape")
    mpc.fused.pack()
    pop = mpc.optimizer.popsize
    acs = torch.rand(n_plans, pop, mpc.plan_hor * 2, device=device) * 2 - 1
    obs = torch.randn(n_plans, 2, device=device)
    mpc.fused.cost(acs, obs)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mpc.fused.cost(acs, obs)
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e-3 / reps, n_plans * pop * mpc.npart * mpc.plan_hor


def cpu_baseline(budget_s=15.0):
    """The reference-style loop (1 env, 1 SAC + 1 Q_risk update per env step; experiment.py:396-452)
    on the host cores: C oracle env + oracle replay + the same torch modules on the CPU."""
    from oracle import c_oracle as co
    from recovery_rl_amd.sac import SAC
    from recovery_rl_amd.spaces import Box
    cfg = arg_utils.get_args([a for a in config2_argv(1, 1) if a != "--cuda"])
    # batch-256 MLP updates do not scale past a few threads; oversubscribing a 128-core host is slower
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    torch.manual_seed(1)
    act_space = Box(-np.ones(2), np.ones(2))
    obs_space = Box(-np.ones(2) * np.inf, np.ones(2) * np.inf)
    agent = SAC(obs_space, act_space, cfg, "/tmp")
    mem, rmem = co.OracleReplay(100000), co.OracleReplay(100000)
    s, a, c, s2, m = co.nav_offline("navigation1", 20000, 1)
    rmem.push(s, a, c, s2, m)
    tt = lambda arrs: tuple(torch.from_numpy(x) for x in arrs)
    pos, obs, t = co.nav_reset("navigation1", 1, seed=1, counter=0)
    steps = updates = 0
    counter = 1
    t0 = time.perf_counter()
    timed_steps = timed_updates = 0
    t_start = None
    while True:
        if len(mem) > cfg.batch_size:
            if t_start is None:                       # time the steady state only
                t_start, timed_steps, timed_updates = time.perf_counter(), 0, 0
            agent.update_parameters(None, cfg.batch_size, updates, safety_critic=agent.safety_critic,
                                    batch=tt(mem.sample(cfg.batch_size, 1, counter)))
            agent.safety_critic.update_parameters(policy=agent.policy,
                                                  batch=tt(rmem.sample(cfg.batch_size, 2, counter)))
            updates += 1
            timed_updates += 1
        st = torch.from_numpy(obs)
        action = agent.select_action(st) if steps >= cfg.start_steps else torch.rand(1, 2) * 2 - 1
        risk = agent.safety_critic.get_value(st, action)
        real = agent.safety_critic.select_action(st) if float(risk) > cfg.eps_safe else action
        o = co.nav_step("navigation1", pos, real.numpy(), t, seed=1, counter=counter, auto_reset=True)
        mask = 1.0 - o["done"].astype(np.float32)
        mem.push(obs, action.numpy(), o["reward"], o["next_obs"], mask)
        rmem.push(obs, real.numpy(), o["constraint"].astype(np.float32), o["next_obs"], mask)
        pos, t, obs = o["pos"], o["t"], o["obs"]
        steps += 1
        timed_steps += 1
        counter += 1
        now = time.perf_counter()
        if t_start is not None and now - t_start > budget_s:
            break
        if now - t0 > 4 * budget_s:
            break
    dt = time.perf_counter() - (t_start or t0)
    # attribution (SURVEY 8d "CPU side-by-side"): the env alone, batched on ONE core of the same host
    n_env = 4096
    p4, _, t4 = co.nav_reset("navigation1", n_env, seed=1, counter=0)
    a4 = np.random.RandomState(0).uniform(-1, 1, (n_env, 2)).astype(np.float32)
    te = time.perf_counter()
    reps = 0
    while time.perf_counter() - te < 1.0:
        o = co.nav_step("navigation1", p4, a4, t4, seed=1, counter=reps + 1, auto_reset=True)
        p4, t4 = o["pos"], o["t"]
        reps += 1
    env_only = reps * n_env / (time.perf_counter() - te)
    return {"value": timed_steps / dt, "unit": "env-steps/s", "cores": torch.get_num_threads(),
            "kind": "port", "grad_steps_per_s": timed_updates / dt,
            "env_only_env_steps_per_s_1core": env_only,
            "sample": "%d env-steps of the reference-order loop (1 env, 1 SAC + 1 Q_risk/recovery update "
                      "per env-step, B=256, H=256) in %.1f s: C oracle env + oracle replay + torch CPU nets"
                      % (timed_steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--num_envs", type=int, default=NUM_ENVS)
    ap.add_argument("--no_graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--planner", action="store_true",
                    help="time the fused planner kernel of config 4 (MFMA roofline) -> roofline_planner; "
                         "on by default for single-GPU runs")
    ap.add_argument("--no_planner", action="store_true")
    ap.add_argument("--sweep", action="store_true",
                    help="also time rrl_nav_step at N = 2^12..2^24 (the bandwidth regime of the env kernel); off by "
                         "default so that every nav_step_kernel launch of the default command has the bench size")
    ap.add_argument("--no_sweep", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--autograd_updates", action="store_true",
                    help="PyTorch autograd + vendor GEMMs for the updates instead of the fused HIP kernels")
    a = ap.parse_args()

    rank, local_rank, world = dist_utils.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = dist_utils.local_device(local_rank)
    torch.cuda.set_device(device)

    cfg = arg_utils.get_args(config2_argv(dist_utils.rank_seed(1, rank), a.num_envs))
    loop = build_loop(cfg, device, fast=not a.autograd_updates)
    step = loop.replay if not a.no_graph else (lambda: loop.vector_step(True, False, True))
    if not a.no_graph:
        loop.capture(online_qrisk=True)
    for _ in range(a.warmup):
        step()
    stats0 = loop.read_stats()

    dist_utils.barrier(world)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(device)
    dist_utils.barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed = dist_utils.max_over_ranks(elapsed, world, device)

    stats1 = loop.read_stats()
    local = {k: stats1[k] - stats0[k] for k in stats1}
    agg = dist_utils.aggregate_stats(local, world, device)
    assert local["env_steps"] == a.steps * a.num_envs, (local["env_steps"], a.steps, a.num_envs)
    assert local["sac_updates"] == a.steps * cfg.updates_per_step

    extra = {}
    if rank == 0:
        t_k = time_nav_step_kernel(device, a.num_envs)
        extra["roofline"] = {
            "kernel": "nav_step_kernel<0,false> (rrl_nav_step)", "bound": "hbm",
            "achieved": a.num_envs * NAV_STEP_ALGO_BYTES / t_k / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": a.num_envs * NAV_STEP_ALGO_BYTES / t_k / 1e9 / HBM_PEAK_GBS,
            "traffic": pmc_traffic(a.num_envs), "launch_us": t_k * 1e6,
            "note": "N=%d moves only %d KB per launch: latency-bound; bandwidth regime (N up to 2^24, "
                    "`bench.py --sweep`): profiles/round1_roofline_sweep.json"
                    % (a.num_envs, a.num_envs * NAV_STEP_ALGO_BYTES // 1024)}
        if a.sweep and not a.no_sweep:
            sweep = []
            for logn in (12, 16, 20, 24):
                n = 1 << logn
                tk = time_nav_step_kernel(device, n, reps=200 if logn <= 16 else 20)
                sweep.append({"n_envs": n, "launch_us": tk * 1e6, "env_steps_per_s": n / tk,
                              "achieved_GBs": n * NAV_STEP_ALGO_BYTES / tk / 1e9,
                              "frac": n * NAV_STEP_ALGO_BYTES / tk / 1e9 / HBM_PEAK_GBS})
            extra["roofline_sweep"] = sweep
            n_r, t_r = 1 << 20, 100
            tr = time_nav_rollout_kernel(device, n_r, t_r)
            extra["roofline_rollout"] = {
                "kernel": "nav_rollout_kernel<0> (rrl_nav_rollout), %d envs x %d steps in one launch" % (n_r, t_r),
                "launch_ms": tr * 1e3, "env_steps_per_s": n_r * t_r / tr, "bytes_per_env_step": 13,
                "achieved_GBs": n_r * t_r * 13 / tr / 1e9, "frac": n_r * t_r * 13 / tr / 1e9 / HBM_PEAK_GBS,
                "note": "8 B action read + 4 B reward + 1 B constraint written per env-step; the state never leaves "
                        "registers, so the f64 Philox / Box-Muller arithmetic, not HBM, is the limit"}
        if (a.planner or world == 1) and not a.no_planner:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):       # the controller announces itself: stdout carries the JSON line only
                t_p, row_steps = time_planner_kernel(device)
            tf = row_steps * PLAN_FLOPS_PER_ROW_STEP / t_p / 1e12
            extra["roofline_planner"] = {
                "kernel": "plan_cost_kernel (rrl_plan_cost, model-based recovery of config 4)", "bound": "mfma",
                "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF,
                "traffic": planner_traffic(256), "launch_ms": t_p * 1e3, "row_steps_per_s": row_steps / t_p,
                "note": "f32-in/f32-acc MFMA (exact f32); algorithmic %d FLOP per particle-step"
                        % PLAN_FLOPS_PER_ROW_STEP}
        if not a.no_cpu_baseline and world == 1:
            extra["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        env_rate = agg["env_steps"] / elapsed
        out = {
            "metric": "env-steps/sec + SAC grad-steps/sec, Navigation1 4096 envs, 1/2/4/8 GPU",
            "value": env_rate, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "sac_grad_steps_per_s": agg["sac_updates"] / elapsed,
            "qrisk_grad_steps_per_s": agg["qrisk_updates"] / elapsed,
            "config": {"workload": "Navigation1, %d vectorised envs/GPU, SAC + Q_risk + model-free recovery "
                                   "(scripts/navigation1.sh:7 + --num_envs %d), batch 256, hidden 256, "
                                   "updates_per_step 1 (UTD 1/%d), one seed per GPU"
                                   % (a.num_envs, a.num_envs, a.num_envs),
                       "num_envs_per_gpu": a.num_envs, "batch_size": cfg.batch_size,
                       "hidden_size": cfg.hidden_size, "updates_per_step": cfg.updates_per_step,
                       "launch": "eager" if a.no_graph else "hipGraph replay",
                       "updates": "autograd + vendor GEMM" if a.autograd_updates else
                                  "hand-written HIP forward/backward (f32 MFMA) + fused Adam",
                       "parallelism": "replicas x%d (RCCL metric all-reduce only)" % world},
            "episodes": agg["episodes"], "violations": agg["num_viols"], "successes": agg["num_successes"],
            # the MLP side of the iteration against the f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
            "roofline_mlp": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3,
                             "achieved": iteration_flops(a.num_envs) * a.steps * world / elapsed / 1e12,
                             "frac": iteration_flops(a.num_envs) * a.steps * world / elapsed / 1e12 / (157.3 * world),
                             "note": "algorithmic FLOPs of SAC + Q_risk updates (B=256) and acting (N envs) per "
                                     "iteration / iteration time; tiny problems: launch- and latency-bound"},
        }
        out.update(extra)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
