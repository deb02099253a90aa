"""bench.py -- headline benchmark (BASELINE.json): env-steps/s + SAC grad-steps/s,
Navigation1, 4096 vectorised envs, SAC + Q_risk safety critic with model-free recovery
(configs[1] = scripts/navigation1.sh:7 + --num_envs 4096), one process per GPU.

A "step" is one lock-step iteration of the hot path over all envs: replay sample -> SAC update
-> Q_risk (+ recovery policy) update -> policy / Q_risk forward + recovery select -> env step
-> two replay pushes -> counters, replayed from ONE hipGraph.  Inputs are synthetic and
already resident in HBM.  Prints one JSON line (rank 0).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python bench.py --gpus 8                       # spawns 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --env maze                     # config 3 / the Maze leg of config 5 (scripts/maze.sh:7)
    python bench.py --utd_sweep                    # U = 1, 4, 16, 64 updates per lock-step iteration

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize; blocks are
repeated until >= MIN_TIMED_S of timed work exists (a 20-step block is 6 ms -- too short to report on), the
number of blocks is agreed over ranks, and every figure is over all timed steps (max over ranks of the time).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from recovery_rl_amd import runtime as rrl_runtime  # noqa: E402

# the timed graph is replayed through the runtime's regular command path (2.8 % faster than pre-captured packets for this
# chain of tiny kernels on ROCm 7.2); the same request rrl_main.py makes, reported in the JSON line ("runtime"), RRL_GRAPH_PACKET_CAPTURE=1 or
# an explicit DEBUG_CLR_GRAPH_PACKET_CAPTURE in the environment wins
RUNTIME = rrl_runtime.configure(graph_packet_capture=rrl_runtime.LAUNCHER_GRAPH_PACKET_CAPTURE, log=False)

NUM_ENVS = 4096
MIN_TIMED_S = 3.0                # timed region of the headline leg (an external SMI sampler must be able to see it)
PREWARM_S = 1.5       # untimed busy time before the warm-up steps of the headline run (cold-box clocks)
MIN_TIMED_LEG_S = 1.0            # ... of the secondary legs (U = 16, config 4)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
NAV_STEP_ALGO_BYTES = 39         # algorithmic bytes per env-step (f32 contract)
STEP_PUSH_ALGO_BYTES = 39 + 32 + 32   # + one 32-byte replay row into each of the two buffers (section 8d "replay")
F32_MFMA_PEAK_TF = 157.3         # MI355X_MICROARCH.md; 155.4 measured on this pool (profiles/mfma_peak.hip)
F16_MFMA_PEAK_TF = 2500.0                       # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
PLAN_Q_FLOPS, PLAN_E_FLOPS = 267264, 163200     # per row: twin Q_risk (4-256-256-1 x2), one ensemble member (4-200-200-200-4)
PLAN_FLOPS_PER_ROW_STEP = PLAN_Q_FLOPS + PLAN_E_FLOPS   # the LITERAL loop of MPC.py:404-412: both networks, every row, every step


def plan_flops(plans, pop=400, npart=20, nets=5, plan_hor=5):
    """FLOPs of `plans` evaluations of MPC._compile_cost (one per planning env and CEM iteration): (needed, literal).
    needed = what the costs depend on and what the kernels execute since round 6: at t = 0 Q_risk once per candidate and each
    member once per (candidate, member) -- the particles share (cur_obs, ac_0), MPC.py:393-402 --, no prediction at the last
    step (it feeds a cur_obs nothing reads, :406-412): 12.90 GFLOP at the config-4 shape; literal = every particle row through
    both networks at every step, as the reference's loop spends them: 17.22 GFLOP."""
    rows = pop * npart
    q = PLAN_Q_FLOPS * (pop + rows * (plan_hor - 1))
    e = PLAN_E_FLOPS * (pop * nets + rows * (plan_hor - 2)) if plan_hor > 1 else 0
    return plans * float(q + e), plans * float(rows * plan_hor * PLAN_FLOPS_PER_ROW_STEP)


def plan_f16_products(plans, pop=400, npart=20, nets=5, plan_hor=5):
    """f16 MFMA FLOPs the f16x3 kernels execute for `plans` evaluations: three products per hidden-layer product (Q_risk
    256 x 256 per head, the member's two 200 x 200 layers) on the rows plan_flops counts as needed."""
    rows = pop * npart
    q_rows = pop + rows * (plan_hor - 1)
    e_rows = pop * nets + rows * (plan_hor - 2) if plan_hor > 1 else 0
    return plans * 3.0 * (2 * 2 * 256 * 256 * q_rows + 2 * 2 * 200 * 200 * e_rows)

CONFIG_ARGV = {
    # configs[1]: scripts/navigation1.sh:7
    "navigation1": ["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8",
                    "--eps_safe", "0.3", "--num_unsafe_transitions", "20000"],
    # configs[2] / the Maze leg of configs[4]: scripts/maze.sh:7
    "maze": ["--env-name", "maze", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe",
             "0.15", "--pos_fraction", "0.3"],
}


def config_argv(env, seed, num_envs=NUM_ENVS, updates_per_step=1):
    return CONFIG_ARGV[env] + ["--num_envs", str(num_envs), "--seed", str(seed), "--updates_per_step",
                               str(updates_per_step)]


def config2_argv(seed, num_envs=NUM_ENVS):
    return config_argv("navigation1", seed, num_envs)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `--gpus N` with no torchrun environment re-executes this script as N ranks, one per GPU
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, n_ranks, port):
    """The reference's unit of parallelism is the seed loop (scripts/navigation1.sh:4-8): rank g = seed base + g."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(gpus, environ):
    """(spawn?, world) from --gpus and the torchrun environment; raises when they contradict each other."""
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but launched with WORLD_SIZE=%d; start it as `python bench.py --gpus "
                             "%d` or under torch.distributed.run with --nproc-per-node %d" % (gpus, world, gpus, gpus))
        return False, world
    return gpus > 1, gpus


def spawn_ranks(argv, n_ranks):
    import torch
    backend = os.environ.get("RRL_DIST_BACKEND") or "nccl"
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n_ranks:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one device per rank; set "
                         "RRL_DIST_BACKEND=gloo to dry-run the multi-rank path on fewer devices)" % (n_ranks, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launch_command(argv, n_ranks, free_port())
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
def build_loop(cfg, device, fast=True, pretrain=50, episode_log=True):
    import torch
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
    recovery_memory.pin()                 # as Experiment.pretrain_critic_recovery does for the lock-step loop
    if cfg.num_envs > 1 and cfg.pos_fraction < 0:
        # ... and the demonstration share of the Q_risk batch (Experiment._apply_demo_share: 0.5 by default)
        share = float(getattr(cfg, "demo_share", -1.0))
        agent.safety_critic.demo_share = (0.5 if share < 0 else share) or None
    for _ in range(pretrain):
        agent.safety_critic.update_parameters(memory=recovery_memory, policy=agent.policy,
                                              batch_size=cfg.batch_size)
    loop = VectorLoop(cfg, env, agent, memory, recovery_memory)
    if episode_log and cfg.num_envs > 1:
        # what Experiment.run_vectorized installs: the per-episode table, advanced by the env-step launch itself
        from recovery_rl_amd.episode_log import EpisodeLog
        loop.episode_log = EpisodeLog(cfg.num_envs, cfg.num_envs * (LOG_EVERY + 4), device)
    loop.start()
    # leave the random-action / empty-buffer regime (start_steps=100, batch=256) eagerly
    while not (len(memory) > cfg.batch_size and loop.total_numsteps >= cfg.start_steps):
        loop.vector_step(do_update=False, random_actions=True)
    return loop


LOG_EVERY = 100      # Experiment.run_vectorized's default logging cadence (--log_every 0)


def production_step(step, loops, every=None, advance=None):
    """`step` plus what the lock-step driver does every LOG_EVERY iterations INSIDE its loop (experiment.py run_vectorized):
    read the counters and the samplers' error flags, drain the episode table -- two host synchronisations per 100
    iterations, part of the timed region because they are part of every real run.  `advance(n)` (VectorLoop.advance /
    PackedLoop.advance): n iterations at once from the loops' many-iteration graphs; `run.many(n)` uses it exactly as the
    driver does -- whole graphs up to the next log point, single iterations for what is left of a block."""
    count = [0]
    ev = LOG_EVERY if every is None else every

    def log_point():
        for loop in loops:
            loop.read_stats()
            if loop.episode_log is not None:
                loop.episode_log.drain()

    def run():
        step()
        count[0] += 1
        if ev and count[0] % ev == 0:
            log_point()

    def many(n):
        while n > 0:
            m = min(n, ev - count[0] % ev) if ev else n
            if advance is not None:
                advance(m)
            else:
                for _ in range(m):
                    step()
            count[0] += m
            n -= m
            if ev and count[0] % ev == 0:
                log_point()
    run.many = many
    return run


def device_update_counts(loop):
    """Adam step counters kept ON THE DEVICE by the optimiser kernels (graph replays advance them): the witness that
    the timed region really ran its optimiser steps.  (SAC critic, Q_risk critic)."""
    fast = getattr(loop.agent, "fast", None)
    if fast is None:
        return None
    return int(fast.critic.step[0].item()), int(fast.qrisk.step[0].item())


def timed_blocks(step, steps, world, device, min_seconds=MIN_TIMED_S, max_blocks=100000):
    """Blocks of exactly `steps` steps, each bracketed by barrier + synchronize on both sides; repeated until
    `min_seconds` of timed work (the count is the same on all ranks).  Returns (sum over blocks of the max-over-ranks
    block time, number of blocks)."""
    import torch
    from recovery_rl_amd import distributed as dist_utils
    total, blocks = 0.0, 0
    while True:
        dist_utils.barrier(world)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        if hasattr(step, "many"):
            step.many(steps)
        else:
            for _ in range(steps):
                step()
        torch.cuda.synchronize(device)
        dist_utils.barrier(world)
        dt = dist_utils.max_over_ranks(time.perf_counter() - t0, world, device)
        total += dt
        blocks += 1
        if total >= min_seconds or blocks >= max_blocks:       # `total` is all-reduced: every rank stops together
            return total, blocks


def _graph_of(launch, reps, device):
    import torch
    for _ in range(10):
        launch()
    torch.cuda.synchronize(device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):          # back-to-back launches from one captured graph: no host launch gaps
        for _ in range(reps):
            launch()
    g.replay()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                        # events on torch's current stream = the stream the kernels are launched on
    g.replay()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e-3 / reps


def step_push_launcher(device, env_name, n, compact=True, log=False):
    """A closure launching ONE step_push_kernel with the bench's buffer shapes.  compact (what the timed graph launches):
    u16 status word instead of step count + four flag arrays, stored state from pos, no per-env output arrays; otherwise the
    reference-shaped arrays with every optional output.  log: the per-episode table advanced by the same launch (what the
    lock-step driver and the timed graph do)."""
    import ctypes as C
    import torch
    from recovery_rl_amd import _lib
    from recovery_rl_amd.env import make_vec_env
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    env = make_vec_env(env_name, n, device=device, seed=1)
    env.reset()
    lib = _lib.load()
    hi = float(env.action_space.high[0])
    act = (torch.rand(n, 2, device=device) * 2 - 1) * hi
    real = (torch.rand(n, 2, device=device) * 2 - 1) * hi
    rec = (torch.rand(n, device=device) < 0.2).to(torch.uint8)
    cap = max(1000000, 2 * n)
    mem, rmem = ReplayMemory(cap, 1, device=device), ConstraintReplayMemory(cap, 1, device=device)
    stats = torch.zeros(10, dtype=torch.int64, device=device)
    sums = torch.zeros(2, dtype=torch.float64, device=device)
    ep_reward = torch.zeros(n, device=device)
    p = _lib.ptr
    a = _lib.rrl_step_push_t()
    a.n, a.pos, a.obs = n, p(env.pos), p(env.obs)
    if compact:
        a.status = p(env.use_status())
    else:
        a.t = p(env.t)
        a.next_obs, a.reward = p(env.next_obs), p(env.reward)
        a.done, a.constraint, a.success, a.ep_done = p(env.done), p(env.constraint), p(env.success), p(env.ep_done)
    a.task_action, a.ld_task, a.real_action, a.recovery = p(act), 2, p(real), p(rec)
    a.seed, a.counter, a.counter_dev, a.counter_inc = env.seed_value, 0, p(env.tick), 1
    a.horizon, a.auto_reset, a.reward_penalty, a.push_real_action = env.horizon, 1, 0.0, 0
    a.memory, a.recovery_memory = C.pointer(mem._desc), C.pointer(rmem._desc)
    a.stats, a.reward_sums, a.ep_reward = p(stats), p(sums), p(ep_reward)
    ep_log = None
    if log:
        from recovery_rl_amd.episode_log import EpisodeLog
        ep_log = EpisodeLog(n, n * 8, device)      # overflowing records are dropped by the kernel (drain() would report it)
        ep_log.attach(a)
    keep = (env, act, real, rec, mem, rmem, stats, sums, ep_reward, a, ep_log)

    def launch():
        keep[0].num_envs        # (the closure owns the buffers)
        if env_name == "maze":
            return lib.rrl_maze_step_push_x(C.byref(a), _lib.current_stream())
        return lib.rrl_nav_step_push_x(env.kind, C.byref(a), _lib.current_stream())
    return launch


def time_step_push_kernel(device, env_name, n, reps=200, compact=True, log=False):
    """Average duration of ONE step_push_kernel launch (the env-step + replay-push kernel of the timed iteration)
    over `reps` back-to-back launches with the bench's own buffers shapes: HIP events on the launch stream."""
    return _graph_of(step_push_launcher(device, env_name, n, compact, log), reps, device)


def time_nav_step_kernel(device, n, reps=200):
    """Average duration of ONE rrl_nav_step launch (the stand-alone env step; the sweep's kernel)."""
    import torch
    from recovery_rl_amd import _lib
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
    return _graph_of(launch, reps, device)


def time_nav_step_compact_kernel(device, n, reps=200):
    """Average duration of ONE rrl_nav_step_compact launch: the same env step in the 56 B/env-step layout (u16 status
    words instead of four u8 masks + an i32 count; no second observation array: the post-reset observation of the ~1 %
    finished rows is float(pos))."""
    import torch
    from recovery_rl_amd import _lib
    from recovery_rl_amd.env import make_vec_env
    env = make_vec_env("navigation1", n, device=device, seed=1)
    env.reset()
    lib = _lib.load()
    act = torch.rand(n, 2, device=device) * 2 - 1
    status = torch.zeros(n, dtype=torch.int16, device=device)

    def launch():
        return lib.rrl_nav_step_compact(0, n, _lib.ptr(env.pos), _lib.ptr(act), None, 1, 0, _lib.ptr(env.tick), 1,
                                        _lib.ptr(env.next_obs), None, _lib.ptr(env.reward),
                                        _lib.ptr(status), 100, 1, _lib.current_stream())
    return _graph_of(launch, reps, device)


def time_nav_rollout_kernel(device, n=1 << 20, T=100):
    """The fused-rollout variant of SURVEY 8d: T scripted steps per env in ONE launch with the state in registers
    (rrl_nav_rollout); per env-step only the 8-byte action is read and the reward + constraint flag written."""
    import torch
    from recovery_rl_amd import _lib
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


def latest_sweep():
    """The newest committed `bench.py --sweep` record under profiles/."""
    for rnd in range(9, 0, -1):
        rel = os.path.join("profiles", "round%d_roofline_sweep.json" % rnd)
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return "profiles/ (no sweep committed)"


def committed_pmc(name, key):
    """(HBM bytes per launch, source file) from the committed rocprofv3 PMC passes under profiles/ (PMC passes cannot run
    inside this process: the figure is NOT measured by this run, `traffic_source` in the JSON line says where it comes
    from).  (None, None) when no measurement exists for this size."""
    for rnd in ("round6", "round5", "round4", "round3", "round2", "round1"):
        rel = os.path.join("profiles", "%s_%s.json" % (rnd, name))
        try:
            rec = json.load(open(os.path.join(ROOT, rel))).get(str(key))
            if rec is not None:
                return rec["fetch_bytes"] + rec["write_bytes"], rel
        except (OSError, ValueError):
            pass
    return None, None


def iteration_flops(num_envs, updates_per_step=1, batch=256, hidden=256):
    """MLP FLOPs of one lock-step iteration of config 2: (executed, survey_model).
    executed = what the iteration's launches compute (2 M K N per product; equals the sum of `roofline_stages` -- checked in
    tests/test_full_size_gpu.py): per update pair 0.549 + 0.548 GFLOP, acting 2 x (policy 67 072 + twin Q_risk 133 632 +
    recovery policy 66 560) MAC per env.  survey_model = SURVEY.md section 8(d)'s 0.685 + 0.62 GFLOP per update pair, which
    prices the reference's literal call list -- including `safety_critic(s, pi)` of recovery_rl/sac.py:216-231, evaluated on
    every update and used only under the DGD / RCPO / LBAC flags.  This stack does not run that call (nor its backward) when
    no flag reads it, so the figure the iteration is priced with is `executed`; the other one is kept for comparison."""
    B, H = batch, hidden

    def fwd(G, M, din, dout):
        return 2.0 * G * M * (din * H + H * H + H * dout)

    def bwd(G, weights, dout, din):        # dh1 (+ dW2) through the hidden layer, dW3 / dh2 of the head, first-layer partials
        return 2.0 * G * B * H * H * (2 if weights else 1) + 4.0 * G * B * H * dout + 4.0 * G * B * H * din
    sac = (fwd(1, 2 * B, 2, 4) + 3 * fwd(2, B, 4, 1)                       # pi(s'), pi(s); Q_t(s', a'), Q(s, a), Q(s, pi)
           + bwd(2, True, 1, 4) + bwd(2, False, 1, 4) + bwd(1, True, 4, 2))  # critic loss; policy loss through Q; the policy
    qrisk = (fwd(1, B, 2, 4) + fwd(1, B, 2, 2) + 2 * fwd(2, B, 4, 1)       # pi(s'), rec(s); Qr_t(s', a'), Qr(s, a)
             + bwd(2, True, 1, 4) + fwd(2, B, 4, 1) + bwd(2, False, 1, 4) + bwd(1, True, 2, 2))
    acting = fwd(1, num_envs, 2, 4) + fwd(2, num_envs, 4, 1) + fwd(1, num_envs, 2, 2)
    executed = updates_per_step * (sac + qrisk) + acting
    survey = updates_per_step * (0.685e9 + 0.62e9) + num_envs * 2.0 * (67072 + 133632 + 66560)
    return executed, survey


def roofline_stages(a, device, iteration_ms):
    """Where the timed iteration's time goes, measured in this process: the iteration's launches are recorded once (the tape
    the seed-packing code uses), then every launch is timed on its own -- `reps` back-to-back launches of that one stage in a
    captured graph, HIP events on the launch stream (as time_step_push_kernel) -- and priced against the roof that bounds it:
    MLP stages by their algorithmic FLOPs against the f32 MFMA peak, the optimiser / env / replay stages by their algorithmic
    bytes against HBM.  `iteration_us` is the timed graph's iteration (incl. the driver's log points) for comparison."""
    import ctypes as C
    import torch
    import arg_utils
    from recovery_rl_amd import _lib, fast_update
    cfg = arg_utils.get_args(config_argv(a.env, 1, a.num_envs, 1))
    loop = build_loop(cfg, device)
    for _ in range(3):
        loop.vector_step(True, False, True)
    tape = []
    fast_update.set_tape(tape)
    try:
        loop.vector_step(True, False, True)
    finally:
        fast_update.set_tape(None)
    torch.cuda.synchronize(device)
    lib, st = _lib.load(), _lib.current_stream

    def mlp_flops(G, M, H, din, dout):
        return 2.0 * G * M * (din * H + H * H + H * dout)

    rows = []
    for pos, op in enumerate(tape):
        kind = op[0]
        if kind == "pair_bwd":
            # what the solo iteration launches for the head + hidden backward of its stacks (fast_update.backward_multi):
            # rrl_mlp_backward_pair_multi -- ONE launch whose tiles derive dh2 (critic-loss kinds and, since round 5, the
            # policy-head kinds)
            heads, hid, n = op[1], op[2], op[3]
            fl = 0.0
            for k in range(n):
                h = hid[k]
                fl += 2.0 * h.G * h.B * h.H * h.H * (2 if h.dW2 else 1) + 4.0 * h.G * h.B * h.H * heads[k].dout
                if h.first.x:
                    fl += 4.0 * h.G * h.B * h.H * h.first.din
            policy = any(heads[k].loss.kind > 3 for k in range(n))
            dout = max(heads[k].dout for k in range(n))
            rows.append(("head + hidden backward x%d (one launch, %s)" % (n, "policy head" if policy else "critic loss"), "backward",
                         lambda heads=heads, hid=hid, n=n: lib.rrl_mlp_backward_pair_multi(n, heads, hid, st()), fl, None,
                         "backward_pair_kernel<%d>" % dout))
            continue
        if kind == "forward":
            arr, n = op[1], op[2]
            fl = sum(mlp_flops(arr[k].G, arr[k].M, arr[k].H, arr[k].din, arr[k].dout) for k in range(n))
            m_max = max(arr[k].M for k in range(n))
            name = "forward x%d (%s rows)" % (n, "/".join(str(arr[k].M) for k in range(n)))
            launch = lambda arr=arr, n=n: lib.rrl_mlp3_forward_multi(n, arr, st())
            m_min = min(arr[k].M for k in range(n))
            # (members of different sizes -- an acting forward riding with 256-row update forwards -- go on the flat grid)
            kernel = "mlp3_fwd_split_flat_group_kernel" if m_min <= 1024 < m_max else \
                "mlp3_fwd_split_group_kernel<%d>" % (2 if m_max > 1024 else 1)
            rows.append((name, "acting forward" if m_max > 1024 else "update forward", launch, fl, None, kernel))
        elif kind == "head_bwd":
            arr, n = op[1], op[2]
            # thin (dout <= 4): a streaming kernel -- h2 read, dh2 written, the stack outputs and W3 read
            by = sum(4.0 * arr[k].G * arr[k].B * (2 * arr[k].H + 4 * arr[k].dout) + 8.0 * arr[k].G * arr[k].H * arr[k].dout
                     for k in range(n))
            rows.append(("head backward x%d" % n, "head backward", lambda arr=arr, n=n: lib.rrl_mlp_head_backward_multi(n, arr, st()), None, by,
                         "head_bwd_group_kernel"))
        elif kind == "hidden_bwd":
            arr, n = op[1], op[2]
            fl = 0.0
            for k in range(n):
                h = arr[k]
                fl += 2.0 * h.G * h.B * h.H * h.H * (2 if h.dW2 else 1)           # dh1 (NN) + dW2 (TN)
                if h.first.x:
                    fl += 4.0 * h.G * h.B * h.H * h.first.din                     # first-layer backward out of the tiles
            rows.append(("hidden backward x%d" % n, "hidden backward", lambda arr=arr, n=n: lib.rrl_mlp_hidden_backward_multi(n, arr, st()), fl, None,
                         "gemm16_group_kernel"))
        elif kind == "adam":
            segs, n, lr, b1, b2, eps = op[1:7]
            by = 0.0
            for k in range(n):
                sgm = segs[k]
                by += 4.0 * sgm.n * (7 + (2 if sgm.target else 0)) + 4.0 * sgm.n_part * sgm.part_elems
            rows.append(("Adam x%d" % n, "optimiser", lambda segs=segs, n=n, lr=lr, b1=b1, b2=b2, eps=eps:
                         lib.rrl_adam_step_multi(n, segs, lr, b1, b2, eps, st()), None, by, "adam_multi_kernel"))
        elif kind == "sample":
            g = op[1]
            by = 2 * cfg.batch_size * (32 + 32 + 3 * 16) + 8.0 * g.noise_pairs
            launch = lambda g=g: lib.rrl_sample_multi(g.first, g.second, g.noise_pairs, g.noise_seed, g.noise_counter,
                                                      g.noise_counter_dev, g.noise_counter_inc, g.noise_out, st())
            rows.append(("replay draws x2 + policy noise", "replay draw", launch, None, by, "sample_group_kernel"))
        elif kind == "step":
            env_name, env_kind, sa = op[1], op[2], op[3]
            launch = (lambda sa=sa: lib.rrl_maze_step_push_x(C.byref(sa), st())) if env_name == "maze" else \
                (lambda sa=sa, env_kind=env_kind: lib.rrl_nav_step_push_x(env_kind, C.byref(sa), st()))
            rows.append(("env step + 2 replay pushes + episode table", "env step", launch, None,
                         float(a.num_envs) * STEP_PUSH_ALGO_BYTES, "step_push_kernel"))
        else:
            rows.append((kind, kind, None, None, None, kind))
    out, total = [], 0.0
    for name, group, launch, fl, by, kernel in rows:
        if launch is None:
            continue
        t = _graph_of(launch, 50, device)
        total += t
        r = {"stage": name, "group": group, "kernel": kernel, "us": t * 1e6,
             "launches": 2 if name.endswith("(two launches)") else 1}
        if fl is not None:
            r.update(bound="mfma", flops=fl, achieved_TFLOPs=fl / t / 1e12, frac=fl / t / 1e12 / F32_MFMA_PEAK_TF)
        else:
            r.update(bound="hbm", bytes=by, achieved_GBs=by / t / 1e9, frac=by / t / 1e9 / HBM_PEAK_GBS)
        out.append(r)
    groups = {}
    for r in out:
        gsum = groups.setdefault(r["group"], {"launches": 0, "us": 0.0, "flops": 0.0, "bytes": 0.0})
        gsum["launches"] += r["launches"]
        gsum["us"] += r["us"]
        gsum["flops"] += r.get("flops", 0.0)
        gsum["bytes"] += r.get("bytes", 0.0)
    summary = []
    for gname, gs in sorted(groups.items(), key=lambda kv: -kv[1]["us"]):
        row = {"group": gname, "launches": gs["launches"], "us": gs["us"], "share_of_stand_alone_sum": gs["us"] / (total * 1e6)}
        if gs["flops"]:
            row.update(bound="mfma", frac=gs["flops"] / (gs["us"] * 1e-6) / 1e12 / F32_MFMA_PEAK_TF)
        else:
            row.update(bound="hbm", frac=gs["bytes"] / (gs["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
        summary.append(row)
    # per KERNEL (the granularity of a rocprofv3 --stats table): the one with the most time is the line's `roofline.dominant`
    kernels = {}
    for r in out:
        ks = kernels.setdefault(r["kernel"], {"kernel": r["kernel"], "launches": 0, "us": 0.0, "flops": 0.0, "bytes": 0.0})
        ks["launches"] += r["launches"]
        ks["us"] += r["us"]
        ks["flops"] += r.get("flops", 0.0)
        ks["bytes"] += r.get("bytes", 0.0)
    by_kernel = []
    for ks in sorted(kernels.values(), key=lambda k: -k["us"]):
        row = {"kernel": ks["kernel"], "launches": ks["launches"], "us": ks["us"], "avg_us": ks["us"] / ks["launches"],
               "share_of_stand_alone_sum": ks["us"] / (total * 1e6)}
        if ks["flops"]:
            tf = ks["flops"] / (ks["us"] * 1e-6) / 1e12
            row.update(bound="mfma", flops=ks["flops"], achieved=tf, peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", frac=tf / F32_MFMA_PEAK_TF)
        else:
            gb = ks["bytes"] / (ks["us"] * 1e-6) / 1e9
            row.update(bound="hbm", bytes=ks["bytes"], achieved=gb, peak=HBM_PEAK_GBS, unit="GB/s", frac=gb / HBM_PEAK_GBS)
        by_kernel.append(row)
    return {"launches": sum(r["launches"] for r in out), "stand_alone_sum_us": total * 1e6, "iteration_us": iteration_ms * 1e3,
            "mlp_flops": sum(r.get("flops", 0.0) for r in out), "by_kernel": by_kernel,
            "method": "each recorded launch of one iteration re-issued 50x back to back in its own graph, HIP events on the "
                      "launch stream; FLOPs / bytes are algorithmic (2 M K N per product; parameter + state bytes for Adam)",
            "dominant": summary[0]["group"], "by_group": summary, "stages": out}


def time_planner_kernel(device, n_plans=256, reps=3, precision="f32"):
    """rrl_plan_cost / rrl_plan_cost_f16x3 (MPC._compile_cost of config 4: 400 candidates x 20 particles x 5 steps per
    planning env): seconds per launch from HIP events on the launch stream."""
    import torch
    import arg_utils
    from recovery_rl_amd import _lib
    from recovery_rl_amd.MPC import MPC
    from recovery_rl_amd.config import create_config
    from recovery_rl_amd.env import make_vec_env
    from recovery_rl_amd.sac import SAC
    env = make_vec_env("navigation2", 4, device=device, seed=1)
    mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1, plan_precision=precision)
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


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY.md section 8d "CPU side-by-side"): the reference-order loop on the host cores
# ------------------------------------------------------------------------------------------------------------------
def _cpu_loop(cfg, env_name, seconds, with_recovery, mode="full"):
    """The reference's per-iteration body (experiment.py:396-452) for ONE env on the CPU: C oracle env + oracle
    replay + the same torch modules.  mode: "full" (update + act + step + push), "update" (updates on a fixed
    replay, no env), returns (env_steps/s or None, grad_steps/s)."""
    import numpy as np
    import torch
    from oracle import c_oracle as co
    from recovery_rl_amd.sac import SAC
    from recovery_rl_amd.spaces import Box
    torch.manual_seed(1)
    agent = SAC(Box(-np.ones(2) * np.inf, np.ones(2) * np.inf), Box(-np.ones(2), np.ones(2)), cfg, "/tmp")
    mem, rmem = co.OracleReplay(100000), co.OracleReplay(100000)
    tt = lambda arrs: tuple(torch.from_numpy(x) for x in arrs)
    if with_recovery:
        rmem.push(*co.nav_offline(env_name, 20000, 1))
    pos, obs, t = co.nav_reset(env_name, 1, seed=1, counter=0)
    steps = updates = 0
    counter = 1
    t_start, timed_steps, timed_updates = None, 0, 0
    t0 = time.perf_counter()
    if mode == "update":                                   # a filled buffer, then only update_parameters calls
        rng = np.random.RandomState(0)
        k = 4096
        mem.push(rng.randn(k, 2).astype(np.float32) + [-50, 0], rng.uniform(-1, 1, (k, 2)).astype(np.float32),
                 -50 + rng.randn(k).astype(np.float32), rng.randn(k, 2).astype(np.float32) + [-50, 0],
                 np.ones(k, np.float32))
    while True:
        if len(mem) > cfg.batch_size:
            if t_start is None:                            # time the steady state only
                t_start, timed_steps, timed_updates = time.perf_counter(), 0, 0
            agent.update_parameters(None, cfg.batch_size, updates, safety_critic=agent.safety_critic,
                                    batch=tt(mem.sample(cfg.batch_size, 1, counter)))
            if with_recovery:
                agent.safety_critic.update_parameters(policy=agent.policy,
                                                      batch=tt(rmem.sample(cfg.batch_size, 2, counter)))
            updates += 1
            timed_updates += 1
        if mode == "full":
            st = torch.from_numpy(obs)
            action = agent.select_action(st) if steps >= cfg.start_steps else torch.rand(1, 2) * 2 - 1
            real = action
            if with_recovery:
                risk = agent.safety_critic.get_value(st, action)
                real = agent.safety_critic.select_action(st) if float(risk) > cfg.eps_safe else action
            o = co.nav_step(env_name, pos, real.numpy(), t, seed=1, counter=counter, auto_reset=True)
            mask = 1.0 - o["done"].astype(np.float32)
            mem.push(obs, action.numpy(), o["reward"], o["next_obs"], mask)
            if with_recovery:
                rmem.push(obs, real.numpy(), o["constraint"].astype(np.float32), o["next_obs"], mask)
            pos, t, obs = o["pos"], o["t"], o["obs"]
            steps += 1
            timed_steps += 1
        counter += 1
        now = time.perf_counter()
        if t_start is not None and now - t_start > seconds:
            break
        if now - t0 > 6 * seconds + 20:
            break
    dt = time.perf_counter() - (t_start or t0)
    return (timed_steps / dt if mode == "full" else None), timed_updates / dt


def cpu_baseline(budget_s=24.0):
    """Config 1 (Navigation1, 1 env, SAC only: scripts/navigation1.sh:21 without --cuda) and the RRL-MF loop of config 2
    at one env, on 1 and min(8, cores) host threads, plus the update-only and env-only rates that attribute the
    time.  `value` = config-1 full-loop env-steps/s at the better thread count."""
    import numpy as np
    import torch
    import arg_utils
    from oracle import c_oracle as co
    cores = os.cpu_count() or 1
    many = min(8, cores)                 # batch-256 MLP updates do not scale past a few threads
    cfg1 = arg_utils.get_args(["--env-name", "navigation1", "--num_unsafe_transitions", "20000", "--seed", "1"])
    cfg2 = arg_utils.get_args([a for a in config_argv("navigation1", 1, 1) if a != "--cuda"])
    slot = budget_s / 6.0
    out = {"threads": {}}
    for nt in sorted({1, many}):
        torch.set_num_threads(nt)
        env_rate, grad_rate = _cpu_loop(cfg1, "navigation1", slot, with_recovery=False)
        _, upd_only = _cpu_loop(cfg1, "navigation1", slot / 2, with_recovery=False, mode="update")
        out["threads"][str(nt)] = {"config1_env_steps_per_s": env_rate, "config1_sac_grad_steps_per_s": grad_rate,
                                   "config1_update_only_grad_steps_per_s": upd_only}
    best = max(out["threads"], key=lambda k: out["threads"][k]["config1_env_steps_per_s"])
    torch.set_num_threads(int(best))
    mf_env, mf_grad = _cpu_loop(cfg2, "navigation1", slot, with_recovery=True)
    # the env alone, batched on ONE core of the same host
    n_env = 4096
    p4, _, t4 = co.nav_reset("navigation1", n_env, seed=1, counter=0)
    a4 = np.random.RandomState(0).uniform(-1, 1, (n_env, 2)).astype(np.float32)
    te, reps = time.perf_counter(), 0
    while time.perf_counter() - te < 1.0:
        o = co.nav_step("navigation1", p4, a4, t4, seed=1, counter=reps + 1, auto_reset=True)
        p4, t4 = o["pos"], o["t"]
        reps += 1
    env_only = reps * n_env / (time.perf_counter() - te)
    b = out["threads"][best]
    out.update({
        "value": b["config1_env_steps_per_s"], "unit": "env-steps/s", "cores": int(best), "kind": "port",
        "host_cores": cores, "grad_steps_per_s": b["config1_sac_grad_steps_per_s"],
        "update_only_grad_steps_per_s": b["config1_update_only_grad_steps_per_s"],
        "env_only_env_steps_per_s_1core": env_only,
        "rrl_mf_env_steps_per_s": mf_env, "rrl_mf_grad_steps_per_s": mf_grad,
        "sample": "config 1 = Navigation1, 1 env, SAC only, 1 update per env-step (experiment.py:396-452, B=256, "
                  "H=256): ~%.0f s per thread count (1 and %d threads) of the full loop + ~%.0f s update-only each; "
                  "the RRL-MF loop (SAC + Q_risk + recovery policy update per env-step) ~%.0f s at %s thread(s); "
                  "env-only 1 s on one core.  C oracle env + oracle replay + the same torch modules on the CPU"
                  % (slot, many, slot / 2, slot, best)})
    return out


# ------------------------------------------------------------------------------------------------------------------
def run_config(a, cfg, device, world, rank, updates_per_step=1, min_seconds=MIN_TIMED_S):
    """Build the loop for `cfg`, capture, warm up, time.  Returns (result dict, loop)."""
    import torch
    from recovery_rl_amd import distributed as dist_utils
    loop = build_loop(cfg, device, fast=not a.autograd_updates)
    step = loop.replay if not a.no_graph else (lambda: loop.vector_step(True, False, True))
    step = production_step(step, [loop], every=a.log_every, advance=None if a.no_graph else loop.advance)
    if not a.no_graph:
        loop.capture(online_qrisk=True)
    # untimed: bring a cold box to its working clocks (the first process on a fresh box measured 0.197 ms per iteration for
    # three seconds, the same command a minute later 0.187), then the W warm-up steps the caller asked for
    t_pre = time.perf_counter()
    while min_seconds > 0 and time.perf_counter() - t_pre < PREWARM_S:
        for _ in range(50):
            step()
        torch.cuda.synchronize(device)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(device)
    stats0, dev0 = loop.read_stats(), device_update_counts(loop)
    elapsed, blocks = timed_blocks(step, a.steps, world, device, min_seconds)
    stats1, dev1 = loop.read_stats(), device_update_counts(loop)
    n_steps = a.steps * blocks
    local = {k: stats1[k] - stats0[k] for k in stats1}
    assert local["env_steps"] == n_steps * cfg.num_envs, (local["env_steps"], n_steps, cfg.num_envs)
    if dev0 is not None:      # the device-side optimiser step counters, not the host mirrors
        assert dev1[0] - dev0[0] == n_steps * updates_per_step, (dev0, dev1, n_steps)
        assert dev1[1] - dev0[1] == n_steps * updates_per_step, (dev0, dev1, n_steps)
        local["sac_updates"], local["qrisk_updates"] = dev1[0] - dev0[0], dev1[1] - dev0[1]
    agg = dist_utils.aggregate_stats(local, world, device)
    return {"elapsed": elapsed, "blocks": blocks, "steps_total": n_steps, "agg": agg,
            "device_counters": dev0 is not None}, loop


def run_config_packed(a, device, world, rank, S, updates_per_step=1, min_seconds=MIN_TIMED_S):
    """The headline leg with S seeds per GPU (`--seeds_per_gpu S`): rank g packs seeds 1 + g S .. 1 + g S + S - 1 (the
    reference's seed loop, scripts/navigation1.sh:4-8, folded onto the GPUs: recovery_rl_amd/packed.py) and every launch of
    the lock-step iteration serves all S of them.  Same shape of result as run_config; the witness of the grad-steps is every
    seed's device-side Adam step counter."""
    import torch
    import arg_utils
    from recovery_rl_amd import distributed as dist_utils
    from recovery_rl_amd.packed import PackedLoop
    U = updates_per_step
    first = 1 + rank * S
    # set-up on every rank, then ONE agreement over the ranks before anybody enters the barriers of the timed region: a rank
    # that failed here (e.g. out of memory) must not leave the others waiting in a collective
    failed, loops, packed, step = None, None, None, None
    try:
        loops = [build_loop(arg_utils.get_args(config_argv(a.env, first + k, a.num_envs, U)), device) for k in range(S)]
        packed = PackedLoop(loops)
        packed.capture()
        step = production_step(packed.replay, loops, advance=packed.advance)
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize(device)
    except Exception as e:      # noqa: BLE001
        failed = "%s: %s" % (type(e).__name__, e)
    if dist_utils.max_over_ranks(1.0 if failed else 0.0, world, device) > 0:
        raise RuntimeError("packed set-up failed on %s" % ("this rank: " + failed if failed else "another rank"))

    def totals():
        st = [l.read_stats() for l in loops]
        return {k: sum(x[k] for x in st) for k in st[0]}
    stats0 = totals()
    c0 = [(int(l.agent.fast.critic.step[0].item()), int(l.agent.fast.qrisk.step[0].item())) for l in loops]
    elapsed, blocks = timed_blocks(step, a.steps, world, device, min_seconds)
    stats1 = totals()
    c1 = [(int(l.agent.fast.critic.step[0].item()), int(l.agent.fast.qrisk.step[0].item())) for l in loops]
    n_steps = a.steps * blocks
    local = {k: stats1[k] - stats0[k] for k in stats1}
    assert local["env_steps"] == n_steps * a.num_envs * S, (local["env_steps"], n_steps, S)
    assert all(y[0] - x[0] == n_steps * U and y[1] - x[1] == n_steps * U for x, y in zip(c0, c1)), (c0, c1, n_steps)
    local["sac_updates"] = sum(y[0] - x[0] for x, y in zip(c0, c1))
    local["qrisk_updates"] = sum(y[1] - x[1] for x, y in zip(c0, c1))
    agg = dist_utils.aggregate_stats(local, world, device)
    witness = {"rank": rank, "seeds": list(range(first, first + S)),
               "adam_steps_per_seed": [y[0] - x[0] for x, y in zip(c0, c1)],
               "launches_per_packed_iteration": packed.launches}
    return {"elapsed": elapsed, "blocks": blocks, "steps_total": n_steps, "agg": agg, "device_counters": True,
            "seed_pack": witness}, loops[0]


def run_seed_pack_leg(a, device, seeds=(1, 2, 4, 8), min_seconds=MIN_TIMED_LEG_S, updates_per_step=1):
    """S independent learners of configs[1] (seeds 1..S: own envs, replay rings, networks, Philox keys) sharing every launch
    of the lock-step iteration on ONE GPU (recovery_rl_amd/packed.py): aggregate env-steps/s and grad-steps/s.  The reference
    runs its ten seeds one after the other (scripts/navigation1.sh:4-8); every packed seed equals its solo run bit for bit
    (tests/test_packed_gpu.py).  Witness of the grad-steps: each seed's device-side Adam step counter."""
    import torch
    import arg_utils
    from recovery_rl_amd.packed import PackedLoop
    out = []
    for S in seeds:
        U = updates_per_step
        loops = [build_loop(arg_utils.get_args(config_argv(a.env, 1 + k, a.num_envs, U)), device) for k in range(S)]
        packed = PackedLoop(loops)
        packed.capture()
        for _ in range(a.warmup):
            packed.replay()
        torch.cuda.synchronize(device)
        c0 = [int(l.agent.fast.critic.step[0].item()) for l in loops]
        elapsed, blocks = timed_blocks(production_step(packed.replay, loops, advance=packed.advance), a.steps, 1, device,
                                       min_seconds)
        c1 = [int(l.agent.fast.critic.step[0].item()) for l in loops]
        n_steps = a.steps * blocks
        assert all(y - x == n_steps * U for x, y in zip(c0, c1)), (c0, c1, n_steps)
        out.append({"seeds_per_gpu": S, "updates_per_step": U, "launches_per_packed_iteration": packed.launches,
                    "ms_per_packed_iteration": elapsed / n_steps * 1e3,
                    "aggregate_env_steps_per_s": S * a.num_envs * n_steps / elapsed,
                    "aggregate_sac_grad_steps_per_s": S * U * n_steps / elapsed,
                    "aggregate_qrisk_grad_steps_per_s": S * U * n_steps / elapsed, "timed_seconds": elapsed})
        del packed, loops
        torch.cuda.empty_cache()
    base = out[0]["aggregate_env_steps_per_s"]
    for r in out:
        r["speedup_vs_one_seed"] = r["aggregate_env_steps_per_s"] / base
    return out


CONFIG4_ARGV = ["--env-name", "navigation2", "--cuda", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                "--num_unsafe_transitions", "20000"]                       # configs[3]: scripts/navigation2.sh:14


def run_config4_leg(device, precision, num_envs=NUM_ENVS, iters=30, graph=True):
    """BASELINE configs[3] (Navigation2, 4096 envs, model-based recovery: PETS/CEM through rrl_plan_cost) with the gate
    the reference runs with -- Q_risk pre-trained for the default 10 000 steps on 20 000 offline transitions, ensemble
    pre-trained for 50 epochs (experiment.py:261-305) -- so that only the envs whose Q_risk exceeds eps_safe plan.
    `iters` lock-step iterations (SAC + Q_risk update, act, plan for the recovery set, step, push) are timed; the size of
    the recovery set of every iteration is recorded ON THE DEVICE (MPC.act does not synchronise)."""
    import torch
    import arg_utils
    from recovery_rl_amd.experiment import Experiment
    cfg = arg_utils.get_args(CONFIG4_ARGV + ["--num_envs", str(num_envs), "--seed", "1", "--logdir", "/tmp/rrl_bench_c4",
                                             "--plan_precision", precision])
    exp = Experiment(cfg)
    t0 = time.perf_counter()
    exp.pretrain_critic_recovery()
    torch.cuda.synchronize(device)
    pre_s = time.perf_counter() - t0
    loop, mpc = exp.loop, exp.recovery_policy
    loop.start()
    while not (len(exp.memory) > cfg.batch_size and loop.total_numsteps >= cfg.start_steps):
        loop.vector_step(do_update=False, random_actions=True)
    for _ in range(2):
        loop.vector_step(do_update=True, online_qrisk=True)
    # the steady-state iteration in ONE hipGraph, as Experiment.run_vectorized replays it (MPC.act reads nothing on the host)
    graph = bool(mpc.device_count and mpc.fused is not None and getattr(loop.agent, "fast", None) is not None) and graph
    if graph:
        loop.capture(online_qrisk=True, warmup=0)     # (two eager steady iterations above: the capture executes nothing)
    step = loop.replay if graph else (lambda: loop.vector_step(do_update=True, online_qrisk=True))
    sizes = torch.zeros(iters, dtype=torch.int32, device=device)
    stats0 = loop.read_stats()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(iters):
        step()
        if mpc.last_count is not None:
            sizes[k:k + 1].copy_(mpc.last_count)
    ev1.record()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    stats1 = loop.read_stats()
    sizes = sizes.cpu().tolist()
    assert stats1["env_steps"] - stats0["env_steps"] == iters * num_envs
    assert stats1["recovery_steps"] - stats0["recovery_steps"] == sum(sizes), (stats1["recovery_steps"], sum(sizes))
    row_steps = sum(sizes) * mpc.optimizer.popsize * mpc.npart * mpc.plan_hor * mpc.optimizer.max_iters
    plans = sum(sizes) * mpc.optimizer.max_iters          # evaluations of _compile_cost: one per planning env and CEM iteration
    shape = dict(pop=mpc.optimizer.popsize, npart=mpc.npart, nets=mpc.model.num_nets, plan_hor=mpc.plan_hor)
    needed, literal = plan_flops(plans, **shape)
    t_ev = ev0.elapsed_time(ev1) * 1e-3
    peak = F32_MFMA_PEAK_TF if precision == "f32" else F16_MFMA_PEAK_TF
    executed = needed if precision == "f32" else plan_f16_products(plans, **shape)
    return {"workload": "Navigation2, %d envs, model-based recovery (scripts/navigation2.sh:14 + --num_envs %d), "
                        "pre-trained gate, planner kernel %s" % (num_envs, num_envs, precision),
            "iterations": iters, "timed_seconds": dt, "ms_per_step": dt / iters * 1e3,
            "env_steps_per_s": iters * num_envs / dt, "grad_steps_per_s": iters / dt,
            "recovery_set_sizes": sizes, "planned_actions": sum(sizes),
            "planner_row_steps_per_s": row_steps / dt,
            "planner_TFLOPs_over_whole_iteration": needed / dt / 1e12,
            "pretrain_seconds": pre_s, "graph": graph,
            # the planner's NEEDED FLOPs (plan_flops) over the WHOLE iteration's time (HIP events around the timed iterations on
            # the launch stream): a lower bound of the planner kernels' own rate (the updates and the env step are in the time)
            "roofline": {"bound": "mfma", "unit": "TFLOP/s",
                         "kernel": "plan_first_step_kernel + plan_cost_kernel inside the lock-step iteration",
                         "achieved": needed / t_ev / 1e12, "peak": peak, "frac": executed / t_ev / 1e12 / peak,
                         "literal_rollout_TFLOPs": literal / t_ev / 1e12,
                         "note": "f32: needed FLOPs (12.90 GFLOP per planning env and CEM iteration: first step once per "
                                 "distinct row, no prediction at the last step) / f32 MFMA peak; f16x3: executed f16 products (3 "
                                 "per hidden-layer product) / f16 MFMA peak; time = the whole iteration.  literal_rollout_TFLOPs "
                                 "prices the same time with the 17.22 GFLOP of the reference's literal loop (the figure of "
                                 "rounds 1-5), for comparison only"},
            "host_syncs_per_iteration": 0 if mpc.device_count else 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--num_envs", type=int, default=NUM_ENVS)
    ap.add_argument("--env", choices=sorted(CONFIG_ARGV), default="navigation1",
                    help="navigation1 = configs[1] (the headline); maze = configs[2] and the Maze leg of configs[4]")
    ap.add_argument("--updates_per_step", type=int, default=1, help="SAC (+ Q_risk) updates per lock-step iteration")
    ap.add_argument("--seeds_per_gpu", type=int, default=1,
                    help="S > 1: every rank packs S consecutive seeds into its launches (rank g: seeds 1 + g S ...); the "
                         "line's value is then the aggregate over all N x S seeds")
    ap.add_argument("--utd_sweep", action="store_true",
                    help="also time U = 4, 16, 64 updates per iteration (update-to-data ratio U / num_envs)")
    ap.add_argument("--no_graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--log_every", type=int, default=LOG_EVERY,
                    help="iterations between the driver's log points inside the timed region (counters read, episode table "
                         "drained); 0 = never (what the GPU-side iteration alone costs)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--planner", action="store_true",
                    help="time the fused planner kernel of config 4 (MFMA roofline) -> roofline_planner; "
                         "on by default for single-GPU runs")
    ap.add_argument("--no_planner", action="store_true")
    ap.add_argument("--sweep", action="store_true",
                    help="also time rrl_nav_step / step_push at N = 2^12..2^24 (the bandwidth regime of the env kernels)")
    ap.add_argument("--min_seconds", type=float, default=MIN_TIMED_S,
                    help="timed blocks of --steps steps are repeated until this much timed work exists")
    ap.add_argument("--no_legs", action="store_true",
                    help="skip the secondary legs of a single-GPU run: U = 16 updates per iteration (UTD 1/256, the "
                         "learning regime) and config 4 (Navigation2, model-based recovery, trained gate)")
    ap.add_argument("--autograd_updates", action="store_true",
                    help="PyTorch autograd + vendor GEMMs for the updates instead of the fused HIP kernels")
    a = ap.parse_args()

    spawn, world_expected = resolve_world(a.gpus, os.environ)
    if spawn:
        raise SystemExit(spawn_ranks(sys.argv[1:], a.gpus))

    import torch
    import arg_utils
    from recovery_rl_amd import distributed as dist_utils

    rank, local_rank, world = dist_utils.init()
    assert world == world_expected
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = dist_utils.local_device(local_rank)
    torch.cuda.set_device(device)

    U = a.updates_per_step
    cfg = arg_utils.get_args(config_argv(a.env, dist_utils.rank_seed(1, rank), a.num_envs, U))
    S = max(1, a.seeds_per_gpu)
    if S > 1:
        res, loop = run_config_packed(a, device, world, rank, S, U, min_seconds=a.min_seconds)
    else:
        res, loop = run_config(a, cfg, device, world, rank, U, min_seconds=a.min_seconds)
    elapsed, agg, n_steps = res["elapsed"], res["agg"], res["steps_total"]

    extra = {}
    if S > 1:
        extra["seed_pack_headline"] = res["seed_pack"]
    elif world > 1 and not a.no_legs and not a.no_graph:
        # a multi-GPU node is best used with several seeds per GPU (the solo iteration is latency-bound): S = 4 on every rank
        del loop
        torch.cuda.empty_cache()
        try:
            r4, loop = run_config_packed(a, device, world, rank, 4, U, min_seconds=MIN_TIMED_LEG_S)
            extra["seed_pack_multi_gpu"] = {
                "seeds_per_gpu": 4, "seeds_total": 4 * world, "ms_per_packed_iteration": r4["elapsed"] / r4["steps_total"] * 1e3,
                "aggregate_env_steps_per_s": r4["agg"]["env_steps"] / r4["elapsed"],
                "aggregate_sac_grad_steps_per_s": r4["agg"]["sac_updates"] / r4["elapsed"],
                "aggregate_qrisk_grad_steps_per_s": r4["agg"]["qrisk_updates"] / r4["elapsed"],
                "timed_seconds": r4["elapsed"], "rank0_witness": r4["seed_pack"]}
        except Exception as e:      # noqa: BLE001  (every rank raises or none does: the legs are deterministic)
            extra["seed_pack_multi_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if a.utd_sweep:
        sweep = []
        for u in (1, 4, 16, 64):
            if u == U:
                r = res
            else:
                del loop
                torch.cuda.empty_cache()
                cfg_u = arg_utils.get_args(config_argv(a.env, dist_utils.rank_seed(1, rank), a.num_envs, u))
                r, loop = run_config(a, cfg_u, device, world, rank, u, min_seconds=0.3)
            sweep.append({"updates_per_step": u, "utd": "%d/%d" % (u, a.num_envs),
                          "ms_per_step": r["elapsed"] / r["steps_total"] * 1e3,
                          "env_steps_per_s": r["agg"]["env_steps"] / r["elapsed"],
                          "sac_grad_steps_per_s": r["agg"]["sac_updates"] / r["elapsed"],
                          "qrisk_grad_steps_per_s": r["agg"]["qrisk_updates"] / r["elapsed"]})
        extra["utd_sweep"] = sweep

    if world == 1 and not a.no_legs and not a.no_graph:
        import contextlib
        # the learning regime: 16 updates per lock-step iteration (UTD 1/256), same graph mechanism
        if U != 16:
            del loop
            torch.cuda.empty_cache()
            cfg_u = arg_utils.get_args(config_argv(a.env, dist_utils.rank_seed(1, rank), a.num_envs, 16))
            r16, loop = run_config(a, cfg_u, device, world, rank, 16, min_seconds=min(a.min_seconds, MIN_TIMED_LEG_S))
            extra["utd_1_256"] = {
                "updates_per_step": 16, "utd": "16/%d" % a.num_envs, "ms_per_step": r16["elapsed"] / r16["steps_total"] * 1e3,
                "env_steps_per_s": r16["agg"]["env_steps"] / r16["elapsed"],
                "sac_grad_steps_per_s": r16["agg"]["sac_updates"] / r16["elapsed"],
                "qrisk_grad_steps_per_s": r16["agg"]["qrisk_updates"] / r16["elapsed"],
                "timed_seconds": r16["elapsed"], "timed_steps_total": r16["steps_total"],
                "grad_step_witness": "device-side Adam step counters" if r16["device_counters"] else "host counters"}
        del loop
        torch.cuda.empty_cache()
        loop = None
        # secondary legs never take the headline line down with them: a failure is reported in the leg's place
        def leg(name, fn):
            try:
                extra[name] = fn()
            except Exception as e:      # noqa: BLE001
                extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
        leg("roofline_stages", lambda: roofline_stages(a, device, elapsed / n_steps * 1e3))
        leg("seed_pack", lambda: run_seed_pack_leg(a, device))
        leg("seed_pack_utd_1_256", lambda: run_seed_pack_leg(a, device, seeds=(1, 4, 8, 16), updates_per_step=16))
        if a.env == "navigation1" and a.num_envs == NUM_ENVS:
            def config4():
                with contextlib.redirect_stdout(sys.stderr):   # the driver announces itself: stdout carries the JSON line only
                    return {prec: run_config4_leg(device, prec) for prec in ("f32", "f16x3")}
            leg("config4", config4)

    if rank == 0:
        t_k = time_step_push_kernel(device, a.env, a.num_envs, log=True)
        gbs = a.num_envs * STEP_PUSH_ALGO_BYTES / t_k / 1e9
        traffic, traffic_src = committed_pmc("step_push_pmc", a.num_envs) if a.env == "navigation1" else (None, None)
        extra["roofline"] = {
            "kernel": "step_push_kernel<%s> (rrl_%s_step_push_x): env step + two replay pushes + episode counters, the "
                      "env kernel of the timed iteration" % ("MazeEnv" if a.env == "maze" else "NavEnv<0>",
                                                             "maze" if a.env == "maze" else "nav"),
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": (
                "committed rocprofv3 PMC passes (TCC fetch / write bytes, separate passes), %s entry '%d'; not "
                "re-measured by this run" % (traffic_src, a.num_envs)) if traffic is not None else None,
            "launch_us": t_k * 1e6, "algorithmic_bytes_per_env_step": STEP_PUSH_ALGO_BYTES,
            "layout": "compact env state (u16 status word, state from pos, no per-env output arrays) + the per-episode "
                      "table advanced by the same launch (40 B of accumulators per env-step, not part of the 103 algorithmic "
                      "bytes): what the timed graph and the lock-step driver launch",
            "note": "N=%d moves only %d KB per launch: latency-bound by construction; bandwidth regime "
                    "(N up to 2^24, `bench.py --sweep`): %s"
                    % (a.num_envs, a.num_envs * STEP_PUSH_ALGO_BYTES // 1024, latest_sweep())}
        if a.sweep:
            sweep, sweep_sp, sweep_c = [], [], []
            for logn in (12, 16, 20, 24):
                n = 1 << logn
                tk = time_nav_step_kernel(device, n, reps=200 if logn <= 16 else 20)
                sweep.append({"n_envs": n, "launch_us": tk * 1e6, "env_steps_per_s": n / tk,
                              "achieved_GBs": n * NAV_STEP_ALGO_BYTES / tk / 1e9,
                              "frac": n * NAV_STEP_ALGO_BYTES / tk / 1e9 / HBM_PEAK_GBS})
                tc = time_nav_step_compact_kernel(device, n, reps=200 if logn <= 16 else 20)
                sweep_c.append({"n_envs": n, "launch_us": tc * 1e6, "env_steps_per_s": n / tc,
                                "achieved_GBs": n * NAV_STEP_ALGO_BYTES / tc / 1e9,
                                "frac": n * NAV_STEP_ALGO_BYTES / tc / 1e9 / HBM_PEAK_GBS})
                if logn <= 22:
                    ts = time_step_push_kernel(device, "navigation1", n, reps=200 if logn <= 16 else 20)
                    ta = time_step_push_kernel(device, "navigation1", n, reps=200 if logn <= 16 else 20, compact=False)
                    sweep_sp.append({"n_envs": n, "launch_us": ts * 1e6, "env_steps_per_s": n / ts,
                                     "achieved_GBs": n * STEP_PUSH_ALGO_BYTES / ts / 1e9,
                                     "frac": n * STEP_PUSH_ALGO_BYTES / ts / 1e9 / HBM_PEAK_GBS,
                                     "array_layout_launch_us": ta * 1e6,
                                     "array_layout_frac": n * STEP_PUSH_ALGO_BYTES / ta / 1e9 / HBM_PEAK_GBS})
                torch.cuda.empty_cache()
            extra["roofline_sweep"] = sweep
            extra["roofline_sweep_step_push"] = sweep_sp
            extra["roofline_sweep_compact"] = {
                "kernel": "nav_step_compact_kernel<0,false> (rrl_nav_step_compact): the env step in the 56 B/env-step "
                          "layout", "moved_bytes_per_env_step": 56, "rows": sweep_c}
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
            plans = row_steps // (400 * 20 * 5)               # time_planner_kernel: n_plans evaluations at the config-4 shape
            needed, literal = plan_flops(plans)
            tf = needed / t_p / 1e12
            extra["roofline_planner"] = {
                "kernel": "plan_first_step_kernel + plan_cost_kernel (rrl_plan_cost, model-based recovery of config 4)",
                "bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF,
                "traffic": committed_pmc("planner_traffic", 256)[0],
                "traffic_source": committed_pmc("planner_traffic", 256)[1], "launch_ms": t_p * 1e3,
                "row_steps_per_s": row_steps / t_p, "needed_GFLOP_per_plan": needed / plans / 1e9,
                "literal_GFLOP_per_plan": literal / plans / 1e9, "literal_rollout_TFLOPs": literal / t_p / 1e12,
                "note": "f32-in/f32-acc MFMA (exact f32); priced with the NEEDED FLOPs of MPC._compile_cost (first step once "
                        "per candidate / per (candidate, member), no prediction at the last step: what the kernels execute "
                        "since round 6); literal_rollout_TFLOPs = the same time priced with every particle row through both "
                        "networks at every step (%d FLOP per particle-step, the figure of rounds 1-5), for comparison only"
                        % PLAN_FLOPS_PER_ROW_STEP}
            with contextlib.redirect_stdout(sys.stderr):
                t_h, row_steps = time_planner_kernel(device, precision="f16x3")
            tf_h = needed / t_h / 1e12
            # three f16 products per algorithmic product: the matrix pipe executes 3x the algorithmic flops of the hidden
            # layers, priced against the dense f16 MFMA peak
            f16 = plan_f16_products(plans)
            extra["roofline_planner_f16x3"] = {
                "kernel": "plan_first_step_kernel<f16x3> + plan_cost_kernel<f16x3> (rrl_plan_cost_f16x3, opt-in --plan_precision "
                          "f16x3): hidden layers as three v_mfma_f32_16x16x32_f16 products of hi/lo splits",
                "bound": "mfma", "achieved": tf_h, "unit": "TFLOP/s (needed, f32-equivalent)",
                "vs_f32_mfma_peak": tf_h / F32_MFMA_PEAK_TF, "launch_ms": t_h * 1e3, "row_steps_per_s": row_steps / t_h,
                "speedup_vs_f32_kernel": t_p / t_h,
                "executed_f16_TFLOPs": f16 / t_h / 1e12, "peak": F16_MFMA_PEAK_TF,
                "frac": f16 / t_h / 1e12 / F16_MFMA_PEAK_TF,
                "note": "costs agree with the f32 kernel to < 2e-5 (tests/test_plan_gpu.py); the K=32 f16 MFMA shape sustains 2460 TF on "
                        "this pool (profiles/mfma_f16_rate.hip)"}
        if not a.no_cpu_baseline and world == 1:
            extra["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        env_rate = agg["env_steps"] / elapsed
        flops_exec, flops_survey = (f * n_steps * world * S for f in iteration_flops(a.num_envs, U, cfg.batch_size, cfg.hidden_size))
        label = {"navigation1": "Navigation1", "maze": "Maze"}[a.env]
        out = {
            "metric": "env-steps/sec + SAC grad-steps/sec, Navigation1 4096 envs, 1/2/4/8 GPU",
            "value": env_rate, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / n_steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "prewarm_seconds": PREWARM_S, "timed_blocks": res["blocks"], "timed_steps_total": n_steps, "timed_seconds": elapsed,
            "sac_grad_steps_per_s": agg["sac_updates"] / elapsed,
            "qrisk_grad_steps_per_s": agg["qrisk_updates"] / elapsed,
            "grad_step_witness": "device-side Adam step counters" if res["device_counters"] else "host counters",
            "runtime": RUNTIME, "collective_backend": dist_utils.backend_name(world),
            "config": {"workload": "%s, %d vectorised envs/GPU, SAC + Q_risk + model-free recovery "
                                   "(scripts/%s + --num_envs %d), batch 256, hidden 256, "
                                   "updates_per_step %d (UTD %d/%d), one seed per GPU"
                                   % (label, a.num_envs, "navigation1.sh:7" if a.env == "navigation1" else "maze.sh:7",
                                      a.num_envs, U, U, a.num_envs),
                       "num_envs_per_gpu": a.num_envs, "batch_size": cfg.batch_size,
                       "hidden_size": cfg.hidden_size, "updates_per_step": cfg.updates_per_step,
                       "launch": "eager" if a.no_graph else "hipGraph replay",
                       "loop": "the iteration Experiment.run_vectorized replays: compact env state, per-episode table "
                               "advanced by the env-step launch, counters read and table drained every %d iterations "
                               "inside the timed region; replayed as the driver does (--graph_iterations %d iterations per "
                               "hipGraph up to each log point / block end, single-iteration graphs for the rest)"
                               % (LOG_EVERY, getattr(cfg, "graph_iterations", 1)),
                       "updates": "autograd + vendor GEMM" if a.autograd_updates else
                                  "hand-written HIP forward/backward (f32 MFMA) + fused Adam",
                       "seeds_per_gpu": S,
                       "parallelism": "replicas x%d%s (RCCL metric all-reduce only)"
                                      % (world, ", %d seeds packed per GPU" % S if S > 1 else "")},
            "episodes": agg["episodes"], "violations": agg["num_viols"], "successes": agg["num_successes"],
            # the MLP side of the iteration against the f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
            "roofline_mlp": {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TF,
                             "achieved": flops_exec / elapsed / 1e12,
                             "frac": flops_exec / elapsed / 1e12 / (F32_MFMA_PEAK_TF * world),
                             "flops_per_iteration": flops_exec / (n_steps * world * S),
                             "survey_model_frac": flops_survey / elapsed / 1e12 / (F32_MFMA_PEAK_TF * world),
                             "note": "EXECUTED MLP FLOPs of the iteration (SAC + Q_risk updates at B rows, acting at N envs: "
                                     "iteration_flops, = the sum of roofline_stages) / iteration time; tiny problems: launch- "
                                     "and latency-bound.  survey_model_frac prices SURVEY 8(d)'s 0.685 + 0.62 GFLOP per update "
                                     "pair, which includes the reference's always-evaluated, never-used safety_critic(s, pi) of "
                                     "recovery_rl/sac.py:216-231 -- a call this stack does not run"},
        }
        out.update(extra)
        stages = out.get("roofline_stages")
        if isinstance(stages, dict) and stages.get("by_kernel") and isinstance(out.get("roofline"), dict):
            # the section-8(d) entry above is the ENV kernel (6 % of the iteration); the kernel with the most time in the
            # iteration, with its own roof, rides along so that the line names the kernel that matters
            out["roofline"]["dominant"] = dict(stages["by_kernel"][0], source="roofline_stages.by_kernel (measured in this run: "
                                               "each recorded launch re-issued 50x in its own graph, HIP events)")
        print(json.dumps(out))
    dist_utils.shutdown()


if __name__ == "__main__":
    main()
