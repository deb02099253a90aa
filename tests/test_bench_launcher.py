"""bench.py's multi-GPU launcher: `--gpus N` without a torchrun environment re-executes the script as N ranks (one per
GPU, RCCL); under torchrun the world size must agree with --gpus.  The timed-block loop stops on the same block on every
rank (its stopping time is all-reduced): world_size-2 gloo run on the CPU.  The real 2-rank run of the whole bench on one
GPU (gloo dry run) is in the -m gpu part."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_flag_decides_the_world_size():
    assert bench.resolve_world(1, {}) == (False, 1)
    assert bench.resolve_world(8, {}) == (True, 8)                      # no torchrun env: bench spawns the ranks itself
    assert bench.resolve_world(4, {"WORLD_SIZE": "4"}) == (False, 4)    # the driver's torchrun launch
    with pytest.raises(SystemExit):
        bench.resolve_world(8, {"WORLD_SIZE": "1"})                     # a 1-rank run must not report itself as 8 GPUs
    with pytest.raises(SystemExit):
        bench.resolve_world(1, {"WORLD_SIZE": "2"})


def test_bench_and_product_launcher_ask_for_the_same_runtime_mode():
    """The timed configuration is the one `python -m rrl_main` runs: both launchers pass the same constant to
    runtime.configure (hipGraph replay mode), and the bench line reports what is in force."""
    from recovery_rl_amd import runtime
    want = "configure(graph_packet_capture=%s.LAUNCHER_GRAPH_PACKET_CAPTURE"
    assert want % "rrl_runtime" in open(os.path.join(ROOT, "bench.py")).read()
    main = open(os.path.join(ROOT, "rrl_main.py")).read()
    assert want % "runtime" in main and main.index("runtime.configure(") < main.index("from recovery_rl_amd.experiment")
    assert runtime.LAUNCHER_GRAPH_PACKET_CAPTURE in (0, 1)
    r = subprocess.run([sys.executable, "-c", "import rrl_main, os, json; from recovery_rl_amd import runtime; "
                        "print(json.dumps(runtime.settings()))"], cwd=ROOT, capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if "GRAPH_PACKET_CAPTURE" not in k})
    got = json.loads(r.stdout.strip().splitlines()[-1])
    measured = runtime.rocm_version() in runtime.MEASURED_ROCM       # the choice is made on the release it was measured on
    assert got == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": str(runtime.LAUNCHER_GRAPH_PACKET_CAPTURE) if measured else
                   "runtime default", "set_by_launcher": measured, "rocm": runtime.rocm_version()}, r.stderr
    assert bench.RUNTIME["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == got["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]
    # any other release: the launchers leave the runtime default alone; an explicit request is honoured everywhere
    probe = ("from recovery_rl_amd import runtime; runtime.MEASURED_ROCM = ('0.0',); import json; "
             "print(json.dumps(runtime.configure(graph_packet_capture=0, log=False)))")
    for extra, want_set in (({}, False), ({"RRL_GRAPH_PACKET_CAPTURE": "0"}, True)):
        env = {k: v for k, v in os.environ.items() if "GRAPH_PACKET_CAPTURE" not in k}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, capture_output=True, text=True, env=env)
        assert json.loads(r.stdout.strip().splitlines()[-1])["set_by_launcher"] is want_set, r.stderr


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20"], 8, 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "20"] and cmd[-5].endswith("bench.py")


def test_spawn_refuses_more_ranks_than_gpus(monkeypatch):
    monkeypatch.delenv("RRL_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(["--gpus", "4"], 4)
    assert "only 1 GPU" in str(e.value)


def test_config_argv_are_the_reference_command_lines():
    import arg_utils
    c2 = arg_utils.get_args(bench.config_argv("navigation1", 3, 4096))
    assert (c2.env_name, c2.use_recovery, c2.MF_recovery, c2.gamma_safe, c2.eps_safe) == ("navigation1", True, True, 0.8, 0.3)
    assert c2.num_unsafe_transitions == 20000 and c2.num_envs == 4096 and c2.seed == 3       # scripts/navigation1.sh:7
    c3 = arg_utils.get_args(bench.config_argv("maze", 1, 4096, 16))
    assert (c3.env_name, c3.gamma_safe, c3.eps_safe, c3.pos_fraction) == ("maze", 0.5, 0.15, 0.3)   # scripts/maze.sh:7
    assert c3.updates_per_step == 16


def _blocks_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from recovery_rl_amd import distributed as du
    du.init(backend="gloo")
    import time
    calls = [0]

    def step():                                   # rank 1 is 3x slower: the slow rank sets the time, both stop together
        calls[0] += 1
        time.sleep(0.001 * (1 + 2 * rank))
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        total, blocks = bench.timed_blocks(step, 10, world, torch.device("cpu"), min_seconds=0.2)
    finally:
        torch.cuda.synchronize = real_sync
    q.put((rank, total, blocks, calls[0]))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_timed_blocks_agree_over_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = bench.free_port()
    procs = [ctx.Process(target=_blocks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, t0, b0, c0), (_, t1, b1, c1) = res
    assert b0 == b1 and c0 == c1 == 10 * b0                  # exactly K steps per block, same number of blocks
    assert t0 == t1 and 0.2 <= t0 < 0.2 + 0.1                # max over ranks, >= the minimum duration
    assert t0 >= 0.003 * 10 * b0 * 0.9                       # the slow rank's time


def test_production_step_advances_whole_graphs_up_to_each_log_point():
    """bench.production_step: `run.many(n)` hands n iterations to the loop's advance() exactly as the lock-step driver does
    (experiment.py run_vectorized): never across a log point, the log-point work every `every` iterations, whatever the block
    size; one-by-one calls of `run()` reach the same log points."""
    calls, logs = [], []

    class Loop:
        episode_log = None

        def read_stats(self):
            logs.append(sum(calls))

    run = bench.production_step(lambda: calls.append(1), [Loop()], every=10, advance=lambda m: calls.append(m))
    run.many(7)            # 7
    run.many(7)            # 3 (log at 10) + 4
    run()                  # a single step in between
    run.many(26)           # 5 (log at 20) + 10 (log at 30) + 10 (log at 40) + 1
    assert calls == [7, 3, 4, 1, 5, 10, 10, 1] and sum(calls) == 41
    assert logs == [10, 20, 30, 40]
    # without advance(): n single steps, same log points
    calls.clear(); logs.clear()
    run = bench.production_step(lambda: calls.append(1), [Loop()], every=4)
    run.many(9)
    assert calls == [1] * 9 and logs == [4, 8]


@pytest.mark.gpu
def test_bench_gpus_2_runs_two_ranks_and_reports_them(tmp_path):
    """`python bench.py --gpus 2` end to end on the test box's single GPU (gloo for the 96-byte metric all-reduce, both
    ranks on cuda:0): the line says n_gpus 2, two seeds' worth of env-steps, device-side grad-step counters."""
    env = dict(os.environ, RRL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--num_envs", "512", "--no_cpu_baseline", "--no_planner", "--no_legs", "--min_seconds", "0.5"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["scaling"] == "weak"
    assert r["timed_steps_total"] == 20 * r["timed_blocks"] and r["timed_seconds"] >= 0.5
    assert abs(r["value"] - 2 * 512 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["value"]
    assert r["grad_step_witness"] == "device-side Adam step counters"
    assert abs(r["sac_grad_steps_per_s"] - 2 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["sac_grad_steps_per_s"]
    assert r["roofline"]["kernel"].startswith("step_push_kernel")


@pytest.mark.gpu
def test_bench_maze_leg_and_contradicting_world_size():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--env", "maze", "--steps", "20", "--warmup",
                          "5", "--num_envs", "1024", "--no_cpu_baseline", "--no_planner", "--no_legs", "--min_seconds", "0.5"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["config"]["workload"].startswith("Maze, 1024")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="1"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


RCCL_PROBE = """
import os, sys, json
sys.path.insert(0, %r)
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
import torch
import torch.distributed as dist
from recovery_rl_amd import distributed as du
rank, local_rank, world = du.init(backend="nccl", force=True)        # backend "nccl" IS RCCL on ROCm
dev = du.local_device(local_rank)
torch.cuda.set_device(dev)
assert dist.is_initialized() and dist.get_backend() == "nccl" and du.backend_name(world) == "nccl"
stats = {k: i + 1 for i, k in enumerate(du.METRIC_KEYS)}
stats["reward_sum"], stats["episode_return_sum"] = -12.5, 3.25
agg = du.aggregate_stats(stats, world, dev)                          # all_reduce(SUM) of the 12-element f64 vector
mx = du.max_over_ranks(0.125, world, dev)                            # all_reduce(MAX)
du.barrier(world)
big = torch.arange(1 << 20, dtype=torch.float32, device=dev)         # and one bandwidth-sized reduction through RCCL
dist.all_reduce(big)
torch.cuda.synchronize()
print(json.dumps({"agg": agg, "max": mx, "backend": dist.get_backend(), "big_ok": bool(big[12345].item() == 12345.0),
                  "nccl_version": list(torch.cuda.nccl.version())}))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rccl_backend_runs_the_metric_collectives_on_the_box():
    """the only collectives of the path (metric all-reduce at logging cadence, max-over-ranks of the bench
    time, barrier) through backend `nccl` = RCCL on cuda:0 -- a 1-rank process group, i.e. the code path the 8-GPU run
    takes, minus the peers."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", RCCL_PROBE % (ROOT, str(bench.free_port()))], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["big_ok"] and r["max"] == 0.125
    from recovery_rl_amd import distributed as du
    for i, k in enumerate(du.METRIC_KEYS[:10]):
        assert r["agg"][k] == i + 1
    assert r["agg"]["reward_sum"] == -12.5 and r["agg"]["episode_return_sum"] == 3.25


@pytest.mark.gpu
def test_bench_through_rccl_on_one_rank():
    """The whole bench with its barrier / max-over-ranks / metric all-reduce going through RCCL (RRL_DIST_FORCE_INIT=1:
    a 1-rank `nccl` process group): the line reports the backend."""
    env = dict(os.environ, RRL_DIST_FORCE_INIT="1", RRL_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(bench.free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--num_envs",
                          "512", "--no_cpu_baseline", "--no_planner", "--no_legs", "--min_seconds", "0.3"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["collective_backend"] == "nccl" and r["n_gpus"] == 1
    assert r["runtime"]["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"        # the graph replay mode in force is part of the line


@pytest.mark.gpu
def test_bench_gpus_8_dry_run_aggregates_eight_ranks():
    """`python bench.py --gpus 8` as the driver's SCALE run issues it, dry-run on the box's single GPU (gloo for the
    96-byte all-reduces, 8 ranks sharing cuda:0): one JSON line, n_gpus 8, the value is eight seeds' worth of env-steps
    over the max-over-ranks time, eight ranks' device-side grad-step counters."""
    env = dict(os.environ, RRL_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "3",
                          "--num_envs", "256", "--no_cpu_baseline", "--min_seconds", "0.2"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    r = json.loads(line[0])
    assert r["n_gpus"] == 8 and r["collective_backend"] == "gloo" and "utd_1_256" not in r and "config4" not in r
    assert abs(r["value"] - 8 * 256 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["value"]
    assert abs(r["sac_grad_steps_per_s"] - 8 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["sac_grad_steps_per_s"]
    assert r["config"]["parallelism"].startswith("replicas x8")


@pytest.mark.gpu
def test_bench_gpus_2_with_two_seeds_per_gpu():
    """`bench.py --gpus 2 --seeds_per_gpu 2` (gloo dry run, both ranks on the box's GPU): rank g packs seeds 1 + 2 g, 2 + 2 g;
    the line's value is the aggregate over the four seeds; every seed's device-side Adam counter advanced once per iteration."""
    env = dict(os.environ, RRL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--seeds_per_gpu", "2", "--steps", "20",
                          "--warmup", "5", "--num_envs", "512", "--no_cpu_baseline", "--no_planner", "--no_legs",
                          "--min_seconds", "0.5"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["config"]["seeds_per_gpu"] == 2
    assert abs(r["value"] - 4 * 512 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["value"]
    assert abs(r["sac_grad_steps_per_s"] - 4 * r["timed_steps_total"] / r["timed_seconds"]) < 1e-6 * r["sac_grad_steps_per_s"]
    w = r["seed_pack_headline"]
    assert w["rank"] == 0 and w["seeds"] == [1, 2] and w["adam_steps_per_seed"] == [r["timed_steps_total"]] * 2
    assert "destroy_process_group" not in out.stderr            # torch's exit warning about a live process group


@pytest.mark.gpu
def test_rrl_main_two_ranks_pack_consecutive_seeds(tmp_path):
    """`torchrun --nproc-per-node 2 rrl_main.py --seeds_per_gpu 2 --seed 4` (gloo dry run on one GPU): rank 0 runs seeds 4, 5 and
    rank 1 seeds 6, 7 -- four log directories, each seed's counters equal to the ones the same seed reaches in a one-process
    packed run (whose members equal their solo runs bit for bit: tests/test_packed_gpu.py)."""
    import pickle
    import arg_utils
    from recovery_rl_amd.experiment import run_packed
    argv = ["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe",
            "0.3", "--num_unsafe_transitions", "3000", "--critic_safe_pretraining_steps", "30", "--num_envs", "128",
            "--log_every", "20", "--num_eps", "100000", "--num_steps", str(128 * 60 - 1), "--seeds_per_gpu", "2"]
    env = dict(os.environ, RRL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(bench.free_port()), os.path.join(ROOT, "rrl_main.py")] + argv + \
          ["--seed", "4", "--logdir", str(tmp_path / "two_ranks")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "destroy_process_group" not in out.stderr
    dirs = sorted(os.listdir(tmp_path / "two_ranks"), key=lambda d: int(d.rsplit("_seed", 1)[1]))
    assert len(dirs) == 4 and [d.rsplit("_seed", 1)[1] for d in dirs] == ["4", "5", "6", "7"]
    multi = {d.rsplit("_seed", 1)[1]: pickle.load(open(tmp_path / "two_ranks" / d / "run_stats.pkl", "rb")) for d in dirs}
    for first in (4, 6):                          # what ONE process packing the same two seeds produces
        cfg = arg_utils.get_args(argv + ["--seed", str(first), "--logdir", str(tmp_path / ("one_%d" % first))])
        hists = run_packed(cfg)
        for k, h in enumerate(hists):
            got = multi[str(first + k)]["vector_stats"]
            assert got[-1] == h[-1] and len(got) == len(h), (first + k, got[-1], h[-1])
    assert multi["4"]["vector_stats"][-1] != multi["6"]["vector_stats"][-1]


@pytest.mark.gpu
def test_bench_gpus_2_default_legs_add_the_packed_leg():
    """What the driver's scaling run issues (`bench.py --gpus N --steps K --warmup W`, legs on): at N > 1 the line carries a second
    leg with S = 4 seeds packed per rank (`seed_pack_multi_gpu`) next to the one-seed-per-GPU headline.  Two gloo ranks on the
    box's GPU, small envs."""
    env = dict(os.environ, RRL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--num_envs", "256", "--no_cpu_baseline", "--no_planner", "--min_seconds", "0.3"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["config"]["seeds_per_gpu"] == 1
    leg = r["seed_pack_multi_gpu"]
    assert "error" not in leg, leg
    assert leg["seeds_per_gpu"] == 4 and leg["seeds_total"] == 8 and leg["rank0_witness"]["seeds"] == [1, 2, 3, 4]
    assert leg["aggregate_env_steps_per_s"] > 0 and leg["ms_per_packed_iteration"] > 0


def test_flop_models_price_what_the_kernels_execute():
    """iteration_flops: the executed MLP FLOPs of the config-2 iteration (3.287 GFLOP at 4096 envs, U = 1) beside SURVEY 8(d)'s
    model with the reference's unused safety_critic(s, pi) call; plan_flops: needed (12.90 GFLOP: first step once per distinct
    row, no prediction at the last step) beside the literal loop's 17.22 GFLOP per planning env and CEM iteration."""
    executed, survey = bench.iteration_flops(4096, 1)
    assert abs(executed - 3.287e9) < 1e6 and abs(survey - 3.4944e9) < 1e6
    e16, s16 = bench.iteration_flops(4096, 16)
    assert abs((e16 - executed) / 15 - (0.54919e9 + 0.54841e9)) < 1e5 and abs((s16 - survey) / 15 - 1.305e9) < 1e3
    needed, literal = bench.plan_flops(1)
    assert abs(needed - 12.9026e9) < 1e5 and literal == 400 * 20 * 5 * 430464
    assert abs(needed / literal - 0.749) < 1e-3
    one_step = bench.plan_flops(1, plan_hor=1)                     # a one-step plan: Q_risk on the 400 candidates, nothing else
    assert one_step[0] == 400 * 267264
