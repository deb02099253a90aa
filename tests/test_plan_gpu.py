"""rrl_plan_cost (fused MFMA candidate evaluation) against the PyTorch restatement of MPC._compile_cost that the
reference KAT G7c pins (tests/test_mpc_gpu.py): same weights, same action sequences, same particle noise.
f32 on both sides; tolerance = summation-order differences through 5 x (3 + 4) layers."""
import numpy as np
import pytest
import torch

import arg_utils
from oracle import c_oracle
from recovery_rl_amd.MPC import MPC
from recovery_rl_amd.config import create_config
from recovery_rl_amd.env import make_vec_env
from recovery_rl_amd.sac import SAC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 2e-4, 2e-4          # costs are sums of plan_hor sigmoids, O(1)


def build(seed=0, weight_scale=2.5, f16x3=False):
    torch.manual_seed(seed)
    env = make_vec_env("navigation2", 4, device=DEV, seed=1)
    cfg = create_config("navigation2", "MPC", {}, [], "/tmp", env=env)
    mpc = MPC(cfg.ctrl_cfg, seed=1)
    args = arg_utils.get_args(["--env-name", "navigation2", "--cuda", "--use_recovery", "--gamma_safe", "0.65",
                               "--eps_safe", "0.2"])
    agent = SAC(env.observation_space, env.action_space, args, "/tmp")
    with torch.no_grad():      # spread the outputs: default inits give sigmoid ~ 0.5 and tiny dynamics everywhere
        for p in agent.safety_critic.safety_critic.parameters():
            p.mul_(weight_scale)
        for name in ("lin0_b", "lin1_b", "lin2_b", "lin3_b"):
            getattr(mpc.model, name).normal_(0, 0.1)
        mpc.model.lin3_w.mul_(3.0)
    data = torch.randn(500, 4, device=DEV) * torch.tensor([1.5, 1.0, 0.6, 0.6], device=DEV) \
        + torch.tensor([-0.5, 0.3, 0.0, 0.0], device=DEV)
    mpc.model.fit_input_stats(data)
    mpc.has_been_trained = True
    mpc.update_value_func(agent.safety_critic)
    assert mpc.fused is not None
    if f16x3:                  # the hidden layers as three f16 MFMA products of hi / lo splits (rrl_plan_cost_f16x3)
        from recovery_rl_amd.planner import FusedPlanner
        mpc.fused = FusedPlanner(mpc, f16x3=True)
    assert mpc.fused.f16x3 == f16x3
    mpc.fused.pack()
    return env, mpc, agent


def inputs(mpc, M, pop, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    acs = torch.rand(M, pop, mpc.plan_hor * 2, device=DEV, generator=g) * 2 - 1
    obs = torch.randn(M, 2, device=DEV, generator=g) * torch.tensor([1.5, 1.0], device=DEV) \
        + torch.tensor([-0.5, 0.3], device=DEV)
    noise = torch.randn(mpc.plan_hor, M * pop * mpc.npart, 2, device=DEV, generator=g)
    return acs, obs, noise


@pytest.mark.parametrize("f16x3", [False, True])
@pytest.mark.parametrize("M,pop", [(1, 400), (3, 400), (2, 30), (5, 7)])
def test_fused_cost_equals_the_pytorch_path(M, pop, f16x3):
    env, mpc, _ = build(f16x3=f16x3)
    acs, obs, noise = inputs(mpc, M, pop, seed=M * 1000 + pop)
    want = mpc._compile_cost(acs, obs, noise=noise, fused=False)
    got = mpc._compile_cost(acs, obs, noise=noise, fused=True)
    assert got.shape == want.shape == (M, pop)
    assert float(want.std()) > 0.05          # the comparison is not vacuous
    torch.testing.assert_close(got, want, rtol=RTOL, atol=ATOL)


def test_f16x3_kernel_agrees_with_the_f32_kernel_far_inside_the_tolerance():
    """hi + lo carries 22 bits: the two fused kernels differ by rounding noise, not by f16 precision.  Also with weights
    and activations large enough that a single f16 product (11 bits) would be off by ~1e-3."""
    for scale in (2.5, 6.0):
        env, mpc32, _ = build(weight_scale=scale)
        _, mpc16, _ = build(weight_scale=scale, f16x3=True)
        acs, obs, noise = inputs(mpc32, 4, 400, seed=77)
        a = mpc32._compile_cost(acs, obs, noise=noise, fused=True)
        b = mpc16._compile_cost(acs, obs, noise=noise, fused=True)
        assert float(a.std()) > 0.05
        assert float((a - b).abs().max()) < 2e-5, float((a - b).abs().max())


@pytest.mark.parametrize("f16x3", [False, True])
def test_fused_cost_nan_particles_count_as_1e6(f16x3):
    env, mpc, _ = build(f16x3=f16x3)
    acs, obs, noise = inputs(mpc, 2, 32, seed=5)
    obs[1, 0] = float("nan")
    want = mpc._compile_cost(acs, obs, noise=noise, fused=False)
    got = mpc._compile_cost(acs, obs, noise=noise, fused=True)
    assert torch.all(got[1] == 1e6) and torch.all(want[1] == 1e6)
    torch.testing.assert_close(got[0], want[0], rtol=RTOL, atol=ATOL)


def test_in_kernel_noise_is_the_documented_philox_stream():
    """noise = NULL: particle noise is N(0,1) from Philox (stream RRL_STREAM_PLAN = 7, row, tick*16 + t); the
    same draws from the C checker fed to the PyTorch path give the same costs, and the tick advances."""
    env, mpc, _ = build()
    M, pop = 1, 16
    acs, obs, _ = inputs(mpc, M, pop, seed=9)
    rows = M * pop * mpc.npart
    for tick in range(2):
        z = np.stack([c_oracle.normals(mpc.fused.seed, rows, 7, tick * 16 + t) for t in range(mpc.plan_hor)])
        noise = torch.as_tensor(z.astype(np.float32), device=DEV)
        want = mpc._compile_cost(acs, obs, noise=noise, fused=False)
        got = mpc._compile_cost(acs, obs, fused=True)
        torch.testing.assert_close(got, want, rtol=RTOL, atol=ATOL)
    assert int(mpc.fused.tick[0].item()) == 2


def test_planner_acts_through_the_fused_kernel_and_repacks_after_updates():
    env, mpc, agent = build()
    obs = env.reset()
    a1 = mpc.act(obs, 0)
    assert a1.shape == (4, 2) and (a1.abs() <= 1).all()
    assert int(mpc.fused.tick[0].item()) == mpc.optimizer.max_iters
    before = mpc.fused.packed.clone()
    with torch.no_grad():
        agent.safety_critic.safety_critic.linear2.weight.add_(0.01)
    mpc.act(obs, 0)
    assert not torch.equal(before, mpc.fused.packed)


@pytest.mark.parametrize("f16x3", [False, True])
def test_fused_cost_at_config4_scale_4096_planning_envs(f16x3):
    """BASELINE config 4's worst case: all 4096 envs plan at once (M = 4096, pop = 400, 20 particles, 5 steps = 164 M
    particle-steps per CEM iteration, through the ragged-tail / chunking / finish path of the kernel).  The PyTorch path
    (pinned to the reference by KAT G7c) evaluates a strided subset of the envs with the same noise rows."""
    env, mpc, _ = build(seed=3, f16x3=f16x3)
    M, pop = 4096, 400
    g = torch.Generator(device=DEV).manual_seed(77)
    acs = torch.rand(M, pop, mpc.plan_hor * 2, device=DEV, generator=g) * 2 - 1
    obs = torch.randn(M, 2, device=DEV, generator=g) * torch.tensor([1.5, 1.0], device=DEV) + \
        torch.tensor([-0.5, 0.3], device=DEV)
    per_env = pop * mpc.npart
    noise = torch.randn(mpc.plan_hor, M * per_env, 2, device=DEV, generator=g)        # 1.3 GB
    got = mpc._compile_cost(acs, obs, noise=noise, fused=True)
    assert got.shape == (M, pop) and torch.isfinite(got).all()
    rows = torch.tensor([0, 1, 255, 256, 1023, 2048, 3000, 4094, 4095], device=DEV)   # first / last groups, interior
    sub_noise = torch.cat([noise[:, int(m) * per_env:(int(m) + 1) * per_env] for m in rows], dim=1)
    want = mpc._compile_cost(acs[rows], obs[rows], noise=sub_noise, fused=False)
    assert float(want.std()) > 0.05
    torch.testing.assert_close(got[rows], want, rtol=RTOL, atol=ATOL)
    # in-kernel Philox noise at the same size: finite, different draws per env, tick advanced once
    t0 = int(mpc.fused.tick[0].item())
    free = mpc._compile_cost(acs, obs, fused=True)
    assert torch.isfinite(free).all() and int(mpc.fused.tick[0].item()) == t0 + 1
    assert float((free - got).abs().mean()) > 1e-4


@pytest.mark.parametrize("f16x3", [False, True])
@pytest.mark.parametrize("n,frac", [(64, 0.3), (64, 0.0), (64, 1.0), (700, 0.05)])
def test_act_with_the_planning_set_counted_on_the_device_equals_the_host_count_path(n, frac, f16x3):
    """MPC.act(obs, t, mask): the device-count path (rrl_cem_begin -> rrl_cem_sample_n / rrl_plan_cost_n /
    rrl_cem_update_n -> rrl_cem_finish; no host synchronisation) against the path that compacts with mask.nonzero() on the
    host: same actions, same shifted solutions, same RNG ticks, bit for bit -- incl. an empty and a full planning set, over
    two consecutive calls (prev_sol carried between them)."""
    _, mpc, agent = build(seed=5, f16x3=f16x3)
    g = torch.Generator(device=DEV).manual_seed(n)
    results = []
    for device_count in (False, True):
        mpc.device_count = device_count
        mpc.prev_sol = torch.zeros(n, mpc.plan_hor * 2, dtype=torch.float64, device=DEV)
        mpc.optimizer.tick.zero_()
        mpc.fused.tick.zero_()
        g.manual_seed(n)
        outs = []
        for call in range(2):
            obs = torch.randn(n, 2, device=DEV, generator=g) * torch.tensor([1.5, 1.0], device=DEV)
            mask = torch.rand(n, device=DEV, generator=g) < frac
            outs.append(mpc.act(obs, 0, mask=mask).clone())
            if frac not in (0.0, 1.0):
                assert 0 < int(mask.sum()) < n
            if device_count:
                assert int(mpc.last_count.item()) == int(mask.sum())
                assert (outs[-1][~mask] == 0).all()
        results.append((outs, mpc.prev_sol.clone(), mpc.optimizer.tick.clone(), mpc.fused.tick.clone()))
    (o0, p0, t0, f0), (o1, p1, t1, f1) = results
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(p0, p1)
    # (the host path skips the CEM for an empty set; the device path launches it for zero problems, which leaves the ticks alone)
    assert torch.equal(t0, t1) and torch.equal(f0, f1)
    if frac > 0:
        assert float(o0[0].abs().max()) > 0


@pytest.mark.parametrize("f16x3", [False, True])
def test_device_counted_planning_stays_on_the_host_count_path_after_an_empty_set(f16x3):
    """A run of calls with an EMPTY recovery set in the middle (advisor, round 3): the host-count path returns before it
    draws, so its Philox ticks stand still; the device-count path used to advance them (same bits per call, different rows
    from the next call on).  Five calls, masks 30 % / empty / 30 % / empty / 100 %: actions, solutions and ticks stay equal."""
    n = 96
    _, mpc, agent = build(seed=7, f16x3=f16x3)
    g = torch.Generator(device=DEV)
    results = []
    for device_count in (False, True):
        mpc.device_count = device_count
        mpc.prev_sol = torch.zeros(n, mpc.plan_hor * 2, dtype=torch.float64, device=DEV)
        mpc.optimizer.tick.zero_()
        mpc.fused.tick.zero_()
        g.manual_seed(11)
        outs = []
        for frac in (0.3, 0.0, 0.3, 0.0, 1.0):
            obs = torch.randn(n, 2, device=DEV, generator=g) * torch.tensor([1.5, 1.0], device=DEV)
            mask = torch.rand(n, device=DEV, generator=g) < frac
            outs.append(mpc.act(obs, 0, mask=mask).clone())
        results.append((outs, mpc.prev_sol.clone(), mpc.optimizer.tick.clone(), mpc.fused.tick.clone()))
    (o0, p0, t0, f0), (o1, p1, t1, f1) = results
    for k, (a, b) in enumerate(zip(o0, o1)):
        assert torch.equal(a, b), k
    assert torch.equal(p0, p1) and torch.equal(t0, t1) and torch.equal(f0, f1)
    assert int(t0[0].item()) > 0


def test_a_captured_pack_sees_the_input_statistics_of_a_later_refit():
    """The model-based iteration is one hipGraph that re-packs the planner's weights on every replay (experiment.py
    `mb_graph`); the online re-fit (experiment.py:464-480 -> MPC.train -> fit_input_stats, navigation1.py:61-69) runs between
    replays.  The statistics must therefore be rewritten IN PLACE: a graph captured before the re-fit has to plan with the
    normaliser of the re-fitted ensemble, exactly as the eager path does."""
    env, mpc, _ = build()
    acs, obs, noise = inputs(mpc, 3, 400, seed=5)
    out = torch.empty(3, 400, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        mpc.fused.pack()
        out.copy_(mpc.fused.cost(acs, obs, noise=noise))
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mpc.fused.pack()
        out.copy_(mpc.fused.cost(acs, obs, noise=noise))
    before = out.clone()
    ptrs = (mpc.model.inputs_mu.data_ptr(), mpc.model.inputs_sigma.data_ptr())
    shifted = torch.randn(700, 4, device=DEV) * torch.tensor([0.7, 2.0, 0.3, 0.9], device=DEV) \
        + torch.tensor([1.5, -0.8, 0.2, -0.1], device=DEV)
    mpc.model.fit_input_stats(shifted)                          # what recovery_policy.train does between replays
    assert (mpc.model.inputs_mu.data_ptr(), mpc.model.inputs_sigma.data_ptr()) == ptrs
    assert tuple(mpc.model.inputs_mu.shape) == (1, 4)           # the reference's shape after a fit
    g.replay()
    torch.cuda.synchronize()
    mpc.fused.pack()
    want = mpc.fused.cost(acs, obs, noise=noise)
    assert float((want - before).abs().max()) > 1e-3            # the re-fit matters for these inputs
    assert torch.equal(out, want)
