"""GPU tests of the fused-kernel SAC / Q_risk updates (fast_update.FastUpdater) against the autograd
path on identical weights, batch and noise, and against the reference KATs (model_golden.npz)."""
import copy
import os

import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd import _lib
from recovery_rl_amd.sac import SAC
from recovery_rl_amd.spaces import Box

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ACT = Box(-np.ones(2), np.ones(2))
OBS = Box(-np.ones(2) * np.inf, np.ones(2) * np.inf)


def make_pair(hidden, extra=()):
    argv = ["--env-name", "navigation1", "--cuda", "--hidden_size", str(hidden), "--use_recovery",
            "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3"] + list(extra)
    args = arg_utils.get_args(argv)
    torch.manual_seed(0)
    a = SAC(OBS, ACT, args, "/tmp")
    for net in (a.critic, a.policy, a.safety_critic.safety_critic, a.safety_critic.policy):
        for n_, p in net.named_parameters():
            if n_.endswith("bias") and "bn" not in n_:
                p.data.uniform_(-0.2, 0.2)
    a.critic_target.load_state_dict(a.critic.state_dict())
    a.safety_critic.safety_critic_target.load_state_dict(a.safety_critic.safety_critic.state_dict())
    torch.manual_seed(0)
    b = SAC(OBS, ACT, args, "/tmp")
    for dst, src in ((b.critic, a.critic), (b.critic_target, a.critic_target), (b.policy, a.policy),
                     (b.safety_critic.safety_critic, a.safety_critic.safety_critic),
                     (b.safety_critic.safety_critic_target, a.safety_critic.safety_critic_target),
                     (b.safety_critic.policy, a.safety_critic.policy)):
        dst.load_state_dict(copy.deepcopy(src.state_dict()))
    return a, b, args


def batch(B, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    s = r(B, 2) * torch.tensor([20.0, 3.0], device=DEV) + torch.tensor([-30.0, 0.0], device=DEV)
    a = torch.rand(B, 2, device=DEV, generator=g) * 2 - 1
    rew = -r(B).abs() * 30
    s2 = s + a + 0.05 * r(B, 2)
    m = (torch.rand(B, device=DEV, generator=g) < 0.8).float()
    c = (torch.rand(B, device=DEV, generator=g) < 0.3).float()
    return (s, a, rew, s2, m), (s, a, c, s2, m), r(B, 2), r(B, 2)


def assert_nets_close(x, y, rtol=2e-4, atol=3e-5):
    """Parameters after Adam steps.  atol = 0.1 * lr: where a gradient entry is ~1e-8 (dead ReLU
    units) Adam's m / (sqrt(v) + eps) amplifies f32 summation-order noise to a fraction of lr."""
    for (k, v), (_, w) in zip(x.state_dict().items(), y.state_dict().items()):
        if "num_batches_tracked" in k:
            continue
        assert torch.allclose(v, w, rtol=rtol, atol=atol), (k, float((v - w).abs().max()))


def close_scaled(g, g_ref, rel=1e-4):
    """max |g - g_ref| <= rel * max |g_ref|: sums with cancellation are compared on the tensor's scale."""
    scale = float(g_ref.abs().max()) + 1e-12
    return float((g - g_ref).abs().max()) <= rel * scale + 1e-9


def assert_grads_close(slow_net, flat_net, names):
    """Gradients of the autograd path (param.grad) vs the hand-written backward (flat grad views)."""
    for pname, (key, head) in names.items():
        g_ref = dict(slow_net.named_parameters())[pname].grad
        g = flat_net.g[key][head].reshape(g_ref.shape)
        scale = float(g_ref.abs().max()) + 1e-12
        assert float((g - g_ref).abs().max()) <= 1e-4 * scale + 1e-9, (pname, float((g - g_ref).abs().max()), scale)


@pytest.mark.parametrize("hidden,B", ((32, 64), (256, 256)))
def test_fast_path_equals_autograd_path(hidden, B):
    slow, fast, args = make_pair(hidden)
    fast.enable_fast_path(B)
    for step in range(3):
        b_sac, b_qr, e1, e2 = batch(B, 10 + step)
        ls = slow.update_parameters(None, B, step, safety_critic=slow.safety_critic, batch=b_sac, eps_next=e1,
                                    eps_pi=e2)
        lf = fast.update_parameters(None, B, step, safety_critic=fast.safety_critic, batch=b_sac, eps_next=e1,
                                    eps_pi=e2)
        for x, y in zip(ls[:3], lf[:3]):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-5), (step, float(x), float(y))
        slow.safety_critic.update_parameters(policy=slow.policy, batch=b_qr, eps_next=e1, eps_pi=e2)
        fast.safety_critic.update_parameters(policy=fast.policy, batch=b_qr, eps_next=e1, eps_pi=e2)
        for x, y in zip(slow.safety_critic.last_losses, fast.safety_critic.last_losses):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6), (step, float(x), float(y))
        fast.fast.gather_first_grads()      # (dW1, db1) live as row-tile partials when the first layer is fused
        twin = {"linear1.weight": ("W1", 0), "linear4.weight": ("W1", 1), "linear2.weight": ("W2", 0),
                "linear5.weight": ("W2", 1), "linear3.weight": ("W3", 0), "linear6.bias": ("b3", 1),
                "linear2.bias": ("b2", 0), "linear4.bias": ("b1", 1)}
        assert_grads_close(slow.critic, fast.fast.critic, twin)
        assert_grads_close(slow.safety_critic.safety_critic, fast.fast.qrisk, twin)
        assert_grads_close(slow.policy, fast.fast.policy,
                           {"linear1.weight": ("W1", 0), "linear2.weight": ("W2", 0), "linear2.bias": ("b2", 0)})
        assert close_scaled(fast.fast.policy.g["W3"][0, 0:2], slow.policy.mean_linear.weight.grad)
        assert close_scaled(fast.fast.policy.g["W3"][0, 2:4], slow.policy.log_std_linear.weight.grad)
        assert_grads_close(slow.safety_critic.policy, fast.fast.recpolicy,
                           {"linear1.weight": ("W1", 0), "linear2.weight": ("W2", 0), "mean.weight": ("W3", 0)})
        assert close_scaled(fast.fast.recpolicy.g["log_std"], slow.safety_critic.policy.log_std.grad)
        assert_nets_close(slow.critic, fast.critic)
        assert_nets_close(slow.critic_target, fast.critic_target)
        assert_nets_close(slow.policy, fast.policy)
        assert_nets_close(slow.safety_critic.safety_critic, fast.safety_critic.safety_critic)
        assert_nets_close(slow.safety_critic.safety_critic_target, fast.safety_critic.safety_critic_target)
        assert_nets_close(slow.safety_critic.policy, fast.safety_critic.policy)
    assert int(fast.fast.critic.step[0].item()) == 3 and fast.safety_critic.updates == 3
    # the modules still see the live weights (their parameters are views of the flat buffers)
    s = torch.randn(5, 2, device=DEV)
    a = torch.rand(5, 2, device=DEV)
    q1f, _ = fast.critic(s, a)
    q1s, _ = slow.critic(s, a)
    assert torch.allclose(q1f, q1s, rtol=1e-3, atol=1e-4)


def load(module, G, prefix):
    want = set(module.state_dict().keys())
    sd = {k[len(prefix) + 1:]: torch.as_tensor(G[k], device=DEV) for k in G.files
          if k.startswith(prefix + ".") and k[len(prefix) + 1:] in want}
    module.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("name", ("sac", "mf"))
def test_fast_path_matches_reference_kats(golden_dir, name):
    """G4 through the fused kernels: one SAC update / one Q_risk + recovery update, H=16, B=8."""
    G = np.load(os.path.join(golden_dir, "model_golden.npz"))
    argv = ["--env-name", "navigation1", "--cuda", "--hidden_size", "16"] + str(G[name + ".argv"]).split()
    args = arg_utils.get_args(argv)
    agent = SAC(OBS, ACT, args, "/tmp")
    pre = name + ".pre"
    load(agent.critic, G, pre + ".critic"); load(agent.critic_target, G, pre + ".critic")
    load(agent.policy, G, pre + ".policy")
    load(agent.safety_critic.safety_critic, G, pre + ".qrisk")
    load(agent.safety_critic.safety_critic_target, G, pre + ".qrisk")
    load(agent.safety_critic.policy, G, pre + ".recpolicy")
    agent.enable_fast_path(8)
    T = lambda k: torch.as_tensor(G[k], device=DEV)
    b = [T("g4.batch." + k) for k in ("s", "a", "r", "s2", "m")]
    e1, e2 = T("g4.eps_next"), T("g4.eps_pi")

    def check(module, prefix):
        n = 0
        for k, v in module.state_dict().items():
            key = prefix + "." + k
            if key in G.files and "num_batches_tracked" not in k:
                assert np.allclose(v.cpu().numpy(), G[key], rtol=1e-4, atol=2e-6), key
                n += 1
        assert n >= 4
    post = name + ".post"
    if name == "sac":
        res = agent.update_parameters(None, 8, 0, safety_critic=agent.safety_critic, batch=tuple(b), eps_next=e1,
                                      eps_pi=e2, as_floats=True)
        assert np.allclose(res, G["sac.returns"], rtol=1e-4, atol=2e-6)
        check(agent.critic, post + ".critic"); check(agent.critic_target, post + ".critic_target")
        check(agent.policy, post + ".policy")
    else:
        b[2] = T("g4.cbatch.c")
        agent.safety_critic.update_parameters(policy=agent.policy, batch=tuple(b), eps_next=e1, eps_pi=e2)
        check(agent.safety_critic.safety_critic, post + ".qrisk")
        check(agent.safety_critic.safety_critic_target, post + ".qrisk_target")
        check(agent.safety_critic.policy, post + ".recpolicy")


def test_fast_actor_matches_module_path():
    """FastActor (fused kernels) vs the nn.Module path of VectorLoop.act on the same weights and noise."""
    from recovery_rl_amd.fast_update import FastActor
    _, agent, args = make_pair(256)
    fast = agent.enable_fast_path(256)
    n = 1000
    actor = FastActor(fast, n)
    obs = torch.randn(n, 2, device=DEV) * torch.tensor([20.0, 4.0], device=DEV) + torch.tensor([-30.0, 0.0], device=DEV)
    noise = torch.randn(2, n, 2, device=DEV)
    a_ref, _, _ = agent.policy.sample(obs, noise[0])
    risk = agent.safety_critic.get_value(obs, a_ref).squeeze(1)
    thr = float(risk.median())                       # a threshold that splits the batch
    task, real, rec = actor.act(obs, thr, True, True, noise=noise)
    r_ref, _, _ = agent.safety_critic.policy.sample(obs, noise[1])
    assert torch.allclose(task, a_ref, rtol=1e-4, atol=1e-5)
    gate = risk > thr
    border = (risk - thr).abs() < 1e-5
    assert torch.equal(rec.bool()[~border], gate[~border])
    want = torch.where(rec.bool().unsqueeze(1), r_ref, a_ref)
    assert torch.allclose(real, want, rtol=1e-4, atol=1e-5)
    assert 0 < int(rec.sum()) < n
    task2, real2, rec2 = actor.act(obs, thr, False, False, noise=noise)
    assert rec2 is None and torch.allclose(task2, a_ref, rtol=1e-4, atol=1e-5)


def test_normal_fill_is_the_documented_philox_stream():
    """rrl_normal_fill: pair i = Philox normal (seed, i, stream 8, tick), f32-rounded; bit-exact vs the C checker;
    the device tick advances per launch (graph-replay safe)."""
    from oracle import c_oracle
    lib = _lib.load()
    n, seed = 1500, 0xABCDEF12345
    out = torch.zeros(n, 2, device=DEV)
    tick = torch.zeros(2, dtype=torch.int64, device=DEV)
    for t in range(2):
        _lib.check(lib.rrl_normal_fill(n, seed, 0, _lib.ptr(tick), 1, _lib.ptr(out), _lib.current_stream()), "fill")
        want = c_oracle.normals(seed, n, 8, t).astype(np.float32)
        assert np.array_equal(out.cpu().numpy(), want)
    assert int(tick[0].item()) == 2
    z = torch.zeros(1 << 18, 2, device=DEV)
    _lib.check(lib.rrl_normal_fill(1 << 18, 7, 3, None, 0, _lib.ptr(z), _lib.current_stream()), "fill")
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1) < 5e-3


def test_adam_multi_equals_separate_adam_steps():
    from recovery_rl_amd.fast_update import FlatNet, adam_multi
    torch.manual_seed(0)
    shapes = [[("a", (300, 7)), ("b", (11,))], [("w", (40000,))], [("c", (5, 5))]]
    nets_a = [FlatNet(sh, DEV) for sh in shapes]
    nets_b = [FlatNet(sh, DEV) for sh in shapes]
    tgt_a, tgt_b = FlatNet(shapes[0], DEV), FlatNet(shapes[0], DEV)
    for na, nb in zip(nets_a, nets_b):
        na.flat.normal_()
        nb.flat.copy_(na.flat)
    tgt_a.flat.normal_()
    tgt_b.flat.copy_(tgt_a.flat)
    for step in range(3):
        for na, nb in zip(nets_a, nets_b):
            na.grad.normal_()
            nb.grad.copy_(na.grad)
        nets_a[0].adam(3e-4, target=tgt_a, tau=0.005)
        nets_a[1].adam(3e-4)
        nets_a[2].adam(3e-4)
        adam_multi(3e-4, [(nets_b[0], tgt_b, 0.005), (nets_b[1], None, 0.0), (nets_b[2], None, 0.0)])
    for na, nb in zip(nets_a, nets_b):
        assert torch.equal(na.flat, nb.flat) and torch.equal(na.m, nb.m) and torch.equal(na.v, nb.v)
        assert int(nb.step[0].item()) == 3 and int(nb.step[1].item()) == 0
    assert torch.equal(tgt_a.flat, tgt_b.flat)


@pytest.mark.parametrize("hidden,B", ((32, 64), (256, 256)))
def test_loss_fused_head_backward_is_bit_identical_to_the_separate_grad_kernels(hidden, B):
    """rrl_mlp_head_backward_loss evaluates the formulas of the stand-alone *_grad / *_head_bwd kernels inside the
    head-backward kernel, and rrl_mlp_hidden_backward runs the dW2 / dh1 tiles of rrl_gemm_f32 in one launch: every
    parameter after 3 updates must be bit-identical to the unfused launches, the logged losses close."""
    _, a, _ = make_pair(hidden)
    _, b, _ = make_pair(hidden)
    for dst, src in ((b.critic, a.critic), (b.critic_target, a.critic_target), (b.policy, a.policy),
                     (b.safety_critic.safety_critic, a.safety_critic.safety_critic),
                     (b.safety_critic.safety_critic_target, a.safety_critic.safety_critic_target),
                     (b.safety_critic.policy, a.safety_critic.policy)):
        dst.load_state_dict(copy.deepcopy(src.state_dict()))
    a.enable_fast_path(B)
    b.enable_fast_path(B)
    a.fast.fuse_loss, b.fast.fuse_loss = True, False
    a.fast.set_fuse_first(False)       # this test isolates the loss fusion and the paired hidden-layer launch: the
    b.fast.set_fuse_first(False)       # first layer of the backward stays its own launch on both sides
    for name in ("pol_a", "pol_b", "cri_a", "cri_b", "qr_a", "qr_b", "rec_a"):
        getattr(b.fast, name).pair_hidden = False          # ... and the two hidden-layer GEMMs as separate launches
    for step in range(3):
        b_sac, b_qr, e1, e2 = batch(B, 40 + step)
        for ag in (a, b):
            ag.update_parameters(None, B, step, safety_critic=ag.safety_critic, batch=b_sac, eps_next=e1, eps_pi=e2)
            ag.safety_critic.update_parameters(policy=ag.policy, batch=b_qr, eps_next=e1, eps_pi=e2)
        for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
            fa, fb = getattr(a.fast, name), getattr(b.fast, name)
            assert torch.equal(fa.flat, fb.flat), (step, name)
            assert torch.equal(fa.grad, fb.grad), (step, name)
        torch.testing.assert_close(a.fast.losses, b.fast.losses, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("env_name,extra", (
    ("navigation1", ["--gamma_safe", "0.8", "--eps_safe", "0.3", "--num_unsafe_transitions", "4000"]),
    ("maze", ["--gamma_safe", "0.5", "--eps_safe", "0.15", "--pos_fraction", "0.3", "--num_unsafe_transitions", "4000"]),
))
def test_grouped_launches_equal_the_separate_ones(env_name, extra):
    """The lock-step iteration with independent kernels sharing launches (rrl_sample_multi, rrl_mlp3_forward_multi,
    rrl_mlp_*_backward_multi, rrl_policy_heads_fwd_multi, rrl_*_step_push_select: ~30 launches) against the same
    iteration issued kernel by kernel (~49 launches): after eager iterations AND hipGraph replays every parameter,
    Adam moment, replay row, env state and counter must be bit-identical."""
    import bench
    loops = []
    for grouped in (True, False):
        cfg = arg_utils.get_args(["--env-name", env_name, "--cuda", "--use_recovery", "--MF_recovery", "--num_envs", "512",
                                  "--seed", "4"] + extra)
        loop = bench.build_loop(cfg, torch.device(DEV), pretrain=5)
        loop.agent.fast.grouped = grouped
        loops.append(loop)
    for phase in range(2):
        for loop in loops:
            if phase == 0:
                for _ in range(4):
                    loop.vector_step(True, False, True)
            else:
                loop.capture(online_qrisk=True)
                for _ in range(5):
                    loop.replay()
        torch.cuda.synchronize()
        a, b = loops
        for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
            fa, fb = getattr(a.agent.fast, name), getattr(b.agent.fast, name)
            assert torch.equal(fa.flat, fb.flat), (phase, name)
            assert torch.equal(fa.m, fb.m) and torch.equal(fa.v, fb.v) and torch.equal(fa.step, fb.step), (phase, name)
        for ma, mb in ((a.memory, b.memory), (a.recovery_memory, b.recovery_memory)):
            assert torch.equal(ma.state, mb.state) and torch.equal(ma.tick, mb.tick)
            for fa, fb in ((ma.s, mb.s), (ma.a, mb.a), (ma.r, mb.r), (ma.s2, mb.s2), (ma.m, mb.m)):
                assert torch.equal(fa, fb), phase
            ma.check_error()
        assert torch.equal(a.env.pos, b.env.pos) and torch.equal(a.env.t, b.env.t) and torch.equal(a.env.obs, b.env.obs)
        assert torch.equal(a.env.status, b.env.status) and a.env._status_live == b.env._status_live
        assert torch.equal(a.stats, b.stats) and torch.equal(a.reward_sums, b.reward_sums)
        assert torch.equal(a.agent.fast.noise_tick, b.agent.fast.noise_tick)
        assert torch.equal(a._actor.recovery, b._actor.recovery) and torch.equal(a._actor.real_action, b._actor.real_action)
        assert a.read_stats() == b.read_stats()
    assert int(loops[0].stats[6].item()) > 0                 # the recovery gate fired on some env-steps


def test_grouped_entry_points_match_their_members():
    """rrl_mlp3_forward_multi / rrl_policy_heads_fwd_multi / rrl_sample_multi on their own: outputs equal the
    stand-alone launches', incl. a mixed group on the non-split path (the acting pass at 4096 rows)."""
    from recovery_rl_amd.fast_update import Stack, forward_multi, heads_multi
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    a, _, _ = make_pair(256)
    f = a.enable_fast_path(256)
    for n_rows in (256, 4096):
        x = torch.randn(n_rows, 4, device=DEV)
        s1, s2, s3 = Stack(f.critic, n_rows), Stack(f.qrisk, n_rows), Stack(f.critic, n_rows)
        r1, r2, r3 = Stack(f.critic, n_rows), Stack(f.qrisk, n_rows), Stack(f.critic, n_rows)
        forward_multi([s1.forward_desc(x), s2.forward_desc(x, save=False), s3.forward_desc(x, params=f.critic_target)])
        r1.forward(x), r2.forward(x, save=False), r3.forward(x, params=f.critic_target)
        for g, r in ((s1, r1), (s2, r2), (s3, r3)):
            tg, tr = g.parts[0], r.parts[0]
            assert g.parts[1:] == r.parts[1:] and torch.equal(tg, tr)
        assert torch.equal(s1.h1, r1.h1) and torch.equal(s1.h2, r1.h2) and torch.equal(s3.h2, r3.h2)
    # draws + noise
    rng = np.random.RandomState(0)
    rows = lambda n: [torch.as_tensor(v, device=DEV) for v in (
        rng.randn(n, 2).astype(np.float32), rng.randn(n, 2).astype(np.float32),
        (rng.uniform(size=n) < 0.2).astype(np.float32), rng.randn(n, 2).astype(np.float32), np.ones(n, np.float32))]
    data = rows(5000)
    mems = [(ReplayMemory(8192, 3, device=DEV), ConstraintReplayMemory(8192, 3, device=DEV)) for _ in range(2)]
    for m, c in mems:
        m.push(*data), c.push(*data)
    (m1, c1), (m2, c2) = mems
    lib = _lib.load()
    import ctypes as C
    noise_a, noise_b = torch.zeros(3000, 2, device=DEV), torch.zeros(3000, 2, device=DEV)
    tick_a, tick_b = torch.zeros(2, dtype=torch.int64, device=DEV), torch.zeros(2, dtype=torch.int64, device=DEV)
    for call in range(3):
        d1, out1 = m1.draw_desc(256)
        d2, out2 = c1.draw_desc(256, pos_fraction=0.3)
        _lib.check(lib.rrl_sample_multi(C.byref(d1), C.byref(d2), 3000, 77, 0, _lib.ptr(tick_a), 1, _lib.ptr(noise_a),
                                        _lib.current_stream()), "rrl_sample_multi")
        want1 = [t.clone() for t in m2.sample(256)]
        want2 = [t.clone() for t in c2.sample(256, pos_fraction=0.3)]
        _lib.check(lib.rrl_normal_fill(3000, 77, 0, _lib.ptr(tick_b), 1, _lib.ptr(noise_b), _lib.current_stream()), "fill")
        assert all(torch.equal(x, y) for x, y in zip(out1, want1)) and all(torch.equal(x, y) for x, y in zip(out2, want2))
        assert torch.equal(noise_a, noise_b) and torch.equal(tick_a, tick_b)
        assert torch.equal(m1._batch(256)[5], m2._batch(256)[5]) and torch.equal(c1._batch(256)[5], c2._batch(256)[5])
    m1.check_error(), c1.check_error()


@pytest.mark.parametrize("B", (256,))
def test_first_layer_backward_inside_the_hidden_launch(B):
    """rrl_first_layer_t: (dW1, db1) as row-tile partials summed by Adam and dx as column-tile partials summed by the
    policy-head backward, against the stand-alone rrl_mlp_input_backward launch: the partial sums add up to its outputs
    (summation order differs: float tolerance), and three updates give the same parameters within Adam's noise floor."""
    _, a, _ = make_pair(256)
    _, b, _ = make_pair(256)
    for dst, src in ((b.critic, a.critic), (b.critic_target, a.critic_target), (b.policy, a.policy),
                     (b.safety_critic.safety_critic, a.safety_critic.safety_critic),
                     (b.safety_critic.safety_critic_target, a.safety_critic.safety_critic_target),
                     (b.safety_critic.policy, a.safety_critic.policy)):
        dst.load_state_dict(copy.deepcopy(src.state_dict()))
    a.enable_fast_path(B)
    b.enable_fast_path(B)
    assert a.fast.cri_a.fuse_first and a.fast.pol_b.fuse_first
    b.fast.set_fuse_first(False)
    for step in range(3):
        b_sac, b_qr, e1, e2 = batch(B, 70 + step)
        for ag in (a, b):
            ag.update_parameters(None, B, step, safety_critic=ag.safety_critic, batch=b_sac, eps_next=e1, eps_pi=e2)
            ag.safety_critic.update_parameters(policy=ag.policy, batch=b_qr, eps_next=e1, eps_pi=e2)
        if step == 0:
            fa, fb = a.fast, b.fast
            for sa, sb, net in ((fa.cri_a, fb.cri_a, "critic"), (fa.pol_b, fb.pol_b, "policy"), (fa.qr_a, fb.qr_a, "qrisk"),
                                (fa.rec_a, fb.rec_a, "recpolicy")):
                summed = sa.first_part.sum(0)
                want = getattr(fb, net).grad[:sa.n_first]
                scale = float(want.abs().max())
                assert float((summed - want).abs().max()) <= 2e-6 * scale + 1e-9, net
            dxa = fa.cri_b.dx_part.sum(0)
            assert float((dxa - fb.cri_b.dx).abs().max()) <= 2e-6 * float(fb.cri_b.dx.abs().max()) + 1e-12
        for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
            pa, pb = getattr(a.fast, name).flat, getattr(b.fast, name).flat
            assert float((pa - pb).abs().max()) < 3e-5, (step, name, float((pa - pb).abs().max()))      # 0.1 lr
    assert int(a.fast.critic.step[0].item()) == 3


@pytest.mark.parametrize("hidden,B", ((256, 256), (128, 128)))
def test_paired_head_and_hidden_backward_equals_the_two_launches(monkeypatch, hidden, B):
    """rrl_mlp_backward_pair_multi: with full aligned tiles (256 x 256, 128 x 128) the head backward and the hidden backward
    go out as ONE launch whose tiles derive dh2 from h2, the loss description and W3 -- the critic-loss kinds (one output) and,
    since round 5, the policy-head kinds (tanh-Gaussian: four outputs, stochastic: two; dh2 = the fmaf chain over the outputs,
    W3 of the NN tiles through LDS).  Against the same updates issued as rrl_mlp_head_backward_multi +
    rrl_mlp_hidden_backward_multi: every parameter and gradient bit for bit (sac.py:216-239, qrisk.py:150-158)."""
    from recovery_rl_amd import fast_update
    _, a, _ = make_pair(hidden)
    _, b, _ = make_pair(hidden)
    for dst, src in ((b.critic, a.critic), (b.critic_target, a.critic_target), (b.policy, a.policy),
                     (b.safety_critic.safety_critic, a.safety_critic.safety_critic),
                     (b.safety_critic.safety_critic_target, a.safety_critic.safety_critic_target),
                     (b.safety_critic.policy, a.safety_critic.policy)):
        dst.load_state_dict(copy.deepcopy(src.state_dict()))
    a.enable_fast_path(B)
    b.enable_fast_path(B)
    # the paired launches fold the critic's dx partials over four column tiles inside the workgroup (rrl_first_layer_t.dx_fold);
    # one-tile workgroups cannot: the two-launch side keeps the 16 tile partials and its consumer adds them in the same
    # grouped order (rrl_loss_t.da_group = 4) -- the same bits
    assert a.fast.cri_b.fold_dx == (B <= 256)
    for st in b.fast.stacks():
        st.fold_dx = False
    paired = fast_update.backward_multi
    calls = {"pair": 0, "two": 0}

    def two_launches(triples):
        lib, st, n = _lib.load(), _lib.current_stream(), len(triples)
        heads = (_lib.rrl_head_bwd_t * n)(*[t[0] for t in triples])
        hiddens = (_lib.rrl_hidden_bwd_t * n)(*[t[1] for t in triples])
        rest = [t[2] for t in triples if t[2] is not None]
        _lib.check(lib.rrl_mlp_head_backward_multi(n, heads, st), "rrl_mlp_head_backward_multi")
        _lib.check(lib.rrl_mlp_hidden_backward_multi(n, hiddens, st), "rrl_mlp_hidden_backward_multi")
        if rest:
            inputs = (_lib.rrl_input_bwd_t * len(rest))(*rest)
            _lib.check(lib.rrl_mlp_input_backward_multi(len(rest), inputs, st), "rrl_mlp_input_backward_multi")
        calls["two"] += 1

    def counted(triples):
        calls["pair"] += 1
        return paired(triples)

    for step in range(3):
        b_sac, b_qr, e1, e2 = batch(B, 90 + step)
        for ag, fn in ((a, counted), (b, two_launches)):
            monkeypatch.setattr(fast_update, "backward_multi", fn)
            ag.update_parameters(None, B, step, safety_critic=ag.safety_critic, batch=b_sac, eps_next=e1, eps_pi=e2)
            ag.safety_critic.update_parameters(policy=ag.policy, batch=b_qr, eps_next=e1, eps_pi=e2)
        for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
            fa, fb = getattr(a.fast, name), getattr(b.fast, name)
            assert torch.equal(fa.grad, fb.grad), (step, name)
            assert torch.equal(fa.flat, fb.flat), (step, name)
        assert torch.equal(a.fast.losses, b.fast.losses)
    assert calls["pair"] == calls["two"] > 0


def _frag_order(W2):
    """rrl_w2_pack's layout in torch: W2p[g][n][j][q][i][c] = W2[g][16 n + i][16 j + 4 q + c]."""
    G, H, _ = W2.shape
    return W2.reshape(G, H // 16, 16, H // 16, 4, 4).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)


@pytest.mark.parametrize("batch", (256, 64, 200))
def test_fragment_order_copy_of_w2_follows_the_parameters(batch):
    """rrl_stack_t.W2p: the forward kernels read W2 a second time in MFMA fragment order.  The copy is made by rrl_w2_pack (every
    eager forward re-makes it: torch code may have written the parameters), kept in step by the fused optimiser launch (own
    parameters AND Polyak target) -- so after eager updates, after graph replays and after a write through the modules the copies
    of all six networks are the permutation of their parameters, and the forward results do not depend on which layout was read."""
    import arg_utils
    import bench
    cfg = arg_utils.get_args(bench.config_argv("navigation1", 5, 256, 1) + ["--num_unsafe_transitions", "3000", "--batch_size",
                                                                           str(batch)])
    loop = bench.build_loop(cfg, torch.device("cuda:0"), pretrain=10)
    f = loop.agent.fast
    # batch 64 / 200: the first layer's backward is NOT fused into the hidden-layer launch (grad_part is None), the step is
    # FlatNet.adam(part=None) -- which must still be the launch that keeps the copies current inside a captured graph
    assert (f.qr_a.grad_part is not None) == (batch == 256)
    nets = {k: getattr(f, k) for k in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy")}
    assert all(n.w2p is not None for n in nets.values())

    def check(tag):
        torch.cuda.synchronize()
        for k, n in nets.items():
            assert torch.equal(n.w2p, _frag_order(n.p["W2"])), (tag, k)
    for _ in range(3):
        loop.vector_step(True, False, True)
    check("eager")
    loop.capture(online_qrisk=True)
    before = nets["critic"].p["W2"].clone()
    loop.advance(9)
    check("graph replays")
    assert not torch.equal(before, nets["critic"].p["W2"])
    # a write through the module's view (load_state_dict, a test harness): the next eager forward re-makes the copy
    with torch.no_grad():
        loop.agent.policy.linear2.weight.mul_(1.5)
    assert not torch.equal(nets["policy"].w2p, _frag_order(nets["policy"].p["W2"]))
    loop.vector_step(True, False, True)
    check("after a write through the module")
    # the two layouts give the same forward: bit-identical outputs with and without the copy
    st = f.q_stack if hasattr(f, "q_stack") else None
    x = torch.randn(256, 4, device="cuda")
    net = nets["critic"]
    from recovery_rl_amd.fast_update import Stack, forward_multi
    s1, s2 = Stack(net, 256), Stack(net, 256)
    d1 = s1.forward_desc(x)
    keep, net.w2p = net.w2p, None
    d2 = s2.forward_desc(x)
    net.w2p = keep
    assert d1.W2p and not d2.W2p
    forward_multi([d1]); forward_multi([d2])
    torch.cuda.synchronize()
    assert torch.equal(s1.scratch, s2.scratch) and torch.equal(s1.h2, s2.h2)
