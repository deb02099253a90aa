"""rrl_ens_train_grad (one-launch gather + forward + loss + backward of the PETS ensemble step) against autograd
of the PyTorch restatement of MPC.train's loss (recovery_rl/MPC.py:270-287, config/navigation1.py:52-96)."""
import copy

import numpy as np
import pytest
import torch

from recovery_rl_amd.MPC import MPC
from recovery_rl_amd.config import create_config
from recovery_rl_amd.ensemble_train import DECAY, PARAMS, FusedEnsembleTrainer
from recovery_rl_amd.env import make_vec_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(seed=0):
    torch.manual_seed(seed)
    env = make_vec_env("navigation2", 2, device=DEV, seed=1)
    mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1)
    g = torch.Generator(device=DEV).manual_seed(seed)
    n = 700
    s = torch.rand(n, 2, device=DEV, generator=g) * torch.tensor([40.0, 30.0], device=DEV) - \
        torch.tensor([45.0, 15.0], device=DEV)
    ac = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
    d = ac + 0.05 * torch.randn(n, 2, device=DEV, generator=g)
    mpc.train_in, mpc.train_targs = torch.cat([s, ac], 1).contiguous(), d.contiguous()
    mpc.model.fit_input_stats(mpc.train_in)
    with torch.no_grad():                       # leave the symmetric initialisation: biases and bounds in play
        for name in ("lin0_b", "lin1_b", "lin2_b", "lin3_b"):
            getattr(mpc.model, name).normal_(0, 0.05)
        mpc.model.max_logvar.copy_(torch.tensor([[0.4, -1.0]], device=DEV))
        mpc.model.min_logvar.copy_(torch.tensor([[-3.0, -6.0]], device=DEV))
    idxs = torch.randint(n, (mpc.model.num_nets, 3 * 32), device=DEV, generator=g)
    return mpc, idxs


def torch_grads(mpc, bi):
    m = mpc.model
    for p in m.parameters():
        p.grad = None
    loss = 0.01 * (m.max_logvar.sum() - m.min_logvar.sum()) + m.compute_decays()
    mean, logvar = m(mpc.train_in[bi], ret_logvar=True)
    tl = ((mean - mpc.train_targs[bi]) ** 2) * torch.exp(-logvar) + logvar
    nll = tl.mean(-1).mean(-1)
    (loss + nll.sum()).backward()
    return {n: getattr(m, n).grad.clone() for n in PARAMS}, nll.detach()


def test_gradients_equal_autograd():
    mpc, idxs = build()
    tr = FusedEnsembleTrainer(mpc.model)
    assert FusedEnsembleTrainer.supported(mpc.model, 32)
    tr.begin(mpc.train_in, mpc.train_targs)
    for b, width in ((0, 32), (1, 32), (2, 32), (0, 13), (1, 1)):      # full batches and shorter last batches
        bi = idxs[:, 32 * b:32 * b + width]                    # strided view: no copy on the fused path
        want, nll = torch_grads(mpc, bi)
        tr.gradients(bi)
        torch.testing.assert_close(tr.loss, nll, rtol=1e-5, atol=1e-6)
        for k, (name, g) in enumerate(zip(PARAMS, tr.grads)):
            if k < 8:                                   # the two row halves of the batch
                g = g + tr.grads2[k]
            if name in DECAY:                           # the decay gradient is added by the Adam kernel
                g = g + DECAY[name] * getattr(mpc.model, name).data
            scale = float(want[name].abs().max()) + 1e-12
            err = float((g - want[name]).abs().max())
            assert err <= 2e-5 * scale + 1e-9, (name, err, scale)


def test_fused_steps_track_the_pytorch_optimiser():
    """20 Adam steps on the same bootstrap batches: same loss trajectory, parameters within Adam's noise floor."""
    mpc_a, idxs = build(3)
    mpc_b, _ = build(3)
    tr = FusedEnsembleTrainer(mpc_a.model)
    losses = []
    tr.begin(mpc_a.train_in, mpc_a.train_targs)
    for step in range(20):
        bi = idxs[:, 32 * (step % 3):32 * (step % 3 + 1)]
        tr.step(bi)
        losses.append(tr.loss.clone())
        mpc_b._train_step(bi)
    for name in PARAMS:
        pa, pb = getattr(mpc_a.model, name), getattr(mpc_b.model, name)
        assert torch.allclose(pa, pb, rtol=1e-3, atol=2e-4), (name, float((pa - pb).abs().max()))   # 0.2 lr
    assert int(tr.steps[0][0].item()) == 20
    _, nll_b = torch_grads(mpc_b, idxs[:, :32])
    tr.gradients(idxs[:, :32])
    torch.testing.assert_close(tr.loss, nll_b, rtol=2e-3, atol=1e-4)
    assert float(tr.loss.sum()) < float(losses[0].sum())             # and it learns


def test_epoch_loop_in_c_equals_the_per_step_calls():
    mpc_a, idxs = build(5)
    mpc_b, _ = build(5)
    idxs = idxs[:, :77].contiguous()                    # 2 full batches + one of 13 rows
    ta, tb = FusedEnsembleTrainer(mpc_a.model), FusedEnsembleTrainer(mpc_b.model)
    ta.begin(mpc_a.train_in, mpc_a.train_targs)
    tb.begin(mpc_b.train_in, mpc_b.train_targs)
    for _ in range(2):
        ta.epoch(idxs, 32)
        for lo in range(0, 77, 32):
            tb.step(idxs[:, lo:lo + 32])
    for name in PARAMS:
        assert torch.equal(getattr(mpc_a.model, name), getattr(mpc_b.model, name)), name
    assert int(ta.steps[0][0].item()) == 6


# ---- pinned to the REFERENCE's MPC.train (tests/golden/mpc_train_golden.npz, gen_mpc_train_golden.py) --------------
import os  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_controller(rows):
    """Controller on the GPU with the reference's initial ensemble weights (mpc_golden.npz pt.*)."""
    G = np.load(os.path.join(GOLDEN, "mpc_golden.npz"))
    T = np.load(os.path.join(GOLDEN, "mpc_train_golden.npz"))
    env = make_vec_env("navigation2", 2, device=DEV, seed=1)
    mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1)
    sd = mpc.model.state_dict()
    mpc.model.load_state_dict({k: torch.as_tensor(G["pt." + k]).to(DEV) for k in sd})
    f32 = lambda k: torch.as_tensor(T[k][:rows], dtype=torch.float32, device=DEV)
    return mpc, T, (f32("data.s"), f32("data.a"), f32("data.s2"))


def golden_view(T, name, t):
    arr = t.detach().cpu().numpy()
    return arr.reshape(-1)[::int(T["stride"])] if name in ("lin1_w", "lin2_w") else arr


def test_fused_gradients_equal_the_reference_step():
    """rrl_ens_train_grad on the reference's first batch (MPC.py:266-292): loss and every gradient."""
    mpc, T, (s, a, s2) = reference_controller(32)
    mpc.train_in = torch.cat([s, a], 1).contiguous()
    mpc.train_targs = (s2 - s).contiguous()
    mpc.model.fit_input_stats(mpc.train_in)
    assert np.allclose(mpc.model.inputs_sigma.cpu().numpy(), T["step.sigma"], rtol=1e-6, atol=1e-6)
    tr = FusedEnsembleTrainer(mpc.model)
    tr.begin(mpc.train_in, mpc.train_targs)
    tr.gradients(torch.as_tensor(T["step.idxs"], device=DEV))
    m = mpc.model
    total = 0.01 * (m.max_logvar.sum() - m.min_logvar.sum()) + m.compute_decays() + tr.loss.sum()
    assert np.isclose(float(total.detach()), float(T["step.loss"]), rtol=1e-5)
    for k, (name, g) in enumerate(zip(PARAMS, tr.grads)):
        if k < 8:
            g = g + tr.grads2[k]
        if name in DECAY:
            g = g + DECAY[name] * getattr(m, name).data
        # (the 0.01 (sum max_logvar - sum min_logvar) term of MPC.py:270 is part of the kernel's bound gradients)
        want = T["step.grad." + name]
        got = golden_view(T, name, g)
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-9, name


@pytest.mark.parametrize("fused", (True, False))
def test_training_run_equals_the_reference(fused, monkeypatch):
    """MPC.train, 2 epochs over 200 rows with the reference's bootstrap table and shuffles injected: the trained
    parameters of the fused kernel path (and of the PyTorch path) against the reference's."""
    import recovery_rl_amd.MPC as mod
    mpc, T, (s, a, s2) = reference_controller(200)
    mpc.fused_train = fused
    tables = [torch.as_tensor(t, device=DEV) for t in T["train.shuffled"]]
    monkeypatch.setattr(torch, "randint", lambda n, size, **k: torch.as_tensor(T["train.idxs"], device=DEV))
    monkeypatch.setattr(mod, "shuffle_rows", lambda arr: tables.pop(0))
    mpc.train(s, a, random=True, next_obs=s2, epochs=2)
    assert (mpc._trainer is not None) == fused
    if fused:
        assert int(mpc._trainer.steps[0][0].item()) == 14
    for name in PARAMS:
        post, want = golden_view(T, name, getattr(mpc.model, name)), T["train.post." + name]
        assert np.abs(post - want).max() < 2e-4, (name, np.abs(post - want).max())       # 0.2 lr


def test_adam_state_moves_with_the_training_path():
    """The fused trainer and torch.optim.Adam (the general path: other shapes, `fused_train = False`) each hold moments
    and a step count: they are handed over, so a re-fit on the other path does not restart the bias correction (ONE
    optimiser in the reference)."""
    mpc, idxs = build(7)
    n = mpc.train_in.shape[0]
    s, a = mpc.train_in[:, :2].contiguous(), mpc.train_in[:, 2:].contiguous()
    s2 = (s + mpc.train_targs).contiguous()
    mpc.train_in, mpc.train_targs = mpc.train_in[:0], mpc.train_targs[:0]
    mpc.train(s, a, random=True, next_obs=s2, epochs=1, batch_size=32)            # fused: ceil(700/32) = 22 steps
    assert mpc._optim_owner == "fused" and int(mpc._trainer.steps[0][0].item()) == 22
    m_fused = mpc._trainer.m[0].clone()
    mpc.fused_train = False
    mpc.train(s[:64], a[:64], random=True, next_obs=s2[:64], epochs=1, batch_size=128)   # torch: ceil(764/128) = 6
    mpc.fused_train = True
    assert mpc._optim_owner == "torch"
    st = mpc.model.optim.state[mpc.model.lin0_w]
    assert int(float(st["step"])) == 28
    assert not torch.equal(st["exp_avg"], m_fused)
    mpc.train(s[:32], a[:32], random=True, next_obs=s2[:32], epochs=1, batch_size=32)    # back: ceil(796/32) = 25
    assert mpc._optim_owner == "fused" and int(mpc._trainer.steps[0][0].item()) == 53
    assert n == 700


# ---- large-batch kernels (rrl_ens_train_grad_big): the lock-step loop's online re-fit -------------------------------
def big_grads_vs(want, tr, mpc, rel):
    for name, g in zip(PARAMS, tr.grads):
        if name in DECAY:
            g = g + DECAY[name] * getattr(mpc.model, name).data
        scale = float(want[name].abs().max()) + 1e-12
        err = float((g - want[name]).abs().max())
        assert err <= rel * scale + 1e-9, (name, err, scale)


@pytest.mark.parametrize("batch", (1, 31, 64, 65, 200, 1000, 4096 + 17))
def test_large_batch_gradients_equal_autograd(batch):
    """Ragged sizes: one row, less than a 64-row tile, exactly one, one more, several workgroups per member, a size
    whose last tile is short and whose chunks of the weight-gradient pass are uneven."""
    mpc, _ = build(11)
    g = torch.Generator(device=DEV).manual_seed(batch)
    bi = torch.randint(mpc.train_in.shape[0], (mpc.model.num_nets, batch), device=DEV, generator=g)
    tr = FusedEnsembleTrainer(mpc.model)
    assert FusedEnsembleTrainer.supported(mpc.model, batch if batch > 32 else 33)
    tr.begin(mpc.train_in, mpc.train_targs)
    want, nll = torch_grads(mpc, bi)
    tr.gradients_big(bi)
    torch.testing.assert_close(tr.loss, nll, rtol=2e-5, atol=1e-6)
    big_grads_vs(want, tr, mpc, 5e-5)
    tr.gradients_big(bi)                                       # deterministic: fixed-order reductions everywhere
    again = [t.clone() for t in tr.grads]
    tr.gradients_big(bi)
    for x, y in zip(again, tr.grads):
        assert torch.equal(x, y)


def test_large_batch_kernels_equal_the_batch_32_kernel():
    mpc, idxs = build(13)
    tr = FusedEnsembleTrainer(mpc.model)
    tr.begin(mpc.train_in, mpc.train_targs)
    bi = idxs[:, :32]
    tr.gradients(bi)
    small = [g.clone() + (tr.grads2[k] if k < 8 else 0) for k, g in enumerate(tr.grads)]
    loss = tr.loss.clone()
    tr.gradients_big(bi)
    torch.testing.assert_close(tr.loss, loss, rtol=1e-5, atol=1e-6)
    for name, a, b in zip(PARAMS, small, tr.grads):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, name


def test_large_batch_gradients_at_the_config4_refit_size():
    """Batch 131 072 = 32 x 4096 envs (experiment.py:659 at BASELINE config 4) against autograd, f32 on both sides:
    sums over 131 072 rows in different orders agree to ~1e-4 of scale."""
    torch.manual_seed(17)
    env = make_vec_env("navigation2", 2, device=DEV, seed=1)
    mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1)
    g = torch.Generator(device=DEV).manual_seed(17)
    n, batch = 424000, 131072
    s = torch.rand(n, 2, device=DEV, generator=g) * torch.tensor([40.0, 30.0], device=DEV) - \
        torch.tensor([45.0, 15.0], device=DEV)
    ac = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
    d = ac + 0.05 * torch.randn(n, 2, device=DEV, generator=g)
    mpc.train_in, mpc.train_targs = torch.cat([s, ac], 1).contiguous(), d.contiguous()
    mpc.model.fit_input_stats(mpc.train_in)
    with torch.no_grad():
        for name in ("lin0_b", "lin1_b", "lin2_b", "lin3_b"):
            getattr(mpc.model, name).normal_(0, 0.05)
    idxs = torch.randint(n, (mpc.model.num_nets, n), device=DEV, generator=g)
    bi = idxs[:, batch:2 * batch]                              # a strided view of the bootstrap table
    tr = FusedEnsembleTrainer(mpc.model)
    tr.begin(mpc.train_in, mpc.train_targs)
    want, nll = torch_grads(mpc, bi)
    tr.gradients_big(bi)
    torch.testing.assert_close(tr.loss, nll, rtol=1e-4, atol=1e-6)
    big_grads_vs(want, tr, mpc, 3e-4)


def test_large_batch_epoch_tracks_the_pytorch_optimiser_and_the_training_golden():
    """(a) MPC.train with batch 128 on 700 rows, 3 epochs: fused large-batch path vs the PyTorch path, same bootstrap
    table and shuffles; (b) the reference's 2-epoch run (mpc_train_golden.npz, batch 32) driven through step_big."""
    mpc_a, idxs = build(19)
    mpc_b, _ = build(19)
    ta = FusedEnsembleTrainer(mpc_a.model)
    ta.begin(mpc_a.train_in, mpc_a.train_targs)
    table = idxs[:, :96].repeat(1, 4)[:, :300].contiguous()    # 300 columns: batches of 128, 128, 44
    for _ in range(3):
        ta.epoch(table, 128)
        for lo in range(0, 300, 128):
            mpc_b._train_step(table[:, lo:lo + 128])
    assert int(ta.steps[0][0].item()) == 9
    for name in PARAMS:
        pa, pb = getattr(mpc_a.model, name), getattr(mpc_b.model, name)
        assert torch.allclose(pa, pb, rtol=1e-3, atol=2e-4), (name, float((pa - pb).abs().max()))   # 0.2 lr
    # (b) reference run
    mpc, T, (s, a, s2) = reference_controller(200)
    mpc.train_in = torch.cat([s, a], 1).contiguous()
    mpc.train_targs = (s2 - s).contiguous()
    mpc.model.fit_input_stats(mpc.train_in)
    tr = FusedEnsembleTrainer(mpc.model)
    tr.begin(mpc.train_in, mpc.train_targs)
    tables = [torch.as_tensor(T["train.idxs"], device=DEV)] + [torch.as_tensor(t, device=DEV) for t in T["train.shuffled"]]
    for ep in range(2):
        tab = tables[ep]
        for lo in range(0, tab.shape[1], 32):
            tr.step_big(tab[:, lo:lo + 32])
    for name in PARAMS:
        post, want = golden_view(T, name, getattr(mpc.model, name)), T["train.post." + name]
        assert np.abs(post - want).max() < 2e-4, (name, np.abs(post - want).max())
