"""CPU suite: PETS ensemble, TS-infinity index map and the CEM oracle against the KATs captured
from the reference (tests/golden/mpc_golden.npz; generator tests/golden/gen_mpc_golden.py)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from recovery_rl_amd.MPC import MPC
from recovery_rl_amd.config import ENV_CONSTANTS, OPT_CFG, PtModel


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "mpc_golden.npz"))


def load_ptmodel(G):
    m = PtModel(5, 4, 4)
    sd = {k[3:]: torch.as_tensor(G[k]) for k in G.files if k.startswith("pt.") and k[3:] in m.state_dict()}
    m.load_state_dict(sd, strict=True)
    return m


def test_ptmodel_forward_decays_and_input_stats(G):
    m = load_ptmodel(G)
    ptrs = (m.inputs_mu.data_ptr(), m.inputs_sigma.data_ptr())
    m.fit_input_stats(G["pt.data"])
    # re-fits rewrite the statistics in place (a captured hipGraph packs the planner's weights from these addresses) and keep
    # the reference's [1, in] shape
    assert (m.inputs_mu.data_ptr(), m.inputs_sigma.data_ptr()) == ptrs and tuple(m.inputs_mu.shape) == (1, 4)
    m.fit_input_stats(np.asarray(G["pt.data"]) * 2.0 + 1.0)
    assert (m.inputs_mu.data_ptr(), m.inputs_sigma.data_ptr()) == ptrs
    m.fit_input_stats(G["pt.data"])
    assert np.allclose(m.inputs_mu.numpy(), G["pt.fit_mu"], rtol=1e-6, atol=1e-6)
    assert np.allclose(m.inputs_sigma.numpy(), G["pt.fit_sigma"], rtol=1e-6, atol=1e-6)
    assert m.inputs_sigma[0, 3] == 1.0                       # constant column: sigma < 1e-12 -> 1
    x = torch.as_tensor(G["pt.x"])
    mean, var = m(x)
    _, logvar = m(x, ret_logvar=True)
    assert np.allclose(mean.detach().numpy(), G["pt.mean"], rtol=1e-4, atol=1e-5)
    assert np.allclose(var.detach().numpy(), G["pt.var"], rtol=1e-4, atol=1e-7)
    assert np.allclose(logvar.detach().numpy(), G["pt.logvar"], rtol=1e-4, atol=1e-5)
    assert np.isclose(m.compute_decays().item(), G["pt.decays"], rtol=1e-5)
    n_params = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert n_params == 5 * (4 * 200 + 200 + 2 * (200 * 200 + 200) + 200 * 4 + 4) + 4


def test_fresh_ensemble_init_statistics():
    g = torch.Generator().manual_seed(0)
    m = PtModel(5, 4, 4, generator=g)
    assert m.lin1_w.abs().max() <= 2 * (1 / (2 * np.sqrt(200))) + 1e-7      # truncated at 2 std
    assert abs(m.lin1_w.std().item() - 0.88 / (2 * np.sqrt(200))) < 0.002   # std of truncnorm(-2,2) = 0.88
    assert m.lin0_b.abs().sum() == 0 and m.max_logvar.tolist() == [[0.5, 0.5]]
    assert m.min_logvar.tolist() == [[-10.0, -10.0]]


def test_ts_infinity_index_map(G):
    stub = SimpleNamespace(model=SimpleNamespace(num_nets=5), npart=20)
    rows = 6 * 20
    mat = torch.arange(rows * 3, dtype=torch.float32).reshape(rows, 3)
    exp = MPC._expand_to_ts_format(stub, mat)
    assert np.array_equal(exp.numpy(), G["ts.expanded"])
    assert torch.equal(MPC._flatten_to_matrix(stub, exp), mat)
    # particle p of every candidate is bound to net p // 4
    r = torch.arange(rows)
    nets = MPC._expand_to_ts_format(stub, ((r % 20) // 4).float().reshape(rows, 1))
    for e in range(5):
        assert (nets[e] == e).all()


def test_config_constants_match_reference(G):
    assert ENV_CONSTANTS["navigation1"]["PLAN_HOR"] == ENV_CONSTANTS["navigation2"]["PLAN_HOR"] == 5
    assert ENV_CONSTANTS["maze"]["PLAN_HOR"] == 15
    assert OPT_CFG["CEM"] == {"popsize": 400, "num_elites": 40, "max_iters": 5, "alpha": 0.1}
    assert np.allclose(G["mpc.init_var"], np.tile(np.square(2.0) / 16, 10))     # (ub-lb)^2/16 tiled
    assert np.allclose(G["mpc.prev_sol"], 0)


@pytest.mark.parametrize("case", ("mid", "edge"))
def test_cem_update_oracle_matches_reference_iteration(G, case):
    """G8: given the reference's samples and costs, elites / new mean / new var must agree
    (numpy reduces the float32 elites in float32; the oracle accumulates in float64)."""
    pre = "cem." + case + "."
    samples, costs = G[pre + "samples"][None], G[pre + "costs"][None]
    mean, var = co.cem_update(samples, costs, G[pre + "init_mean"][None], G[pre + "init_var"][None],
                              int(G["cem.num_elites"]), float(G["cem.alpha"]))
    assert np.allclose(mean[0], G[pre + "new_mean"], rtol=1e-5, atol=1e-6)
    assert np.allclose(var[0], G[pre + "new_var"], rtol=1e-4, atol=1e-7)
    # and the reference's samples are z * sqrt(constrained_var) + mean (optimizers.py:95-102)
    mu, v, lb, ub = G[pre + "init_mean"], G[pre + "init_var"], G["cem.lb"], G["cem.ub"]
    cv = np.minimum(np.minimum(np.square((mu - lb) / 2), np.square((ub - mu) / 2)), v)
    assert np.allclose(samples[0], (G[pre + "z"] * np.sqrt(cv) + mu).astype(np.float32))
    if case == "edge":
        assert cv[0] < v[0] and cv[1] < v[1]                     # clamped near the bounds


def test_cem_sample_oracle_distribution_and_clamping():
    M, pop, dim = 3, 400, 10
    mean = np.zeros((M, dim))
    mean[1, 0], mean[1, 1] = 0.97, -0.99
    var = np.full((M, dim), 0.25)
    var[2] = 1e-4                                                 # max(var) <= epsilon: inactive
    lb, ub = -np.ones(dim), np.ones(dim)
    samples, active = co.cem_sample(mean, var, lb, ub, pop, epsilon=1e-3, seed=4, counter=2)
    assert active.tolist() == [1, 1, 0] and not samples[2].any()
    cv = np.minimum(np.minimum(np.square((mean - lb) / 2), np.square((ub - mean) / 2)), var)
    z = (samples - mean[:, None]) / np.sqrt(cv)[:, None]
    z = z[:2]
    assert np.abs(z).max() <= 2 + 1e-5
    assert abs(z.mean()) < 0.03 and abs(z.std() - 0.8796) < 0.02      # std of N(0,1) truncated at +-2
    assert (samples[1, :, 0] <= 1.0).all() and (samples[1, :, 1] >= -1.0).all()
    # counter / seed change the draw; same inputs reproduce it
    s2, _ = co.cem_sample(mean, var, lb, ub, pop, epsilon=1e-3, seed=4, counter=3)
    s3, _ = co.cem_sample(mean, var, lb, ub, pop, epsilon=1e-3, seed=4, counter=2)
    assert not np.array_equal(s2, samples) and np.array_equal(s3, samples)
    # sticky: an env that went inactive stays inactive
    _, act = co.cem_sample(mean, var, lb, ub, pop, sticky=True, active=np.array([0, 1, 1], np.uint8))
    assert act.tolist() == [0, 1, 0]


def test_cem_update_rejects_more_elites_than_population():
    with pytest.raises(ValueError):
        co.cem_update(np.zeros((1, 4, 2), np.float32), np.zeros((1, 4), np.float32), np.zeros((1, 2)),
                      np.ones((1, 2)), 5, 0.1)


# ---- MPC.train pinned to the reference (tests/golden/mpc_train_golden.npz, gen_mpc_train_golden.py) ----------------
TRAIN_PARAMS = ("lin0_w", "lin0_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "lin3_w", "lin3_b", "max_logvar",
                "min_logvar")


@pytest.fixture(scope="module")
def T(golden_dir):
    return np.load(os.path.join(golden_dir, "mpc_train_golden.npz"))


def golden_view(T, name, arr):
    """The fixture stores the two 5x200x200 matrices on a strided subset."""
    arr = np.asarray(arr)
    return arr.reshape(-1)[::int(T["stride"])] if name in ("lin1_w", "lin2_w") else arr


def cpu_controller(G):
    """An MPC with only what train() touches, on the CPU (the planner side needs the GPU)."""
    from recovery_rl_amd.config import NN_TRAIN_CFG, targ_proc
    mpc = MPC.__new__(MPC)
    mpc.device = torch.device("cpu")
    mpc.model = load_ptmodel(G)
    mpc.model.optim = torch.optim.Adam(mpc.model.parameters(), lr=0.001)
    mpc.targ_proc, mpc.model_train_cfg = targ_proc, dict(NN_TRAIN_CFG)
    mpc.train_in, mpc.train_targs = torch.zeros(0, 4), torch.zeros(0, 2)
    mpc.has_been_trained, mpc.fused_train, mpc.graph_train, mpc._trainer = False, False, False, None
    return mpc


def run_train_with_tables(mpc, monkeypatch, T, case, rows, epochs):
    """MPC.train with the reference's bootstrap table and shuffles injected (it drew them from np.random)."""
    import recovery_rl_amd.MPC as mod
    tables = [torch.as_tensor(t) for t in (T[case + ".shuffled"] if case == "train" else [T["step.idxs"]])]
    monkeypatch.setattr(torch, "randint", lambda n, size, **k: torch.as_tensor(T[case + ".idxs"]))
    monkeypatch.setattr(mod, "shuffle_rows", lambda arr: tables.pop(0))
    f32 = lambda k: torch.as_tensor(T[k][:rows], dtype=torch.float32)
    mpc.train(f32("data.s"), f32("data.a"), random=True, next_obs=f32("data.s2"), epochs=epochs)


def test_one_reference_optimiser_step(G, T, monkeypatch):
    """MPC.py:250-298: loss, gradients and post-step parameters of the reference's first batch-32 step."""
    mpc = cpu_controller(G)
    losses = []
    real = torch.Tensor.backward
    monkeypatch.setattr(torch.Tensor, "backward", lambda self, *a, **k: (losses.append(float(self.detach())), real(self, *a, **k))[1])
    run_train_with_tables(mpc, monkeypatch, T, "step", 32, 1)
    assert np.allclose(mpc.model.inputs_mu.numpy(), T["step.mu"], rtol=1e-6, atol=1e-6)
    assert np.allclose(mpc.model.inputs_sigma.numpy(), T["step.sigma"], rtol=1e-6, atol=1e-6)
    assert len(losses) == 1 and np.isclose(losses[0], float(T["step.loss"]), rtol=1e-5)
    for name in TRAIN_PARAMS:
        p = getattr(mpc.model, name)
        g, want = golden_view(T, name, p.grad.numpy()), T["step.grad." + name]
        assert np.abs(g - want).max() <= 1e-4 * np.abs(want).max() + 1e-9, name
        post, want = golden_view(T, name, p.detach().numpy()), T["step.post." + name]
        # first Adam step = -lr * sign(g): entries whose gradient is ~0 may land on either side
        assert np.mean(np.abs(post - want) > 1e-6) < 1e-3, name


def test_reference_training_run_two_epochs(G, T, monkeypatch):
    """14 optimiser steps (2 epochs over 200 rows, last batch of an epoch = 8 rows) with the reference's bootstrap
    table and shuffles: per-step losses and the trained parameters."""
    mpc = cpu_controller(G)
    losses = []
    real = torch.Tensor.backward
    monkeypatch.setattr(torch.Tensor, "backward", lambda self, *a, **k: (losses.append(float(self.detach())), real(self, *a, **k))[1])
    run_train_with_tables(mpc, monkeypatch, T, "train", 200, 2)
    assert np.allclose(mpc.train_in.numpy(), T["train.train_in"]) and np.allclose(mpc.train_targs.numpy(),
                                                                                T["train.train_targs"], atol=1e-5)
    assert np.allclose(losses, T["train.losses"], rtol=2e-4, atol=2e-4)
    for name in TRAIN_PARAMS:
        post, want = golden_view(T, name, getattr(mpc.model, name).detach().numpy()), T["train.post." + name]
        assert np.abs(post - want).max() < 2e-4, (name, np.abs(post - want).max())       # 0.2 lr
