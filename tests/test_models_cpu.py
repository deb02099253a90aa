"""CPU suite: MLP modules, one SAC update and one Q_risk update against the KATs captured from
the reference (tests/golden/model_golden.npz; generator tests/golden/gen_model_golden.py).
float32 throughout; tolerance rel 1e-4 / abs 2e-6 (SURVEY.md section 8d parity gates)."""
import os

import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd.model import (DeterministicPolicy, GaussianPolicy, QNetwork, QNetworkConstraint,
                                   StochasticPolicy)
from recovery_rl_amd.sac import SAC
from recovery_rl_amd.spaces import Box

RTOL, ATOL = 1e-4, 2e-6
ACT = Box(-np.ones(2), np.ones(2))
OBS = Box(-np.ones(2) * np.inf, np.ones(2) * np.inf)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden.npz"))


def load(module, G, prefix):
    want = set(module.state_dict().keys())
    sd = {k[len(prefix) + 1:]: torch.as_tensor(G[k]) for k in G.files
          if k.startswith(prefix + ".") and k[len(prefix) + 1:] in want}
    module.load_state_dict(sd, strict=True)          # reference state_dicts load unchanged
    return module


def close(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return np.allclose(a, b, rtol=RTOL, atol=ATOL)


def assert_params(module, G, prefix):
    n = 0
    for k, v in module.state_dict().items():
        key = prefix + "." + k
        if key in G.files and "num_batches_tracked" not in k:
            assert np.allclose(v.cpu().numpy(), G[key], rtol=RTOL, atol=ATOL), key
            n += 1
    assert n >= 4


def test_forward_and_sample_kats(G):
    s, a, eps = (torch.as_tensor(G["g3." + k]) for k in ("s", "a", "eps"))
    q = load(QNetwork(2, 2, 16), G, "g3.q")
    q1, q2 = q(s, a)
    assert close(q1, G["g3.q.out1"]) and close(q2, G["g3.q.out2"])
    qc = load(QNetworkConstraint(2, 2, 16), G, "g3.qc")
    c1, c2 = qc(s, a)
    assert close(c1, G["g3.qc.out1"]) and close(c2, G["g3.qc.out2"])
    assert (c1 > 0).all() and (c1 < 1).all()
    gp = load(GaussianPolicy(2, 2, 16, ACT), G, "g3.gp")
    act, logp, mean = gp.sample(s, eps)
    assert close(act, G["g3.gp.action"]) and close(logp, G["g3.gp.logp"]) and close(mean, G["g3.gp.mean"])
    assert logp.shape == (8, 1)
    sp = load(StochasticPolicy(2, 2, 16, ACT), G, "g3.sp")
    act, logp, mean = sp.sample(s, eps)
    assert close(act, G["g3.sp.action"]) and close(logp, G["g3.sp.logp"]) and close(mean, G["g3.sp.mean"])
    assert logp.shape == (8,) and sp.log_std.dtype == torch.float32
    dp = load(DeterministicPolicy(2, 2, 16, ACT), G, "g3.dp")
    assert close(dp(s), G["g3.dp.mean"])


def test_parameter_counts_match_survey():
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(QNetwork(2, 2, 256)) == 134658
    assert count(QNetworkConstraint(2, 2, 256)) == 134666
    assert count(GaussianPolicy(2, 2, 256, ACT)) == 67588
    assert count(StochasticPolicy(2, 2, 256, ACT)) == 67076


def make_agent(G, name, device="cpu"):
    argv = ["--env-name", "navigation1", "--hidden_size", "16"] + str(G[name + ".argv"]).split()
    if device != "cpu" and "--cuda" not in argv:
        argv.append("--cuda")
    args = arg_utils.get_args(argv)
    agent = SAC(OBS, ACT, args, "/tmp")
    pre = name + ".pre"
    load(agent.critic, G, pre + ".critic")
    load(agent.critic_target, G, pre + ".critic")
    load(agent.policy, G, pre + ".policy")
    load(agent.safety_critic.safety_critic, G, pre + ".qrisk")
    load(agent.safety_critic.safety_critic_target, G, pre + ".qrisk")
    load(agent.safety_critic.policy, G, pre + ".recpolicy")
    return agent, args


def batch_of(G, constraint=False):
    b = [torch.as_tensor(G["g4.batch." + k]) for k in ("s", "a", "r", "s2", "m")]
    if constraint:
        b[2] = torch.as_tensor(G["g4.cbatch.c"])
    return tuple(b)


@pytest.mark.parametrize("name", ("sac", "sac_autoent", "sac_dgd", "sac_rcpo"))
def test_one_sac_update_matches_reference(G, name):
    agent, args = make_agent(G, name)
    eps_next, eps_pi = torch.as_tensor(G["g4.eps_next"]), torch.as_tensor(G["g4.eps_pi"])
    res = agent.update_parameters(None, 8, 0, nu=args.nu, safety_critic=agent.safety_critic,
                                  batch=batch_of(G), eps_next=eps_next, eps_pi=eps_pi, as_floats=True)
    assert np.allclose(res, G[name + ".returns"], rtol=RTOL, atol=ATOL), (res, G[name + ".returns"])
    post = name + ".post"
    assert_params(agent.critic, G, post + ".critic")
    assert_params(agent.critic_target, G, post + ".critic_target")       # G5: soft update
    assert_params(agent.policy, G, post + ".policy")
    if name == "sac_autoent":
        assert close(agent.log_alpha, G[post + ".log_alpha"])
    assert np.isclose(agent.log_nu.item(), G[post + ".log_nu"], rtol=RTOL)
    assert np.isclose(agent.log_lambda_RCPO.item(), G[post + ".log_lambda"], rtol=RTOL)
    # the step changed something
    assert not np.allclose(agent.critic.linear1.weight.detach().numpy(), G[name + ".pre.critic.linear1.weight"])


def test_one_qrisk_update_with_mf_recovery_matches_reference(G):
    name = "mf"
    agent, args = make_agent(G, name)
    eps_next, eps_pi = torch.as_tensor(G["g4.eps_next"]), torch.as_tensor(G["g4.eps_pi"])
    qr = agent.safety_critic
    qr.update_parameters(policy=agent.policy, batch=batch_of(G, constraint=True), eps_next=eps_next,
                         eps_pi=eps_pi)
    post = name + ".post"
    assert_params(qr.safety_critic, G, post + ".qrisk")
    assert_params(qr.safety_critic_target, G, post + ".qrisk_target")
    assert_params(qr.policy, G, post + ".recpolicy")
    b = batch_of(G)
    assert close(qr.get_value(b[0], b[1]), G[name + ".get_value"])
    q1, q2 = qr(b[0], b[1])
    assert torch.equal(torch.max(q1, q2), qr.get_value(b[0], b[1]))
    assert qr.updates == 1


def test_qrisk_batch_clamp_and_pos_fraction_gate():
    args = arg_utils.get_args(["--env-name", "maze", "--pos_fraction", "0.3", "--MF_recovery", "--use_recovery"])
    agent = SAC(Box(-0.3, 0.3, shape=(2,)), Box(-0.1 * np.ones(2), 0.1 * np.ones(2)), args, "/tmp")
    qr = agent.safety_critic
    assert qr.pos_fraction == 0.3
    assert qr.clamp_batch_size(256, 300) == 210            # int(0.7 * 300), qrisk.py:100-102
    assert qr.clamp_batch_size(256, 10 ** 6) == 256
    args = arg_utils.get_args(["--env-name", "navigation1"])
    assert SAC(OBS, ACT, args, "/tmp").safety_critic.pos_fraction is None    # pos_fraction -1 -> None


def test_select_action_shapes_and_eval():
    args = arg_utils.get_args(["--env-name", "navigation1", "--hidden_size", "16", "--use_recovery", "--MF_recovery"])
    agent = SAC(OBS, ACT, args, "/tmp")
    s = torch.randn(32, 2)
    a = agent.select_action(s)
    assert a.shape == (32, 2) and (a.abs() <= 1).all()
    m1, m2 = agent.select_action(s, eval=True), agent.select_action(s, eval=True)
    assert torch.equal(m1, m2)
    one = agent.select_action(np.array([-50.0, 0.0]))
    assert isinstance(one, np.ndarray) and one.shape == (2,)
    r = agent.safety_critic.select_action(s)
    assert r.shape == (32, 2)
    args = arg_utils.get_args(["--env-name", "navigation1", "--hidden_size", "16", "--use_recovery",
                               "--Q_sampling_recovery"])
    agent = SAC(OBS, ACT, args, "/tmp")
    r = agent.safety_critic.select_action(s[:4])
    assert r.shape == (4, 2) and (r.abs() <= 1).all()
    args = arg_utils.get_args(["--env-name", "navigation1", "--hidden_size", "16", "--DGD_constraints",
                               "--use_constraint_sampling"])
    agent = SAC(OBS, ACT, args, "/tmp")
    a = agent.select_action(s)
    assert a.shape == (32, 2) and (a.abs() <= 1).all()


# ---- sampling-based selectors pinned to the reference (tests/golden/select_golden.npz, gen_select_golden.py) --------
@pytest.fixture(scope="module")
def S(golden_dir):
    return np.load(os.path.join(golden_dir, "select_golden.npz"))


def _load_prefixed(module, S, prefix):
    sd = {k[len(prefix) + 1:]: torch.as_tensor(S[k]) for k in S.files if k.startswith(prefix + ".")}
    module.load_state_dict(sd, strict=True)


def sqrl_agent(S, device="cpu"):
    args = arg_utils.get_args(["--env-name", "navigation1", "--hidden_size", "16", "--DGD_constraints",
                               "--use_constraint_sampling"] + (["--cuda"] if device != "cpu" else []))
    agent = SAC(OBS, ACT, args, "/tmp")
    _load_prefixed(agent.policy, S, "sqrl.policy")
    _load_prefixed(agent.safety_critic.safety_critic, S, "sqrl.qrisk")
    return agent


def check_sqrl(S, device):
    agent = sqrl_agent(S, device)
    n = len(S["sqrl.u"])
    seen = set()
    for c in range(n):
        agent.eps_safe = float(S["sqrl.eps_safe"][c])
        st = torch.as_tensor(S["sqrl.state"][c], dtype=torch.float32, device=device).unsqueeze(0)
        eps = torch.as_tensor(S["sqrl.noise"][c], device=device).unsqueeze(0)
        a = agent._sqrl_action(st, eps=eps, draw=S["sqrl.idx"][c:c + 1])[0].detach().cpu().numpy()
        assert np.allclose(a, S["sqrl.action"][c], rtol=1e-5, atol=1e-6), (c, a, S["sqrl.action"][c], S["sqrl.n_safe"][c])
        seen.add(int(S["sqrl.n_safe"][c]))
    assert 0 in seen and 100 in seen and any(0 < k < 100 for k in seen)       # argmin branch, all safe, mixed
    # batched call == the per-state calls (the vectorised loop evaluates all envs at once)
    agent.eps_safe = float(S["sqrl.eps_safe"][4])
    rows = [c for c in range(n) if S["sqrl.eps_safe"][c] == S["sqrl.eps_safe"][4]] or [4]
    st = torch.as_tensor(S["sqrl.state"][rows], dtype=torch.float32, device=device)
    out = agent._sqrl_action(st, eps=torch.as_tensor(S["sqrl.noise"][rows], device=device),
                             draw=S["sqrl.idx"][rows])
    assert np.allclose(out.detach().cpu().numpy(), S["sqrl.action"][rows], rtol=1e-5, atol=1e-6)


def check_q_sampling(S, device):
    args = arg_utils.get_args(["--env-name", "navigation1", "--hidden_size", "16", "--use_recovery",
                               "--Q_sampling_recovery"] + (["--cuda"] if device != "cpu" else []))
    agent = SAC(OBS, ACT, args, "/tmp")
    _load_prefixed(agent.safety_critic.safety_critic, S, "qs.qrisk")
    st = torch.as_tensor(S["qs.state"], dtype=torch.float32, device=device)
    out = agent.safety_critic.select_action(st, candidates=S["qs.candidates"])
    assert np.array_equal(out.cpu().numpy(), S["qs.action"])               # the reference's argmin candidate, row by row
    one = agent.safety_critic.select_action(S["qs.state"][1], candidates=S["qs.candidates"][1:2])
    assert isinstance(one, np.ndarray) and np.array_equal(one, S["qs.action"][1])


def test_sqrl_constraint_sampling_matches_reference(S):
    """sac.py:139-161 incl. the reference's indexing of the full candidate list with the safe-list position."""
    check_sqrl(S, "cpu")


def test_q_sampling_recovery_matches_reference(S):
    """qrisk.py:214-225: argmin of Q_risk over 1000 injected uniform candidates."""
    check_q_sampling(S, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ("sac", "sac_autoent", "sac_dgd", "sac_rcpo"))
def test_one_sac_update_matches_reference_on_the_gpu(G, name):
    """The LR / RCPO / auto-entropy KATs of G4 through the modules on cuda (autograd path of the comparison algorithms)."""
    agent, args = make_agent(G, name, device="cuda")
    dev = agent.device
    eps_next, eps_pi = torch.as_tensor(G["g4.eps_next"], device=dev), torch.as_tensor(G["g4.eps_pi"], device=dev)
    res = agent.update_parameters(None, 8, 0, nu=args.nu, safety_critic=agent.safety_critic,
                                  batch=tuple(t.to(dev) for t in batch_of(G)), eps_next=eps_next, eps_pi=eps_pi,
                                  as_floats=True)
    assert np.allclose(res, G[name + ".returns"], rtol=RTOL, atol=ATOL), (res, G[name + ".returns"])
    post = name + ".post"
    assert_params(agent.critic, G, post + ".critic")
    assert_params(agent.critic_target, G, post + ".critic_target")
    assert_params(agent.policy, G, post + ".policy")
    assert np.isclose(agent.log_nu.item(), G[post + ".log_nu"], rtol=RTOL)
    assert np.isclose(agent.log_lambda_RCPO.item(), G[post + ".log_lambda"], rtol=RTOL)


@pytest.mark.gpu
def test_sampling_selectors_match_reference_on_the_gpu(S):
    check_sqrl(S, "cuda")
    check_q_sampling(S, "cuda")
