"""`--policy Deterministic` (recovery_rl/model.py:447-485, sac.py:115-123) against fixtures captured from
the reference (tests/golden/gen_det_policy_golden.py): forward, sample with the reference's noise draws (incl. the +-0.25
clamp), one SAC.update_parameters step, select_action -- on the CPU modules and, `-m gpu`, on cuda; plus a whole
`rrl_main`-style run of the flag on the GPU (one env, reference-order loop) and in the lock-step loop."""
import os
import pickle

import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd.model import DeterministicPolicy
from recovery_rl_amd.sac import SAC
from recovery_rl_amd.spaces import Box

ACT = Box(-np.ones(2), np.ones(2))
OBS = Box(-np.ones(2) * np.inf, np.ones(2) * np.inf)
RTOL, ATOL = 1e-4, 1e-6


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "det_policy_golden.npz"))


def load(module, G, prefix, dev):
    want = set(module.state_dict().keys())
    sd = {k[len(prefix) + 1:]: torch.as_tensor(G[k], device=dev) for k in G.files
          if k.startswith(prefix + ".") and k[len(prefix) + 1:] in want}
    module.load_state_dict(sd, strict=True)
    return module


def close(t, ref):
    return np.allclose(t.detach().cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


def check_policy(G, dev):
    dp = load(DeterministicPolicy(2, 2, 16, ACT).to(dev), G, "dp", dev)
    s = torch.as_tensor(G["s"], device=dev)
    assert close(dp(s), G["dp.forward"])
    for k in range(3):
        eps = torch.as_tensor(G["sample%d.noise" % k], device=dev) / 0.1      # sample() scales N(0,1) by 0.1 (:476)
        act, logp, mean = dp.sample(s, eps)
        assert close(act, G["sample%d.action" % k]) and close(mean, G["sample%d.mean" % k]) and float(logp) == 0.0
    act, _, _ = dp.sample(s, torch.as_tensor(G["clamp.noise"], device=dev) / 0.1)
    assert close(act, G["clamp.action"])                                     # noise clamped to +-0.25 (:477)
    assert np.abs(G["clamp.action"] - G["dp.forward"]).max() <= 0.25 + 1e-6


def check_update(G, dev):
    argv = ["--env-name", "navigation1", "--hidden_size", "16"] + str(G["upd.argv"]).split()
    if dev != "cpu":
        argv.append("--cuda")
    args = arg_utils.get_args(argv)
    agent = SAC(OBS, ACT, args, "/tmp")
    assert isinstance(agent.policy, DeterministicPolicy) and agent.alpha == 0 and agent.fast is None
    load(agent.critic, G, "upd.pre.critic", agent.device)
    load(agent.critic_target, G, "upd.pre.critic", agent.device)
    load(agent.policy, G, "upd.pre.policy", agent.device)
    batch = tuple(torch.as_tensor(G["upd.batch." + k], device=agent.device) for k in ("s", "a", "r", "s2", "m"))
    e1 = torch.as_tensor(G["upd.noise_next"], device=agent.device) / 0.1
    e2 = torch.as_tensor(G["upd.noise_pi"], device=agent.device) / 0.1
    res = agent.update_parameters(None, 8, 0, nu=args.nu, safety_critic=agent.safety_critic, batch=batch, eps_next=e1,
                                  eps_pi=e2, as_floats=True)
    assert np.allclose(res, G["upd.returns"], rtol=RTOL, atol=ATOL), (res, G["upd.returns"])
    for name, mod in (("critic", agent.critic), ("critic_target", agent.critic_target), ("policy", agent.policy)):
        for k, v in mod.state_dict().items():
            ref = G["upd.post.%s.%s" % (name, k)]
            assert np.allclose(v.cpu().numpy(), ref, rtol=2e-4, atol=3e-5), (name, k)      # atol = 0.1 lr (Adam)
    # select_action (sac.py:133-168): eval = the mean; train = mean + clamped noise
    st = G["select.state"]
    assert np.allclose(agent.select_action(st, eval=True), G["select.eval"], rtol=RTOL, atol=ATOL)
    noisy = G["select.eval"] + np.clip(G["select.noise"], -0.25, 0.25)
    assert np.allclose(G["select.train"], noisy, rtol=RTOL, atol=ATOL)        # the fixture is self-consistent
    got = agent.select_action(st)
    assert np.abs(got - G["select.eval"]).max() <= 0.25 + 1e-6


def test_deterministic_policy_matches_reference_cpu(G):
    check_policy(G, "cpu")


def test_sac_update_with_deterministic_policy_matches_reference_cpu(G):
    check_update(G, "cpu")


@pytest.mark.gpu
def test_deterministic_policy_matches_reference_on_the_gpu(G):
    check_policy(G, "cuda:0")
    check_update(G, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", ([], ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3"]))
def test_deterministic_policy_script_line_runs_on_the_gpu(tmp_path, extra):
    """`python -m rrl_main --cuda --env-name navigation1 --policy Deterministic ...` (the reference's CLI, sac.py:115-123):
    one env in the reference-order loop, then the same flags in the lock-step loop (autograd update path, hipGraph replay)."""
    from recovery_rl_amd.experiment import Experiment
    base = ["--env-name", "navigation1", "--cuda", "--policy", "Deterministic", "--hidden_size", "32", "--batch_size", "4",
            "--start_steps", "3", "--critic_safe_pretraining_steps", "10", "--num_unsafe_transitions", "600",
            "--eval", "", "--seed", "2"] + extra
    exp = Experiment(arg_utils.get_args(base + ["--num_eps", "2", "--logdir", str(tmp_path / "one")]))
    exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert len(data["train_stats"]) == 2 and exp.updates > 0 and isinstance(exp.agent.policy, DeterministicPolicy)
    w0 = exp.agent.policy.mean.weight.clone()
    vec = Experiment(arg_utils.get_args(base + ["--num_envs", "64", "--num_eps", "1000", "--log_every", "20", "--logdir",
                                                 str(tmp_path / "vec")]))
    hist = vec.run()
    assert hist and hist[-1]["sac_updates"] > 20 and vec.loop.graph is not None
    assert all(torch.isfinite(p).all() for p in vec.agent.policy.parameters())
    assert not torch.equal(vec.agent.policy.mean.weight, w0)
