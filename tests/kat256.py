"""Shared checker of the production-shape known-answer tests (tests/golden/model_golden_256.npz, produced by the imported
reference: gen_model_golden_256.py; inputs re-created by tests/golden/kat256_inputs.py).  Used by the CPU test (module /
autograd path) and the GPU tests (the fused kernels the bench times)."""
import os
import sys

import numpy as np
import torch

import arg_utils
from recovery_rl_amd.sac import SAC
from recovery_rl_amd.spaces import Box

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import kat256_inputs as K  # noqa: E402

ACT = Box(-np.ones(2), np.ones(2))
OBS = Box(-np.ones(2) * np.inf, np.ones(2) * np.inf)
REL = 1e-4                    # the tolerance north_star states for the f32 updates (SURVEY 8d parity gates)


def golden():
    return np.load(os.path.join(HERE, "golden", "model_golden_256.npz"))


def build_agent(device):
    """SAC + Q_risk + model-free recovery at --hidden_size 256, --batch_size 256 with the fixture's weights."""
    argv = ["--env-name", "navigation1", "--hidden_size", str(K.H), "--batch_size", str(K.B)] + K.ARGV
    if device != "cpu":
        argv.append("--cuda")
    agent = SAC(OBS, ACT, arg_utils.get_args(argv), "/tmp")
    qr = agent.safety_critic
    for module, tag in ((agent.critic, "critic"), (agent.policy, "policy"), (qr.safety_critic, "qrisk"),
                        (qr.policy, "recpolicy")):
        sd = module.state_dict()
        for k, v in K.weights(sd, tag).items():
            sd[k] = torch.as_tensor(v, device=device)
        module.load_state_dict(sd, strict=True)
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    qr.safety_critic_target.load_state_dict(qr.safety_critic.state_dict())
    return agent


def inputs(device):
    (s, a, r, s2, m), c, eps_next, eps_pi = K.batch()
    t = lambda x: torch.as_tensor(x, device=device)
    return tuple(t(x) for x in (s, a, r, s2, m)), t(c), t(eps_next), t(eps_pi)


def check_tensor(G, prefix, key, value, grad_like, big=None):
    """`value` (this stack) against the reference's record of tensor prefix.key: sampled entries and the sum.
    Gradients (grad_like): |x - ref| <= REL * max|ref| per sampled entry and REL * sum|ref| for the sum.
    Post-update weights: one Adam step from zero moments moves every entry by lr * g / (|g| + 1e-8) -- the sampled entries
    whose reference gradient is not rounding noise (`big`) must agree to rtol REL / atol 2e-6, all of them to 0.1 lr."""
    x = value.detach().float().cpu().numpy().ravel()
    ref = G["%s.%s.at" % (prefix, key)]
    idx = K.sample_index(prefix + "." + key, x.size)
    got = x[idx]
    if grad_like:
        scale = float(np.abs(ref).max()) + 1e-12
        assert float(np.abs(got - ref).max()) <= REL * scale + 1e-9, (prefix, key, float(np.abs(got - ref).max()), scale)
        total = float(G["%s.%s.abs" % (prefix, key)])
        assert abs(float(x.astype(np.float64).sum()) - float(G["%s.%s.sum" % (prefix, key)])) <= REL * total + 1e-9, (prefix, key)
    else:
        assert np.allclose(got, ref, rtol=REL, atol=3e-5), (prefix, key, float(np.abs(got - ref).max()))
        if big is not None and big.any():
            assert np.allclose(got[big], ref[big], rtol=REL, atol=2e-6), (prefix, key, float(np.abs(got[big] - ref[big]).max()))
        total = float(G["%s.%s.abs" % (prefix, key)])
        assert abs(float(x.astype(np.float64).sum()) - float(G["%s.%s.sum" % (prefix, key)])) <= REL * total, (prefix, key)


def check_post(G, prefix, module, grad_prefix=None):
    """Every recorded tensor of `module` after the update; `grad_prefix`: the reference gradients of the same keys decide
    which sampled entries get the tight tolerance."""
    n = 0
    for k, v in module.state_dict().items():
        if "%s.%s.at" % (prefix, k) not in G.files:
            continue
        big = None
        gk = "%s.%s.at" % (grad_prefix, k) if grad_prefix else None
        if gk in G.files:
            big = np.abs(G[gk]) > 1e-6
        check_tensor(G, prefix, k, v, False, big)
        n += 1
    assert n >= 6, (prefix, n)


def check_grads(G, prefix, named_grads):
    n = 0
    for k, g in named_grads.items():
        assert "%s.%s.at" % (prefix, k) in G.files, (prefix, k)
        check_tensor(G, prefix, k, g, True)
        n += 1
    assert n >= 6, (prefix, n)


TWIN = {"linear1": ("W1", "b1", 0), "linear4": ("W1", "b1", 1), "linear2": ("W2", "b2", 0), "linear5": ("W2", "b2", 1),
        "linear3": ("W3", "b3", 0), "linear6": ("W3", "b3", 1)}


def flat_grads_twin(flat):
    """{module parameter name: gradient view} of a twin-critic FlatNet (fast_update.flatten_twin_q)."""
    out = {}
    for lin, (w, b, head) in TWIN.items():
        out[lin + ".weight"] = flat.g[w][head]
        out[lin + ".bias"] = flat.g[b][head]
    return out


def flat_grads_policy(flat, gaussian):
    out = {"linear1.weight": flat.g["W1"][0], "linear1.bias": flat.g["b1"][0], "linear2.weight": flat.g["W2"][0],
           "linear2.bias": flat.g["b2"][0]}
    if gaussian:
        out.update({"mean_linear.weight": flat.g["W3"][0, 0:2], "log_std_linear.weight": flat.g["W3"][0, 2:4],
                    "mean_linear.bias": flat.g["b3"][0, 0:2], "log_std_linear.bias": flat.g["b3"][0, 2:4]})
    else:
        out.update({"mean.weight": flat.g["W3"][0], "mean.bias": flat.g["b3"][0], "log_std": flat.g["log_std"]})
    return out


def check_acting(G, task, risk, rec):
    """One acting pass at 4096 rows against the reference's policy.sample / get_value / recovery policy.sample."""
    for name, got, ref in (("task_action", task, G["act.task_action"]), ("risk", risk, G["act.risk"]),
                           ("rec_action", rec, G["act.rec_action"])):
        if got is None:
            continue
        got = got.detach().float().cpu().numpy().reshape(ref.shape)
        assert np.allclose(got, ref, rtol=REL, atol=1e-5), (name, float(np.abs(got - ref).max()))
    rows = G["act.rows"]
    assert np.allclose(G["act.rows_out"][:, 0:2], G["act.task_action"][rows], rtol=1e-5, atol=1e-6)   # fixture self-check
    assert np.allclose(G["act.rows_out"][:, 2], G["act.risk"][rows], rtol=1e-5, atol=1e-6)
    assert np.allclose(G["act.rows_out"][:, 3:5], G["act.rec_action"][rows], rtol=1e-5, atol=1e-6)
