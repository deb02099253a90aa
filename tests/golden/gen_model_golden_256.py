"""Known answers at the PRODUCTION shape (hidden 256, batch 256: arg_utils.py:77,89 defaults; 4096 acting rows), produced by
IMPORTING the reference (recovery_rl/sac.py:170-277, qrisk.py:86-182, model.py, experiment.py:546-577) in this container:

  * one SAC update followed by one Q_risk + recovery-policy update on the same agent (the order of one iteration of
    experiment.py:397-416), with the replay batch and the policy noise injected;
  * one acting pass: task action, Q_risk(s, a_task) and the model-free recovery action for 4096 observations.

Weights, batch, noise and observations are re-created from seeded numpy streams (kat256_inputs.py), so the fixture holds only
what the reference PRODUCED: the five returned scalars, and per tensor of every network after the update -- and of every
gradient the optimisers saw -- the sum, the sum of magnitudes and 192 sampled entries.

Run: python tests/golden/gen_model_golden_256.py  ->  tests/golden/model_golden_256.npz (numbers only).
Harness patches as in gen_model_golden.py: (b) the critic step deferred behind policy_loss.backward(), (c) float32 log_std.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402

import kat256_inputs as K  # noqa: E402
from gen_model_golden import NoiseFeed, StubMemory  # noqa: E402


def ref_args():
    import arg_utils
    argv = sys.argv
    sys.argv = ["rrl_main", "--env-name", "navigation1", "--hidden_size", str(K.H), "--batch_size", str(K.B)] + K.ARGV
    try:
        return arg_utils.get_args()
    finally:
        sys.argv = argv


def load(module, tag):
    sd = module.state_dict()
    for k, v in K.weights(sd, tag).items():
        sd[k] = torch.as_tensor(v)
    module.load_state_dict(sd, strict=True)


def record(out, prefix, named):
    """sum, sum of magnitudes and sampled entries of every float tensor of `named` ({key: tensor})."""
    for k, v in named.items():
        if not torch.is_floating_point(v) or v.numel() == 0:
            continue
        x = v.detach().cpu().numpy().astype(np.float32).ravel()
        idx = K.sample_index(prefix + "." + k, x.size)
        out["%s.%s.sum" % (prefix, k)] = np.float64(x.astype(np.float64).sum())
        out["%s.%s.abs" % (prefix, k)] = np.float64(np.abs(x.astype(np.float64)).sum())
        out["%s.%s.at" % (prefix, k)] = x[idx].copy()


def grads_of(module):
    return {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}


def main():
    from env.navigation1 import Navigation1
    from recovery_rl.sac import SAC
    out = {}
    env = Navigation1()
    args = ref_args()
    assert args.hidden_size == 256 and args.batch_size == 256
    torch.manual_seed(1)
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    qr = agent.safety_critic
    qr.policy.log_std.data = qr.policy.log_std.data.float()                       # (c)
    load(agent.critic, "critic"); load(agent.policy, "policy")
    load(qr.safety_critic, "qrisk"); load(qr.policy, "recpolicy")
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    qr.safety_critic_target.load_state_dict(qr.safety_critic.state_dict())
    (s, a, r, s2, m), c, eps_next, eps_pi = K.batch()
    t = torch.as_tensor

    # ---- one SAC update (sac.py:170-277), patch (b) -----------------------------------------------------------------
    real_cstep, real_pstep = agent.critic_optim.step, agent.policy_optim.step
    snap = {}

    def deferred_cstep():
        snap["critic"] = grads_of(agent.critic)

    def pstep_then_critic():
        snap["policy"] = grads_of(agent.policy)
        real_pstep()
        for k, p in agent.critic.named_parameters():
            p.grad = snap["critic"][k]
        real_cstep()
    agent.critic_optim.step, agent.policy_optim.step = deferred_cstep, pstep_then_critic
    with NoiseFeed([t(eps_next), t(eps_pi)]):
        res = agent.update_parameters(StubMemory((s, a, r, s2, m)), K.B, 0, nu=args.nu, safety_critic=qr)
    out["sac.returns"] = np.array(res, dtype=np.float64)
    record(out, "sac.grad.critic", snap["critic"]); record(out, "sac.grad.policy", snap["policy"])
    record(out, "sac.post.critic", agent.critic.state_dict())
    record(out, "sac.post.critic_target", agent.critic_target.state_dict())
    record(out, "sac.post.policy", agent.policy.state_dict())

    # ---- one Q_risk + recovery-policy update (qrisk.py:86-182) with the UPDATED task policy ---------------------------------
    real_q, real_p = qr.safety_critic_optim.step, qr.policy_optim.step

    def qstep():
        snap["qrisk"] = grads_of(qr.safety_critic)
        real_q()

    def rstep():
        snap["recpolicy"] = grads_of(qr.policy)
        real_p()
    qr.safety_critic_optim.step, qr.policy_optim.step = qstep, rstep
    with NoiseFeed([t(eps_next), t(eps_pi)]):
        qr.update_parameters(memory=StubMemory((s, a, c, s2, m)), policy=agent.policy, batch_size=K.B)
    record(out, "mf.grad.qrisk", snap["qrisk"]); record(out, "mf.grad.recpolicy", snap["recpolicy"])
    record(out, "mf.post.qrisk", qr.safety_critic.state_dict())
    record(out, "mf.post.qrisk_target", qr.safety_critic_target.state_dict())
    record(out, "mf.post.recpolicy", qr.policy.state_dict())
    out["mf.get_value"] = qr.get_value(t(s), t(a)).numpy().ravel()              # on the updated nets, all 256 rows

    # ---- one acting pass at 4096 rows (experiment.py:546-577 -> sac.py:133-168, qrisk.py:184-213) on the updated nets -------
    obs, noise = K.acting()
    with torch.no_grad():
        with NoiseFeed([t(noise[0])]):
            task, _, _ = agent.policy.sample(t(obs))
        risk = qr.get_value(t(obs), task)
        with NoiseFeed([t(noise[1])]):
            rec, _, _ = qr.policy.sample(t(obs))
    out["act.task_action"], out["act.risk"], out["act.rec_action"] = task.numpy(), risk.numpy().ravel(), rec.numpy()
    # a handful of rows through the reference's per-state entry points (what get_action calls), to pin the batched use above
    rows = [0, 1, 777, 4095]
    one = []
    for i in rows:
        with NoiseFeed([t(noise[0][i:i + 1])]):
            a_i = agent.select_action(obs[i])
        v_i = float(qr.get_value(t(obs[i]).unsqueeze(0), t(a_i).unsqueeze(0)))
        with NoiseFeed([t(noise[1][i:i + 1])]):
            r_i = qr.select_action(obs[i])
        one.append(np.concatenate([a_i, [v_i], r_i]))
    out["act.rows"], out["act.rows_out"] = np.array(rows), np.array(one, dtype=np.float32)
    assert np.allclose(out["act.rows_out"][:, 0:2], out["act.task_action"][rows], rtol=1e-5, atol=1e-6)
    np.savez_compressed(os.path.join(HERE, "model_golden_256.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "model_golden_256.npz")), "bytes")
    print("returns", out["sac.returns"], "risk quantiles", np.quantile(out["act.risk"], [0.1, 0.5, 0.9]))


if __name__ == "__main__":
    main()
