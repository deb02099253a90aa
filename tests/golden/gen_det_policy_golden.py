"""Golden fixture for `--policy Deterministic`: DeterministicPolicy.forward / .sample
(recovery_rl/model.py:447-485) and ONE SAC.update_parameters step with that policy (recovery_rl/sac.py:115-123,170-277:
alpha = 0, log_pi = 0), captured by IMPORTING the reference in this container.

Run: python tests/golden/gen_det_policy_golden.py  ->  tests/golden/det_policy_golden.npz (data only).

The policy's exploration noise is `self.noise.normal_(0., std=0.1)` on the global torch generator (model.py:476): the
generator is seeded right before each call and the SAME draws are recorded by replaying the seed, so the fixture carries
the noise the reference used.  Harness patch (b) of SURVEY 8c (critic step deferred) as in gen_model_golden.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402

from gen_model_golden import B, H, StubMemory, ref_args, sd  # noqa: E402


def draws(seed, k):
    torch.manual_seed(seed)
    return [torch.Tensor(2).normal_(0., std=0.1).clone() for _ in range(k)]


def main():
    from env.navigation1 import Navigation1
    from recovery_rl.model import DeterministicPolicy
    from recovery_rl.sac import SAC
    out = {}
    rng = np.random.RandomState(321)
    env = Navigation1()
    torch.manual_seed(9)
    s = torch.tensor(rng.randn(B, 2) * [20, 3] + [-30, 0], dtype=torch.float32)
    dp = DeterministicPolicy(2, 2, H, env.action_space)
    for n_, p in dp.named_parameters():
        if n_.endswith("bias"):
            p.data.uniform_(-0.3, 0.3)
    sd("dp", dp, out)
    out["s"] = s.numpy()
    out["dp.forward"] = dp(s).detach().numpy()
    for k, seed in enumerate((3, 4, 77)):           # seed 77: a draw beyond the +-0.25 clamp is unlikely; the clamp is
        torch.manual_seed(seed)                     # exercised separately below with an injected 5-sigma noise
        act, logp, mean = dp.sample(s)
        out["sample%d.noise" % k] = draws(seed, 1)[0].numpy()
        out["sample%d.action" % k], out["sample%d.mean" % k] = act.detach().numpy(), mean.detach().numpy()
        assert float(logp) == 0.0
    big = torch.tensor([0.9, -0.4])                 # what normal_ would have to return to hit the clamp
    real = torch.Tensor.normal_
    torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(big)
    try:
        act, _, mean = dp.sample(s)
    finally:
        torch.Tensor.normal_ = real
    out["clamp.noise"], out["clamp.action"] = big.numpy(), act.detach().numpy()

    # one SAC update with the deterministic policy
    extra = ["--policy", "Deterministic"]
    args = ref_args(extra)
    torch.manual_seed(11)
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    assert isinstance(agent.policy, DeterministicPolicy) and agent.alpha == 0
    for mod in (agent.critic, agent.policy):
        for n_, p in mod.named_parameters():
            if n_.endswith("bias"):
                p.data.uniform_(-0.2, 0.2)
    agent.critic_target.load_state_dict(agent.critic.state_dict())
    sd("upd.pre.critic", agent.critic, out)
    sd("upd.pre.policy", agent.policy, out)
    batch = (rng.randn(B, 2) * [20, 3] + [-30, 0], rng.uniform(-1, 1, (B, 2)), -np.abs(rng.randn(B)) * 30,
             rng.randn(B, 2) * [20, 3] + [-30, 0], (rng.uniform(size=B) < 0.8).astype(np.float64))
    for i, name in enumerate(("s", "a", "r", "s2", "m")):
        out["upd.batch." + name] = np.asarray(batch[i], dtype=np.float32)
    real_cstep, real_pstep = agent.critic_optim.step, agent.policy_optim.step
    snap = {}

    def deferred_cstep():
        snap["g"] = [p.grad.clone() for p in agent.critic.parameters()]

    def pstep_then_critic():
        real_pstep()
        for p, g in zip(agent.critic.parameters(), snap["g"]):
            p.grad = g
        real_cstep()

    agent.critic_optim.step, agent.policy_optim.step = deferred_cstep, pstep_then_critic      # patch (b)
    torch.manual_seed(21)
    res = agent.update_parameters(StubMemory(batch), B, 0, nu=args.nu, safety_critic=agent.safety_critic)
    n_next, n_pi = draws(21, 2)                      # sample(next_state) then sample(state): sac.py:192,217
    out["upd.noise_next"], out["upd.noise_pi"] = n_next.numpy(), n_pi.numpy()
    out["upd.returns"] = np.array(res, dtype=np.float64)
    sd("upd.post.critic", agent.critic, out)
    sd("upd.post.critic_target", agent.critic_target, out)
    sd("upd.post.policy", agent.policy, out)
    out["upd.argv"] = np.array(" ".join(extra))
    # select_action: train (noisy) and eval (mean) (sac.py:133-168)
    st = np.array([-30.0, 1.5])
    torch.manual_seed(31)
    out["select.state"] = st
    out["select.train"] = np.asarray(agent.select_action(st))
    out["select.noise"] = draws(31, 1)[0].numpy()
    out["select.eval"] = np.asarray(agent.select_action(st, eval=True))
    np.savez_compressed(os.path.join(HERE, "det_policy_golden.npz"), **out)
    print("wrote", len(out), "arrays; returns", res)


if __name__ == "__main__":
    main()
