"""Generate golden fixtures G3 (MLP forward/sample KATs), G4 (one SAC update, one Q_risk update)
and G5 (soft update, via the post-step target nets) by IMPORTING the reference
(recovery_rl/model.py, sac.py, qrisk.py) in this container.

Run: python tests/golden/gen_model_golden.py  ->  tests/golden/model_golden.npz (data only:
weights, inputs, injected noise, outputs).

Harness patches, applied from outside, reference files untouched:
  (b) critic_optim.step() is deferred until after policy_loss.backward(), using a snapshot of
      the critic gradients taken at the original call site (torch>=1.5 rejects the reference's
      order; this is the mathematically intended update);
  (c) StochasticPolicy.log_std is cast to float32.
Policy noise is injected by replacing torch.distributions.normal._standard_normal.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402
import torch.distributions.normal as tdn  # noqa: E402

H, B = 16, 8


def ref_args(extra=()):
    import arg_utils
    argv = sys.argv
    sys.argv = ["rrl_main", "--env-name", "navigation1", "--hidden_size", str(H)] + list(extra)
    try:
        return arg_utils.get_args()
    finally:
        sys.argv = argv


class NoiseFeed:
    """Replaces _standard_normal with a queue of preset tensors."""

    def __init__(self, tensors):
        self.q = list(tensors)
        self.real = tdn._standard_normal

    def __enter__(self):
        def fake(shape, dtype, device):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t.to(dtype)
        tdn._standard_normal = fake
        return self

    def __exit__(self, *a):
        tdn._standard_normal = self.real
        assert not self.q, "unused noise"


def sd(prefix, module, out):
    for k, v in module.state_dict().items():
        out[prefix + "." + k] = v.detach().cpu().numpy().copy()


class StubMemory:
    def __init__(self, batch):
        self.batch = batch

    def sample(self, batch_size, pos_fraction=None):
        return tuple(np.array(x) for x in self.batch)

    def __len__(self):
        return 10 ** 6


def main():
    from env.navigation1 import Navigation1
    from recovery_rl.model import GaussianPolicy, QNetwork, QNetworkConstraint, StochasticPolicy, DeterministicPolicy
    from recovery_rl.sac import SAC
    out = {}
    rng = np.random.RandomState(123)
    env = Navigation1()

    # ---------------- G3: forward / sample KATs ----------------
    torch.manual_seed(5)
    s = torch.tensor(rng.randn(B, 2) * [20, 3] + [-30, 0], dtype=torch.float32)
    a = torch.tensor(rng.uniform(-1, 1, (B, 2)), dtype=torch.float32)
    eps = torch.tensor(rng.randn(B, 2), dtype=torch.float32)
    out["g3.s"], out["g3.a"], out["g3.eps"] = s.numpy(), a.numpy(), eps.numpy()
    q = QNetwork(2, 2, H)
    qc = QNetworkConstraint(2, 2, H)
    gp = GaussianPolicy(2, 2, H, env.action_space)
    sp = StochasticPolicy(2, 2, H, env.action_space)
    sp.log_std.data = sp.log_std.data.float()                      # patch (c)
    dp = DeterministicPolicy(2, 2, H, env.action_space)
    for m in (q, qc, gp, sp, dp):                                   # non-trivial biases
        for n_, p in m.named_parameters():
            if n_.endswith("bias"):
                p.data.uniform_(-0.3, 0.3)
    sd("g3.q", q, out); sd("g3.qc", qc, out); sd("g3.gp", gp, out); sd("g3.sp", sp, out); sd("g3.dp", dp, out)
    q1, q2 = q(s, a)
    out["g3.q.out1"], out["g3.q.out2"] = q1.detach().numpy(), q2.detach().numpy()
    c1, c2 = qc(s, a)
    out["g3.qc.out1"], out["g3.qc.out2"] = c1.detach().numpy(), c2.detach().numpy()
    with NoiseFeed([eps]):
        act, logp, mean = gp.sample(s)
    out["g3.gp.action"], out["g3.gp.logp"], out["g3.gp.mean"] = (x.detach().numpy() for x in (act, logp, mean))
    with NoiseFeed([eps]):
        act, logp, mean = sp.sample(s)
    out["g3.sp.action"], out["g3.sp.logp"], out["g3.sp.mean"] = (x.detach().numpy() for x in (act, logp, mean))
    out["g3.dp.mean"] = dp(s).detach().numpy()

    # ---------------- G4: one SAC update + one Q_risk update ----------------
    variants = {
        "sac": [],
        "sac_autoent": ["--automatic_entropy_tuning", "1"],
        "sac_dgd": ["--DGD_constraints", "--nu", "50", "--update_nu", "--gamma_safe", "0.8", "--eps_safe", "0.3"],
        "sac_rcpo": ["--RCPO", "--lambda_RCPO", "10", "--gamma_safe", "0.8", "--eps_safe", "0.3"],
        "mf": ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3"],
    }
    batch = (rng.randn(B, 2) * [20, 3] + [-30, 0], rng.uniform(-1, 1, (B, 2)),
             -np.abs(rng.randn(B)) * 30, rng.randn(B, 2) * [20, 3] + [-30, 0],
             (rng.uniform(size=B) < 0.8).astype(np.float64))
    cbatch = (batch[0], batch[1], (rng.uniform(size=B) < 0.4).astype(np.float64), batch[3], batch[4])
    for i, name in enumerate(("s", "a", "r", "s2", "m")):
        out["g4.batch." + name] = np.asarray(batch[i], dtype=np.float32)
    out["g4.cbatch.c"] = np.asarray(cbatch[2], dtype=np.float32)
    eps_next = torch.tensor(rng.randn(B, 2), dtype=torch.float32)
    eps_pi = torch.tensor(rng.randn(B, 2), dtype=torch.float32)
    out["g4.eps_next"], out["g4.eps_pi"] = eps_next.numpy(), eps_pi.numpy()

    for name, extra in variants.items():
        args = ref_args(extra)
        torch.manual_seed(11)
        agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
        agent.safety_critic.policy.log_std.data = agent.safety_critic.policy.log_std.data.float()   # (c)
        for mod in (agent.critic, agent.policy, agent.safety_critic.safety_critic, agent.safety_critic.policy):
            for n_, p in mod.named_parameters():
                if n_.endswith("bias") and "bn" not in n_:
                    p.data.uniform_(-0.2, 0.2)
        agent.critic_target.load_state_dict(agent.critic.state_dict())
        agent.safety_critic.safety_critic_target.load_state_dict(agent.safety_critic.safety_critic.state_dict())
        pre = name + ".pre"
        sd(pre + ".critic", agent.critic, out); sd(pre + ".policy", agent.policy, out)
        sd(pre + ".qrisk", agent.safety_critic.safety_critic, out)
        sd(pre + ".recpolicy", agent.safety_critic.policy, out)

        if name.startswith("sac"):
            # patch (b)
            real_cstep, real_pstep = agent.critic_optim.step, agent.policy_optim.step
            snap = {}

            def deferred_cstep():
                snap["g"] = [p.grad.clone() for p in agent.critic.parameters()]

            def pstep_then_critic():
                real_pstep()
                for p, g in zip(agent.critic.parameters(), snap["g"]):
                    p.grad = g
                real_cstep()

            agent.critic_optim.step = deferred_cstep
            agent.policy_optim.step = pstep_then_critic
            with NoiseFeed([eps_next, eps_pi]):
                res = agent.update_parameters(StubMemory(batch), B, 0, nu=args.nu,
                                              safety_critic=agent.safety_critic)
            out[name + ".returns"] = np.array(res, dtype=np.float64)
            post = name + ".post"
            sd(post + ".critic", agent.critic, out); sd(post + ".critic_target", agent.critic_target, out)
            sd(post + ".policy", agent.policy, out)
            if agent.automatic_entropy_tuning:
                out[post + ".log_alpha"] = agent.log_alpha.detach().numpy().copy()
            out[post + ".log_nu"] = np.array(agent.log_nu.item())
            out[post + ".log_lambda"] = np.array(agent.log_lambda_RCPO.item())
        else:
            with NoiseFeed([eps_next, eps_pi]):
                agent.safety_critic.update_parameters(memory=StubMemory(cbatch), policy=agent.policy,
                                                      batch_size=B)
            post = name + ".post"
            sd(post + ".qrisk", agent.safety_critic.safety_critic, out)
            sd(post + ".qrisk_target", agent.safety_critic.safety_critic_target, out)
            sd(post + ".recpolicy", agent.safety_critic.policy, out)
            # get_value / __call__ on the updated nets
            st = torch.tensor(batch[0], dtype=torch.float32)
            ac = torch.tensor(batch[1], dtype=torch.float32)
            out[name + ".get_value"] = agent.safety_critic.get_value(st, ac).numpy()
        out[name + ".argv"] = np.array(" ".join(extra))
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
