"""Known-answer fixtures for the two sampling-based action selectors, captured by IMPORTING the reference:
  * SQRL constraint sampling, SAC.select_action with --use_constraint_sampling (recovery_rl/sac.py:139-161)
  * Q-sampling recovery, QRiskWrapper.select_action with --Q_sampling_recovery (recovery_rl/qrisk.py:214-225)
with every random input injected from outside (policy noise, the categorical draw, the candidate actions).

Run: python tests/golden/gen_select_golden.py -> tests/golden/select_golden.npz (data only).
Note on sac.py:150-158: the reference draws `sampled_idx` from a Categorical over the SAFE candidates only and then
indexes the FULL candidate list with it (`pi[sampled_idx]`, not `pi[thresh_idxs[sampled_idx]]`); the fixture records
what the reference returns, i.e. that behaviour.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402

from gen_model_golden import NoiseFeed, ref_args, sd  # noqa: E402


def main():
    from env.navigation1 import Navigation1
    from recovery_rl.sac import SAC
    out = {}
    rng = np.random.RandomState(321)
    env = Navigation1()
    torch.manual_seed(9)

    # ---- SQRL (sac.py:139-161) ----
    args = ref_args(["--DGD_constraints", "--use_constraint_sampling", "--eps_safe", "0.5"])
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    for net in (agent.policy, agent.safety_critic.safety_critic):
        for n_, p in net.named_parameters():
            if n_.endswith("bias") and "bn" not in n_:
                p.data.uniform_(-0.5, 0.5)
    sd("sqrl.policy", agent.policy, out)
    sd("sqrl.qrisk", agent.safety_critic.safety_critic, out)
    cases = []
    states = [np.array([-30.0, 1.0]), np.array([-3.0, 4.5]), np.array([-60.0, -2.0]), np.array([10.0, 0.3])]
    real_sample = torch.distributions.Categorical.sample
    for k, st in enumerate(states):
        for mode in ("median", "none", "q90", "all"):          # mixed / none safe (argmin branch) / mixed / all safe
            noise = torch.tensor(rng.randn(100, 2), dtype=torch.float32)
            with torch.no_grad(), NoiseFeed([noise]):            # dry pass: where do the Q_risk values of these samples lie
                sb = torch.FloatTensor(st).unsqueeze(0).repeat(100, 1)
                pi_dry, _, _ = agent.policy.sample(sb)
                q_dry = agent.safety_critic.get_value(sb, pi_dry).numpy().ravel()
            eps_safe = {"median": float(np.median(q_dry)), "none": 1e-9, "q90": float(np.quantile(q_dry, 0.9)),
                        "all": 2.0}[mode]
            agent.eps_safe = eps_safe
            u = float(rng.uniform())
            seen = {}

            def fake_sample(self, sample_shape=torch.Size()):
                probs = self.probs.detach().numpy().astype(np.float64)
                seen["probs"] = probs.copy()
                idx = int(np.searchsorted(np.cumsum(probs), u * probs.sum(), side="right"))
                seen["idx"] = min(idx, len(probs) - 1)
                return torch.tensor(seen["idx"])
            torch.distributions.Categorical.sample = fake_sample
            try:
                with NoiseFeed([noise]):
                    action = agent.select_action(st)
            finally:
                torch.distributions.Categorical.sample = real_sample
            cases.append({"state": st, "eps_safe": eps_safe, "noise": noise.numpy(), "u": u,
                          "n_safe": len(seen.get("probs", [])), "idx": seen.get("idx", -1), "action": action})
    for key in ("state", "eps_safe", "noise", "u", "n_safe", "idx", "action"):
        out["sqrl." + key] = np.array([c[key] for c in cases])
    print("SQRL cases", len(cases), "safe counts", [c["n_safe"] for c in cases])

    # ---- Q-sampling recovery (qrisk.py:214-225) ----
    args = ref_args(["--use_recovery", "--Q_sampling_recovery", "--eps_safe", "0.3"])
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    for n_, p in agent.safety_critic.safety_critic.named_parameters():
        if n_.endswith("bias") and "bn" not in n_:
            p.data.uniform_(-0.5, 0.5)
    sd("qs.qrisk", agent.safety_critic.safety_critic, out)
    qs_states, qs_cands, qs_actions = [], [], []
    for st in states:
        cands = rng.uniform(-1, 1, (1000, 2)).astype(np.float32)
        it = iter(cands)
        agent.safety_critic.ac_space.sample = lambda: next(it)
        action = agent.safety_critic.select_action(st)
        qs_states.append(st), qs_cands.append(cands), qs_actions.append(action)
    out["qs.state"], out["qs.candidates"], out["qs.action"] = np.array(qs_states), np.array(qs_cands), np.array(qs_actions)
    np.savez_compressed(os.path.join(HERE, "select_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
