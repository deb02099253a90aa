"""Known answers for the planner kernel at its PRODUCTION shape, produced by IMPORTING the reference in this container:
MPC._compile_cost (recovery_rl/MPC.py:374-416, _predict_next_obs :421-439, config/navigation2.py PtModel.forward :71-96,
QRiskWrapper.get_value recovery_rl/qrisk.py:184-196) at the default --hidden_size 256 with the 5 x 200 ensemble, 400
candidates x 20 particles x 5 steps, for three planning problems.

Weights, candidates, observations and the particle noise are re-created from seeded numpy streams
(kat256_plan_inputs.py); the noise reaches the reference through torch.randn_like, in the TS-infinity layout the reference's
own _expand_to_ts_format gives the flat rows.  The fixture holds what the reference PRODUCED: costs [3, 400], the fitted
input statistics, and the safety critic's value on sampled particle rows at every step (what the rollout passed through).

Run: python tests/golden/gen_mpc_golden_256.py  ->  tests/golden/mpc_golden_256.npz (numbers only).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402

import kat256_plan_inputs as P  # noqa: E402


def ref_args():
    import arg_utils
    argv = sys.argv
    sys.argv = ["rrl_main", "--env-name", "navigation2"] + P.ARGV
    try:
        return arg_utils.get_args()
    finally:
        sys.argv = argv


def main():
    out = {}
    np.random.seed(5)
    torch.manual_seed(5)
    from env.make_utils import register_env
    register_env("navigation2")
    from config import create_config
    from dotmap import DotMap
    from env.navigation2 import Navigation2
    from recovery_rl.MPC import MPC
    from recovery_rl.sac import SAC

    with contextlib.redirect_stdout(io.StringIO()):
        cfg = create_config("navigation2", "MPC", DotMap(), [], "/tmp")
        mpc = MPC(cfg.ctrl_cfg)
    model = mpc.model
    assert (mpc.npart, mpc.plan_hor, model.num_nets, mpc.optimizer.popsize) == (P.NPART, P.PLAN_HOR, P.NETS, P.POP)
    assert tuple(model.lin1_w.shape) == (P.NETS, P.HE, P.HE)
    for k, v in P.ensemble_weights().items():
        getattr(model, k).data = torch.as_tensor(v)
    model.fit_input_stats(P.stats_data())
    out["fit_mu"] = model.inputs_mu.detach().numpy().copy()
    out["fit_sigma"] = model.inputs_sigma.detach().numpy().copy()

    args = ref_args()
    assert args.hidden_size == P.HQ
    env = Navigation2()
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    net = agent.safety_critic.safety_critic
    sd = net.state_dict()
    for k, v in P.qrisk_weights(sd).items():
        sd[k] = torch.as_tensor(v)
    net.load_state_dict(sd, strict=True)
    assert tuple(net.linear2.weight.shape) == (P.HQ, P.HQ)
    mpc.update_value_func(agent.safety_critic)

    acs, noise, rows = P.candidates(), P.noise(), P.q_sample_rows()
    costs, q_steps = [], []
    real_randn, real_value = torch.randn_like, agent.safety_critic.get_value
    for m, cur_obs in enumerate(P.CUR_OBS):
        feed = [mpc._expand_to_ts_format(torch.as_tensor(noise[t, m])) for t in range(P.PLAN_HOR)]
        seen = []

        def randn_like(t, **k):
            z = feed.pop(0)
            assert z.shape == t.shape
            return z

        def get_value(states, actions, **k):
            v = real_value(states, actions, **k)
            seen.append(v.detach().numpy().reshape(-1)[rows].copy())
            return v
        torch.randn_like, agent.safety_critic.get_value = randn_like, get_value
        try:
            mpc.sy_cur_obs = cur_obs
            costs.append(mpc._compile_cost(acs[m]))
        finally:
            torch.randn_like, agent.safety_critic.get_value = real_randn, real_value
        assert not feed and len(seen) == P.PLAN_HOR
        q_steps.append(np.stack(seen))
    out["costs"] = np.stack(costs).astype(np.float32)               # [3, 400]
    out["q_steps"] = np.stack(q_steps).astype(np.float32)           # [3, 5, 96]
    path = os.path.join(HERE, "mpc_golden_256.npz")
    np.savez_compressed(path, **out)
    print("wrote", len(out), "arrays,", os.path.getsize(path), "bytes")
    print("costs: mean", out["costs"].mean(1), "std", out["costs"].std(1), "min/max", out["costs"].min(), out["costs"].max())
    print("q at t=0 / t=4 (std over rows):", out["q_steps"][:, 0].std(1), out["q_steps"][:, 4].std(1))


if __name__ == "__main__":
    main()
