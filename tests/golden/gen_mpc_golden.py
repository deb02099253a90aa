"""Generate golden fixtures G7 (PtModel forward / decays / input stats, TS-infinity index map,
MPC._compile_cost) and G8 (one CEM iteration with injected samples) by IMPORTING the reference
(config/navigation2.py, recovery_rl/MPC.py, recovery_rl/optimizers.py) in this container.

Run: python tests/golden/gen_mpc_golden.py -> tests/golden/mpc_golden.npz (data only).
Noise is injected by replacing torch.randn_like / scipy's truncnorm.rvs from outside.
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


def ref_args(extra=()):
    import arg_utils
    argv = sys.argv
    sys.argv = ["rrl_main", "--env-name", "navigation2", "--hidden_size", "16"] + list(extra)
    try:
        return arg_utils.get_args()
    finally:
        sys.argv = argv


def main():
    out = {}
    rng = np.random.RandomState(77)
    np.random.seed(5)
    torch.manual_seed(5)
    from env.make_utils import register_env
    register_env("navigation2")
    from config import create_config
    from dotmap import DotMap
    from recovery_rl.MPC import MPC
    import recovery_rl.optimizers as ro
    from recovery_rl.sac import SAC

    with contextlib.redirect_stdout(io.StringIO()):
        cfg = create_config("navigation2", "MPC", DotMap(), [], "/tmp")
        mpc = MPC(cfg.ctrl_cfg)
    model = mpc.model

    # ---------------- G7a: PtModel ----------------
    for k, v in model.state_dict().items():
        out["pt." + k] = v.detach().numpy().copy()
    data = rng.randn(500, 4) * [20, 5, 0.6, 0.6] + [-30, 0, 0, 0]
    data[:, 3] = 0.25                                  # a constant column: sigma < 1e-12 -> 1
    model.fit_input_stats(data)
    out["pt.data"] = data
    out["pt.fit_mu"] = model.inputs_mu.detach().numpy().copy()
    out["pt.fit_sigma"] = model.inputs_sigma.detach().numpy().copy()
    x = torch.tensor(rng.randn(5, 7, 4) * [20, 5, 0.6, 0.6] + [-30, 0, 0, 0.25], dtype=torch.float32)
    mean, var = model(x)
    _, logvar = model(x, ret_logvar=True)
    out["pt.x"], out["pt.mean"], out["pt.var"], out["pt.logvar"] = (t.detach().numpy() for t in (x, mean, var, logvar))
    out["pt.decays"] = np.array(model.compute_decays().item())
    assert mpc.npart == 20 and mpc.plan_hor == 5 and model.num_nets == 5
    out["mpc.init_var"] = mpc.init_var
    out["mpc.prev_sol"] = mpc.prev_sol

    # ---------------- G7b: TS-infinity index map ----------------
    rows = 6 * 20
    mat = torch.arange(rows * 3, dtype=torch.float32).reshape(rows, 3)
    exp = mpc._expand_to_ts_format(mat)
    out["ts.expanded"] = exp.numpy()
    assert torch.equal(mpc._flatten_to_matrix(exp), mat)

    # ---------------- G7c: _compile_cost with a Q_risk value function and injected noise ----------------
    from env.navigation2 import Navigation2
    env = Navigation2()
    args = ref_args(["--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2"])
    agent = SAC(env.observation_space, env.action_space, args, "/tmp", tmp_env=None)
    for n_, p in agent.safety_critic.safety_critic.named_parameters():
        if n_.endswith("bias") and "bn" not in n_:
            p.data.uniform_(-0.3, 0.3)
    for k, v in agent.safety_critic.safety_critic.state_dict().items():
        out["cc.qrisk." + k] = v.detach().numpy().copy()
    mpc.update_value_func(agent.safety_critic)
    nopt = 6
    ac_seqs = rng.uniform(-1, 1, (nopt, 10)).astype(np.float32)
    cur_obs = np.array([-33.0, 2.5])
    noises = [torch.tensor(rng.randn(5, nopt * 20 // 5, 2), dtype=torch.float32) for _ in range(5)]
    real = torch.randn_like
    q = list(noises)
    torch.randn_like = lambda t, **k: q.pop(0)
    try:
        mpc.sy_cur_obs = cur_obs
        costs = mpc._compile_cost(ac_seqs)
    finally:
        torch.randn_like = real
    out["cc.ac_seqs"], out["cc.cur_obs"], out["cc.costs"] = ac_seqs, cur_obs, costs
    out["cc.noise"] = np.stack([n.numpy() for n in noises])

    # ---------------- G8: one CEM iteration ----------------
    pop, dim, ne, alpha = 400, 10, 40, 0.1
    lb, ub = -np.ones(dim), np.ones(dim)
    target = rng.uniform(-0.8, 0.8, dim)
    rec = {"var": [], "mean": []}

    class NpProxy:
        def __getattr__(self, k):
            return getattr(np, k)

        def var(self, *a, **k):
            v = np.var(*a, **k)
            rec["var"].append(v)
            return v

    ro.np = NpProxy()
    for case, init_mean in (("mid", np.zeros(dim)), ("edge", np.r_[0.97, -0.99, np.zeros(dim - 2)])):
        init_var = np.full(dim, 0.25)
        z = np.clip(rng.randn(pop, dim), -2, 2)
        seen = {}

        def cost_fn(samples):
            seen["samples"] = samples.copy()
            seen["costs"] = ((samples - target.astype(np.float32)) ** 2).sum(1)
            return seen["costs"]

        class FakeX:
            def rvs(self, size):
                assert tuple(size) == (pop, dim)
                return z

        real_tn = ro.stats.truncnorm
        ro.stats.truncnorm = lambda *a, **k: FakeX()
        try:
            opt = ro.CEMOptimizer(dim, 1, pop, ne, cost_fn, upper_bound=ub, lower_bound=lb, alpha=alpha)
            rec["var"].clear()
            new_mean = opt.obtain_solution(init_mean, init_var)
        finally:
            ro.stats.truncnorm = real_tn
        new_var = alpha * init_var + (1 - alpha) * rec["var"][0]
        pre = "cem." + case + "."
        out[pre + "init_mean"], out[pre + "init_var"], out[pre + "z"] = init_mean, init_var, z
        out[pre + "samples"], out[pre + "costs"] = seen["samples"], seen["costs"].astype(np.float32)
        out[pre + "elite_idx"] = np.argsort(seen["costs"])[:ne]
        out[pre + "new_mean"], out[pre + "new_var"] = new_mean, new_var
    out["cem.target"], out["cem.lb"], out["cem.ub"] = target, lb, ub
    out["cem.alpha"], out["cem.num_elites"] = np.array(alpha), np.array(ne)
    np.savez_compressed(os.path.join(HERE, "mpc_golden.npz"), **out)
    print("wrote", len(out), "arrays; costs", costs)


if __name__ == "__main__":
    main()
