"""MB-recovery configuration (reference: config/default.py:15-119 + config/<env>.py ConfigModules).

The reference builds a DotMap tree whose `ctrl_args` / `overrides` are parsed but never applied
(config/default.py has no apply step); the constants it ends up with are reproduced here:
5 nets (default.py:91), TSinf with 20 particles (:108-109), CEM popsize 400 / 40 elites / 5
iterations / alpha 0.1 (config/navigation1.py:121-126), plan horizon 5 (nav) / 15 (maze),
5 training epochs (:116), Adam lr 1e-3 for the ensemble (:167).
"""
import torch

from .ensemble import PtModel  # noqa: F401


class AttrDict(dict):
    """dict with attribute access and `.get` -- the part of DotMap the controller uses."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


ENV_CONSTANTS = {  # config/navigation1.py:105-111, config/navigation2.py, config/maze.py:105-112
    "navigation1": dict(TASK_HORIZON=100, PLAN_HOR=5, MODEL_IN=4, MODEL_OUT=2, AC_COST=0.0),
    "navigation2": dict(TASK_HORIZON=100, PLAN_HOR=5, MODEL_IN=4, MODEL_OUT=2, AC_COST=0.0),
    "maze": dict(TASK_HORIZON=150, PLAN_HOR=15, MODEL_IN=4, MODEL_OUT=2, AC_COST=0.01),
}
NN_TRAIN_CFG = {"epochs": 5}
OPT_CFG = {"CEM": {"popsize": 400, "num_elites": 40, "max_iters": 5, "alpha": 0.1}}


def obs_postproc(obs, pred):
    return obs + pred                       # config/navigation1.py:131-133


def targ_proc(obs, next_obs):
    return next_obs - obs                   # config/navigation1.py:135-137


def nn_constructor_for(env_name, device):
    c = ENV_CONSTANTS[env_name]

    def nn_constructor(model_init_cfg):
        """config/navigation1.py:155-169: ensemble with 2x outputs (mean + log-variance), Adam 1e-3."""
        if model_init_cfg.get("load_model", False):
            raise AssertionError('Has yet to support loading model')
        model = PtModel(model_init_cfg["num_nets"], c["MODEL_IN"], c["MODEL_OUT"] * 2).to(device)
        model.optim = torch.optim.Adam(model.parameters(), lr=0.001,
                                       capturable=torch.device(device).type == "cuda", foreach=True)
        return model
    return nn_constructor


def create_config(env_name, ctrl_type, ctrl_args, overrides, logdir, env=None):
    """Counterpart of config/default.py:15-51,64-119.  `env` is the (vector) env the controller
    plans for; the reference instantiates a second env object for this (config/navigation1.py:114)."""
    assert ctrl_type == 'MPC'
    c = ENV_CONSTANTS[env_name]
    device = env.device if env is not None else "cuda"
    cfg = AttrDict()
    cfg.exp_cfg = AttrDict(sim_cfg=AttrDict(env=env, task_hor=c["TASK_HORIZON"]),
                           log_cfg=AttrDict(logdir=logdir))
    ac = c["AC_COST"]
    cfg.ctrl_cfg = AttrDict(
        env=env,
        prop_cfg=AttrDict(
            model_init_cfg=AttrDict(num_nets=5, model_constructor=nn_constructor_for(env_name, device)),
            model_train_cfg=dict(NN_TRAIN_CFG), mode="TSinf", npart=20,
            obs_postproc=obs_postproc, targ_proc=targ_proc),
        opt_cfg=AttrDict(mode="CEM", plan_hor=c["PLAN_HOR"], cfg=dict(OPT_CFG["CEM"]),
                         obs_cost_fn=lambda obs: obs.norm(dim=1),          # never evaluated (MPC.py:147-151)
                         ac_cost_fn=lambda acs: ac * (acs ** 2).sum(dim=1)),
        log_cfg=AttrDict())
    return cfg
