"""Checkpoint / resume of one training process (absent in the reference, whose
`SAC.save_model` (recovery_rl/sac.py:279-290) writes actor/critic weights only and has no caller).

Everything the lock-step loop reads lives on the device, so a checkpoint is the set of those buffers plus a
few host counters: networks and Adam moments (flat buffers of the fused path, or the torch optimisers of the
autograd path), both replay rings (filled rows only), env state and in-kernel RNG ticks, loop counters, the
episode-log accumulators, the model-based controller, and the host/device generator states.  A run resumed
from a checkpoint continues bit-for-bit like the uninterrupted run (tests/test_checkpoint_gpu.py).
"""
import os
import random

import numpy as np
import torch

from .fast_update import FLAT_NETS as _FLAT_NETS

FORMAT = 1

_ENV_FIELDS = ("pos", "t", "obs", "prev_obs", "next_obs", "reward", "action_clipped", "_flags", "tick")
_DUALS = ("alpha", "nu", "lambda_RCPO")


def _cpu(t):
    return t.detach().to("cpu", copy=True)


# -- pieces ------------------------------------------------------------------------------------------
def replay_state(mem):
    size = int(mem.state[1].item())
    out = {"capacity": mem.capacity, "size": size, "state": _cpu(mem.state), "tick": _cpu(mem.tick),
           "len": mem._len, "len_exact": mem._len_exact, "pinned": mem.pinned}
    for name in ("s", "a", "r", "s2", "m"):
        out[name] = _cpu(getattr(mem, name)[:size])
    if mem.pos_cnt is not None:
        out["pos_cnt"] = _cpu(mem.pos_cnt)
        out["abi"] = int(mem.lib.rrl_abi_version())      # the count-table layout belongs to the library version
    return out


def capacity_fits(sd, capacity):
    """A buffer of `capacity` rows can take the checkpointed ring `sd`: same capacity, or a larger one while the ring has
    never reached its end (rows [0, size) and the write position mean the same in both)."""
    return sd["capacity"] == capacity or (capacity > sd["capacity"] and sd["size"] < sd["capacity"])


def _capacity_error(name, sd, capacity):
    return ValueError("%s capacity differs: checkpoint %d, run %d (a run resumed through --resume takes the checkpoint's "
                      "capacities by itself; otherwise pass --replay_size / --safe_replay_size with --keep_replay_size)"
                      % (name, sd["capacity"], capacity))


def peek_capacities(path):
    """{buffer: (capacity, wrapped)} of a checkpoint, read before the run allocates its buffers."""
    sd = torch.load(path, map_location="cpu", weights_only=False, mmap=True)
    return {name: (int(sd[name]["capacity"]), sd[name]["size"] >= sd[name]["capacity"])
            for name in ("memory", "recovery_memory")}


def load_replay_state(mem, sd):
    if not capacity_fits(sd, mem.capacity):
        raise _capacity_error("replay", sd, mem.capacity)
    size = sd["size"]
    for name in ("s", "a", "r", "s2", "m"):
        getattr(mem, name)[:size].copy_(sd[name])
    mem.state.copy_(sd["state"])
    mem.tick.copy_(sd["tick"])
    mem.pin(sd.get("pinned", 0))
    if mem.pos_cnt is not None:
        same = sd.get("abi") == int(mem.lib.rrl_abi_version()) and sd["pos_cnt"].shape == mem.pos_cnt.shape
        if same:
            mem.pos_cnt.copy_(sd["pos_cnt"])
        else:       # written by a library with another table layout: the table is a function of the rows
            mem.rebuild_pos_cnt()
    mem._len, mem._len_exact = sd["len"], sd["len_exact"]


def env_state(env):
    if hasattr(env, "refresh_arrays"):
        env.refresh_arrays()             # the loop may be running on the compact status word: decode it for the record
    return {"num_envs": env.num_envs, "seed": env.seed_value,
            **{f: _cpu(getattr(env, f)) for f in _ENV_FIELDS}}


def load_env_state(env, sd):
    if sd["num_envs"] != env.num_envs:
        raise ValueError("num_envs differs: checkpoint %d, run %d" % (sd["num_envs"], env.num_envs))
    env.seed_value = sd["seed"]
    for f in _ENV_FIELDS:
        getattr(env, f).copy_(sd[f])
    if hasattr(env, "_status_live"):
        env._status_live = False         # the arrays are the record; the loop re-encodes the status word when it needs it


def agent_state(agent):
    qr = agent.safety_critic
    out = {"modules": {"critic": agent.critic.state_dict(), "critic_target": agent.critic_target.state_dict(),
                       "policy": agent.policy.state_dict(), "safety_critic": qr.safety_critic.state_dict(),
                       "safety_critic_target": qr.safety_critic_target.state_dict(),
                       "recovery_policy": qr.policy.state_dict()},
           "updates": (agent.updates, qr.updates)}
    out["duals"] = {name: _cpu(getattr(agent, "log_" + name)) for name in _DUALS if hasattr(agent, "log_" + name)}
    out["modules"] = {k: {n: _cpu(v) for n, v in sd.items()} for k, sd in out["modules"].items()}
    if agent.fast is not None:
        out["flat"] = {name: {"m": _cpu(net.m), "v": _cpu(net.v), "step": _cpu(net.step)}
                       for name, net in ((n, getattr(agent.fast, n)) for n in _FLAT_NETS)}
        out["noise_tick"] = _cpu(agent.fast.noise_tick)
        out["actor_rows"] = agent.fast.actor_rows        # layout of the per-iteration noise fill
    out["optim"] = {k: o.state_dict() for k, o in _optimisers(agent).items()}
    return out


def _optimisers(agent):
    qr = agent.safety_critic
    optims = {"critic": agent.critic_optim, "policy": agent.policy_optim,
              "safety_critic": qr.safety_critic_optim, "recovery_policy": qr.policy_optim}
    for name in _DUALS:
        if hasattr(agent, name + "_optim"):
            optims[name] = getattr(agent, name + "_optim")
    return optims


def load_agent_state(agent, sd):
    qr = agent.safety_critic
    mods = {"critic": agent.critic, "critic_target": agent.critic_target, "policy": agent.policy,
            "safety_critic": qr.safety_critic, "safety_critic_target": qr.safety_critic_target,
            "recovery_policy": qr.policy}
    for k, m in mods.items():
        m.load_state_dict(sd["modules"][k])            # in place: parameters stay views of the flat buffers
    agent.updates, qr.updates = sd["updates"]
    for name, val in sd["duals"].items():
        log_param = getattr(agent, "log_" + name)
        with torch.no_grad():
            log_param.copy_(val)
        learned = {"alpha": agent.automatic_entropy_tuning, "nu": agent.update_nu, "lambda_RCPO": agent.RCPO}
        if learned[name]:                                   # the live exp(log) value the losses read
            agent._set_dual(name, log_param)
    if ("flat" in sd) != (agent.fast is not None):
        raise ValueError("checkpoint and run disagree on the fused update path (--no_fast_path)")
    if agent.fast is not None:
        for name in _FLAT_NETS:
            net = getattr(agent.fast, name)
            net.m.copy_(sd["flat"][name]["m"])
            net.v.copy_(sd["flat"][name]["v"])
            net.step.copy_(sd["flat"][name]["step"])
        agent.fast.noise_tick.copy_(sd["noise_tick"])
        agent.fast.actor_rows = sd["actor_rows"]
    for k, o in _optimisers(agent).items():
        o.load_state_dict(sd["optim"][k])


def loop_state(loop):
    out = {"stats": _cpu(loop.stats), "reward_sums": _cpu(loop.reward_sums), "ep_reward": _cpu(loop.ep_reward),
           "total_numsteps": loop.total_numsteps, "updates": loop.updates, "host_updates": list(loop.host_updates),
           "num_constraint_violations": loop.num_constraint_violations}
    log = loop.episode_log
    if log is not None:
        out["episode_log"] = {"ep_len": _cpu(log.ep_len), "ep_ret": _cpu(log.ep_ret), "ep_viol": _cpu(log.ep_viol),
                              "ep_rec": _cpu(log.ep_rec), "state": _cpu(log.state),
                              "rec_i32": _cpu(log.rec_i32[:int(log.state[0].item())]),
                              "rec_f64": _cpu(log.rec_f64[:int(log.state[0].item())])}
    return out


def load_loop_state(loop, sd):
    loop.stats.copy_(sd["stats"])
    loop.reward_sums.copy_(sd["reward_sums"])
    loop.ep_reward.copy_(sd["ep_reward"])
    loop.total_numsteps, loop.updates = sd["total_numsteps"], sd["updates"]
    loop.host_updates = list(sd["host_updates"])
    loop.num_constraint_violations = sd["num_constraint_violations"]
    loop.obs = loop.env.obs
    loop.graph = None                                   # captured graphs hold the old stream position
    if loop.episode_log is not None and "episode_log" in sd:
        log, e = loop.episode_log, sd["episode_log"]
        for f in ("ep_len", "ep_ret", "ep_viol", "ep_rec", "state"):
            getattr(log, f).copy_(e[f])
        k = e["rec_i32"].shape[0]
        log.rec_i32[:k].copy_(e["rec_i32"])
        log.rec_f64[:k].copy_(e["rec_f64"])


def mpc_state(mpc):
    return {"model": {n: _cpu(v) for n, v in mpc.model.state_dict().items()},
            "optim": mpc.model.optim.state_dict() if hasattr(mpc.model, "optim") else None,
            "trainer": mpc._trainer.state_dict() if mpc._trainer is not None else None,
            "train_in": _cpu(mpc.train_in), "train_targs": _cpu(mpc.train_targs),
            "has_been_trained": mpc.has_been_trained, "prev_sol": _cpu(mpc.prev_sol),
            "cem_tick": _cpu(mpc.optimizer.tick)}


def load_mpc_state(mpc, sd):
    mpc.model.load_state_dict(sd["model"])
    if sd["optim"] is not None:
        mpc.model.optim.load_state_dict(sd["optim"])
    if sd.get("trainer") is not None:
        from .ensemble_train import FusedEnsembleTrainer
        if mpc._trainer is None:
            mpc._trainer = FusedEnsembleTrainer(mpc.model, lr=mpc.model.optim.param_groups[0]["lr"])
        mpc._trainer.load_state_dict(sd["trainer"])
    dev = mpc.device
    mpc.train_in, mpc.train_targs = sd["train_in"].to(dev), sd["train_targs"].to(dev)
    mpc.has_been_trained = sd["has_been_trained"]
    mpc.prev_sol.copy_(sd["prev_sol"])
    mpc.optimizer.tick.copy_(sd["cem_tick"])


# -- whole experiment -------------------------------------------------------------------------------
def experiment_state(exp, extra=None):
    dev = exp.device
    sd = {"format": FORMAT, "env_name": exp.exp_cfg.env_name, "agent": agent_state(exp.agent),
          "memory": replay_state(exp.memory), "recovery_memory": replay_state(exp.recovery_memory),
          "env": env_state(exp.env), "loop": loop_state(exp.loop),
          "counters": {k: getattr(exp, k) for k in ("total_numsteps", "updates", "num_constraint_violations",
                                                    "num_unsafe_transitions", "num_viols", "num_successes",
                                                    "viol_and_recovery", "viol_and_no_recovery")},
          "rng": {"torch": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state(dev),
                  "numpy": np.random.get_state(), "python": random.getstate(),
                  "loop_actions": exp.loop.action_rng.get_state()},
          "extra": extra or {}}
    if exp.recovery_policy is not None:
        sd["mpc"] = mpc_state(exp.recovery_policy)
    if getattr(exp, "_eval_env", None) is not None:
        sd["eval_env"] = env_state(exp._eval_env)
    return sd


def load_experiment_state(exp, sd):
    if sd.get("format") != FORMAT:
        raise ValueError("unknown checkpoint format %r" % (sd.get("format"),))
    if sd["env_name"] != exp.exp_cfg.env_name:
        raise ValueError("checkpoint is for env %r, run is %r" % (sd["env_name"], exp.exp_cfg.env_name))
    # validate everything before the first buffer is overwritten
    if sd["env"]["num_envs"] != exp.env.num_envs:
        raise ValueError("num_envs differs: checkpoint %d, run %d" % (sd["env"]["num_envs"], exp.env.num_envs))
    for name in ("memory", "recovery_memory"):
        if not capacity_fits(sd[name], getattr(exp, name).capacity):
            raise _capacity_error(name, sd[name], getattr(exp, name).capacity)
    if ("flat" in sd["agent"]) != (exp.agent.fast is not None):
        raise ValueError("checkpoint and run disagree on the fused update path (--no_fast_path)")
    if ("mpc" in sd) != (exp.recovery_policy is not None):
        raise ValueError("checkpoint and run disagree on model-based recovery")
    load_agent_state(exp.agent, sd["agent"])
    load_replay_state(exp.memory, sd["memory"])
    load_replay_state(exp.recovery_memory, sd["recovery_memory"])
    load_env_state(exp.env, sd["env"])
    load_loop_state(exp.loop, sd["loop"])
    for k, v in sd["counters"].items():
        setattr(exp, k, v)
    if exp.recovery_policy is not None:
        load_mpc_state(exp.recovery_policy, sd["mpc"])
    if "eval_env" in sd:
        load_env_state(exp.eval_env(), sd["eval_env"])
    torch.set_rng_state(sd["rng"]["torch"])
    torch.cuda.set_rng_state(sd["rng"]["cuda"], exp.device)
    np.random.set_state(sd["rng"]["numpy"])
    random.setstate(sd["rng"]["python"])
    if "loop_actions" in sd["rng"]:
        exp.loop.action_rng.set_state(sd["rng"]["loop_actions"])
    torch.cuda.synchronize(exp.device)
    return sd.get("extra", {})


def save(exp, path, extra=None):
    """Write atomically (tmp + rename): a job killed mid-write leaves the previous checkpoint intact."""
    torch.cuda.synchronize(exp.device)
    tmp = path + ".tmp"
    torch.save(experiment_state(exp, extra), tmp)
    os.replace(tmp, path)
    return path


def load(exp, path):
    return load_experiment_state(exp, torch.load(path, map_location="cpu", weights_only=False))
