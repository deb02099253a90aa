"""Checkpoint / resume: a run continued from checkpoint.pt ends in exactly the state of the
uninterrupted run -- every tensor of the final checkpoints (networks, Adam moments, both replay rings, env
state, RNG ticks, counters, episode records) is compared bit-for-bit."""
import os
import pickle

import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd import checkpoint
from recovery_rl_amd.experiment import Experiment

pytestmark = pytest.mark.gpu


def _cfg(tmp, num_eps, extra=()):
    return arg_utils.get_args(["--env-name", "navigation1", "--cuda", "--hidden_size", "32", "--logdir", str(tmp),
                               "--seed", "5", "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps",
                               "20", "--num_envs", "64", "--log_every", "10", "--num_eps", str(num_eps)]
                              + list(extra))


def _diff(a, b, path=""):
    """Paths at which two checkpoint trees differ."""
    if isinstance(a, dict):
        if set(a) != set(b):
            return [path + ": keys %s" % sorted(set(a) ^ set(b))]
        return [d for k in a for d in _diff(a[k], b[k], path + "/" + str(k))]
    if isinstance(a, (list, tuple)):
        if len(a) != len(b):
            return [path + ": length %d vs %d" % (len(a), len(b))]
        return [d for i, (x, y) in enumerate(zip(a, b)) for d in _diff(x, y, path + "/%d" % i)]
    if torch.is_tensor(a):
        return [] if a.shape == b.shape and torch.equal(a, b) else [path]
    if isinstance(a, np.ndarray):
        return [] if a.shape == b.shape and a.tobytes() == b.tobytes() else [path]
    return [] if a == b else [path + ": %r vs %r" % (a, b)]


FLAGS = {"mf_recovery": ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3"],
         "sac_only": [],
         "lagrangian_autograd": ["--DGD_constraints", "--update_nu", "--nu", "100", "--gamma_safe", "0.8",
                                 "--eps_safe", "0.3"]}


@pytest.mark.parametrize("variant", sorted(FLAGS))
def test_resumed_run_equals_uninterrupted_run(tmp_path, variant):
    flags = FLAGS[variant]
    full = Experiment(_cfg(tmp_path / "full", 400, flags))
    full.run()
    part = Experiment(_cfg(tmp_path / "part", 150, flags))
    part.run()
    ck = os.path.join(part.logdir, "checkpoint.pt")
    mid = torch.load(ck, map_location="cpu", weights_only=False)
    assert mid["extra"]["iteration"] % 10 == 0
    assert 150 < mid["extra"]["history"][-1]["episodes"] < 400
    cont = Experiment(_cfg(tmp_path / "cont", 400, flags + ["--resume", ck]))
    cont.run()
    a = torch.load(os.path.join(full.logdir, "checkpoint.pt"), map_location="cpu", weights_only=False)
    b = torch.load(os.path.join(cont.logdir, "checkpoint.pt"), map_location="cpu", weights_only=False)
    assert a["extra"]["iteration"] == b["extra"]["iteration"] > mid["extra"]["iteration"]
    d = _diff(a, b)
    assert not d, "\n".join(d)
    ra = pickle.load(open(os.path.join(full.logdir, "run_stats.pkl"), "rb"))
    rb = pickle.load(open(os.path.join(cont.logdir, "run_stats.pkl"), "rb"))
    assert ra["vector_stats"] == rb["vector_stats"]
    assert ra["episode_stats"].tobytes() == rb["episode_stats"].tobytes()
    assert os.path.getsize(os.path.join(cont.logdir, "episode_stats.bin")) == ra["episode_stats"].nbytes


def test_checkpoint_rejects_a_different_run(tmp_path):
    part = Experiment(_cfg(tmp_path / "a", 100, FLAGS["mf_recovery"]))
    part.run()
    ck = os.path.join(part.logdir, "checkpoint.pt")
    other = Experiment(_cfg(tmp_path / "b", 100, FLAGS["mf_recovery"] + ["--num_envs", "32"]))
    with pytest.raises(ValueError, match="num_envs"):
        checkpoint.load(other, ck)
    slow = Experiment(_cfg(tmp_path / "c", 100, FLAGS["mf_recovery"] + ["--no_fast_path"]))
    with pytest.raises(ValueError, match="fused update path"):
        checkpoint.load(slow, ck)
    assert not os.path.exists(ck + ".tmp")


def test_resume_with_a_larger_step_budget_keeps_or_grows_the_checkpoints_buffers(tmp_path):
    """Vectorisation rule 4 sizes both buffers from --num_steps; continuing a run with a larger --num_steps (the usual way to
    continue) must not be refused for that: the resumed run takes the checkpoint's capacities, and grows them while the
    checkpoint's rings have not wrapped.  The continued run equals the run that had the larger budget from the start."""
    flags = FLAGS["mf_recovery"] + ["--replay_size", "1000", "--safe_replay_size", "3000", "--num_eps", "100000"]
    full = Experiment(_cfg(tmp_path / "full", 0, flags + ["--num_steps", "12799"]))
    assert full.memory.capacity == 12799 + 128 and full.recovery_memory.capacity == 12799 + 128 + 2000
    full.run()
    part = Experiment(_cfg(tmp_path / "part", 0, flags + ["--num_steps", "6399"]))
    assert part.memory.capacity == 6399 + 128
    part.run()
    ck = os.path.join(part.logdir, "checkpoint.pt")
    assert checkpoint.peek_capacities(ck) == {"memory": (6527, False), "recovery_memory": (8527, False)}
    cont = Experiment(_cfg(tmp_path / "cont", 0, flags + ["--num_steps", "12799", "--resume", ck]))
    assert cont.memory.capacity == full.memory.capacity and cont.recovery_memory.capacity == full.recovery_memory.capacity
    cont.run()
    a = torch.load(os.path.join(full.logdir, "checkpoint.pt"), map_location="cpu", weights_only=False)
    b = torch.load(os.path.join(cont.logdir, "checkpoint.pt"), map_location="cpu", weights_only=False)
    assert a["extra"]["iteration"] == b["extra"]["iteration"] == 200
    assert not _diff(a["agent"], b["agent"]) and not _diff(a["memory"], b["memory"])
    assert not _diff(a["recovery_memory"], b["recovery_memory"]) and not _diff(a["env"], b["env"])
    # the same budget again: the checkpoint's capacities are kept as they are
    same = Experiment(_cfg(tmp_path / "same", 0, flags + ["--num_steps", "6399", "--resume", ck]))
    assert same.memory.capacity == 6527 and same.recovery_memory.capacity == 8527
    # a wrapped ring cannot move into another capacity: the message names the flags that settle it
    wrapped = dict(a["memory"], size=a["memory"]["capacity"])
    assert not checkpoint.capacity_fits(wrapped, a["memory"]["capacity"] + 64)
    with pytest.raises(ValueError, match="--keep_replay_size"):
        checkpoint.load_replay_state(same.memory, wrapped)
