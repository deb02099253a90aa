"""--dp_mode env_shard: ONE learner, envs and replay split over ranks, gradients averaged before
every optimiser step.  Two ranks share the single GPU of the test box and reduce through gloo
(RRL_DIST_BACKEND=gloo; RCCL refuses two ranks on one device) -- the code path is the one RCCL runs on 8 GPUs."""
import glob
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(tmp_path, mode, port):
    env = dict(os.environ, RRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "rrl_main.py"), "--env-name", "navigation1",
           "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3", "--hidden_size", "32",
           "--batch_size", "64", "--num_envs", "32", "--num_eps", "60", "--log_every", "10", "--seed", "4",
           "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps", "20", "--logdir", str(tmp_path),
           "--dp_mode", mode]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    # one directory per rank; the names start with a wall-clock stamp, so order them by the seed suffix
    dirs = sorted(glob.glob(os.path.join(str(tmp_path), "*_seed*")), key=lambda d: d.rsplit("_seed", 1)[1])
    assert len(dirs) == 2 and dirs[0].endswith("_seed4") and dirs[1].endswith("_seed5")
    return [torch.load(os.path.join(d, "checkpoint.pt"), map_location="cpu", weights_only=False) for d in dirs], out


def test_env_shard_ranks_hold_one_learner(tmp_path):
    (a, b), out = _launch(tmp_path, "env_shard", 29631)
    for net in a["agent"]["modules"]:
        for k, v in a["agent"]["modules"][net].items():
            assert torch.equal(v, b["agent"]["modules"][net][k]), (net, k)
    for net in a["agent"]["flat"]:
        for k in ("m", "v", "step"):
            assert torch.equal(a["agent"]["flat"][net][k], b["agent"]["flat"][net][k]), (net, k)
    assert int(a["agent"]["flat"]["critic"]["step"][0]) > 10                 # updates did run
    assert not torch.equal(a["env"]["pos"], b["env"]["pos"])                  # own envs ...
    assert not torch.equal(a["memory"]["s"][:64], b["memory"]["s"][:64])      # ... and own replay
    assert a["extra"]["iteration"] == b["extra"]["iteration"]                 # left the loop together
    assert "Iter:" in out.stdout


def test_replica_ranks_are_independent_learners_that_stop_together(tmp_path):
    (a, b), _ = _launch(tmp_path, "replicas", 29632)
    assert not torch.equal(a["agent"]["modules"]["critic"]["linear1.weight"],
                           b["agent"]["modules"]["critic"]["linear1.weight"])
    assert a["extra"]["iteration"] == b["extra"]["iteration"]
