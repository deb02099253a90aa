"""What the demonstration-share rule is for, on the device (rule 3).

At 4096 lock-step envs the safety buffer's ring holds ~1e6 safe online rows after 245 iterations and the 20 000 pinned
constraint demonstrations are 2 % of ONE uniform draw.  In round 3 that ended seeds 6 / 8 of config 4 (Navigation2, model-based
recovery) at 100 % / 45-100 % violation rate: the online rows pull Q_risk down on the only rows that show violations, the gate
opens, the envs ram the obstacle.  The full-length evidence is profiles/round4_learning_vec4096_config4_*.json (8 seeds, 1 650
iterations, minutes per seed); this test checks the MECHANISM in seconds: the same pre-trained safety critic, the ring flooded
with safe rows of random-policy episodes, then K Q_risk updates under either draw (profiles/qrisk_mix_probe.py is the long form:
0.63 -> 0.48 under the uniform draw, 0.63 -> 0.71 under the split draw after 12 000 updates)."""
import numpy as np
import pytest
import torch

import arg_utils

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_split_draw_keeps_the_safety_critic_on_the_violating_demonstrations():
    from recovery_rl_amd.env import make_vec_env, register_env
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory
    from recovery_rl_amd.sac import SAC
    n, K, pre = 4096, 4000, 2000
    cfg = arg_utils.get_args(["--env-name", "navigation2", "--cuda", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                              "--num_unsafe_transitions", "20000", "--num_envs", str(n), "--seed", "8"])
    register_env(cfg.env_name)
    env = make_vec_env(cfg.env_name, n, device=DEV, seed=cfg.seed)
    s, a, c, s2, m = (x.contiguous() for x in env.transition_function(cfg.num_unsafe_transitions))
    viol = c > 0
    assert 1000 < int(viol.sum()) < 3000
    q_viol, q_safe, demo_rows = {}, {}, {}
    for name, share in (("uniform", None), ("split", 0.5)):
        torch.manual_seed(cfg.seed)
        agent = SAC(env.observation_space, env.action_space, cfg, "/tmp")
        agent.enable_fast_path(cfg.batch_size)
        qr = agent.safety_critic
        mem = ConstraintReplayMemory(cfg.safe_replay_size, cfg.seed, device=DEV)
        mem.push(s, a, c, s2, m)
        mem.pin()
        for _ in range(pre):
            qr.update_parameters(memory=mem, policy=agent.policy, batch_size=cfg.batch_size)
        obs = env.reset()
        while len(mem) < mem.capacity:                 # the ring after 245 iterations of a policy that stays near the start
            act = env.sample_actions()
            state = obs.clone()
            obs, rew, done, info = env.step(act)
            mem.push(state, info["action"], info["constraint"].float(), info["next_state"], 1.0 - done.float())
        assert int((mem.r[mem.pinned:] > 0).sum()) < 100             # the online rows are (almost) all safe
        qr.demo_share = share
        rows = 0
        for k in range(K):
            qr.update_parameters(memory=mem, policy=agent.policy, batch_size=cfg.batch_size)
            if k % 500 == 0:
                rows += int((mem._batch(cfg.batch_size)[5] < mem.pinned).sum())
        mem.check_error()
        with torch.no_grad():
            q = qr.get_value(s, a).squeeze(1)
        q_viol[name], q_safe[name], demo_rows[name] = float(q[viol].mean()), float(q[~viol].mean()), rows / (K // 500)
    # what each draw showed the critic: ~2 % of 256 rows against exactly half
    assert demo_rows["uniform"] < 12 and demo_rows["split"] == 128
    # the split draw keeps the critic on the violating demonstrations, the uniform draw lets it slide
    assert q_viol["split"] > q_viol["uniform"] + 0.06, (q_viol, q_safe)
    assert q_viol["split"] - q_safe["split"] > q_viol["uniform"] - q_safe["uniform"] + 0.05, (q_viol, q_safe)
    assert q_viol["split"] > 0.5 > 0.2                              # far above eps_safe
