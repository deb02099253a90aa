"""Scripted env / agent for fixture G9 (driver loop semantics).  Pure Python + numpy, shared by the
generator (which drives the REFERENCE's Experiment with it) and by tests/test_loop_semantics_cpu.py
(which drives this repo's Experiment with the same script).  Nothing here comes from the reference."""
import numpy as np

HORIZON = 6
# per global env-step (0-based, training steps only): (done_by_env, constraint, success)
# episode 1: 6 steps, time-out only; episode 2: constraint at its 3rd step; episode 3: success at its 2nd;
# episode 4: time-out; episode 5: constraint at its 1st step; then time-outs
EPISODES = [(6, None), (3, "constraint"), (2, "success"), (6, None), (1, "constraint"), (6, None), (4, "constraint"),
            (6, None), (6, None), (2, "success"), (6, None), (6, None)]
VARIANTS = {
    "mf": ["--use_recovery", "--MF_recovery"],
    "mf_norelabel": ["--use_recovery", "--MF_recovery", "--disable_action_relabeling"],
    "mf_both": ["--use_recovery", "--MF_recovery", "--add_both_transitions"],
    "penalty": ["--constraint_reward_penalty", "5"],
    "online_off": ["--use_recovery", "--MF_recovery", "--disable_online_updates"],
}
BASE_ARGV = ["--env-name", "navigation1", "--num_eps", "8", "--start_steps", "4", "--batch_size", "5",
             "--num_unsafe_transitions", "9", "--critic_safe_pretraining_steps", "3", "--eps_safe", "0.5",
             "--updates_per_step", "2", "--eval", "", "--seed", "1"]


class Script:
    """Deterministic source of everything the stubs return."""

    def __init__(self):
        self.ep = -1
        self.k = 0
        self.n_random = 0
        self.n_task = 0
        self.n_rec = 0
        self.n_risk = 0

    def start_episode(self):
        self.ep += 1
        self.k = 0
        return np.array([float(self.ep + 1), 0.0])

    def transition(self, state, action):
        length, kind = EPISODES[self.ep % len(EPISODES)]
        self.k += 1
        last = self.k == length
        cons = int(last and kind == "constraint")
        succ = bool(last and kind == "success")
        done = bool(cons or succ)
        nxt = np.asarray(state, dtype=np.float64) + np.asarray(action, dtype=np.float64)
        reward = -float(self.k) - 0.25 * self.ep
        return nxt, reward, done, cons, succ

    def random_action(self):
        self.n_random += 1
        return np.array([0.01 * self.n_random, -0.5], dtype=np.float32)

    def task_action(self):
        self.n_task += 1
        return np.array([0.1 * self.n_task, 0.25], dtype=np.float32)

    def recovery_action(self):
        self.n_rec += 1
        return np.array([-0.125 * self.n_rec, -0.75], dtype=np.float32)

    def risk(self):
        self.n_risk += 1
        return 0.9 if self.n_risk % 3 == 0 else 0.1          # every third query triggers recovery

    def offline_data(self, num):
        return [(np.array([-float(i), 1.0]), np.array([0.5, float(i)], dtype=np.float32), i % 2,
                 np.array([-float(i) + 0.5, 2.0]), not (i % 2)) for i in range(num + 3)]
