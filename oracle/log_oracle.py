"""TEST INFRASTRUCTURE -- CPU checker for the episode log (not part of the product).

Sequential numpy restatement of what the reference's driver and plotting derive per episode from the per-step
info dicts: episode length and sum of rewards (recovery_rl/experiment.py:423-425), whether any step violated
a constraint and the last reward (plotting/plot_runs.py:203-211), recovery use (experiment.py:421).
"""
import numpy as np


class EpisodeLogOracle:
    def __init__(self, n):
        self.n = n
        self.len = np.zeros(n, np.int32)
        self.ret = np.zeros(n, np.float64)
        self.viol = np.zeros(n, np.int32)
        self.rec = np.zeros(n, np.int32)
        self.iteration = 0
        self.records = []       # tuples in (iteration, env) order

    def append(self, reward, constraint, success, ep_done, recovery=None):
        for i in range(self.n):
            r = np.float32(reward[i])
            rec = int(recovery[i] != 0) if recovery is not None else 0
            c = int(constraint[i] != 0)
            self.len[i] += 1
            self.ret[i] = self.ret[i] + np.float64(r)
            self.viol[i] += c
            self.rec[i] += rec
            if ep_done[i]:
                flags = (1 if success[i] else 0) | (2 if c else 0) | (4 if rec else 0)
                self.records.append((i, self.iteration, int(self.len[i]), int(self.viol[i]), int(self.rec[i]),
                                     flags, float(self.ret[i]), float(r)))
                self.len[i] = 0
                self.ret[i] = 0.0
                self.viol[i] = 0
                self.rec[i] = 0
        self.iteration += 1
