"""Minimal action/observation space (the subset of gym.spaces.Box the reference touches:
`.low`, `.high`, `.shape`, `.sample()`, `.seed()`; env/navigation1.py:60-64, env/maze.py:123-132)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=np.float64)
            high = np.asarray(high, dtype=np.float64)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=np.float64)
            high = np.full(shape, high, dtype=np.float64)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.min(), self.high.max(), self.shape)
