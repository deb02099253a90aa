"""In-container harness that makes the *reference* importable so that golden
fixtures can be captured from it.

This module never travels as "the reference": it only installs tiny stand-ins
for third-party packages the image lacks (gym, dotmap, cv2, moviepy, mujoco_py,
torchvision) so that `/root/reference`'s own Python runs unmodified, and it is
imported ONLY by the `gen_*.py` fixture generators in this directory.  Nothing
under `tests/` that runs on the GPU box imports it (the reference tree does not
exist there).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("RRL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "recovery_rl"))


class _Box:
    """Minimal gym.spaces.Box: low/high/shape/sample/seed."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=np.float64)
            high = np.asarray(high, dtype=np.float64)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=np.float64)
            high = np.full(shape, high, dtype=np.float64)
        self.low, self.high, self.shape = low, high, tuple(shape)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(np.float32)


class _Env:
    def seed(self, seed=None):
        return [seed]


class _EzPickle:
    def __init__(self, *a, **k):
        pass


class _DotMap(dict):
    """Auto-vivifying attribute dict (the subset of dotmap the reference uses)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, val in dict(*a, **k).items():
            self[key] = val

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        if key not in self:
            self[key] = _DotMap()
        return self[key]

    def __setattr__(self, key, val):
        self[key] = val

    def pprint(self):
        pass


_REGISTRY = {}


def _register(id, entry_point):
    _REGISTRY[id] = entry_point


def _make(id):
    import importlib
    mod, cls = _REGISTRY[id].split(":")
    return getattr(importlib.import_module(mod), cls)()


def install():
    """Install the shims and put the reference on sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    if not hasattr(np, "float"):
        np.float = float  # env/navigation1.py:63 uses np.float('inf')
    if not hasattr(np, "int"):
        np.int = int

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "gym" not in sys.modules:
        gym = mod("gym", Env=_Env, make=_make)
        gym.utils = mod("gym.utils", EzPickle=_EzPickle)
        gym.spaces = mod("gym.spaces", Box=_Box)
        gym.envs = mod("gym.envs")
        gym.envs.registration = mod("gym.envs.registration", register=_register)
    if "dotmap" not in sys.modules:
        mod("dotmap", DotMap=_DotMap)
    for name in ("cv2", "moviepy", "moviepy.editor", "mujoco_py", "torchvision",
                 "torchvision.utils"):
        if name not in sys.modules:
            mod(name)
    sys.modules["mujoco_py"].load_model_from_path = None
    sys.modules["mujoco_py"].MjSim = None
    sys.modules["torchvision.utils"].save_image = None
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    import matplotlib
    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
