"""Golden rows of the REFERENCE's navigation get_offline_data (env/navigation1.py:133-164, env/navigation2.py:133-243)
together with the random draws it consumed, so that the C oracle's generator can be fed the same draws and must then
reproduce the rows one for one (the existing nav_offline_golden.npz pins the numpy oracle through np.random.seed;
the C generator is Philox-driven and could so far only be compared in distribution).

Run: python tests/golden/gen_nav_offline_draws_golden.py -> tests/golden/nav_offline_draws_golden.npz (data only).
np.random.uniform / np.random.randn are replaced from outside by recording streams: uniform(lo, hi) = lo + (hi - lo) * u
(numpy's formula), randn = the recorded standard normals.
"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()


class Streams:
    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.u, self.z = [], []

    def uniform(self, low=0.0, high=1.0, size=None):
        n = 1 if size is None else int(np.prod(size))
        u = self.rng.random_sample(n)
        self.u.extend(u.tolist())
        out = low + (high - low) * u
        return float(out[0]) if size is None else out.reshape(size)

    def randn(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        z = self.rng.standard_normal(n)
        self.z.extend(z.tolist())
        return z.reshape(shape) if shape else float(z[0])


def main():
    out = {}
    for env_name in ("navigation1", "navigation2"):
        mod = importlib.import_module("env." + env_name)
        for num in (1000, 250):
            st = Streams(len(env_name) * 1000 + num + int(env_name[-1]))
            real_u, real_n = np.random.uniform, np.random.randn
            np.random.uniform, np.random.randn = st.uniform, st.randn
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    tr = mod.get_offline_data(num)
            finally:
                np.random.uniform, np.random.randn = real_u, real_n
            pre = "%s_n%d_" % (env_name, num)
            out[pre + "u"], out[pre + "z"] = np.array(st.u), np.array(st.z)
            out[pre + "s"] = np.array([t[0] for t in tr], dtype=np.float64)
            out[pre + "a"] = np.array([t[1] for t in tr], dtype=np.float64)
            out[pre + "c"] = np.array([int(t[2]) for t in tr], dtype=np.uint8)
            out[pre + "s2"] = np.array([t[3] for t in tr], dtype=np.float64)
            out[pre + "m"] = np.array([int(bool(t[4])) for t in tr], dtype=np.uint8)
            print(env_name, num, "->", len(tr), "rows,", int(out[pre + "c"].sum()), "violations,", len(st.u), "uniforms,",
                  len(st.z), "normals")
    np.savez_compressed(os.path.join(HERE, "nav_offline_draws_golden.npz"), **out)


if __name__ == "__main__":
    main()
