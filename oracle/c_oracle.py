"""ctypes binding of oracle/librrl_oracle.so (the C CPU oracle) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
numpy arrays in, numpy arrays out; mirrors the argument order of include/rrl_hip.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RRL_ORACLE_SANITIZE=1: load the ASan + UBSan build instead (oracle/Makefile target `sanitize`; run the tests with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0, see tests/sanitize_oracle.sh)
_SANITIZE = os.environ.get("RRL_ORACLE_SANITIZE", "") not in ("", "0")
_SO = os.path.join(_HERE, "librrl_oracle_san.so" if _SANITIZE else "librrl_oracle.so")

ENV_KIND = {"navigation1": 0, "navigation2": 1, "maze": 2}
STREAM_STEP, STREAM_RESET, STREAM_OFFLINE, STREAM_SAMPLE, STREAM_SAMPLE_NEG, STREAM_CEM, STREAM_ACTION = range(7)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] + (["sanitize"] if _SANITIZE else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.rrl_oracle_uniform01.restype = C.c_double
        _lib.rrl_oracle_uniform01.argtypes = [C.c_uint64]
        _lib.rrl_oracle_nav_offline.restype = C.c_int64
        _lib.rrl_oracle_maze_offline.restype = C.c_int64
        _lib.rrl_oracle_nav_offline_explicit.restype = C.c_int64
        _lib.rrl_oracle_maze_offline_explicit.restype = C.c_int64
        _lib.rrl_oracle_maze_distance.restype = C.c_double
        _lib.rrl_oracle_maze_distance.argtypes = [C.c_double, C.c_double]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox4x32(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().rrl_oracle_philox4x32(*[C.c_uint32(int(c)) for c in ctr], *[C.c_uint32(int(k)) for k in key], out)
    return [int(x) for x in out]


def normal2(seed, idx, stream, counter):
    z = (C.c_double * 2)()
    lib().rrl_oracle_normal2(C.c_uint64(seed), C.c_uint32(idx), C.c_uint32(stream), C.c_uint64(counter), z)
    return np.array([z[0], z[1]])


def normals(seed, n, stream, counter):
    return np.stack([normal2(seed, i, stream, counter) for i in range(n)])


def obstacle(env_name, x, y):
    return int(lib().rrl_oracle_obstacle(ENV_KIND[env_name], C.c_double(x), C.c_double(y)))


def nav_step(env_name, pos, action, t, noise=None, seed=0, counter=0, horizon=100, auto_reset=False):
    """Returns dict with next_obs, obs, reward, done, constraint, success, ep_done, pos, t,
    next_pos64, reward64.  `pos`/`t` are not modified (copies are advanced)."""
    n = len(pos)
    pos = np.ascontiguousarray(pos, dtype=np.float64).copy()
    action = np.ascontiguousarray(action, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.int32).copy()
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float64)
    o = dict(next_obs=np.zeros((n, 2), np.float32), obs=np.zeros((n, 2), np.float32),
             reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
             constraint=np.zeros(n, np.uint8), success=np.zeros(n, np.uint8),
             ep_done=np.zeros(n, np.uint8), next_pos64=np.zeros((n, 2)), reward64=np.zeros(n))
    rc = lib().rrl_oracle_nav_step(
        ENV_KIND[env_name], C.c_int64(n), _p(pos), _p(action), _p(noise), C.c_uint64(seed),
        C.c_uint64(counter), _p(o["next_obs"]), _p(o["obs"]), _p(o["reward"]), _p(o["done"]),
        _p(o["constraint"]), _p(o["success"]), _p(o["ep_done"]), _p(t), C.c_int32(horizon),
        C.c_int(int(auto_reset)), _p(o["next_pos64"]), _p(o["reward64"]))
    assert rc == 0
    o["pos"], o["t"] = pos, t
    return o


def nav_reset(env_name, n, noise=None, seed=0, counter=0):
    pos = np.zeros((n, 2))
    obs = np.zeros((n, 2), np.float32)
    t = np.zeros(n, np.int32)
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float64)
    rc = lib().rrl_oracle_nav_reset(ENV_KIND[env_name], C.c_int64(n), _p(pos), _p(obs), _p(t),
                                    _p(noise), C.c_uint64(seed), C.c_uint64(counter))
    assert rc == 0
    return pos, obs, t


def nav_rollout(env_name, pos, actions, seed=0, counter=0):
    T, n = actions.shape[0], actions.shape[1]
    pos = np.ascontiguousarray(pos, dtype=np.float64).copy()
    actions = np.ascontiguousarray(actions, dtype=np.float32)
    obs = np.zeros((T, n, 2), np.float32)
    rew = np.zeros((T, n), np.float32)
    cons = np.zeros((T, n), np.uint8)
    done = np.zeros((T, n), np.uint8)
    rc = lib().rrl_oracle_nav_rollout(ENV_KIND[env_name], C.c_int64(n), C.c_int32(T), _p(pos),
                                      _p(actions), C.c_uint64(seed), C.c_uint64(counter),
                                      _p(obs), _p(rew), _p(cons), _p(done))
    assert rc == 0
    return dict(pos=pos, obs=obs, reward=rew, constraint=cons, done=done)


def nav_offline(env_name, num_transitions, seed):
    cap = 10 * (num_transitions // 10 // 3 + 4 * (num_transitions // 10 // 4) + num_transitions // 10) + 16
    s = np.zeros((cap, 2), np.float32)
    a = np.zeros((cap, 2), np.float32)
    c = np.zeros(cap, np.float32)
    s2 = np.zeros((cap, 2), np.float32)
    m = np.zeros(cap, np.float32)
    w = lib().rrl_oracle_nav_offline(ENV_KIND[env_name], C.c_int64(num_transitions),
                                     C.c_uint64(seed), _p(s), _p(a), _p(c), _p(s2), _p(m),
                                     C.c_int64(cap))
    assert w >= 0, w
    return s[:w], a[:w], c[:w], s2[:w], m[:w]


class Draws(C.Structure):
    """rrl_oracle_draws: uniforms / standard normals in the reference's np.random call order."""
    _fields_ = [("u", C.c_void_p), ("n_u", C.c_int64), ("i_u", C.c_int64), ("z", C.c_void_p), ("n_z", C.c_int64),
                ("i_z", C.c_int64), ("exhausted", C.c_int)]

    def __init__(self, u=(), z=()):
        self._u = np.ascontiguousarray(u, dtype=np.float64)
        self._z = np.ascontiguousarray(z, dtype=np.float64)
        super().__init__(self._u.ctypes.data, len(self._u), 0, self._z.ctypes.data, len(self._z), 0, 0)


def _offline_out(cap):
    f32 = lambda *sh: np.zeros(sh, np.float32)
    f64 = lambda *sh: np.zeros(sh, np.float64)
    return (f32(cap, 2), f32(cap, 2), f32(cap), f32(cap, 2), f32(cap)), (f64(cap, 2), f64(cap, 2), f64(cap, 2))


def nav_offline_explicit(env_name, num_transitions, u, z):
    """The offline-data generator fed explicit draws -> (f32 rows, f64 rows (s, a, s2), draws consumed)."""
    cap = 10 * (num_transitions // 10 // 3 + 4 * (num_transitions // 10 // 4) + num_transitions // 10) + 16
    (s, a, c, s2, m), (s64, a64, s2_64) = _offline_out(cap)
    d = Draws(u, z)
    w = lib().rrl_oracle_nav_offline_explicit(ENV_KIND[env_name], C.c_int64(num_transitions), C.byref(d), _p(s),
                                              _p(a), _p(c), _p(s2), _p(m), _p(s64), _p(a64), _p(s2_64),
                                              C.c_int64(cap))
    assert w >= 0, w
    return (s[:w], a[:w], c[:w], s2[:w], m[:w]), (s64[:w], a64[:w], s2_64[:w]), (d.i_u, d.i_z)


def maze_offline_explicit(num_transitions, u, rand_actions):
    cap = max(2 * (num_transitions // 2), 1)
    (s, a, c, s2, m), (s64, a64, s2_64) = _offline_out(cap)
    d = Draws(u)
    ra = np.ascontiguousarray(rand_actions, dtype=np.float32)
    assert ra.shape == (num_transitions // 2, 2)
    w = lib().rrl_oracle_maze_offline_explicit(C.c_int64(num_transitions), C.byref(d), _p(ra), _p(s), _p(a), _p(c),
                                               _p(s2), _p(m), _p(s64), _p(a64), _p(s2_64), C.c_int64(cap))
    assert w >= 0, w
    return (s[:w], a[:w], c[:w], s2[:w], m[:w]), (s64[:w], a64[:w], s2_64[:w]), d.i_u


def maze_reset_explicit(mode, check_constraint, u):
    """One reset from explicit uniforms -> (x, y, uniforms consumed)."""
    d = Draws(u)
    x, y = C.c_double(), C.c_double()
    rc = lib().rrl_oracle_maze_reset_explicit(C.c_int(mode), C.c_int(int(check_constraint)), C.byref(d),
                                              C.byref(x), C.byref(y))
    assert rc == 0, rc
    return x.value, y.value, d.i_u


def maze_step64(x, y, ax, ay, steps, horizon=100):
    """One env step with a float64 action -> dict (env/maze.py:139-168)."""
    cx, cy, st = C.c_double(x), C.c_double(y), C.c_int32(steps)
    rew, dn, cons, succ = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    lib().rrl_oracle_maze_step64(C.byref(cx), C.byref(cy), C.c_double(ax), C.c_double(ay), C.byref(st),
                                 C.c_int32(horizon), C.byref(rew), C.byref(dn), C.byref(cons), C.byref(succ))
    return dict(x=cx.value, y=cy.value, steps=st.value, reward=rew.value, done=dn.value, constraint=cons.value,
                success=succ.value)


def maze_distance(x, y):
    return float(lib().rrl_oracle_maze_distance(C.c_double(x), C.c_double(y)))


def maze_contact(x, y):
    return int(lib().rrl_oracle_maze_contact(C.c_double(x), C.c_double(y)))


def maze_step(pos, action, t, seed=0, counter=0, horizon=100, auto_reset=False):
    n = len(pos)
    pos = np.ascontiguousarray(pos, dtype=np.float64).copy()
    action = np.ascontiguousarray(action, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.int32).copy()
    o = dict(next_obs=np.zeros((n, 2), np.float32), obs=np.zeros((n, 2), np.float32),
             reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
             constraint=np.zeros(n, np.uint8), success=np.zeros(n, np.uint8),
             ep_done=np.zeros(n, np.uint8), next_pos64=np.zeros((n, 2)), reward64=np.zeros(n))
    rc = lib().rrl_oracle_maze_step(
        C.c_int64(n), _p(pos), _p(action), C.c_uint64(seed), C.c_uint64(counter), _p(o["next_obs"]),
        _p(o["obs"]), _p(o["reward"]), _p(o["done"]), _p(o["constraint"]), _p(o["success"]),
        _p(o["ep_done"]), _p(t), C.c_int32(horizon), C.c_int(int(auto_reset)), _p(o["next_pos64"]),
        _p(o["reward64"]))
    assert rc == 0
    o["pos"], o["t"] = pos, t
    return o


def maze_reset(n, mode=0, check_constraint=True, seed=0, counter=0):
    pos = np.zeros((n, 2))
    obs = np.zeros((n, 2), np.float32)
    t = np.zeros(n, np.int32)
    rc = lib().rrl_oracle_maze_reset(C.c_int64(n), _p(pos), _p(obs), _p(t), C.c_int(mode),
                                     C.c_int(int(check_constraint)), C.c_uint64(seed), C.c_uint64(counter))
    assert rc == 0
    return pos, obs, t


def maze_expert_action(x, y):
    act = (C.c_double * 2)()
    lib().rrl_oracle_maze_expert_action(C.c_double(x), C.c_double(y), act)
    return np.array([act[0], act[1]])


def maze_offline(num_transitions, seed):
    cap = max(2 * (num_transitions // 2), 1)
    s = np.zeros((cap, 2), np.float32)
    a = np.zeros((cap, 2), np.float32)
    c = np.zeros(cap, np.float32)
    s2 = np.zeros((cap, 2), np.float32)
    m = np.zeros(cap, np.float32)
    w = lib().rrl_oracle_maze_offline(C.c_int64(num_transitions), C.c_uint64(seed), _p(s), _p(a), _p(c),
                                      _p(s2), _p(m), C.c_int64(cap))
    assert w >= 0, w
    return s[:w], a[:w], c[:w], s2[:w], m[:w]


def cem_sample(mean, var, lb, ub, pop, epsilon=1e-3, sticky=False, active=None, seed=0, counter=0):
    M, dim = mean.shape
    mean, var = (np.ascontiguousarray(x, np.float64) for x in (mean, var))
    lb, ub = (np.ascontiguousarray(x, np.float64) for x in (lb, ub))
    active = np.ones(M, np.uint8) if active is None else np.ascontiguousarray(active, np.uint8).copy()
    samples = np.zeros((M, pop, dim), np.float32)
    rc = lib().rrl_oracle_cem_sample(C.c_int64(M), C.c_int32(pop), C.c_int32(dim), _p(mean), _p(var), _p(lb),
                                     _p(ub), C.c_double(epsilon), C.c_int(int(sticky)), _p(active),
                                     C.c_uint64(seed), C.c_uint64(counter), _p(samples))
    assert rc == 0
    return samples, active


def cem_update(samples, costs, mean, var, num_elites, alpha, active=None):
    M, pop, dim = samples.shape
    samples = np.ascontiguousarray(samples, np.float32)
    costs = np.ascontiguousarray(costs, np.float32)
    mean, var = (np.ascontiguousarray(x, np.float64).copy() for x in (mean, var))
    if active is not None:
        active = np.ascontiguousarray(active, np.uint8)
    rc = lib().rrl_oracle_cem_update(C.c_int64(M), C.c_int32(pop), C.c_int32(dim), C.c_int32(num_elites),
                                     C.c_double(alpha), _p(samples), _p(costs), _p(mean), _p(var), _p(active))
    if rc != 0:
        raise ValueError("Number of elites must be at most the population size.")
    return mean, var


class _Replay(C.Structure):
    _fields_ = [("s", C.c_void_p), ("a", C.c_void_p), ("r", C.c_void_p), ("s2", C.c_void_p),
                ("m", C.c_void_p), ("cap", C.c_int64), ("pos", C.c_int64), ("size", C.c_int64), ("pinned", C.c_int64)]


class OracleReplay:
    """Ring buffer with the reference's push/sample semantics (replay_memory.py)."""

    def __init__(self, capacity):
        self.cap = capacity
        self.s = np.zeros((capacity, 2), np.float32)
        self.a = np.zeros((capacity, 2), np.float32)
        self.r = np.zeros(capacity, np.float32)
        self.s2 = np.zeros((capacity, 2), np.float32)
        self.m = np.zeros(capacity, np.float32)
        self._c = _Replay(self.s.ctypes.data, self.a.ctypes.data, self.r.ctypes.data,
                          self.s2.ctypes.data, self.m.ctypes.data, capacity, 0, 0, 0)

    def pin(self, rows=None):
        """rows [0, rows) are never overwritten (default: everything stored so far)"""
        self._c.pinned = self.size if rows is None else int(rows)

    @property
    def pos(self):
        return int(self._c.pos)

    @property
    def size(self):
        return int(self._c.size)

    def __len__(self):
        return self.size

    def push(self, s, a, r, s2, m, valid=None):
        s, a, s2 = (np.ascontiguousarray(x, np.float32).reshape(-1, 2) for x in (s, a, s2))
        r, m = (np.ascontiguousarray(x, np.float32).reshape(-1) for x in (r, m))
        if valid is not None:
            valid = np.ascontiguousarray(valid, np.uint8)
        rc = lib().rrl_oracle_replay_push(C.byref(self._c), C.c_int64(len(r)), _p(s), _p(a),
                                          _p(r), _p(s2), _p(m), _p(valid))
        assert rc == 0

    def sample_indices(self, B, seed, counter):
        idx = np.zeros(B, np.int64)
        rc = lib().rrl_oracle_sample_indices(C.c_int64(self.size), C.c_int32(B), C.c_uint64(seed),
                                             C.c_uint64(counter), C.c_uint32(STREAM_SAMPLE), _p(idx))
        if rc != 0:
            raise ValueError("Sample larger than population or is negative")
        return idx

    def sample_stratified_indices(self, n_pos, n_neg, seed, counter, clamp=False, return_split=False):
        idx = np.zeros(n_pos + n_neg, np.int64)
        used = C.c_int32(n_pos)
        rc = lib().rrl_oracle_sample_stratified_clamped(C.byref(self._c), C.c_int32(n_pos), C.c_int32(n_neg),
                                                        C.c_int(int(clamp)), C.c_uint64(seed), C.c_uint64(counter),
                                                        _p(idx), C.byref(used))
        if rc != 0:
            raise ValueError("Sample larger than population or is negative")
        return (idx, used.value) if return_split else idx

    def sample_split_indices(self, n_demo, n_online, seed, counter, return_split=False):
        idx = np.zeros(n_demo + n_online, np.int64)
        used = C.c_int32(n_demo)
        rc = lib().rrl_oracle_sample_split(C.byref(self._c), C.c_int32(n_demo), C.c_int32(n_online), C.c_uint64(seed),
                                           C.c_uint64(counter), _p(idx), C.byref(used))
        if rc != 0:
            raise ValueError("Sample larger than population or is negative")
        return (idx, used.value) if return_split else idx

    def gather(self, idx):
        B = len(idx)
        idx = np.ascontiguousarray(idx, np.int64)
        s = np.zeros((B, 2), np.float32)
        a = np.zeros((B, 2), np.float32)
        r = np.zeros(B, np.float32)
        s2 = np.zeros((B, 2), np.float32)
        m = np.zeros(B, np.float32)
        rc = lib().rrl_oracle_gather(C.byref(self._c), C.c_int32(B), _p(idx), _p(s), _p(a), _p(r),
                                     _p(s2), _p(m))
        assert rc == 0
        return s, a, r, s2, m

    def sample(self, B, seed, counter, pos_fraction=None):
        if pos_fraction is not None:
            n_pos = int(B * pos_fraction)
            idx = self.sample_stratified_indices(n_pos, B - n_pos, seed, counter)
        else:
            idx = self.sample_indices(B, seed, counter)
        return self.gather(idx)
