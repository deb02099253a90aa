"""CPU suite: the oracle against the golden vectors captured from the reference
(tests/golden/*.npz, generator tests/golden/gen_env_golden.py)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import nav_oracle as no

ENVS = ("navigation1", "navigation2")


@pytest.fixture(scope="module")
def step_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "nav_step_golden.npz"))


@pytest.fixture(scope="module")
def offline_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "nav_offline_golden.npz"))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert co.philox4x32((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert co.philox4x32((0xffffffff,) * 4, (0xffffffff,) * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert co.philox4x32((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_normal_generator_moments():
    z = co.normals(1234, 40000, co.STREAM_STEP, 5)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert abs(np.corrcoef(z.T)[0, 1]) < 0.02
    assert abs((z ** 4).mean() - 3.0) < 0.15
    assert np.abs(z).max() < 8.5
    # different streams / counters decorrelate
    z2 = co.normals(1234, 40000, co.STREAM_RESET, 5)
    assert abs(np.corrcoef(z[:, 0], z2[:, 0])[0, 1]) < 0.02


@pytest.mark.parametrize("env", ENVS)
def test_c_oracle_step_matches_reference_bit_exact(env, step_golden):
    g = step_golden
    S, A, E = g[env + "_s"], g[env + "_a"], g[env + "_eps"]
    o = co.nav_step(env, S, A, np.zeros(len(S), np.int32), noise=E)
    assert np.array_equal(o["next_pos64"], g[env + "_s2"])
    assert np.array_equal(o["reward64"], g[env + "_reward"])
    for k in ("done", "constraint", "success"):
        assert np.array_equal(o[k], g[env + "_" + k]), k
    assert np.array_equal(o["next_obs"], g[env + "_s2"].astype(np.float32))
    assert np.array_equal(o["reward"], g[env + "_reward"].astype(np.float32))
    # the golden set exercises every branch
    assert g[env + "_constraint"].sum() > 100 and g[env + "_success"].sum() > 100
    assert (g[env + "_noise_drawn"] == 0).sum() > 100


@pytest.mark.parametrize("env", ENVS)
def test_numpy_oracle_step_matches_reference(env, step_golden):
    g = step_golden
    S, A, E = g[env + "_s"], g[env + "_a"], g[env + "_eps"]
    rows = np.r_[0:200, len(S) - 200:len(S)]
    e = no.NavOracleEnv(env)
    real = np.random.randn
    try:
        for i in rows:
            np.random.randn = lambda n, _i=i: E[_i].copy()
            e.state, e.time = S[i].copy(), 0
            nxt, cost, done, info = e.step(A[i])
            assert np.array_equal(nxt, g[env + "_s2"][i])
            assert cost == g[env + "_reward"][i]
            assert done == bool(g[env + "_done"][i])
            assert info["constraint"] == g[env + "_constraint"][i]
            assert info["success"] == bool(g[env + "_success"][i])
    finally:
        np.random.randn = real


def test_anchor_from_survey(step_golden):
    """SURVEY section 8c anchor: nav1, np.random.seed(0)."""
    a = step_golden["anchor"]
    np.random.seed(0)
    e = no.NavOracleEnv("navigation1")
    s0 = e.reset()
    o1, r1, _, _ = e.step(np.array([1, 0]))
    o2, r2, _, _ = e.step(np.array([2, -3]))
    assert np.array_equal(np.r_[s0, o1, r1, o2, r2], a)
    assert np.allclose(a[:2], [-48.23594765, 0.40015721]) and np.isclose(a[4], -48.23760744350776)


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("seed", (0, 1))
def test_numpy_oracle_offline_data_equals_reference(env, seed, offline_golden):
    g = offline_golden
    key = "%s_seed%d_n1000_" % (env, seed)
    np.random.seed(seed)
    tr = no.get_offline_data(env, 1000)
    assert len(tr) == len(g[key + "s"])
    assert np.array_equal(np.array([t[0] for t in tr]), g[key + "s"])
    assert np.array_equal(np.array([t[1] for t in tr]), g[key + "a"])
    assert np.array_equal(np.array([t[2] for t in tr], dtype=np.uint8), g[key + "c"])
    assert np.array_equal(np.array([t[3] for t in tr]), g[key + "s2"])
    assert np.array_equal(np.array([int(t[4]) for t in tr], dtype=np.uint8), g[key + "m"])


@pytest.mark.parametrize("env", ENVS)
def test_c_oracle_offline_data_distribution(env, offline_golden):
    """The Philox-driven generator cannot share the reference's MT19937 stream; its output must
    agree in distribution with the reference's 20000-transition run and be self-consistent."""
    st = offline_golden[env + "_seed1_n20000_stats"]
    s, a, c, s2, m = co.nav_offline(env, 20000, 1)
    assert abs(len(s) - st[0]) / st[0] < 0.04
    assert abs(c.sum() / len(c) - st[1] / st[0]) < 0.012
    assert np.allclose(s.mean(0), st[2:4], atol=1.5) and np.allclose(s.var(0), st[4:6], rtol=0.08)
    assert np.allclose(a.mean(0), st[6:8], atol=0.03) and np.allclose(a.var(0), st[8:10], rtol=0.05)
    assert np.array_equal(m, 1 - c)
    cons = np.array([co.obstacle(env, float(x), float(y)) for x, y in s2.astype(np.float64)])
    # f32-rounded s' can flip membership only within 1 ulp of an edge
    assert (cons != c).sum() <= 2
    assert np.all(np.abs(a) <= 1)


def test_auto_reset_and_horizon_semantics():
    env = "navigation1"
    n = 64
    pos, obs, t = co.nav_reset(env, n, seed=3, counter=0)
    assert np.allclose(pos.mean(0), [-50, 0], atol=0.6)
    t[:] = 99
    o = co.nav_step(env, pos, np.zeros((n, 2), np.float32), t, seed=3, counter=1, auto_reset=True)
    assert o["ep_done"].all() and not o["done"].any()       # time-out is not terminal (experiment.py:434-435)
    assert (o["t"] == 0).all()
    assert not np.array_equal(o["obs"], o["next_obs"])       # obs is post-reset, next_obs pre-reset
    assert np.array_equal(o["obs"], o["pos"].astype(np.float32))
    o2 = co.nav_step(env, pos, np.zeros((n, 2), np.float32), t, seed=3, counter=1, auto_reset=False)
    assert (o2["t"] == 100).all() and np.array_equal(o2["obs"], o2["next_obs"])


def test_replay_oracle_ring_and_sampling():
    rb = co.OracleReplay(10)
    rows = lambda k, n: (np.full((n, 2), k, np.float32), np.full((n, 2), -k, np.float32),
                         np.arange(n, dtype=np.float32) + 100 * k, np.full((n, 2), k + 0.5, np.float32),
                         np.ones(n, np.float32))
    rb.push(*rows(1, 7))
    assert (len(rb), rb.pos) == (7, 7)
    rb.push(*rows(2, 6))                                      # wraps: slots 7,8,9,0,1,2
    assert (len(rb), rb.pos) == (10, 3)
    assert rb.r[0] == 203 and rb.r[3] == 103 and rb.r[9] == 202
    valid = np.array([1, 0, 1, 0], np.uint8)
    rb.push(*rows(3, 4), valid=valid)                         # rows 0 and 2 -> slots 3,4
    assert rb.pos == 5 and rb.r[3] == 300 and rb.r[4] == 302
    idx = rb.sample_indices(10, seed=5, counter=0)
    assert sorted(idx) == list(range(10))                     # B == size: a permutation
    with pytest.raises(ValueError):
        rb.sample_indices(11, seed=5, counter=0)
    big = co.OracleReplay(5000)
    big.push(*rows(1, 5000))
    hits = np.zeros(5000)
    for c in range(300):
        i = big.sample_indices(256, seed=9, counter=c)
        assert len(set(i)) == 256
        hits[i] += 1
    assert abs(hits.mean() - 300 * 256 / 5000) < 1e-9 and hits.std() < 6.0   # ~binomial(300, .0512)


def test_replay_oracle_stratified_matches_reference_composition(golden_dir):
    """G6: 76 positives first, then 180 negatives at pos_fraction 0.3, B=256
    (replay_memory.py:54-72)."""
    g = np.load(os.path.join(golden_dir, "replay_golden.npz"))
    rb = co.OracleReplay(4096)
    n = len(g["constraint"])
    z = np.zeros((n, 2), np.float32)
    rb.push(z, z, g["constraint"].astype(np.float32), z, np.ones(n, np.float32))
    B, pf = int(g["B"]), float(g["pos_fraction"])
    n_pos = int(B * pf)
    assert n_pos == int(g["n_pos_ref"]) == 76
    idx = rb.sample_stratified_indices(n_pos, B - n_pos, seed=1, counter=0)
    assert len(set(idx)) == B
    assert rb.r[idx[:n_pos]].all() and not rb.r[idx[n_pos:]].any()
    # the reference's own batch has the same composition
    assert g["ref_batch_constraint"][:n_pos].all() and not g["ref_batch_constraint"][n_pos:].any()


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("num", (1000, 250))
def test_c_oracle_offline_data_equals_reference_row_for_row(env, num, golden_dir):
    """The C generator (the restatement the HIP kernel is compared with bit-for-bit) fed the draws the REFERENCE's
    get_offline_data consumed (tests/golden/gen_nav_offline_draws_golden.py): same rollouts, same rejection loop,
    same break-on-constraint, float64 rows equal one for one (navigation1.py:133-164, navigation2.py:133-243)."""
    g = np.load(os.path.join(golden_dir, "nav_offline_draws_golden.npz"))
    pre = "%s_n%d_" % (env, num)
    (s, a, c, s2, m), (s64, a64, s2_64), (nu, nz) = co.nav_offline_explicit(env, num, g[pre + "u"], g[pre + "z"])
    assert (nu, nz) == (len(g[pre + "u"]), len(g[pre + "z"]))             # every draw consumed, none missing
    assert len(s64) == len(g[pre + "s"])
    assert np.array_equal(s64, g[pre + "s"]) and np.array_equal(a64, g[pre + "a"])
    assert np.array_equal(s2_64, g[pre + "s2"])
    assert np.array_equal(c, g[pre + "c"].astype(np.float32)) and np.array_equal(m, g[pre + "m"].astype(np.float32))
    assert np.array_equal(s, g[pre + "s"].astype(np.float32)) and np.array_equal(s2, g[pre + "s2"].astype(np.float32))
    assert np.array_equal(a, g[pre + "a"].astype(np.float32))
    assert g[pre + "c"].sum() > 10


def test_stratified_draw_starvation_rule():
    """Too few positives: the reference's random.sample raises (replay_memory.py:61-66) and so does the oracle; the
    clamped variant (the lock-step loop's rule) takes all positives and fills the batch with negatives."""
    rb = co.OracleReplay(512)
    n = 400
    r = np.zeros(n, np.float32)
    r[[3, 77, 200]] = 1.0
    z = np.zeros((n, 2), np.float32)
    rb.push(z, z, r, z, np.ones(n, np.float32))
    with pytest.raises(ValueError):
        rb.sample_stratified_indices(76, 180, seed=1, counter=0)
    idx, used = rb.sample_stratified_indices(76, 180, seed=1, counter=0, clamp=True, return_split=True)
    assert used == 3 and sorted(idx[:3]) == [3, 77, 200] and len(set(idx)) == 256 and not r[idx[3:]].any()
    with pytest.raises(ValueError):                     # more rows than the ring holds stays an error
        rb.sample_stratified_indices(200, 300, seed=1, counter=0, clamp=True)
    ok = rb.sample_stratified_indices(2, 100, seed=1, counter=0)
    assert np.array_equal(ok, rb.sample_stratified_indices(2, 100, seed=1, counter=0, clamp=True))
