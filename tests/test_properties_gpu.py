"""Property tests over random sizes (hypothesis): the HIP path equals the C checker bit-for-bit wherever the sizes
land -- ragged tails, batch == population, tiny rings, collisions in the dedupe table."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import c_oracle as co
from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
from test_nav_gpu import assert_same, hip_step
from test_replay_gpu import dev, rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COMMON = dict(deadline=None, max_examples=25, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large],
              derandomize=True)


@settings(**COMMON)
@given(size=st.integers(1, 3000), batch=st.integers(1, 1024), seed=st.integers(0, 2 ** 40), calls=st.integers(1, 3))
def test_uniform_sampling_is_the_checkers_for_any_size(size, batch, seed, calls):
    batch = min(batch, size)
    rng = np.random.RandomState(size * 7 + batch)
    cap = size + rng.randint(0, 50)
    mem, ora = ReplayMemory(cap, seed, device=DEV), co.OracleReplay(cap)
    b = rows(rng, size)
    mem.push(*dev(b))
    ora.push(*b)
    for call in range(calls):
        out = mem.sample(batch)
        idx = mem._batch(batch)[5].cpu().numpy()
        ref = ora.sample_indices(batch, seed=mem.seed, counter=call)
        assert np.array_equal(idx, ref)
        assert len(set(idx.tolist())) == batch and idx.min() >= 0 and idx.max() < size
        for got, want in zip(out, ora.gather(ref)):
            assert np.array_equal(got.cpu().numpy(), want)


@settings(**COMMON)
@given(size=st.integers(64, 4000), batch=st.integers(2, 512), pos_rate=st.floats(0.05, 0.6), frac=st.floats(0.0, 1.0),
       seed=st.integers(0, 2 ** 40))
def test_stratified_sampling_is_the_checkers_for_any_composition(size, batch, pos_rate, frac, seed):
    rng = np.random.RandomState(size + batch)
    mem, ora = ConstraintReplayMemory(size + 17, seed, device=DEV), co.OracleReplay(size + 17)
    b = rows(rng, size, pos_rate=pos_rate)
    mem.push(*dev(b))
    ora.push(*b)
    n_pos_avail = int((b[2] != 0).sum())
    n_pos = min(int(batch * frac), n_pos_avail)
    n_neg = min(batch - n_pos, size - n_pos_avail)
    if n_pos + n_neg == 0:
        return
    fraction = n_pos / (n_pos + n_neg)
    total = n_pos + n_neg
    if int(total * fraction) != n_pos:          # int(B * pos_fraction) is how both sides split the batch
        return
    out = mem.sample(total, pos_fraction=fraction)
    idx = mem._batch(total)[5].cpu().numpy()
    ref = ora.sample_stratified_indices(n_pos, n_neg, seed=mem.seed, counter=0)
    assert np.array_equal(idx, ref)
    assert len(set(idx.tolist())) == total
    assert np.all(b[2][idx[:n_pos]] != 0) and np.all(b[2][idx[n_pos:]] == 0)
    for got, want in zip(out, ora.gather(ref)):
        assert np.array_equal(got.cpu().numpy(), want)


@settings(**COMMON)
@given(n=st.integers(1, 70000), env=st.sampled_from(["navigation1", "navigation2"]), seed=st.integers(0, 2 ** 50),
       counter=st.integers(0, 2 ** 40), auto=st.booleans())
def test_nav_step_is_the_checkers_for_any_batch(n, env, seed, counter, auto):
    rng = np.random.RandomState(n)
    pos = np.c_[rng.uniform(-60, 10, n), rng.uniform(-12, 12, n)]
    act = rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    t = rng.randint(0, 100, n).astype(np.int32)
    assert_same(hip_step(env, pos, act, t, seed=seed, counter=counter, auto_reset=auto),
                co.nav_step(env, pos, act, t, seed=seed, counter=counter, auto_reset=auto))
