"""GPU parity tests for the device replay kernels vs the C oracle (bit-exact rows and indices)
and vs the reference's golden semantics (tests/golden/replay_golden.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rows(rng, n, pos_rate=0.0):
    s = rng.randn(n, 2).astype(np.float32)
    a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    r = (rng.uniform(size=n) < pos_rate).astype(np.float32) if pos_rate else rng.randn(n).astype(np.float32)
    s2 = rng.randn(n, 2).astype(np.float32)
    m = (rng.uniform(size=n) < 0.9).astype(np.float32)
    return s, a, r, s2, m


def dev(xs):
    return [torch.as_tensor(x, device=DEV) for x in xs]


def assert_buffers_equal(mem, ora):
    assert len(mem) == len(ora) and mem.position == ora.pos
    for got, ref in ((mem.s, ora.s), (mem.a, ora.a), (mem.r, ora.r), (mem.s2, ora.s2), (mem.m, ora.m)):
        assert np.array_equal(got.cpu().numpy(), ref)


def test_ring_semantics_rowwise(golden_dir):
    g = np.load(os.path.join(golden_dir, "replay_golden.npz"))
    mem = ReplayMemory(10, 3, device=DEV)
    for lo, hi in ((0, 7), (7, 13)):
        i = np.arange(lo, hi, dtype=np.float32)
        mem.push(*dev([np.c_[i, i], np.c_[-i, -i], 100 + i, np.c_[i + .5, i + .5], (i % 2).astype(np.float32)]))
    assert len(mem) == int(g["ring_len"]) and mem.position == int(g["ring_position"])
    assert np.array_equal(mem.r.cpu().numpy(), g["ring_rewards_by_slot"].astype(np.float32))
    assert np.array_equal(mem.m.cpu().numpy(), g["ring_masks_by_slot"].astype(np.float32))
    with pytest.raises(ValueError):
        mem.sample(11)                       # random.sample raises (golden: oversample_raises)
    assert int(g["oversample_raises"]) == 1
    big = rows(np.random.RandomState(0), 13)
    with pytest.raises(Exception):
        mem.push(*dev(big))                  # more rows than capacity in one call is rejected


@pytest.mark.parametrize("cap,chunks", ((1000, (300, 300, 300, 300, 1000)), (5000, (4096, 4096, 7)),
                                        (70000, (65536, 4096, 4096))))
def test_push_matches_oracle_including_wraparound(cap, chunks):
    rng = np.random.RandomState(cap)
    mem, ora = ConstraintReplayMemory(cap, 1, device=DEV), co.OracleReplay(cap)
    for n in chunks:
        b = rows(rng, n, pos_rate=0.07)
        mem.push(*dev(b))
        ora.push(*b)
        assert_buffers_equal(mem, ora)
        # per-chunk positive counts stay exact
        filled = ora.r[:len(ora)] != 0
        ref_cnt = np.add.reduceat(np.r_[filled, np.zeros((-len(filled)) % 64, bool)].astype(np.int32),
                                  np.arange(0, len(filled), 64)) if len(filled) else np.zeros(0)
        got = mem.pos_cnt.cpu().numpy()
        n_chunks = (cap + 63) // 64
        base = (n_chunks + 3) // 4 * 4                     # second level: per-1024-slot counts (RRL_POS_CNT_LEN)
        n_super = (cap + 1023) // 1024
        mbase = (base + n_super + 1) // 2 * 2              # third region: one 64-bit mask per chunk
        assert len(got) == mbase + 2 * n_chunks
        assert np.array_equal(got[:len(ref_cnt)], ref_cnt) and not got[len(ref_cnt):base].any()
        first = np.r_[got[:n_chunks], np.zeros((-n_chunks) % 16, np.int32)]
        assert np.array_equal(got[base:base + n_super], first.reshape(-1, 16).sum(1))
        masks = got[mbase:].view(np.uint64)
        want = np.zeros(n_chunks * 64, bool)
        want[:len(filled)] = filled
        assert np.array_equal(masks, np.packbits(want.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).ravel())


def assert_count_tables(mem, ora, cap):
    """chunk counts, super-chunk counts and slot masks of `mem.pos_cnt` against the oracle's rows"""
    filled = ora.r[:len(ora)] != 0
    n_chunks, n_super = (cap + 63) // 64, (cap + 1023) // 1024
    base = (n_chunks + 3) // 4 * 4
    mbase = (base + n_super + 1) // 2 * 2
    got = mem.pos_cnt.cpu().numpy()
    want = np.zeros(n_chunks * 64, bool)
    want[:len(filled)] = filled
    assert np.array_equal(got[:n_chunks], want.reshape(-1, 64).sum(1))
    first = np.r_[got[:n_chunks], np.zeros((-n_chunks) % 16, np.int32)]
    assert np.array_equal(got[base:base + n_super], first.reshape(-1, 16).sum(1))
    assert np.array_equal(got[mbase:].view(np.uint64),
                          np.packbits(want.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).ravel())


def test_masked_push_matches_oracle():
    rng = np.random.RandomState(5)
    mem, ora = ConstraintReplayMemory(9000, 1, device=DEV), co.OracleReplay(9000)
    for n in (4096, 4096, 4096, 1, 2500):
        b = rows(rng, n, pos_rate=0.2)
        valid = (rng.uniform(size=n) < 0.37).astype(np.uint8)
        mem.push(*dev(b), valid=torch.as_tensor(valid, device=DEV))
        ora.push(*b, valid=valid)
        assert_buffers_equal(mem, ora)
        assert_count_tables(mem, ora, 9000)        # masked rows land in non-consecutive lanes: the per-lane mask path
    b = rows(rng, 64, pos_rate=0.2)
    mem.push(*dev(b), valid=torch.zeros(64, dtype=torch.uint8, device=DEV))   # nothing valid
    assert_buffers_equal(mem, ora)


@pytest.mark.parametrize("B", (1, 64, 100, 256, 1024))
def test_uniform_sample_indices_and_rows_match_oracle(B):
    rng = np.random.RandomState(B)
    cap = 50000
    mem, ora = ReplayMemory(cap, 77, device=DEV), co.OracleReplay(cap)
    b = rows(rng, 30000)
    mem.push(*dev(b))
    ora.push(*b)
    for call in range(4):
        s, a, r, s2, m = mem.sample(B)
        idx = mem._batch(B)[5].cpu().numpy()
        ref_idx = ora.sample_indices(B, seed=77, counter=call)
        assert np.array_equal(idx, ref_idx)
        assert len(set(idx)) == B
        for got, ref in zip((s, a, r, s2, m), ora.gather(ref_idx)):
            assert np.array_equal(got.cpu().numpy(), ref)
    assert int(mem.tick[0].item()) == 4


def test_sample_when_batch_equals_population_is_a_permutation():
    mem = ReplayMemory(300, 5, device=DEV)
    rng = np.random.RandomState(0)
    mem.push(*dev(rows(rng, 256)))
    mem.sample(256)
    idx = mem._batch(256)[5].cpu().numpy()
    assert sorted(idx) == list(range(256))
    ora = co.OracleReplay(300)
    ora.push(*rows(np.random.RandomState(0), 256))
    assert np.array_equal(idx, ora.sample_indices(256, seed=5, counter=0))


def test_device_error_flag_when_oversampling_after_masked_push():
    mem = ReplayMemory(300, 5, device=DEV)
    rng = np.random.RandomState(0)
    valid = torch.zeros(100, dtype=torch.uint8, device=DEV)
    valid[:10] = 1
    mem.push(*dev(rows(rng, 100)), valid=valid)
    mem._len_exact = False                       # host does not know the size: the device must flag
    mem.sample(64)
    with pytest.raises(ValueError):
        mem.check_error()


def test_stratified_sample_matches_oracle_and_reference_composition(golden_dir):
    g = np.load(os.path.join(golden_dir, "replay_golden.npz"))
    n = len(g["constraint"])
    z = np.zeros((n, 2), np.float32)
    z[:, 0] = np.arange(n)
    b = (z, z, g["constraint"].astype(np.float32), z, np.ones(n, np.float32))
    mem, ora = ConstraintReplayMemory(4096, 1, device=DEV), co.OracleReplay(4096)
    mem.push(*dev(b))
    ora.push(*b)
    B, pf = int(g["B"]), float(g["pos_fraction"])
    for call in range(3):
        s, a, r, s2, m = mem.sample(B, pos_fraction=pf)
        idx = mem._batch(B)[5].cpu().numpy()
        ref = ora.sample_stratified_indices(76, B - 76, seed=mem.seed, counter=call)
        assert np.array_equal(idx, ref)
        r = r.cpu().numpy()
        assert r[:76].all() and not r[76:].any() and len(set(idx)) == B       # G6 composition
        assert np.array_equal(s.cpu().numpy()[:, 0], idx.astype(np.float32))
    mem.check_error()


@pytest.mark.parametrize("n_positive,n_rows", ((20, 3000), (0, 500), (2990, 3000), (76, 3000)))
def test_starved_stratified_draw_flags_by_default_and_clamps_on_request(n_positive, n_rows):
    """Fewer positives (or negatives) than int(B * pos_fraction) asks for: the reference's random.sample raises
    ValueError (replay_memory.py:61-66) -> error flag; with clamp_stratified (the lock-step loop's rule) the short class
    gives all its rows and the other fills the batch, bit-equal to the checker's clamped draw."""
    rng = np.random.RandomState(n_positive)
    b = list(rows(rng, n_rows, pos_rate=0.0))
    b[2][:] = 0.0
    b[2][rng.permutation(n_rows)[:n_positive]] = 1.0
    B, n_pos = 256, 76
    starved = n_positive < n_pos or n_rows - n_positive < B - n_pos
    mem, ora = ConstraintReplayMemory(4096, 5, device=DEV), co.OracleReplay(4096)
    mem.push(*dev(b))
    ora.push(*b)
    out0 = [t.clone() for t in mem.sample(B, pos_fraction=0.3)]
    if starved:
        with pytest.raises(ValueError):
            mem.check_error()
        with pytest.raises(ValueError):
            ora.sample_stratified_indices(n_pos, B - n_pos, seed=mem.seed, counter=0)
        mem.state[3] = 0
        mem.tick.zero_()
    else:
        mem.check_error()
    mem2 = ConstraintReplayMemory(4096, 5, device=DEV)
    mem2.push(*dev(b))
    mem2.clamp_stratified = True
    assert mem2.clamp_stratified and not mem.clamp_stratified
    s, a, r, s2, m = mem2.sample(B, pos_fraction=0.3)
    mem2.check_error()
    idx = mem2._batch(B)[5].cpu().numpy()
    ref, used = ora.sample_stratified_indices(n_pos, B - n_pos, seed=mem2.seed, counter=0, clamp=True, return_split=True)
    assert np.array_equal(idx, ref) and len(set(idx)) == B
    assert used == (min(n_pos, n_positive) if n_positive < n_pos else max(n_pos, B - (n_rows - n_positive)))
    rr = r.cpu().numpy()
    assert rr[:used].all() and not rr[used:].any()
    if not starved:                                   # a feasible draw is unchanged by the flag
        assert all(torch.equal(x, y) for x, y in zip(out0, (s, a, r, s2, m)))


def test_stratified_sample_on_wrapped_million_row_buffer():
    """Reference default capacity 1e6, pushed past wrap-around in 4096-row vector steps."""
    cap = 1000000
    rng = np.random.RandomState(1)
    mem, ora = ConstraintReplayMemory(cap, 3, device=DEV), co.OracleReplay(cap)
    for k in range(4):
        n = 300000
        b = rows(rng, n, pos_rate=0.03)
        mem.push(*dev(b))
        ora.push(*b)
    assert len(mem) == cap and mem.position == ora.pos
    s, a, r, s2, m = mem.sample(256, pos_fraction=0.3)
    idx = mem._batch(256)[5].cpu().numpy()
    assert np.array_equal(idx, ora.sample_stratified_indices(76, 180, seed=mem.seed, counter=0))
    assert np.array_equal(r.cpu().numpy(), ora.r[idx])
    mem.check_error()


def test_sampling_is_uniform():
    mem = ReplayMemory(4096, 9, device=DEV)
    mem.push(*dev(rows(np.random.RandomState(0), 4096)))
    hits = torch.zeros(4096, device=DEV)
    for _ in range(400):
        mem.sample(256)
        hits[mem._batch(256)[5]] += 1
    h = hits.cpu().numpy()
    assert abs(h.mean() - 25.0) < 1e-6 and 3.5 < h.std() < 6.5    # binomial(400, 1/16): std 4.84


@pytest.mark.parametrize("cap,pinned,n", [(5000, 1200, 900), (4096, 1024, 1024), (100424, 20000, 40000)])
def test_pinned_rows_are_never_overwritten(cap, pinned, n):
    """rrl_replay_t.pinned (the lock-step loop pins the offline demonstrations): past the last slot the ring continues at
    slot `pinned`.  Rows, cursor, size and the three count tables equal the oracle's over several wraps; plain and masked
    pushes; the stratified draw still finds the pinned positives."""
    rng = np.random.RandomState(cap)
    mem, ora = ConstraintReplayMemory(cap, 1, device=DEV), co.OracleReplay(cap)

    def rows(k, pos_rate):
        s, a, s2 = (rng.randn(k, 2).astype(np.float32) for _ in range(3))
        return s, a, (rng.uniform(size=k) < pos_rate).astype(np.float32), s2, rng.randint(0, 2, k).astype(np.float32)

    first = rows(pinned, 0.5)
    mem.push(*dev(first))
    ora.push(*first)
    mem.pin()
    ora.pin()
    assert mem.pinned == pinned == int(ora._c.pinned)
    for k in range(2 * (cap - pinned) // n + 3):
        batch = rows(n, 0.0)                       # the online rows carry no violation at all
        valid = (rng.uniform(size=n) < 0.7).astype(np.uint8) if k % 3 == 2 else None
        mem.push(*dev(batch), valid=None if valid is None else dev([valid])[0])
        ora.push(*batch, valid=valid)
        assert int(mem.state[0].item()) == ora.pos and int(mem.state[1].item()) == ora.size
    assert ora.size == cap and ora.pos >= pinned
    for a_, b_ in ((mem.s, ora.s), (mem.a, ora.a), (mem.r, ora.r), (mem.s2, ora.s2), (mem.m, ora.m)):
        assert np.array_equal(a_.cpu().numpy(), b_)
    assert np.array_equal(mem.r.cpu().numpy()[:pinned], first[2]) and not mem.r.cpu().numpy()[pinned:].any()
    assert_count_tables(mem, ora, cap)
    n_pos = int(first[2].sum())
    s, a, c, s2, m = mem.sample(64, pos_fraction=0.25)
    c = c.reshape(-1).cpu().numpy()
    assert n_pos >= 16 and (c[:16] == 1).all() and (c[16:] == 0).all()
    mem.check_error()


@pytest.mark.parametrize("cap,pinned,online,B,share", [
    (100424, 20000, 40000, 256, 0.5),     # the lock-step loop's default: 128 demonstrations + 128 online rows
    (100424, 20000, 90000, 256, 0.5),     # ring wrapped: the online range is [pinned, cap)
    (5000, 1200, 100, 256, 0.5),          # online range short: it gives all 100 rows, the demonstrations fill the batch
    (5000, 100, 3000, 256, 0.5),          # demonstration range short: all 100 of them, 156 online rows
    (5000, 1200, 0, 256, 0.5),            # no online row yet (pre-training): the whole batch from the demonstrations
    (5000, 1200, 2000, 1024, 0.25),
    (5000, 1200, 2000, 7, 0.3),
])
def test_demo_share_draw_matches_oracle(cap, pinned, online, B, share):
    """rrl_replay_sample_gather_split (vectorisation rule for the safety critic's batch): indices bit-equal to the C
    checker's, first n_demo rows from [0, pinned), the rest from [pinned, size); stand-alone entry and as a member of
    rrl_sample_multi; three consecutive calls (the tick advances)."""
    import ctypes as C
    from recovery_rl_amd import _lib
    rng = np.random.RandomState(cap + online)
    mem, ora = ConstraintReplayMemory(cap, 11, device=DEV), co.OracleReplay(cap)
    first = rows(rng, pinned, pos_rate=0.08)
    mem.push(*dev(first))
    ora.push(*first)
    mem.pin()
    ora.pin()
    left = online
    while left > 0:
        k = min(left, 4096)
        b = rows(rng, k, pos_rate=0.01)
        mem.push(*dev(b))
        ora.push(*b)
        left -= k
    n_demo = int(B * share)
    size = len(ora)
    demo_total = min(pinned, size)
    exp_demo = n_demo
    if B - n_demo > size - demo_total:
        exp_demo = B - (size - demo_total)
    elif n_demo > demo_total:
        exp_demo = demo_total
    for call in range(3):
        ref, used = ora.sample_split_indices(n_demo, B - n_demo, seed=mem.seed, counter=call, return_split=True)
        assert used == exp_demo
        if call == 1:                          # the grouped entry point draws the same rows
            d, batch = mem.draw_desc(B, demo_share=share)
            lib = _lib.load()
            _lib.check(lib.rrl_sample_multi(C.byref(d), None, 0, 0, 0, None, 0, None, _lib.current_stream()), "multi")
            s, a, r, s2, m = batch
        else:
            s, a, r, s2, m = mem.sample(B, demo_share=share)
        mem.check_error()
        idx = mem._batch(B)[5].cpu().numpy()
        assert np.array_equal(idx, ref) and len(set(idx.tolist())) == B
        assert (idx[:used] < pinned).all() and (idx[used:] >= pinned).all() and (idx < size).all()
        gs = ora.gather(ref)
        for got, want in zip((s, a, r, s2, m), gs):
            assert np.array_equal(got.cpu().numpy(), want)


def test_demo_share_draw_oversample_flags():
    mem = ConstraintReplayMemory(4096, 1, device=DEV)
    mem.push(*dev(rows(np.random.RandomState(0), 100)))
    mem.pin()
    mem._len_exact, mem._len = True, 4096         # bypass the host guard: the device flag is what is tested
    mem.sample(256, demo_share=0.5)
    with pytest.raises(ValueError):
        mem.check_error()


def test_demo_share_keeps_the_demonstrations_in_the_batch():
    """What the rule is for: after the ring has wrapped, a uniform draw shows ~2 % demonstration rows, the split draw
    exactly int(B * share), uniformly over the demonstrations."""
    cap, pinned = 100000, 2000
    rng = np.random.RandomState(3)
    mem = ConstraintReplayMemory(cap, 2, device=DEV)
    mem.push(*dev(rows(rng, pinned, pos_rate=0.08)))
    mem.pin()
    for _ in range(30):
        mem.push(*dev(rows(rng, 4096)))
    hits = torch.zeros(cap, device=DEV)
    uni = 0
    for _ in range(300):
        mem.sample(256)
        uni += int((mem._batch(256)[5] < pinned).sum())
        mem.sample(256, demo_share=0.5)
        hits[mem._batch(256)[5]] += 1
    h = hits.cpu().numpy()
    assert h[:pinned].sum() == 300 * 128 and h[pinned:].sum() == 300 * 128
    assert uni / (300 * 256) < 0.04
    assert 3.0 < h[:pinned].std() < 5.5            # binomial(300, 128 / 2000): std 4.24
