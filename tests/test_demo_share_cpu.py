"""The demonstration-share draw of the C checker (oracle/rrl_oracle.c rrl_oracle_sample_split): the vectorisation rule the
lock-step loop uses for the safety critic's batch.  CPU only: properties of the checker, the host-side rule that turns
--demo_share into the draw, and the one-env default (off, G9 semantics untouched)."""
import numpy as np
import pytest

from oracle import c_oracle as co


def filled(cap, pinned, online, seed=0):
    rng = np.random.RandomState(seed)
    ora = co.OracleReplay(cap)

    def rows(k):
        z = rng.randn(k, 2).astype(np.float32)
        return z, z, (rng.uniform(size=k) < 0.1).astype(np.float32), z, np.ones(k, np.float32)
    if pinned:
        ora.push(*rows(pinned))
        ora.pin()
    while online > 0:
        k = min(online, 4096)
        ora.push(*rows(k))
        online -= k
    return ora


@pytest.mark.parametrize("cap,pinned,online,n_demo,n_online,want_demo", [
    (50000, 20000, 25000, 128, 128, 128),
    (50000, 20000, 45000, 128, 128, 128),        # wrapped
    (5000, 1200, 100, 128, 128, 156),            # online range short
    (5000, 100, 3000, 128, 128, 100),            # demonstration range short
    (5000, 1200, 0, 128, 128, 256),              # pre-training: demonstrations only
    (5000, 0, 3000, 128, 128, 0),                # nothing pinned: everything from the online range
])
def test_split_draw_properties(cap, pinned, online, n_demo, n_online, want_demo):
    ora = filled(cap, pinned, online)
    B = n_demo + n_online
    for counter in range(4):
        idx, used = ora.sample_split_indices(n_demo, n_online, seed=9, counter=counter, return_split=True)
        assert used == want_demo
        assert len(set(idx.tolist())) == B and idx.min() >= 0 and idx.max() < len(ora)
        assert (idx[:used] < pinned).all() and (idx[used:] >= pinned).all()
    again = ora.sample_split_indices(n_demo, n_online, seed=9, counter=3)
    assert np.array_equal(idx, again)                           # a function of (seed, counter) only


def test_split_draw_groups_are_the_uniform_draws_of_their_ranges():
    """demo group = rrl_oracle_sample_indices over [0, pinned) on the positive stream; online group = the same over
    size - pinned rows on the negative stream, shifted by pinned"""
    import ctypes as C
    ora = filled(50000, 20000, 12345)
    idx = ora.sample_split_indices(100, 156, seed=5, counter=7)
    lib = co.lib()
    a, b = np.zeros(100, np.int64), np.zeros(156, np.int64)
    assert lib.rrl_oracle_sample_indices(C.c_int64(20000), C.c_int32(100), C.c_uint64(5), C.c_uint64(7),
                                         C.c_uint32(co.STREAM_SAMPLE), a.ctypes.data_as(C.c_void_p)) == 0
    assert lib.rrl_oracle_sample_indices(C.c_int64(12345), C.c_int32(156), C.c_uint64(5), C.c_uint64(7),
                                         C.c_uint32(co.STREAM_SAMPLE_NEG), b.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(idx[:100], a) and np.array_equal(idx[100:], b + 20000)


def test_split_draw_oversample_raises():
    ora = filled(5000, 100, 50)
    with pytest.raises(ValueError):
        ora.sample_split_indices(128, 128, seed=1, counter=0)


def test_demo_share_flag_defaults():
    import arg_utils
    a = arg_utils.get_args(["--env-name", "navigation2", "--use_recovery"])
    assert a.demo_share == -1.0 and a.num_envs == 1
    a = arg_utils.get_args(["--env-name", "navigation2", "--use_recovery", "--num_envs", "4096", "--demo_share", "0.25"])
    assert a.demo_share == 0.25


def test_buffers_cover_the_run_at_many_envs_only():
    """Vectorisation rule 4 (experiment.replay_capacities): one env -> the reference's capacities untouched; N > 1 -> both buffers
    hold the run's step budget (the reference's defaults: replay_size == num_steps), the stratified sampler's 2^21-row bound
    respected, --keep_replay_size restores the rings."""
    import arg_utils
    from recovery_rl_amd.experiment import replay_capacities
    one = arg_utils.get_args(["--env-name", "navigation2", "--use_recovery", "--num_steps", "5000000"])
    assert replay_capacities(one) == (1000000, 1000000)
    n, steps = 4096, 4096 * 1650
    many = arg_utils.get_args(["--env-name", "navigation2", "--use_recovery", "--num_envs", str(n), "--num_steps", str(steps),
                               "--num_unsafe_transitions", "20000"])
    cap, safe = replay_capacities(many)
    assert cap == steps + 2 * n and safe == steps + 2 * n + 20000
    short = arg_utils.get_args(["--env-name", "navigation1", "--num_envs", "64", "--num_steps", "5000"])
    assert replay_capacities(short) == (1000000, 1000000)                 # a run shorter than the buffers: nothing to raise
    default = arg_utils.get_args(["--env-name", "navigation1", "--num_envs", "4096"])
    assert replay_capacities(default) == (1000000, 1000000)               # num_steps == capacities: the reference's defaults
    maze = arg_utils.get_args(["--env-name", "maze", "--num_envs", str(n), "--num_steps", str(steps), "--pos_fraction=0.3"])
    assert replay_capacities(maze) == (steps + 2 * n, 1 << 21)
    keep = arg_utils.get_args(["--env-name", "navigation2", "--num_envs", str(n), "--num_steps", str(steps), "--keep_replay_size"])
    assert replay_capacities(keep) == (1000000, 1000000)
