"""CPU suite: the N>1 path (replicas + metric all-reduce) with world_size 2 on gloo."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from recovery_rl_amd import distributed as du


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = du.init(backend="gloo")
    assert (r, w) == (rank, world)
    stats = {k: (rank + 1) * (i + 1) for i, k in enumerate(du.METRIC_KEYS)}
    stats["reward_sum"] = -1.5 * (rank + 1)
    stats["episode_return_sum"] = -100.25 * (rank + 1)
    agg = du.aggregate_stats(stats, w, torch.device("cpu"))
    mx = du.max_over_ranks(0.5 + rank, w, torch.device("cpu"))
    du.barrier(w)
    q.put((rank, agg, mx, du.rank_seed(1, rank)))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_metric_all_reduce_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, agg, mx, seed in res:
        for i, k in enumerate(du.METRIC_KEYS[:10]):
            assert agg[k] == 3 * (i + 1) and isinstance(agg[k], int)
        assert agg["reward_sum"] == pytest.approx(-4.5)
        assert agg["episode_return_sum"] == pytest.approx(-300.75)
        assert mx == 1.5
        assert seed == 1 + rank


def test_single_process_is_a_no_op():
    stats = {k: i for i, k in enumerate(du.METRIC_KEYS)}
    assert du.aggregate_stats(stats, 1, torch.device("cpu")) == stats
    assert du.max_over_ranks(2.5, 1, torch.device("cpu")) == 2.5
    du.barrier(1)
