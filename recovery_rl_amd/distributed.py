"""Multi-GPU fan-out: replicas only.  One process per GPU, each rank runs
its own seed / env shard, replay and networks; the ONLY collective on the path is a small
all-reduce(sum) of the counter vector at logging cadence (RCCL via torch.distributed backend
"nccl" on the GPU box, gloo in the CPU tests)."""
import os

import torch

METRIC_KEYS = ("env_steps", "episodes", "num_viols", "viol_and_recovery", "viol_and_no_recovery",
               "num_successes", "recovery_steps", "constraint_steps", "sac_updates", "qrisk_updates",
               "reward_sum", "episode_return_sum")


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


_FORCED = False       # a 1-rank process group was asked for: the collectives below then really run


def init(backend=None, force=None):
    """Initialise torch.distributed from the torchrun environment (no-op for one process, unless `force` or
    RRL_DIST_FORCE_INIT=1 asks for a 1-rank process group: then every collective of this module goes through the
    backend -- RCCL on the GPU box -- exactly as it does with 8 ranks; used to prove the RCCL path on a 1-GPU box)."""
    global _FORCED
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if force is None:
        force = os.environ.get("RRL_DIST_FORCE_INIT", "") not in ("", "0")
    if world == 1 and force:
        _FORCED = True
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("RRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def local_device(local_rank):
    """cuda:<local_rank>; with RRL_DIST_BACKEND=gloo several ranks may share one GPU (dry runs of the
    multi-rank path on a single-GPU box), so the index wraps around the visible devices."""
    n = torch.cuda.device_count()
    return torch.device("cuda", local_rank % n if n else 0)


def rank_seed(base_seed, rank):
    """Rank g runs seed base+g -- the reference's unit of parallelism is the seed loop
    (scripts/navigation1.sh:4-8)."""
    return int(base_seed) + int(rank)


def aggregate_stats(stats, world_size, device):
    """Sum the metric vector over ranks: one 96-byte all-reduce (latency-bound; xGMI bandwidth is
    irrelevant at this size, so it runs at logging cadence, never per step)."""
    import torch.distributed as dist
    if not _active(world_size):
        return dict(stats)
    vec = torch.tensor([float(stats[k]) for k in METRIC_KEYS], dtype=torch.float64, device=device)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    out = dict(stats)
    for k, v in zip(METRIC_KEYS, vec.tolist()):
        out[k] = int(round(v)) if k not in ("reward_sum", "episode_return_sum") else v
    return out


def max_over_ranks(value, world_size, device):
    import torch.distributed as dist
    if not _active(world_size):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(world_size):
    import torch.distributed as dist
    if _active(world_size):
        dist.barrier()


def _active(world_size):
    import torch.distributed as dist
    return (world_size > 1 or _FORCED) and dist.is_initialized()


def backend_name(world_size=1):
    """'nccl' (= RCCL on ROCm) / 'gloo' when collectives run, None for a plain single process."""
    import torch.distributed as dist
    return dist.get_backend() if _active(world_size) else None


def shutdown(barrier=True):
    """Tear the process group down (RCCL communicators, the rendezvous store) before the interpreter exits: without it
    torch warns at exit and a rank can be killed while its peers still wait in the communicator's destructor.
    barrier=False after an error on this rank (the peers are not at the barrier)."""
    global _FORCED
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        try:
            if barrier:
                dist.barrier()
        finally:
            dist.destroy_process_group()
    _FORCED = False
