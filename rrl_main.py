"""`python -m rrl_main --env-name navigation1 --cuda ...` -- same entry point as the reference
(rrl_main.py:1-9).  Under torchrun each rank runs seed + rank on its own GPU."""
from arg_utils import get_args
from recovery_rl_amd import runtime
# the runtime mode bench.py times (hipGraph replay through the regular command path; RRL_GRAPH_PACKET_CAPTURE=1 or an explicit
# DEBUG_CLR_GRAPH_PACKET_CAPTURE in the environment wins) -- before anything touches the GPU
runtime.configure(graph_packet_capture=runtime.LAUNCHER_GRAPH_PACKET_CAPTURE)
from recovery_rl_amd import distributed as dist_utils
from recovery_rl_amd.experiment import Experiment

if __name__ == '__main__':
    exp_cfg = get_args()
    rank, local_rank, world = dist_utils.init()
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(dist_utils.local_device(local_rank))
    packed = getattr(exp_cfg, "seeds_per_gpu", 1)
    if world > 1:
        # env, replay and noise streams differ per rank (a rank that packs S seeds takes S consecutive ones)
        exp_cfg.seed = dist_utils.rank_seed(exp_cfg.seed, rank * max(packed, 1))
    ok = False
    try:
        if packed > 1:
            from recovery_rl_amd.experiment import run_packed
            run_packed(exp_cfg, rank=rank, world_size=world)
        else:
            experiment = Experiment(exp_cfg, rank=rank, world_size=world)
            experiment.run()
        ok = True
    finally:
        dist_utils.shutdown(barrier=ok)
