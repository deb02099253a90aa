"""Small helpers of the hot path (reference: recovery_rl/utils.py:46-64)."""
import os

import torch

# RRL_ROCTX=1: roctx ranges (torch.cuda.nvtx maps to roctx on ROCm) around the stages of the lock-step iteration, so a
# `rocprofv3 --marker-trace --kernel-trace` run of the EAGER loop (bench.py --no_graph) attributes kernels to
# sample / sac_update / qrisk_update / act / env_step / cem without name matching.  Off by default: a range is two host
# calls per stage, and ranges inside a captured hipGraph mean nothing (the graph is one launch).
TRACE = os.environ.get("RRL_ROCTX", "") not in ("", "0")


class trace_range:
    """with trace_range("env_step"): ...  -- a roctx range when RRL_ROCTX=1, nothing otherwise."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TRACE:
            torch.cuda.nvtx.range_push(self.name)
        return self

    def __exit__(self, *exc):
        if TRACE:
            torch.cuda.nvtx.range_pop()
        return False


@torch.no_grad()
def soft_update(target, source, tau):
    """theta' <- (1 - tau) theta' + tau theta over parameters() (utils.py:46-49), as ONE fused
    foreach launch instead of one small kernel pair per tensor."""
    tp = [p.data for p in target.parameters()]
    sp = [p.data for p in source.parameters()]
    torch._foreach_mul_(tp, 1.0 - tau)
    torch._foreach_add_(tp, sp, alpha=tau)


@torch.no_grad()
def hard_update(target, source):
    """theta' <- theta (utils.py:52-54)."""
    for tp, sp in zip(target.parameters(), source.parameters()):
        tp.data.copy_(sp.data)


def linear_schedule(startval, endval, endtime):
    """nu schedule for RSPO (utils.py:62-64)."""
    def value(t):
        if t < endtime:
            return startval + t / endtime * (endval - startval)
        return endval
    return value
