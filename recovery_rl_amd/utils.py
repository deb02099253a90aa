"""Small helpers of the hot path (reference: recovery_rl/utils.py:46-64)."""
import torch


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
