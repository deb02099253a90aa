"""Cross-entropy-method optimiser (reference: recovery_rl/optimizers.py:28-124), batched over M
independent problems and resident on the GPU: sampling and the elite update are the HIP kernels
rrl_cem_sample / rrl_cem_update; the cost function is called once per iteration on
[M, popsize, sol_dim]."""
import numpy as np
import torch

from . import _lib


class Optimizer:
    def reset(self):
        raise NotImplementedError("Must be implemented in subclass.")

    def obtain_solution(self, *args, **kwargs):
        raise NotImplementedError("Must be implemented in subclass.")


class CEMOptimizer(Optimizer):
    def __init__(self, sol_dim, max_iters, popsize, num_elites, cost_function, upper_bound=None,
                 lower_bound=None, epsilon=0.001, alpha=0.25, device="cuda", seed=0):
        super().__init__()
        self.sol_dim, self.max_iters, self.popsize, self.num_elites = sol_dim, max_iters, popsize, num_elites
        self.epsilon, self.alpha = epsilon, alpha
        self.cost_function = cost_function
        if num_elites > popsize:                                       # optimizers.py:66-68
            raise ValueError("Number of elites must be at most the population size.")
        self.device = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.ub = torch.as_tensor(np.asarray(upper_bound, dtype=np.float64), device=self.device)
        self.lb = torch.as_tensor(np.asarray(lower_bound, dtype=np.float64), device=self.device)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.tick = torch.zeros(2, dtype=torch.int64, device=self.device)

    def reset(self):
        pass

    def obtain_solution(self, init_mean, init_var, iters=None):
        """init_mean / init_var: [M, sol_dim] (or a single 1-D numpy problem).  Returns the final
        mean, same container type.  Iterates while t < max_iters and max(var) > epsilon, per problem
        (optimizers.py:93-94)."""
        single = not torch.is_tensor(init_mean)
        if single:
            init_mean = torch.as_tensor(np.asarray(init_mean, dtype=np.float64)[None], device=self.device)
            init_var = torch.as_tensor(np.asarray(init_var, dtype=np.float64)[None], device=self.device)
        mean = init_mean.to(torch.float64).contiguous().clone()
        var = init_var.to(torch.float64).contiguous().clone()
        M, dim = mean.shape
        assert dim == self.sol_dim
        samples = torch.zeros(M, self.popsize, dim, dtype=torch.float32, device=self.device)
        active = torch.ones(M, dtype=torch.uint8, device=self.device)
        stream = _lib.current_stream()
        for _ in range(self.max_iters if iters is None else iters):
            rc = self.lib.rrl_cem_sample(M, self.popsize, dim, _lib.ptr(mean), _lib.ptr(var),
                                         _lib.ptr(self.lb), _lib.ptr(self.ub), self.epsilon, 1,
                                         _lib.ptr(active), self.seed, 0, _lib.ptr(self.tick), 1,
                                         _lib.ptr(samples), stream)
            _lib.check(rc, "rrl_cem_sample")
            costs = self.cost_function(samples).to(torch.float32).contiguous()
            rc = self.lib.rrl_cem_update(M, self.popsize, dim, self.num_elites, self.alpha,
                                         _lib.ptr(samples), _lib.ptr(costs), _lib.ptr(mean), _lib.ptr(var),
                                         _lib.ptr(active), stream)
            _lib.check(rc, "rrl_cem_update")
        return mean[0].cpu().numpy() if single else mean

    def obtain_solution_n(self, ws, count, iters=None):
        """obtain_solution for a planning set whose size lives on the device (`count`, int32[1], <= ws.m_max problems;
        MPC.act with a recovery mask): mean / var / samples are the caller's persistent workspace `ws`, the cost function
        is called as cost_function(samples, count=count) and must not synchronise.  Same Philox rows, same arithmetic as
        obtain_solution on the compacted problems."""
        stream = _lib.current_stream()
        dim = self.sol_dim
        for _ in range(self.max_iters if iters is None else iters):
            rc = self.lib.rrl_cem_sample_n(_lib.ptr(count), ws.m_max, self.popsize, dim, _lib.ptr(ws.mean),
                                           _lib.ptr(ws.var), _lib.ptr(self.lb), _lib.ptr(self.ub), self.epsilon, 1,
                                           _lib.ptr(ws.active), self.seed, 0, _lib.ptr(self.tick), 1,
                                           _lib.ptr(ws.samples), stream)
            _lib.check(rc, "rrl_cem_sample_n")
            costs = self.cost_function(ws.samples, count=count)
            rc = self.lib.rrl_cem_update_n(_lib.ptr(count), ws.m_max, self.popsize, dim, self.num_elites, self.alpha,
                                           _lib.ptr(ws.samples), _lib.ptr(costs), _lib.ptr(ws.mean), _lib.ptr(ws.var),
                                           _lib.ptr(ws.active), stream)
            _lib.check(rc, "rrl_cem_update_n")
        return ws.mean


class PlanWorkspace:
    """Persistent buffers of the device-count planning path, sized for m_max problems."""

    def __init__(self, m_max, popsize, sol_dim, device):
        self.m_max = int(m_max)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.idx, self.count = z(m_max, dt=torch.int32), z(1, dt=torch.int32)
        self.mean, self.var = z(m_max, sol_dim, dt=torch.float64), z(m_max, sol_dim, dt=torch.float64)
        self.samples = z(m_max, popsize, sol_dim)
        self.cur_obs = z(m_max, 2)
        self.active = z(m_max, dt=torch.uint8)
        self.costs = z(m_max, popsize)
