"""Fused candidate evaluation for the model-based recovery controller: `rrl_plan_cost` behind
`MPC._compile_cost` (recovery_rl/MPC.py:374-416).  The PyTorch path in MPC.py stays as the general
path (other widths) and as the cross-check (tests/test_plan_gpu.py)."""
import ctypes as C
import os

import torch

from . import _lib


class FusedPlanner:
    """Packs the live Q_risk and ensemble weights into MFMA fragment order and evaluates CEM candidates."""

    def __init__(self, mpc, f16x3=None):
        """f16x3: evaluate the hidden layers as three f16 MFMA products of hi/lo splits (rrl_plan_cost_f16x3) instead of
        f32 MFMA; default from RRL_PLAN_F16X3 (off)."""
        self.mpc = mpc
        self.lib = _lib.load()
        self.f16x3 = (os.environ.get("RRL_PLAN_F16X3", "") not in ("", "0")) if f16x3 is None else bool(f16x3)
        self.device = mpc.device
        model = mpc.model
        self.hq = int(mpc.value_func.safety_critic.linear1.weight.shape[0])
        self.he = int(model.lin1_w.shape[1])
        self.n_nets = int(model.num_nets)
        n = self.lib.rrl_plan_pack_floats(self.hq, self.he, self.n_nets)
        if n <= 0:
            raise _lib.RRLError("planner shape not supported by rrl_plan_cost")
        self.packed = torch.zeros(int(n), dtype=torch.float32, device=self.device)
        self.tick = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.seed = (int(mpc.optimizer.seed) ^ 0x706C616E) & 0xFFFFFFFFFFFFFFFF
        self._scratch = None

    @staticmethod
    def supported(mpc):
        vf = mpc.value_func
        net = getattr(vf, "safety_critic", None)
        if net is None or not hasattr(net, "linear1") or mpc.mb_dynamics != "model":
            return False
        lib = _lib.load()
        return bool(lib.rrl_plan_supported(int(net.linear1.weight.shape[0]), int(mpc.model.lin1_w.shape[1]),
                                           int(mpc.model.num_nets), int(mpc.npart), int(mpc.dO), int(mpc.dU)))

    def pack(self):
        """Re-pack the current weights (they change with every Q_risk update / ensemble re-fit)."""
        net, model = self.mpc.value_func.safety_critic, self.mpc.model
        st = lambda a, b: torch.stack([a.detach(), b.detach()]).contiguous()
        keep = [st(net.linear1.weight, net.linear4.weight), st(net.linear1.bias, net.linear4.bias),
                st(net.linear2.weight, net.linear5.weight), st(net.linear2.bias, net.linear5.bias),
                st(net.linear3.weight, net.linear6.weight), st(net.linear3.bias, net.linear6.bias)]
        ens = [model.lin0_w, model.lin0_b, model.lin1_w, model.lin1_b, model.lin2_w, model.lin2_b, model.lin3_w,
               model.lin3_b, model.inputs_mu, model.inputs_sigma, model.max_logvar, model.min_logvar]
        ens = [t.detach().to(torch.float32).contiguous() for t in ens]
        w = _lib.rrl_plan_weights_t(self.hq, self.he, self.n_nets, *[t.data_ptr() for t in keep + ens])
        pack = self.lib.rrl_plan_pack_f16x3 if self.f16x3 else self.lib.rrl_plan_pack
        _lib.check(pack(C.byref(w), _lib.ptr(self.packed), _lib.current_stream()), "rrl_plan_pack")
        self._keep = keep + ens        # the pack kernels read them asynchronously

    def _scratch_for(self, M, pop):
        """First-step values + per-member cost sums of M planning problems (rrl_plan_scratch_floats); grows, never shrinks
        (a captured graph holds its address)."""
        need = int(self.lib.rrl_plan_scratch_floats(self.n_nets, M, pop))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=self.device)
        return self._scratch

    def cost(self, ac_seqs, cur_obs, noise=None):
        """ac_seqs [M, pop, plan_hor*2], cur_obs [M, 2] -> costs [M, pop] (f32)."""
        mpc = self.mpc
        M, pop = int(ac_seqs.shape[0]), int(ac_seqs.shape[1])
        ac_seqs = ac_seqs.to(torch.float32).contiguous()
        cur_obs = cur_obs.to(torch.float32).contiguous()
        if noise is not None:
            noise = noise.to(torch.float32).contiguous()
            assert tuple(noise.shape) == (mpc.plan_hor, M * pop * mpc.npart, 2)
        scratch = self._scratch_for(M, pop)
        costs = torch.empty(M, pop, dtype=torch.float32, device=self.device)
        entry = self.lib.rrl_plan_cost_f16x3 if self.f16x3 else self.lib.rrl_plan_cost
        rc = entry(_lib.ptr(self.packed), self.hq, self.he, self.n_nets, mpc.npart, M, pop,
                                    mpc.plan_hor, _lib.ptr(cur_obs), _lib.ptr(ac_seqs), _lib.ptr(noise), self.seed, 0,
                                    _lib.ptr(self.tick), 1, _lib.ptr(scratch), _lib.ptr(costs),
                                    _lib.current_stream())
        _lib.check(rc, "rrl_plan_cost")
        return costs

    def cost_n(self, ws, count, costs):
        """cost() for the first count[0] problems of the workspace (count on the device; rrl_plan_cost_n): no host
        synchronisation, the launch covers ws.m_max problems and the workgroups past the live ones exit at once."""
        mpc = self.mpc
        pop = int(ws.samples.shape[1])
        scratch = self._scratch_for(ws.m_max, pop)
        rc = self.lib.rrl_plan_cost_n(int(self.f16x3), _lib.ptr(self.packed), self.hq, self.he, self.n_nets, mpc.npart,
                                      _lib.ptr(count), ws.m_max, pop, mpc.plan_hor, _lib.ptr(ws.cur_obs),
                                      _lib.ptr(ws.samples), None, self.seed, 0, _lib.ptr(self.tick), 1,
                                      _lib.ptr(scratch), _lib.ptr(costs), _lib.current_stream())
        _lib.check(rc, "rrl_plan_cost_n")
        return costs
