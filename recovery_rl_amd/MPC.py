"""PETS / CEM model-based recovery controller (reference: recovery_rl/MPC.py:55-467), batched
over the envs that need a recovery action and resident on the GPU.

Same method names as the reference: `train(obs_trajs, acs_trajs, random, next_obs, epochs)`,
`act(obs, t)`, `update_value_func(vf)`, `reset()`, `_compile_cost`, `_predict_next_obs`,
`_expand_to_ts_format`, `_flatten_to_matrix`.  Differences:
  * `act(obs[N,2], t, mask)` plans for the rows with mask != 0 only (compacted), each with its own
    persistent `prev_sol` row; a single numpy observation works as in the reference;
  * the candidate rollout is chunked over envs to bound activation memory
    (popsize * npart = 8000 rows per planning env);
  * `mb_dynamics="env"` (extension) propagates candidates through the navigation step kernels
    instead of the learned ensemble; the default "model" is the reference behaviour.
"""
import numpy as np
import torch

from . import _lib
from .optimizers import CEMOptimizer


def shuffle_rows(arr):
    """Independent permutation of every row (MPC.py:50-52)."""
    idxs = torch.argsort(torch.rand(arr.shape, device=arr.device), dim=-1)
    return torch.gather(arr, 1, idxs)


def _required(cfg, key, message):
    if cfg.get(key, None) is None:
        raise ValueError(message)                                    # utils.get_required_argument
    return cfg[key]


def _capturable(opt):
    return bool(opt.param_groups[0].get("capturable", False))


class _GraphedTrainStep:
    """The ensemble training step (forward, NLL loss, backward, Adam) captured once per `MPC.train` call and
    replayed per batch: the batch-32 loop of the reference (MPC.py:266-292) is ~40 tiny launches per step, i.e.
    bound by launch latency from Python; a replay issues them back-to-back.  The first steps run eagerly (they are
    ordinary training steps and double as the warm-up torch asks for before a capture)."""

    WARMUP = 3

    def __init__(self, mpc, batch_size):
        self.mpc, self.graph, self.calls = mpc, None, 0
        self.bi = torch.zeros(mpc.model.num_nets, batch_size, dtype=torch.long, device=mpc.device)

    def __call__(self, bi):
        mpc = self.mpc
        if self.calls < self.WARMUP:
            side = torch.cuda.Stream(device=mpc.device)
            side.wait_stream(torch.cuda.current_stream(mpc.device))
            with torch.cuda.stream(side):
                mpc._train_step(bi)
            torch.cuda.current_stream(mpc.device).wait_stream(side)
        else:
            self.bi.copy_(bi)
            if self.graph is None:
                mpc.model.optim.zero_grad(set_to_none=True)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    mpc._train_step(self.bi)            # recorded, not executed
            self.graph.replay()
        self.calls += 1


class MPC:
    optimizers = {"CEM": CEMOptimizer}
    MAX_ROWS = 1 << 21       # rows per rollout chunk (activations: rows x 200 x 4 B x few)

    def __init__(self, params, mb_dynamics="model", seed=0, plan_precision=None):
        env = params.env
        self.env = env
        self.device = env.device
        self.dO, self.dU = env.observation_space.shape[0], env.action_space.shape[0]
        self.ac_ub = np.minimum(env.action_space.high, params.get("ac_ub", env.action_space.high))
        self.ac_lb = np.maximum(env.action_space.low, params.get("ac_lb", env.action_space.low))
        self.per = params.get("per", 1)
        assert self.per == 1, "only per=1 (re-plan at every call) is supported, as config/default.py sets"
        prop, opt = params.prop_cfg, params.opt_cfg
        self.model_init_cfg = prop.get("model_init_cfg", {})
        self.model_train_cfg = prop.get("model_train_cfg", {})
        self.prop_mode = _required(prop, "mode", "Must provide propagation method.")
        self.npart = _required(prop, "npart", "Must provide number of particles.")
        self.obs_postproc = prop.get("obs_postproc", lambda obs, model_out: model_out)
        self.targ_proc = prop.get("targ_proc", lambda obs, next_obs: next_obs)
        self.opt_mode = _required(opt, "mode", "Must provide optimization method.")
        self.plan_hor = _required(opt, "plan_hor", "Must provide planning horizon.")
        self.obs_cost_fn = _required(opt, "obs_cost_fn", "Must provide cost on observations.")
        self.ac_cost_fn = _required(opt, "ac_cost_fn", "Must provide cost on actions.")
        assert self.opt_mode == 'CEM'
        assert self.prop_mode == 'TSinf', 'only TSinf propagation mode is supported'
        assert self.npart % self.model_init_cfg["num_nets"] == 0, \
            "Number of particles must be a multiple of the ensemble size."
        assert mb_dynamics in ("model", "env")
        self.mb_dynamics = mb_dynamics

        self.optimizer = CEMOptimizer(sol_dim=self.plan_hor * self.dU,
                                      lower_bound=np.tile(self.ac_lb, [self.plan_hor]),
                                      upper_bound=np.tile(self.ac_ub, [self.plan_hor]),
                                      cost_function=self._compile_cost, device=self.device, seed=seed,
                                      **opt.get("cfg", {}))
        self.has_been_trained = prop.get("model_pretrained", False)
        n = getattr(env, "num_envs", 1)
        mid = np.tile((self.ac_lb + self.ac_ub) / 2, [self.plan_hor])           # MPC.py:174
        self.prev_sol = torch.as_tensor(mid, dtype=torch.float64, device=self.device).repeat(n, 1)
        self._mid = torch.as_tensor(mid, dtype=torch.float64, device=self.device)
        self.init_var = torch.as_tensor(np.tile(np.square(self.ac_ub - self.ac_lb) / 16, [self.plan_hor]),
                                        dtype=torch.float64, device=self.device)            # :175-176
        self.train_in = torch.zeros(0, self.dO + self.dU, device=self.device)
        self.train_targs = torch.zeros(0, self.dO, device=self.device)
        print("Created an MPC controller, prop mode %s, %d particles. " % (self.prop_mode, self.npart))
        print("Trajectory prediction logging is disabled.")
        self.model = _required(self.model_init_cfg, "model_constructor",
                               "Must provide a model constructor.")(self.model_init_cfg)
        self.value_func = None
        self.graph_train = True            # replay the ensemble training step from a hipGraph (same kernels)
        self.fused_train = True            # ... or, for the reference's shapes, run it as one fused kernel
        self._trainer = None
        self.fused = None                  # FusedPlanner (rrl_plan_cost) when the shapes allow it
        self.device_count = True           # act(obs, t, mask): planning-set size stays on the device (no host sync)
        self._plan_ws = self._plan_out = self.last_count = None
        self.use_fused_planner = True
        if plan_precision not in (None, "f32", "f16x3"):
            raise ValueError("--plan_precision must be 'f32' or 'f16x3'")
        self.plan_f16x3 = None if plan_precision is None else plan_precision == "f16x3"   # None: RRL_PLAN_F16X3
        self._lb = torch.as_tensor(self.ac_lb, dtype=torch.float32, device=self.device)
        self._ub = torch.as_tensor(self.ac_ub, dtype=torch.float32, device=self.device)

    # -- model fitting (MPC.py:213-309) --------------------------------------------------------
    def train(self, obs_trajs, acs_trajs, random=False, next_obs=False, epochs=None, batch_size=32,
              progress=False):
        dev = self.device
        t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev)
        if random:
            assert next_obs is not None
            new_in = [torch.cat([t(obs_trajs), t(acs_trajs)], dim=-1)]
            new_targ = [self.targ_proc(t(obs_trajs), t(next_obs))]
        else:
            new_in, new_targ = [], []
            for obs, acs in zip(obs_trajs, acs_trajs):
                obs, acs = t(obs), t(acs)
                new_in.append(torch.cat([obs[:-1], acs], dim=-1))
                new_targ.append(self.targ_proc(obs[:-1], obs[1:]))
        self.train_in = torch.cat([self.train_in] + new_in, dim=0)
        self.train_targs = torch.cat([self.train_targs] + new_targ, dim=0)
        self.has_been_trained = True
        self.model.fit_input_stats(self.train_in)
        n = self.train_in.shape[0]
        idxs = torch.randint(n, (self.model.num_nets, n), device=dev)          # bootstrap, :255-257
        if epochs is None:
            epochs = self.model_train_cfg['epochs']
        num_batch = int(np.ceil(n / batch_size))
        losses = None
        fused = self._fused_trainer(batch_size)
        self._hand_over_optimiser("fused" if fused is not None else "torch")
        if fused is not None:
            fused.begin(self.train_in, self.train_targs)
        step = None
        if fused is None and self.graph_train and epochs * (n // batch_size) >= 16:
            step = _GraphedTrainStep(self, batch_size)
        for _ in range(epochs):
            if fused is not None:
                fused.epoch(idxs, batch_size)         # every batch: rrl_ens_train_grad + rrl_adam_step_multi
            for b in range(num_batch if fused is None else 0):
                bi = idxs[:, b * batch_size:(b + 1) * batch_size]
                if step is not None and bi.shape[1] == batch_size:
                    step(bi)                          # one hipGraph replay: forward + backward + Adam
                else:
                    self._train_step(bi)
            idxs = shuffle_rows(idxs)
            if progress:
                with torch.no_grad():
                    mean, _ = self.model(self.train_in[idxs[:, :5000]])
                    losses = ((mean - self.train_targs[idxs[:, :5000]]) ** 2).mean(-1).mean(-1)
                print("Network training: MSE per net", losses.cpu().numpy())
        return losses

    def _hand_over_optimiser(self, owner):
        """ONE Adam state per controller (the reference has a single torch.optim.Adam, config/navigation1.py:167):
        the fused trainer (batch <= 32) and model.optim (any batch) each hold moments and a step count, so when
        consecutive `train` calls take different paths the state moves with them instead of restarting the bias
        correction on a fitted model."""
        prev = getattr(self, "_optim_owner", None)
        self._optim_owner = owner
        if prev is None or prev == owner or self._trainer is None:
            return
        from .ensemble_train import PARAMS
        tr, opt = self._trainer, self.model.optim
        params = [getattr(self.model, n) for n in PARAMS]
        with torch.no_grad():
            if owner == "torch":                       # fused -> torch.optim.Adam
                for p, m, v, st in zip(params, tr.m, tr.v, tr.steps):
                    opt.state[p] = {"step": st[0].to(torch.float32).cpu() if not _capturable(opt)
                                    else st[0].to(torch.float32), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
            else:                                      # torch.optim.Adam -> fused
                for p, m, v, st in zip(params, tr.m, tr.v, tr.steps):
                    state = opt.state.get(p)
                    if not state:
                        m.zero_(), v.zero_(), st.zero_()
                        continue
                    m.copy_(state["exp_avg"]), v.copy_(state["exp_avg_sq"])
                    st[0] = int(float(state["step"]))

    def _fused_trainer(self, batch_size):
        """FusedEnsembleTrainer when the shapes are the kernels' (4-200-200-200-4; batch <= 32: the one-launch kernel,
        larger: the large-batch MFMA kernels of the lock-step loop's re-fit), else None."""
        if not self.fused_train or self.train_in.device.type != "cuda":
            return None
        from .ensemble_train import FusedEnsembleTrainer
        if not FusedEnsembleTrainer.supported(self.model, batch_size):
            return None
        if self._trainer is None:
            self._trainer = FusedEnsembleTrainer(self.model, lr=self.model.optim.param_groups[0]["lr"])
        return self._trainer

    def _train_step(self, bi):
        """One optimiser step on the bootstrap rows bi [num_nets, batch] (MPC.py:270-292)."""
        loss = 0.01 * (self.model.max_logvar.sum() - self.model.min_logvar.sum())
        loss = loss + self.model.compute_decays()
        mean, logvar = self.model(self.train_in[bi], ret_logvar=True)
        inv_var = torch.exp(-logvar)
        tl = ((mean - self.train_targs[bi]) ** 2) * inv_var + logvar
        loss = loss + tl.mean(-1).mean(-1).sum()                       # :282-287
        self.model.optim.zero_grad(set_to_none=True)
        loss.backward()
        self.model.optim.step()

    def reset(self):
        mid = np.tile((self.ac_lb + self.ac_ub) / 2, [self.plan_hor])
        self.prev_sol[:] = torch.as_tensor(mid, dtype=torch.float64, device=self.device)
        self.optimizer.reset()

    def forget_plans(self, ended):
        """Vectorisation rule 5 (lock-step loop, N > 1): the envs whose episode just ended start their next plan from the
        mid-point sequence (what prev_sol is before the first plan, MPC.py:174) instead of from the shifted solution of their
        last plan.  The reference never resets prev_sol (experiment.py never calls MPC.reset): its ONE env plans at every
        gate event, a few steps or at most an episode apart, in the part of the arena it is learning in.  Of 4096 envs with a
        trained task policy each plans once in hundreds of iterations: the warm start is a plan for another place, and it sits
        AT the action bound (the recovery direction saturates at -0.98), where CEM's variance clamp (optimizers.py:96-99:
        var <= ((mean - lb) / 2)^2 = 1e-4) freezes it -- five iterations cannot turn it.  Measured on config 4, seed 1
        (profiles/round6_config4_seed1/): inside a violation burst the safety critic rates a safe direction at 0.03 on 100 % of
        the rows under the recovery controller and the executed action at > eps_safe on 16-50 % of them, all at the obstacle's
        far corner (x = -21.5, |y| = 8), all with x-component -0.93 .. -0.98 -- the direction that was 'away' where the env
        planned last.  ended: uint8 / bool [N]."""
        self.prev_sol.copy_(torch.where(ended.bool().unsqueeze(1), self._mid, self.prev_sol))

    def update_value_func(self, value_func):
        self.value_func = value_func
        self.fused = None
        if self.use_fused_planner:
            from .planner import FusedPlanner
            if FusedPlanner.supported(self):
                self.fused = FusedPlanner(self, f16x3=self.plan_f16x3)

    # -- acting (MPC.py:322-347) ---------------------------------------------------------------
    @torch.no_grad()
    def act(self, obs, t, mask=None, get_pred_cost=False):
        single = not torch.is_tensor(obs)
        if single:
            obs = torch.as_tensor(np.asarray(obs, dtype=np.float32)[None], device=self.device)
        n = obs.shape[0]
        if not self.has_been_trained:                                         # :333-334
            out = self._lb + (self._ub - self._lb) * torch.rand(n, self.dU, device=self.device)
            return out[0].cpu().numpy() if single else out
        if (mask is not None and self.fused is not None and self.device_count and self.prev_sol.shape[0] == n
                and self.mb_dynamics == "model"):
            return self._act_device_count(obs, mask)
        out = torch.zeros(n, self.dU, dtype=torch.float32, device=self.device)
        idx = torch.arange(n, device=self.device) if mask is None else mask.nonzero().squeeze(1)
        if idx.numel() == 0:
            return out
        self.sy_cur_obs = obs[idx].contiguous()
        if self.fused is not None:
            self.fused.pack()              # weights moved since the last call (Q_risk update / re-fit)
        rows = idx if self.prev_sol.shape[0] == n else torch.zeros_like(idx)
        from .utils import trace_range
        with trace_range("cem"):
            soln = self.optimizer.obtain_solution(self.prev_sol[rows], self.init_var.expand(idx.numel(), -1))
        shifted = torch.cat([soln[:, self.per * self.dU:],
                             torch.zeros(idx.numel(), self.per * self.dU, dtype=soln.dtype,
                                         device=self.device)], dim=1)          # :342-344
        self.prev_sol[rows] = shifted
        out[idx] = soln[:, :self.dU].to(torch.float32)
        return out[0].cpu().numpy() if single else out

    def _act_device_count(self, obs, mask):
        """act() for a recovery mask without reading the number of planning envs on the host: the mask is compacted on
        the device (rrl_cem_begin), every kernel of the CEM reads the count from device memory and is launched for the
        upper bound n (workgroups past the live problems exit at once), rrl_cem_finish scatters the actions and the
        shifted solutions.  Same index order, Philox rows and arithmetic as the host-count path."""
        from .optimizers import PlanWorkspace
        from .utils import trace_range
        n, dim = obs.shape[0], self.plan_hor * self.dU
        ws = self._plan_ws
        if ws is None or ws.m_max != n:
            ws = self._plan_ws = PlanWorkspace(n, self.optimizer.popsize, dim, self.device)
            self._plan_out = torch.zeros(n, self.dU, dtype=torch.float32, device=self.device)
        mask_u8 = mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)
        obs = obs.to(torch.float32).contiguous()
        lib, st = _lib.load(), _lib.current_stream()
        p = _lib.ptr
        _lib.check(lib.rrl_cem_begin(n, p(mask_u8), dim, p(self.prev_sol), p(self.init_var), p(obs), p(ws.idx),
                                     p(ws.count), p(ws.mean), p(ws.var), p(ws.cur_obs), p(ws.active), st),
                   "rrl_cem_begin")
        self.fused.pack()                  # weights moved since the last call (Q_risk update / re-fit)
        with trace_range("cem"):
            self.optimizer.obtain_solution_n(ws, ws.count)
        _lib.check(lib.rrl_cem_finish(n, p(mask_u8), dim, self.per * self.dU, p(ws.idx), p(ws.count), p(ws.mean),
                                      p(self.prev_sol), p(self._plan_out), st), "rrl_cem_finish")
        self.last_count = ws.count         # device-side size of the planning set of this call (int32[1])
        return self._plan_out

    # -- candidate evaluation (MPC.py:374-416) -------------------------------------------------
    @torch.no_grad()
    def _compile_cost(self, ac_seqs, cur_obs=None, noise=None, fused=None, count=None):
        """ac_seqs [M, pop, plan_hor*dU] -> mean over particles of sum_t Q_risk(obs_t, ac_t): [M, pop].
        `noise` (optional, [plan_hor, M*pop*npart, dO], row = (m*pop + c)*npart + p) replaces the particle
        noise draws; `fused` forces (True) or forbids (False) the rrl_plan_cost kernel."""
        if count is not None:              # device-count planning set (obtain_solution_n): the workspace holds the inputs
            return self.fused.cost_n(self._plan_ws, count, self._plan_ws.costs)
        single = not torch.is_tensor(ac_seqs)
        if single:
            ac_seqs = torch.as_tensor(ac_seqs, dtype=torch.float32, device=self.device)[None]
        if cur_obs is None:
            cur_obs = self.sy_cur_obs
        cur_obs = torch.as_tensor(cur_obs, dtype=torch.float32, device=self.device).reshape(-1, self.dO)
        M, pop = ac_seqs.shape[0], ac_seqs.shape[1]
        use_fused = (self.fused is not None) if fused is None else fused
        if use_fused:
            if self.fused is None:
                raise _lib.RRLError("rrl_plan_cost does not support this planner shape")
            costs = self.fused.cost(ac_seqs, cur_obs, noise)
            return costs[0].cpu().numpy() if single else costs
        per_env = pop * self.npart
        chunk = max(1, self.MAX_ROWS // per_env)
        outs = []
        for lo in range(0, M, chunk):
            nz = None if noise is None else noise[:, lo * per_env:(lo + chunk) * per_env]
            outs.append(self._compile_cost_chunk(ac_seqs[lo:lo + chunk], cur_obs[lo:lo + chunk], nz))
        costs = torch.cat(outs, dim=0)
        return costs[0].cpu().numpy() if single else costs

    def _compile_cost_chunk(self, ac_seqs, cur_obs, noise=None):
        M, pop = ac_seqs.shape[0], ac_seqs.shape[1]
        H, P = self.plan_hor, self.npart
        acs = ac_seqs.reshape(M, pop, H, self.dU).permute(2, 0, 1, 3)            # [H, M, pop, dU]
        acs = acs[:, :, :, None, :].expand(H, M, pop, P, self.dU).reshape(H, M * pop * P, self.dU)
        obs = cur_obs[:, None, None, :].expand(M, pop, P, self.dO).reshape(M * pop * P, self.dO)
        costs = torch.zeros(M * pop, P, device=self.device)
        if self.mb_dynamics == "env":
            obs_seq = self._rollout_through_env(obs, acs.contiguous())
        for t in range(H):
            cur_acs = acs[t]
            cost = self.value_func.get_value(obs, cur_acs).reshape(-1, P)
            costs += cost
            if self.mb_dynamics == "env":
                obs = obs_seq[t]
            else:
                obs = self._predict_next_obs(obs, cur_acs, None if noise is None else noise[t])
        costs = torch.where(costs != costs, torch.full_like(costs, 1e6), costs)     # NaN -> 1e6 (:415)
        return costs.mean(dim=1).reshape(M, pop)

    def _predict_next_obs(self, obs, acs, noise=None):
        """One TS-infinity step through the ensemble (MPC.py:421-439)."""
        inputs = torch.cat((self._expand_to_ts_format(obs), self._expand_to_ts_format(acs)), dim=-1)
        mean, var = self.model(inputs)
        eps = torch.randn_like(mean) if noise is None else self._expand_to_ts_format(noise)
        predictions = mean + eps * var.sqrt()
        return self.obs_postproc(obs, self._flatten_to_matrix(predictions))

    def _expand_to_ts_format(self, mat):
        """[rows, dim] -> [num_nets, rows / num_nets, dim]; particle p of every candidate is bound
        to net p // (npart / num_nets) for the whole rollout (MPC.py:441-455)."""
        dim = mat.shape[-1]
        E = self.model.num_nets
        return mat.reshape(-1, E, self.npart // E, dim).transpose(0, 1).reshape(E, -1, dim)

    def _flatten_to_matrix(self, ts_fmt_arr):
        """Inverse of _expand_to_ts_format (MPC.py:457-467)."""
        dim = ts_fmt_arr.shape[-1]
        E = self.model.num_nets
        return ts_fmt_arr.reshape(E, -1, self.npart // E, dim).transpose(0, 1).reshape(-1, dim)

    def _rollout_through_env(self, obs, acs):
        """Extension (not the reference behaviour): ground-truth dynamics via rrl_nav_rollout."""
        kind = {"navigation1": 0, "navigation2": 1}[self.env.env_name]
        H, rows = acs.shape[0], acs.shape[1]
        pos = obs.to(torch.float64).contiguous()
        obs_seq = torch.empty(H, rows, 2, dtype=torch.float32, device=self.device)
        lib = _lib.load()
        rc = lib.rrl_nav_rollout(kind, rows, H, _lib.ptr(pos), _lib.ptr(acs), self.optimizer.seed ^ 0x5151,
                                 0, _lib.ptr(self.optimizer.tick), _lib.ptr(obs_seq), None, None, None,
                                 _lib.current_stream())
        _lib.check(rc, "rrl_nav_rollout")
        return obs_seq
