"""SAC task agent with the Recovery-RL hooks (reference: recovery_rl/sac.py).

Same constructor and method names as the reference's `SAC` -- `select_action(state, eval)`,
`update_parameters(memory, batch_size, updates, nu, safety_critic)` -- over batched CUDA
tensors.  Differences that are deliberate and documented (DESIGN.md "SAC update order"):
  * critic and policy gradients are BOTH taken at the pre-update weights, then both
    optimisers step.  The reference builds policy_loss before critic_optim.step() and
    back-propagates it afterwards (sac.py:216-239), which torch>=1.5 rejects and torch 1.4
    silently evaluated at mixed weights;
  * the five returned statistics are device tensors unless `as_floats=True` (the reference's
    five `.item()` calls, sac.py:276-277, are five host syncs per update).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.optim import Adam

from .model import DeterministicPolicy, GaussianPolicy, QNetwork
from .qrisk import QRiskWrapper
from .utils import hard_update, soft_update


def _adam(params, lr, capturable):
    return Adam(params, lr=lr, capturable=capturable, foreach=True)


class SAC(object):
    def __init__(self, observation_space, action_space, args, logdir, im_shape=None, tmp_env=None):
        if getattr(args, "cnn", False):
            raise NotImplementedError("image observations are outside the hot path (SURVEY.md section 2)")
        self.gamma = args.gamma
        self.seed = int(getattr(args, "seed", 0))
        self.tau = args.tau
        self.alpha = args.alpha
        self.env_name = args.env_name
        self.logdir = logdir
        self.policy_type = args.policy
        self.target_update_interval = args.target_update_interval
        self.automatic_entropy_tuning = args.automatic_entropy_tuning
        self.device = torch.device("cuda" if args.cuda else "cpu")
        self.updates = 0
        self.lr = args.lr
        self.fast = None
        capturable = self.device.type == "cuda"

        self.gamma_safe = args.gamma_safe
        self.eps_safe = args.eps_safe
        self.DGD_constraints = args.DGD_constraints
        self.nu = args.nu
        self.update_nu = args.update_nu
        self.use_constraint_sampling = args.use_constraint_sampling
        # dual variables (sac.py:58-70): Adam at 0.1 * lr
        self.log_nu = torch.tensor(np.log(self.nu), dtype=torch.float32, requires_grad=True,
                                   device=self.device)
        self.nu_optim = _adam([self.log_nu], 0.1 * args.lr, capturable)
        self.RCPO = args.RCPO
        self.lambda_RCPO = args.lambda_RCPO
        self.log_lambda_RCPO = torch.tensor(np.log(self.lambda_RCPO), dtype=torch.float32,
                                            requires_grad=True, device=self.device)
        self.lambda_RCPO_optim = _adam([self.log_lambda_RCPO], 0.1 * args.lr, capturable)

        d_obs, d_act = observation_space.shape[0], action_space.shape[0]
        self.critic = QNetwork(d_obs, d_act, args.hidden_size).to(self.device)
        self.critic_target = QNetwork(d_obs, d_act, args.hidden_size).to(self.device)
        self.critic_optim = _adam(self.critic.parameters(), args.lr, capturable)
        hard_update(self.critic_target, self.critic)

        if self.policy_type == "Gaussian":
            if self.automatic_entropy_tuning is True:
                self.target_entropy = -float(np.prod(action_space.shape))     # -dim(A), sac.py:94-95
                self.log_alpha = torch.zeros(1, requires_grad=True, device=self.device)
                self.alpha_optim = _adam([self.log_alpha], args.lr, capturable)
            self.policy = GaussianPolicy(d_obs, d_act, args.hidden_size, action_space).to(self.device)
        else:
            self.alpha = 0
            self.automatic_entropy_tuning = False
            self.policy = DeterministicPolicy(d_obs, d_act, args.hidden_size, action_space).to(self.device)
        self.policy_optim = _adam(self.policy.parameters(), args.lr, capturable)

        self.safety_critic = QRiskWrapper(observation_space, action_space, args.hidden_size, logdir,
                                          args, tmp_env=tmp_env)
        # constants returned as statistics; created once (no host->device copy inside a hipGraph)
        self._zero = torch.zeros((), device=self.device)
        self._alpha_const = torch.tensor(float(self.alpha), device=self.device)

    @torch.no_grad()
    def _set_dual(self, name, log_param):
        """self.<name> = exp(log_param), written IN PLACE into one persistent tensor so a captured
        hipGraph keeps reading the live value on replay."""
        cur = getattr(self, name)
        if not torch.is_tensor(cur):
            cur = torch.zeros_like(log_param.detach())
            setattr(self, name, cur)
        cur.copy_(log_param.detach().exp())

    # -- acting ------------------------------------------------------------------------------
    @torch.no_grad()
    def select_action(self, state, eval=False, eps=None):
        """Task action (sac.py:133-168).  [N,2] CUDA tensor in -> tensor out; a single numpy state
        in -> numpy out (the reference's calling convention).  `eps`: the policy's N(0,1) draws (default: torch's global
        generator, as the reference)."""
        single = not torch.is_tensor(state)
        if single:
            state = torch.as_tensor(np.asarray(state, dtype=np.float32), device=self.device).unsqueeze(0)
        if self.use_constraint_sampling:
            action = self._sqrl_action(state)
        else:
            sampled, _, mean = self.policy.sample(state, eps)
            action = mean if eval else sampled
        return action[0].cpu().numpy() if single else action

    @torch.no_grad()
    def _sqrl_action(self, state, safe_samples=100, eps=None, draw=None):
        """SQRL constraint sampling (sac.py:139-161) for every env of the batch at once: draw 100 candidate actions,
        keep those with Q_risk <= eps_safe and draw ONE with probability proportional to pi(a|s) = exp(log pi) over the
        safe ones; no safe candidate -> the argmin of Q_risk.

        Bug-compatible with the reference on purpose (results parity): its Categorical runs over the safe candidates
        only, and the drawn position j -- an index into that SAFE list -- is then used on the FULL candidate list
        (`pi[sampled_idx]`, sac.py:157-158), so the executed action is candidate j, safe or not.  `eps` [n,k,dU]
        injects the candidate noise and `draw` [n] the categorical's result (a position in the safe list) -- KAT tests."""
        n, k = state.shape[0], safe_samples
        sb = state.unsqueeze(1).expand(n, k, state.shape[1]).reshape(n * k, -1)
        pi, log_pi, _ = self.policy.sample(sb, None if eps is None else eps.reshape(n * k, -1))
        q = self.safety_critic.get_value(sb, pi).reshape(n, k)
        safe = q <= self.eps_safe
        n_safe = safe.sum(1)
        if draw is None:
            # Categorical over the safe candidates = multinomial over all candidates with the unsafe weights zeroed;
            # its position within the compacted safe list is the number of safe candidates up to it
            w = torch.exp(log_pi.reshape(n, k)) * safe
            w = torch.where((n_safe > 0).unsqueeze(1), w, torch.ones_like(w))
            cand = torch.multinomial(w, 1)
            j = (safe.to(torch.int64).cumsum(1).gather(1, cand).squeeze(1) - 1).clamp(min=0)
        else:
            j = torch.as_tensor(draw, dtype=torch.int64, device=state.device).clamp(min=0)
        pick = torch.where(n_safe > 0, j, q.argmin(1))                 # sac.py:153-158
        return pi.reshape(n, k, -1)[torch.arange(n, device=state.device), pick]

    # -- learning ----------------------------------------------------------------------------
    def enable_fast_path(self, batch_size):
        """Route update_parameters (and the safety critic's) through the fused HIP kernels
        (fast_update.FastUpdater).  Only for configurations fast_path_supported() accepts."""
        from .fast_update import FastUpdater
        self.fast = FastUpdater(self, batch_size)
        self.safety_critic.fast = self.fast
        return self.fast

    def update_parameters(self, memory, batch_size, updates, nu=None, safety_critic=None,
                          batch=None, eps_next=None, eps_pi=None, as_floats=False):
        """One SAC step (sac.py:170-277).  `batch` / `eps_*` inject a fixed batch and policy
        noise (KAT tests)."""
        if nu is None:
            nu = self.nu
        rows_loaded = False
        if batch is None:
            if self.fast is not None and batch_size == self.fast.B and hasattr(memory, "_desc"):
                batch = memory.sample(batch_size=batch_size, rows=self.fast.rows)   # gather fills the net inputs
                rows_loaded = True
            else:
                batch = memory.sample(batch_size=batch_size)
        if self.fast is not None and batch[2].shape[0] == self.fast.B:
            if eps_next is None:
                eps_next, eps_pi = self.fast.noise(0)
            losses = self.fast.sac_update(batch, eps_next, eps_pi, rows_loaded=rows_loaded)
            out = (losses[0], losses[1], losses[2], self._zero, self._alpha_const)
            return tuple(float(x) for x in out) if as_floats else out
        state, action, reward, next_state, mask = batch
        reward = reward.reshape(-1, 1)
        mask = mask.reshape(-1, 1)

        qsafe = None
        with torch.no_grad():
            next_action, next_log_pi, _ = self.policy.sample(next_state, eps_next)
            q1n, q2n = self.critic_target(next_state, next_action)
            min_qn = torch.min(q1n, q2n) - self.alpha * next_log_pi
            next_q = reward + mask * self.gamma * min_qn                        # :199-201
            if self.RCPO:                                                       # :202-205
                qsafe = torch.max(*safety_critic(state, action))
                next_q = next_q - self.lambda_RCPO * qsafe
        q1, q2 = self.critic(state, action)
        q1_loss = F.mse_loss(q1, next_q)
        q2_loss = F.mse_loss(q2, next_q)

        pi, log_pi, _ = self.policy.sample(state, eps_pi)
        q1_pi, q2_pi = self.critic(state, pi)
        min_q_pi = torch.min(q1_pi, q2_pi)
        max_sqf_pi = None
        if self.DGD_constraints or self.update_nu:
            sq1, sq2 = self.safety_critic(state, pi)
            max_sqf_pi = torch.max(sq1, sq2)
        if self.DGD_constraints:                                                # :224-228
            policy_loss = ((self.alpha * log_pi) + nu * (max_sqf_pi - self.eps_safe) - 1. * min_q_pi).mean()
        else:
            policy_loss = ((self.alpha * log_pi) - min_q_pi).mean()

        # both gradients at the pre-update weights, then both steps
        critic_params = list(self.critic.parameters())
        policy_params = list(self.policy.parameters())
        c_grads = torch.autograd.grad(q1_loss + q2_loss, critic_params)
        p_grads = torch.autograd.grad(policy_loss, policy_params)
        for p, g in zip(critic_params, c_grads):
            p.grad = g
        for p, g in zip(policy_params, p_grads):
            p.grad = g
        self.critic_optim.step()
        self.policy_optim.step()

        if self.automatic_entropy_tuning:                                       # :241-250
            alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean()
            self.alpha_optim.zero_grad(set_to_none=True)
            alpha_loss.backward()
            self.alpha_optim.step()
            self._set_dual("alpha", self.log_alpha)
            alpha_t = self.alpha.reshape(())
        else:
            alpha_loss = self._zero
            alpha_t = self._alpha_const

        if self.update_nu:                                                      # :256-262
            nu_loss = (self.log_nu * (self.eps_safe - max_sqf_pi).detach()).mean()
            self.nu_optim.zero_grad(set_to_none=True)
            nu_loss.backward()
            self.nu_optim.step()
            self._set_dual("nu", self.log_nu)

        if self.RCPO:                                                           # :265-271
            lam_loss = (self.log_lambda_RCPO * (self.eps_safe - qsafe).detach()).mean()
            self.lambda_RCPO_optim.zero_grad(set_to_none=True)
            lam_loss.backward()
            self.lambda_RCPO_optim.step()
            self._set_dual("lambda_RCPO", self.log_lambda_RCPO)

        if updates % self.target_update_interval == 0:                          # :273-274
            soft_update(self.critic_target, self.critic, self.tau)

        out = (q1_loss.detach(), q2_loss.detach(), policy_loss.detach(), alpha_loss.detach().reshape(()),
               alpha_t)
        if as_floats:
            return tuple(float(x) for x in out)
        return out
