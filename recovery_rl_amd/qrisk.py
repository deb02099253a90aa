"""Safety critic Q_risk and the model-free recovery policy (reference: recovery_rl/qrisk.py).

Same public surface as the reference's QRiskWrapper -- `update_parameters(memory, policy,
batch_size, plot)`, `get_value(states, actions)`, `select_action(state, eval)`, `__call__` --
with batched CUDA tensors in place of numpy rows.  The debugging heat-map `plot`
(qrisk.py:229-301) is out of scope (never enabled by the driver, experiment.py:415)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch.optim import Adam

from .model import QNetworkConstraint, StochasticPolicy
from .utils import hard_update, soft_update


def _adam(params, lr, capturable):
    return Adam(params, lr=lr, capturable=capturable, foreach=True)


class QRiskWrapper:
    def __init__(self, obs_space, ac_space, hidden_size, logdir, args, tmp_env=None):
        self.env_name = args.env_name
        self.logdir = logdir
        self.device = torch.device("cuda" if args.cuda else "cpu")
        self.ac_space = ac_space
        if getattr(args, "cnn", False) or getattr(args, "vismpc_recovery", False):
            raise NotImplementedError("image observations are outside the hot path (SURVEY.md section 2)")
        capturable = self.device.type == "cuda"
        d_obs, d_act = obs_space.shape[0], ac_space.shape[0]
        self.safety_critic = QNetworkConstraint(d_obs, d_act, hidden_size).to(self.device)
        self.safety_critic_target = QNetworkConstraint(d_obs, d_act, args.hidden_size).to(self.device)
        self.lr = args.lr
        self.safety_critic_optim = _adam(self.safety_critic.parameters(), args.lr, capturable)
        hard_update(self.safety_critic_target, self.safety_critic)      # qrisk.py:61

        self.tau = args.tau_safe
        self.gamma_safe = args.gamma_safe
        self.updates = 0
        self.target_update_interval = args.target_update_interval
        self.policy = StochasticPolicy(d_obs, d_act, hidden_size, ac_space).to(self.device)
        self.policy_optim = _adam(self.policy.parameters(), args.lr, capturable)
        self.pos_fraction = args.pos_fraction if args.pos_fraction >= 0 else None   # :78
        # lock-step loop: share of the batch drawn from the pinned demonstrations (set by Experiment once they are
        # pinned; None = the reference's one uniform draw)
        self.demo_share = None
        self.MF_recovery = args.MF_recovery
        self.Q_sampling_recovery = args.Q_sampling_recovery
        self.tmp_env = tmp_env
        self.last_losses = None
        self.fast = None

    # -- training ----------------------------------------------------------------------------
    def clamp_batch_size(self, batch_size, memory_len):
        """qrisk.py:100-104."""
        if self.pos_fraction:
            return min(batch_size, int((1 - self.pos_fraction) * memory_len))
        return min(batch_size, memory_len)

    def _share_kw(self):
        return {"demo_share": self.demo_share} if self.demo_share and self.pos_fraction is None else {}

    def update_parameters(self, memory=None, policy=None, batch_size=None, plot=False,
                          batch=None, eps_next=None, eps_pi=None):
        """One Q_risk step (+ one recovery-policy step if MF_recovery), qrisk.py:86-163.
        `policy` is the TASK policy: the target action a' ~ pi_task(s') (:119-120).
        `batch` / `eps_*` inject a fixed batch and policy noise (KAT tests)."""
        rows_loaded = False
        if batch is None:
            batch_size = self.clamp_batch_size(batch_size, len(memory))
            if self.fast is not None and batch_size == self.fast.B and hasattr(memory, "_desc"):
                batch = memory.sample(batch_size=batch_size, pos_fraction=self.pos_fraction, rows=self.fast.rows,
                                      **self._share_kw())
                rows_loaded = True
            else:
                batch = memory.sample(batch_size=batch_size, pos_fraction=self.pos_fraction, **self._share_kw())
        if self.fast is not None and batch[2].shape[0] == self.fast.B:
            if eps_next is None:
                eps_next, eps_pi = self.fast.noise(1)
            losses = self.fast.qrisk_update(batch, eps_next, eps_pi, rows_loaded=rows_loaded)
            self.updates += 1
            self.last_losses = (losses[4], losses[5], losses[6] if self.MF_recovery else None)
            return
        state, action, constraint, next_state, mask = batch
        constraint = constraint.reshape(-1, 1)
        mask = mask.reshape(-1, 1)

        with torch.no_grad():
            next_action, _, _ = policy.sample(next_state, eps_next)
            q1n, q2n = self.safety_critic_target(next_state, next_action)
            target = constraint + mask * self.gamma_safe * torch.max(q1n, q2n)   # :127-129

        q1, q2 = self.safety_critic(state, action)
        q1_loss = F.mse_loss(q1, target)
        q2_loss = F.mse_loss(q2, target)
        self.safety_critic_optim.zero_grad(set_to_none=True)
        (q1_loss + q2_loss).backward()
        self.safety_critic_optim.step()

        policy_loss = None
        if self.MF_recovery:                                              # :150-158
            pi, _, _ = self.policy.sample(state, eps_pi)
            q1p, q2p = self.safety_critic(state, pi)
            policy_loss = torch.max(q1p, q2p).mean()
            self.policy_optim.zero_grad(set_to_none=True)
            grads = torch.autograd.grad(policy_loss, list(self.policy.parameters()))
            for p, g in zip(self.policy.parameters(), grads):
                p.grad = g
            self.policy_optim.step()

        if self.updates % self.target_update_interval == 0:               # :160-163
            soft_update(self.safety_critic_target, self.safety_critic, self.tau)
        self.updates += 1
        self.last_losses = (q1_loss.detach(), q2_loss.detach(),
                            None if policy_loss is None else policy_loss.detach())

    # -- queries -----------------------------------------------------------------------------
    @torch.no_grad()
    def get_value(self, states, actions, encoded=False):
        """Q_risk(s,a) = max(q1,q2) without grad (qrisk.py:184-196)."""
        q1, q2 = self.safety_critic(states, actions)
        return torch.max(q1, q2)

    def __call__(self, states, actions):
        return self.safety_critic(states, actions)                        # :303-307

    @torch.no_grad()
    def select_action(self, state, eval=False, candidates=None, eps=None):
        """Recovery action (qrisk.py:198-227).  Accepts a [N,2] CUDA tensor (returns a tensor) or a
        single numpy state (returns numpy, like the reference).  `candidates` [N,1000,dU] injects the
        action_space.sample() draws of the Q-sampling branch (KAT tests)."""
        single = not torch.is_tensor(state)
        if single:
            state = torch.as_tensor(np.asarray(state, dtype=np.float32), device=self.device).unsqueeze(0)
        if self.MF_recovery:
            action, _, mean = self.policy.sample(state, eps)
            out = mean if eval else action
        elif self.Q_sampling_recovery:
            # 1000 uniform candidate actions per state, keep the argmin of Q_risk (:214-225)
            n, k = state.shape[0], 1000
            if getattr(self, "_ac_bounds", None) is None:        # once: a host -> device copy is not allowed inside a capture
                self._ac_bounds = (torch.as_tensor(self.ac_space.low, dtype=torch.float32, device=self.device),
                                   torch.as_tensor(self.ac_space.high, dtype=torch.float32, device=self.device))
            lo, hi = self._ac_bounds
            cand = lo + (hi - lo) * torch.rand(n, k, lo.numel(), device=self.device) if candidates is None else \
                torch.as_tensor(candidates, dtype=torch.float32, device=self.device).reshape(n, k, -1)
            q = self.get_value(state.unsqueeze(1).expand(n, k, -1).reshape(n * k, -1),
                               cand.reshape(n * k, -1)).reshape(n, k)
            out = cand[torch.arange(n, device=self.device), q.argmin(dim=1)]
        else:
            raise AssertionError("no model-free recovery mode selected")
        return out[0].cpu().numpy() if single else out
