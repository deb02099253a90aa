"""MLP critics and policies of the hot path (reference: recovery_rl/model.py, MLP part).

Parameter names match the reference's modules (linear1..linear6, mean_linear,
log_std_linear, mean, log_std, bn1) so a reference `state_dict()` loads unchanged -- the
golden KATs in tests/golden/model_golden.npz rely on that.  Every sampler takes an
optional explicit noise tensor so results can be pinned without sharing an RNG stream.
All modules are batched: [N, dim] in, [N, ...] out, float32.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Normal

LOG_SIG_MAX = 2       # model.py:14
LOG_SIG_MIN = -20     # model.py:15
EPSILON = 1e-6        # model.py:16


def weights_init_(m):
    """Xavier-uniform weights (gain 1), zero bias for every Linear (model.py:23-26)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)
        nn.init.constant_(m.bias, 0)


def _space_affine(action_space):
    """(scale, bias) that map tanh output onto the action box (model.py:308-315)."""
    if action_space is None:
        return torch.tensor(1.0), torch.tensor(0.0)
    hi = np.asarray(action_space.high, dtype=np.float64)
    lo = np.asarray(action_space.low, dtype=np.float64)
    return (torch.as_tensor((hi - lo) / 2.0, dtype=torch.float32),
            torch.as_tensor((hi + lo) / 2.0, dtype=torch.float32))


class _TwinQ(nn.Module):
    """Two independent 3-layer heads on cat[state, action] (model.py:49-76)."""

    squash = False

    def __init__(self, num_inputs, num_actions, hidden_dim):
        super().__init__()
        d = num_inputs + num_actions
        self._pre_init(d)
        self.linear1 = nn.Linear(d, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.linear4 = nn.Linear(d, hidden_dim)
        self.linear5 = nn.Linear(hidden_dim, hidden_dim)
        self.linear6 = nn.Linear(hidden_dim, 1)
        self.apply(weights_init_)

    def _pre_init(self, d):
        pass

    def forward(self, state, action):
        xu = torch.cat([state, action], 1)
        q1 = self.linear3(F.relu(self.linear2(F.relu(self.linear1(xu)))))
        q2 = self.linear6(F.relu(self.linear5(F.relu(self.linear4(xu)))))
        if self.squash:
            return torch.sigmoid(q1), torch.sigmoid(q2)
        return q1, q2


class QNetwork(_TwinQ):
    """Task critic (model.py:49-76)."""


class QNetworkConstraint(_TwinQ):
    """Safety critic Q_risk: sigmoid outputs (model.py:172-199).  `bn1` is declared by the
    reference (:175) but never used in forward; it is kept so state_dicts and optimiser
    parameter lists line up."""

    squash = True

    def _pre_init(self, d):
        self.bn1 = nn.BatchNorm1d(d)


class _PolicyBase(nn.Module):
    def __init__(self, num_inputs, hidden_dim, action_space):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        scale, bias = _space_affine(action_space)
        # plain attributes in the reference (moved by its own .to()); buffers here, excluded
        # from state_dict so reference checkpoints still load
        self.register_buffer("action_scale", scale, persistent=False)
        self.register_buffer("action_bias", bias, persistent=False)

    def trunk(self, state):
        return F.relu(self.linear2(F.relu(self.linear1(state))))


class GaussianPolicy(_PolicyBase):
    """tanh-Gaussian SAC policy (model.py:295-343)."""

    def __init__(self, num_inputs, num_actions, hidden_dim, action_space=None):
        super().__init__(num_inputs, hidden_dim, action_space)
        self.mean_linear = nn.Linear(hidden_dim, num_actions)
        self.log_std_linear = nn.Linear(hidden_dim, num_actions)
        self.apply(weights_init_)

    def forward(self, state):
        x = self.trunk(state)
        mean = self.mean_linear(x)
        log_std = torch.clamp(self.log_std_linear(x), min=LOG_SIG_MIN, max=LOG_SIG_MAX)
        return mean, log_std

    def sample(self, state, eps=None):
        """-> (action, log_prob[N,1], tanh(mean) action).  `eps` ~ N(0,1) of the mean's shape
        replaces the internal rsample draw."""
        mean, log_std = self.forward(state)
        std = log_std.exp()
        if eps is None:
            eps = torch.randn_like(mean)
        x_t = mean + std * eps                                    # rsample (:329)
        y_t = torch.tanh(x_t)
        action = y_t * self.action_scale + self.action_bias
        # Normal(mean,std).log_prob(x_t) with (x_t-mean)/std == eps
        log_prob = -0.5 * eps.pow(2) - log_std - 0.5 * math.log(2 * math.pi)
        log_prob = log_prob - torch.log(self.action_scale * (1 - y_t.pow(2)) + EPSILON)
        log_prob = log_prob.sum(1, keepdim=True)
        mean_action = torch.tanh(mean) * self.action_scale + self.action_bias
        return action, log_prob, mean_action


class DeterministicPolicy(_PolicyBase):
    """`--policy Deterministic` (model.py:447-485): tanh mean + clipped N(0, 0.1) noise."""

    def __init__(self, num_inputs, num_actions, hidden_dim, action_space=None):
        super().__init__(num_inputs, hidden_dim, action_space)
        self.mean = nn.Linear(hidden_dim, num_actions)
        self.num_actions = num_actions
        self.apply(weights_init_)

    def forward(self, state):
        return torch.tanh(self.mean(self.trunk(state))) * self.action_scale + self.action_bias

    def sample(self, state, eps=None):
        mean = self.forward(state)
        if eps is None:                       # one noise vector shared by the batch (:476-478)
            eps = torch.randn(self.num_actions, device=mean.device)
        noise = (eps * 0.1).clamp(-0.25, 0.25)
        return mean + noise, torch.zeros((), device=mean.device), mean


class StochasticPolicy(_PolicyBase):
    """Model-free recovery policy (model.py:489-530): tanh mean on the action box, one
    learnable state-independent log_std (init log 0.1, floor log 1e-6), unsquashed sample."""

    def __init__(self, num_inputs, num_actions, hidden_dim, action_space=None):
        super().__init__(num_inputs, hidden_dim, action_space)
        self.mean = nn.Linear(hidden_dim, num_actions)
        # float32 (the reference's float64 parameter under numpy>=1.24 breaks its own update)
        self.log_std = nn.Parameter(torch.full((num_actions,), math.log(0.1), dtype=torch.float32))
        self.min_log_std = math.log(1e-6)
        self.apply(weights_init_)

    def forward(self, state):
        mean = torch.tanh(self.mean(self.trunk(state))) * self.action_scale + self.action_bias
        log_std = torch.clamp(self.log_std, min=self.min_log_std)
        std = torch.exp(log_std).unsqueeze(0).expand_as(mean)
        return Normal(mean, std, validate_args=False)  # validation would sync the stream

    def sample(self, state, eps=None):
        dist = self.forward(state)
        if eps is None:
            eps = torch.randn_like(dist.mean)
        action = dist.mean + dist.stddev * eps
        return action, dist.log_prob(action).sum(-1), dist.mean
