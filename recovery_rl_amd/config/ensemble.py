"""Bootstrapped probabilistic ensemble of PETS (reference: PtModel in config/navigation1.py:23-96,
identical copies in config/navigation2.py and config/maze.py).

Same parameter names (lin0_w .. lin3_b, inputs_mu, inputs_sigma, max_logvar, min_logvar) so a
reference state_dict loads unchanged.  All nets of the ensemble run as ONE batched matmul over
the leading ensemble dimension ([E, rows, in] x [E, in, out]) -- MFMA through rocBLAS bmm.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from .utils import get_affine_params, swish


class PtModel(nn.Module):
    HIDDEN = 200

    def __init__(self, ensemble_size, in_features, out_features, generator=None):
        super().__init__()
        self.num_nets = ensemble_size
        self.in_features = in_features
        self.out_features = out_features
        h = self.HIDDEN
        self.lin0_w, self.lin0_b = get_affine_params(ensemble_size, in_features, h, generator)
        self.lin1_w, self.lin1_b = get_affine_params(ensemble_size, h, h, generator)
        self.lin2_w, self.lin2_b = get_affine_params(ensemble_size, h, h, generator)
        self.lin3_w, self.lin3_b = get_affine_params(ensemble_size, h, out_features, generator)
        self.inputs_mu = nn.Parameter(torch.zeros(in_features), requires_grad=False)
        self.inputs_sigma = nn.Parameter(torch.zeros(in_features), requires_grad=False)
        self.max_logvar = nn.Parameter(torch.ones(1, out_features // 2, dtype=torch.float32) / 2.0)
        self.min_logvar = nn.Parameter(-torch.ones(1, out_features // 2, dtype=torch.float32) * 10.0)

    def compute_decays(self):
        """Weight decay terms (navigation1.py:52-59)."""
        return (0.00025 * (self.lin0_w ** 2).sum() + 0.0005 * (self.lin1_w ** 2).sum()
                + 0.0005 * (self.lin2_w ** 2).sum() + 0.00075 * (self.lin3_w ** 2).sum()) / 2.0

    @torch.no_grad()
    def fit_input_stats(self, data):
        """Input standardisation statistics (navigation1.py:61-69): sigma < 1e-12 -> 1."""
        data = torch.as_tensor(data, device=self.inputs_mu.device)
        mu = data.double().mean(0, keepdim=True)
        sigma = data.double().std(0, unbiased=False, keepdim=True)
        sigma = torch.where(sigma < 1e-12, torch.ones_like(sigma), sigma)
        # written IN PLACE: a captured hipGraph (the model-based iteration, experiment.py `mb_graph`) packs the planner's
        # weights from these addresses on every replay, so a re-fit must not move them.  The reference's [1, in] shape
        # (a view of the same storage) is kept.
        for prm, val in ((self.inputs_mu, mu), (self.inputs_sigma, sigma)):
            flat = prm.data.reshape(-1)
            assert flat.data_ptr() == prm.data.data_ptr()
            flat.copy_(val.float().reshape(-1))
            prm.data = flat.view(1, -1)

    def forward(self, inputs, ret_logvar=False):
        """inputs [E, rows, in] -> (mean, var or logvar) each [E, rows, out/2] (navigation1.py:71-96)."""
        x = (inputs - self.inputs_mu) / self.inputs_sigma
        x = swish(torch.baddbmm(self.lin0_b, x, self.lin0_w))
        x = swish(torch.baddbmm(self.lin1_b, x, self.lin1_w))
        x = swish(torch.baddbmm(self.lin2_b, x, self.lin2_w))
        x = torch.baddbmm(self.lin3_b, x, self.lin3_w)
        half = self.out_features // 2
        mean, logvar = x[:, :, :half], x[:, :, half:]
        logvar = self.max_logvar - F.softplus(self.max_logvar - logvar)
        logvar = self.min_logvar + F.softplus(logvar - self.min_logvar)
        if ret_logvar:
            return mean, logvar
        return mean, torch.exp(logvar)
