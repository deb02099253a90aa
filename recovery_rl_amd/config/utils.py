"""Helpers of the PETS ensemble (reference: config/utils.py:6-24)."""
import numpy as np
import torch
from torch import nn


def swish(x):
    return x * torch.sigmoid(x)                                    # config/utils.py:6-7


def truncated_normal(size, std, generator=None):
    """N(0,1) truncated to [-2,2], times std (config/utils.py:10-12; scipy truncnorm.rvs in the
    reference -- here rejection sampling with torch's generator: same distribution)."""
    out = torch.empty(size, dtype=torch.float32)
    out.normal_(generator=generator)
    bad = out.abs() > 2
    while bad.any():
        out[bad] = torch.empty(int(bad.sum()), dtype=torch.float32).normal_(generator=generator)
        bad = out.abs() > 2
    return out * std


def get_affine_params(ensemble_size, in_features, out_features, generator=None):
    """Weights ~ truncnorm * 1/(2 sqrt(in)), zero biases (config/utils.py:15-24)."""
    w = truncated_normal((ensemble_size, in_features, out_features),
                         std=1.0 / (2.0 * np.sqrt(in_features)), generator=generator)
    b = torch.zeros(ensemble_size, 1, out_features, dtype=torch.float32)
    return nn.Parameter(w), nn.Parameter(b)
