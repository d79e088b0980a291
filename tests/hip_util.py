"""Shared helpers for the GPU parity tests (HIP path vs oracle / golden fixtures)."""
import numpy as np
import torch

import dtgan_amd  # noqa: F401  (repo-root shim)
from oracle import recipe
from oracle.tape import T, backward, leaf  # noqa: F401

DEV = "cuda"


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return x.requires_grad_(True) if grad else x


def n(x):
    return x.detach().cpu().numpy()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def load_recipe(net, net_name, seed, flavour):
    """same weights as the oracle/golden side: keyed by the first (unique) parameter name"""
    with torch.no_grad():
        for k, p in dict(net.named_parameters()).items():
            p.copy_(torch.from_numpy(recipe.param(seed, net_name, k, tuple(p.shape), flavour)).to(p.device))
    from dtgan_amd.modules import mark_dirty
    mark_dirty(net)
    return net
