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


def l2rel(a, b):
    """norm-wise relative error: robust to the isolated pointwise differences a flipped ReLU mask produces"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# Two conv arithmetic modes are parity paths (DESIGN.md "Arithmetic"):
#   f32    — v_mfma_f32_32x32x2_f32, every product exact fp32: the tight bounds pin indexing and fusion logic;
#   bf16x3 — fp32 operands split hi+lo into bf16, three bf16 MFMAs per product (operands carry 16 mantissa bits,
#            ~4e-6 rms per conv measured by tools/precision_probe.py): held to the north-star bar itself,
#            1e-3 relative on activations and losses.
PRECISIONS = ["f32", "bf16x3"]


class precision(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from dtgan_amd import ops
        self.before = ops.get_precision()
        ops.set_precision(self.name)

    def __exit__(self, *a):
        from dtgan_amd import ops
        ops.set_precision(self.before)


class Spy(object):
    """records (C-ABI entry point, kernel the dispatcher launched for it) while active"""

    def __enter__(self):
        from dtgan_amd import ops
        self.ops = ops
        self.real = ops._lib.call
        self.seen = []

        def call(name, *a):
            r = self.real(name, *a)
            if name.startswith("acg_conv"):
                self.seen.append((name, ops._lib.query("acg_last_kernel").decode()))
            else:
                self.seen.append((name, ""))
            return r
        ops._lib.call = call
        return self

    def __exit__(self, *a):
        self.ops._lib.call = self.real

    def entries(self):
        return set(n for n, _ in self.seen)

    def kernels(self, entry=None):
        """kernels of the launches whose entry point starts with `entry` (all convolution launches if None)"""
        return [k for n, k in self.seen if k and (entry is None or n.startswith(entry))]


def injected_dropout(seed):
    """--use_dropout fixtures: the k-th Dropout forward takes the seeded keep mask the fixture's generator gave the reference
    (oracle.ops.dropout_keep, NCHW) — packed into the bit words acg_dropout_apply reads (ops.DROPOUT_SOURCE)"""
    import contextlib
    from dtgan_amd import ops
    from oracle import ops as oops

    @contextlib.contextmanager
    def cm():
        if seed is None:
            yield
            return
        k = [0]

        def src(shape, C, p):
            N, H, W, Cp = shape
            keep = oops.dropout_keep(seed, k[0], (N, C, H, W), p)
            k[0] += 1
            return ops.pack_keep_bits(keep, Cp, torch.device("cuda", 0))
        ops.DROPOUT_SOURCE = src
        try:
            yield
        finally:
            ops.DROPOUT_SOURCE = None
    return cm()
