#!/usr/bin/env python3
"""GPU box: what the norm-backward sums cost inside the fp32-operand data-gradient kernels of the full-resolution layers
(acg_conv2d_bwd_data_sums) against the plain data gradient (acg_conv2d_bwd_data) plus the norm's own first pass
(acg_norm_bwd_sums, the same reads as norm_bwd_partial), at the bench geometry — interleaved, HIP events.
    python tools/sums_ab.py [--batch 32] [--size 256] [--iters 20]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402,F401
from dtgan_amd import _lib, ops  # noqa: E402

P = ops._ptr


def timeit(f, iters):
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def layer(name, N, H, W, Ci, Co, K, stride, pad, Cir, Cor, iters):
    dev = torch.device("cuda")
    st = ops._stream()
    pk = ops.PackedConv(torch.randn((Cor, Cir, K, K), device=dev) * 0.05, torch.zeros(Cor, device=dev), ops.cpad(Cir), ops.cpad(Cor))
    d = ops.conv_desc(N, H, W, pk.Cis, pk.Cos, K, stride, pad, 0, Cir, Cor)
    D = ctypes.byref(d)
    if not _lib.query("acg_conv2d_bwd_data_sums_supported", D):
        print("%-28s not supported" % name)
        return
    Ho, Wo = H // stride, W // stride
    dy = torch.randn((N, Ho, Wo, pk.Cos), device=dev) * 1e-2
    xn = torch.randn((N, H, W, pk.Cis), device=dev)
    mean, rstd = torch.randn(N * pk.Cis, device=dev) * 0.1, torch.rand(N * pk.Cis, device=dev) + 0.5
    gamma, beta = torch.randn(pk.Cis, device=dev), torch.randn(pk.Cis, device=dev)
    dx = torch.empty((N, H, W, pk.Cis), device=dev)
    part = torch.empty((N, H * W // 128, 2, pk.Cis), device=dev)
    nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
    ws = ops.workspace(max(nbw, 1))
    ns = _lib.NormSumsDesc()
    ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta, ns.gstride, ns.sign_mask, ns.act, ns.part = P(xn), P(mean), P(rstd), P(gamma), P(beta), 0, None, ops.ACT_RELU, P(part)
    nb = _lib.query("acg_norm_workspace_bytes", N, H * W, pk.Cis)
    nws = ops.workspace(nb, slot=1)
    sums = torch.empty(N * 2 * pk.Cis, device=dev)
    y = torch.relu(xn)

    def fused():
        _lib.call("acg_conv2d_bwd_data_sums", D, P(dy), P(pk.wb), P(dx), P(ws), nbw, ctypes.byref(ns), st)

    def plain():
        _lib.call("acg_conv2d_bwd_data", D, P(dy), P(pk.wb), P(dx), P(ws), nbw, st)

    def first_pass():   # (reads dy, x and y: one stream more than norm_bwd_partial<.., 1>, which recomputes the mask from x)
        _lib.call("acg_norm_bwd_sums", P(dx), P(y), P(xn), P(mean), P(rstd), P(sums), N, H * W, pk.Cis, ops.ACT_RELU, P(nws), nb, st)

    for rnd in range(2):
        tf = timeit(fused, iters); kf = _lib.query("acg_last_kernel").decode()
        tp = timeit(plain, iters); kp = _lib.query("acg_last_kernel").decode()
        tn = timeit(first_pass, iters)
        print("%-28s fused %.4f ms (%s) | plain %.4f ms (%s) + norm first pass %.4f ms = %.4f" % (name, tf, kf, tp, kp, tn, tp + tn), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    N, S = a.batch, a.size
    layer("a3 3x3 s2 64->128 dgrad", N, S, S, 64, 128, 3, 2, 1, 64, 128, a.iters)
    layer("a2 3x3 32->64 dgrad", N, S, S, 32, 64, 3, 1, 1, 32, 64, a.iters)
    layer("a8 7x7 32->3 dgrad", N, S, S, 32, 3, 7, 1, 3, 32, 3, a.iters)
    layer("a7 3x3 64->32 dgrad (rows)", N, S, S, 64, 32, 3, 1, 1, 64, 32, a.iters)


if __name__ == "__main__":
    main()
