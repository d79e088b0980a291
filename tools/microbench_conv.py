#!/usr/bin/env python3
"""Kernel microbenchmark (GPU box): times conv fwd / bwd-data / bwd-weight launches through the C ABI with HIP
events, for the layer shapes of the headline workload.   python tools/microbench_conv.py [--only resblock] [--iters 20]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402
from dtgan_amd import _lib, ops  # noqa: E402

# name, N, H, W, Ci, Co, K, stride, pad, mode(0 zero,1 reflect), real Ci, real Co
SHAPES = [
    ("resblock_3x3_128", 32, 128, 128, 128, 128, 3, 1, 1, 1, 128, 128),
    ("stem_7x7_3to32", 32, 256, 256, 16, 32, 7, 1, 3, 1, 3, 32),
    ("a2_3x3_32to64", 32, 256, 256, 32, 64, 3, 1, 1, 0, 32, 64),
    ("a3_3x3s2_64to128", 32, 256, 256, 64, 128, 3, 2, 1, 0, 64, 128),
    ("a7_3x3_64to32", 32, 256, 256, 64, 32, 3, 1, 1, 0, 64, 32),
    ("a8_7x7_32to3", 32, 256, 256, 32, 16, 7, 1, 3, 0, 32, 3),
    ("DB_4x4_128to256", 32, 64, 64, 128, 256, 4, 1, 1, 0, 128, 256),
    ("DB_4x4_256to256", 32, 63, 63, 256, 256, 4, 1, 1, 0, 256, 256),
    ("DB_4x4s2_3to64", 32, 256, 256, 16, 64, 4, 2, 1, 0, 3, 64),
    ("DB_head_4x4_256to1", 32, 62, 62, 256, 16, 4, 1, 1, 0, 256, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--precision", default="f32")
    a = ap.parse_args()
    dev = torch.device("cuda")
    ops.set_precision(a.precision)
    st = ops._stream()
    for name, N, H, W, Ci, Co, K, s, p, mode, Cir, Cor in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        d = ops.conv_desc(N, H, W, Ci, Co, K, s, p, mode, Cir, Cor)
        x = torch.randn((N, H, W, Ci), device=dev)
        dy = torch.randn((N, d.Ho, d.Wo, Co), device=dev)
        w = torch.randn((Cor, Cir, K, K), device=dev) * 0.05
        b = torch.randn(Cor, device=dev)
        pk = ops.PackedConv(w, b, Ci, Co)
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        nb_d = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
        nb_w = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
        ws = ops.workspace(max(nb_d, nb_w, 1))
        P = ops._ptr
        calls = {
            "fwd": lambda: _lib.call("acg_conv2d_fwd", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y), 1, st),
            "dgrad": lambda: _lib.call("acg_conv2d_bwd_data", ctypes.byref(d), P(dy), P(pk.wb), P(dx), P(ws), nb_d, st),
            "wgrad": lambda: _lib.call("acg_conv2d_bwd_weight", ctypes.byref(d), P(x), P(dy), P(dw), P(db), Cor, Cir, P(ws), nb_w, st),
        }
        flops = 2.0 * N * d.Ho * d.Wo * Cor * Cir * K * K
        line = "%-20s %6.1f GF(real)" % (name, flops / 1e9)
        for what in a.what.split(","):
            f = calls[what]
            for _ in range(2):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            line += " | %s %7.3f ms %6.1f TF" % (what, ms, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
