#!/usr/bin/env python3
"""Kernel microbenchmark (GPU box): times conv fwd / bwd-data / bwd-weight launches through the C ABI with HIP
events, for the layer shapes of the headline workload.
    python tools/microbench_conv.py [--only resblock] [--iters 20] [--precision bf16x3] [--json out.json] [--mark]
--mark: a pad_vector_kernel launch separates the segments, so that a rocprofv3 counter pass over this script can be
cut into (layer, pass) segments by tools/summarize_layer_traffic.py; --json records the segment order, the HIP-event
time and the algorithmic bytes (each tensor once, REAL channel counts) of every segment."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402
from dtgan_amd import _lib, ops  # noqa: E402

# name, N, H, W, Ci, Co, K, stride, pad, mode(0 zero,1 reflect), real Ci, real Co, transposed
# (transposed: the descriptor is the Conv2d whose adjoint the ConvTranspose2d is; H, W = its INPUT = the convT OUTPUT side)
SHAPES = [
    ("resblock_3x3_128", 32, 128, 128, 128, 128, 3, 1, 1, 1, 128, 128, 0),
    ("stem_7x7_3to32", 32, 256, 256, 4, 32, 7, 1, 3, 1, 3, 32, 0),
    ("a2_3x3_32to64", 32, 256, 256, 32, 64, 3, 1, 1, 0, 32, 64, 0),
    ("a3_3x3s2_64to128", 32, 256, 256, 64, 128, 3, 2, 1, 0, 64, 128, 0),
    ("a6_convT_128to64", 32, 256, 256, 64, 128, 3, 2, 1, 0, 64, 128, 1),
    ("a7_3x3_64to32", 32, 256, 256, 64, 32, 3, 1, 1, 0, 64, 32, 0),
    ("a8_7x7_32to3", 32, 256, 256, 32, 4, 7, 1, 3, 0, 32, 3, 0),
    ("DB_4x4_128to256", 32, 64, 64, 128, 256, 4, 1, 1, 0, 128, 256, 0),
    ("DB_4x4_256to256", 32, 63, 63, 256, 256, 4, 1, 1, 0, 256, 256, 0),
    ("DB_4x4s2_3to64", 32, 256, 256, 4, 64, 4, 2, 1, 0, 3, 64, 0),
    ("DB_4x4s2_64to128", 32, 128, 128, 64, 128, 4, 2, 1, 0, 64, 128, 0),
    ("DB_head_4x4_256to1", 32, 62, 62, 256, 4, 4, 1, 1, 0, 256, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--batch", type=int, default=0, help="override N of every shape")
    ap.add_argument("--mark", action="store_true")
    ap.add_argument("--json", default="")
    ap.add_argument("--digest", action="store_true", help="seeded inputs; print a bit-pattern checksum of every output (A/B of two builds / switches)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    ops.set_precision(a.precision)
    st = ops._stream()
    MARK = 1792   # pad_vector_kernel with a grid of 7 x 256 threads = segment marker (bias pads launch one block)
    mark_src, mark_dst = torch.zeros(8, device=dev), torch.zeros(MARK, device=dev)
    segs = []
    for name, N, H, W, Ci, Co, K, s, p, mode, Cir, Cor, tr in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        N = a.batch or N
        if a.digest:
            torch.manual_seed(1234)
        d = ops.conv_desc(N, H, W, Ci, Co, K, s, p, mode, Cir, Cor)
        x = torch.randn((N, H, W, Ci), device=dev)           # conv-input side
        dy = torch.randn((N, d.Ho, d.Wo, Co), device=dev)    # conv-output side
        w = torch.randn((Cor, Cir, K, K), device=dev) * 0.05
        b = torch.randn(Cir if tr else Cor, device=dev)
        pk = ops.PackedConv(w, b, ops.cpad(Ci), ops.cpad(Co))   # packed widths: multiples of 16 (the tensors may be C4)
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        nb_d = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
        nb_w = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
        ws = ops.workspace(max(nb_d, nb_w, 1))
        P = ops._ptr
        if tr:   # ConvTranspose2d: forward maps the small side (dy-shaped) to the large side (x-shaped)
            calls = {
                "fwd": lambda: _lib.call("acg_conv_transpose2d_fwd", ctypes.byref(d), P(dy), P(pk.wb), P(pk.bias), P(dx), 0, st),
                "dgrad": lambda: _lib.call("acg_conv_transpose2d_bwd_data", ctypes.byref(d), P(x), P(pk.wf), P(y), st),
                "wgrad": lambda: _lib.call("acg_conv_transpose2d_bwd_weight", ctypes.byref(d), P(dy), P(x), P(dw), P(db), Cor, Cir, P(ws), nb_w, 0, st),
            }
        else:
            calls = {
                "fwd": lambda: _lib.call("acg_conv2d_fwd", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y), 1, st),
                "dgrad": lambda: _lib.call("acg_conv2d_bwd_data", ctypes.byref(d), P(dy), P(pk.wb), P(dx), P(ws), nb_d, st),
                "wgrad": lambda: _lib.call("acg_conv2d_bwd_weight", ctypes.byref(d), P(x), P(dy), P(dw), P(db), Cor, Cir, P(ws), nb_w, 0, st),
            }
        flops = 2.0 * N * d.Ho * d.Wo * Cor * Cir * K * K
        algo = 4.0 * (N * H * W * Cir + N * d.Ho * d.Wo * Cor + K * K * Cir * Cor)     # each tensor once, real channels
        stored = 4.0 * (N * H * W * Ci + N * d.Ho * d.Wo * Co + K * K * Ci * Co)       # as stored (C16)
        line = "%-20s %6.1f GF(real)" % (name, flops / 1e9)
        for what in a.what.split(","):
            f = calls[what]
            for _ in range(2):
                f()
            if a.mark:
                _lib.call("acg_pad_vector", P(mark_src), 8, P(mark_dst), MARK, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            if a.mark:
                _lib.call("acg_pad_vector", P(mark_src), 8, P(mark_dst), MARK, st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            kern = _lib.query("acg_last_kernel").decode()
            if a.digest:
                outs = {"fwd": (dx if tr else y,), "dgrad": (y if tr else dx,), "wgrad": (dw, db)}[what]
                print("    digest %s %s: %s  [%s]" % (name, what, " ".join("%016x" % (int(o.contiguous().view(torch.int32).to(torch.int64).sum().item()) & (2 ** 64 - 1)) for o in outs), kern), flush=True)
            segs.append(dict(layer=name, what=what, ms=ms, iters=a.iters, flops=flops, algorithmic_bytes=algo,
                             stored_bytes=stored, last_kernel=kern))
            line += " | %s %7.3f ms %6.1f TF %5.2f TB/s" % (what, ms, flops / ms / 1e9, algo / ms / 1e9)
        print(line, flush=True)
    if a.json:
        json.dump(dict(precision=a.precision, segments=segs), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
