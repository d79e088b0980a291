#!/usr/bin/env python3
"""GPU box: the 3x3 stride-2 64 -> 128 downsample forward (+ tile statistics) of the training step at batch 32 (256 x 256
in, 128 x 128 out) and its twin, the ConvTranspose data gradient: time per launch (HIP events) and — with a -DACG_STAMP
build loaded through ACGAN_HIP_LIB — where a tile's cycles go (set-up, main loop, epilogue of consumer wave 0).
    [ACGAN_HIP_LIB=build/lib_stamp.so] python tools/ws_stamps.py [--batch 32] [--iters 20]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dtgan_amd  # noqa: E402,F401
from dtgan_amd import _lib, ops  # noqa: E402

P = ops._ptr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    N, H, W, Ci, Co = a.batch, 256, 256, 64, 128
    dev, st = torch.device("cuda"), ops._stream()
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 2, 1, 0, Ci, Co)
    D = ctypes.byref(d)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, Ci), device=dev, generator=g)
    w = torch.randn((Co, Ci, 3, 3), device=dev, generator=g) * 0.05
    pk = ops.PackedConv(w, torch.zeros(Co, device=dev), Ci, Co)
    y = torch.empty((N, d.Ho, d.Wo, Co), device=dev)
    spart = torch.empty((N, d.Ho * d.Wo // 128, 2, Co), device=dev)
    runs = [
        ("fwd + stats", lambda: _lib.call("acg_conv2d_fwd_stats", D, P(x), P(pk.wf), P(pk.bias), P(y), P(spart), st)),
        ("fwd", lambda: _lib.call("acg_conv2d_fwd", D, P(x), P(pk.wf), P(pk.bias), P(y), 0, st)),
    ]
    lib = _lib.load()
    fn = getattr(lib, "acg_debug_ws_tile", None) if hasattr(lib, "acg_debug_ws_tile") else None
    for name, f in runs:
        f()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        line = []
        for name, f in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            line.append("%s %.3f ms" % (name, e0.elapsed_time(e1) / a.iters))
        print("round %d: %s  [%s]" % (r, "  ".join(line), lib.acg_last_kernel().decode() if hasattr(lib, "acg_last_kernel") else ""), flush=True)
    if fn is not None:
        for name, f in runs:
            f()
            torch.cuda.synchronize()
            nwg = N * d.Ho * d.Wo // 128
            buf = (ctypes.c_ulonglong * (nwg * 4))()
            if fn(buf, nwg * 4) != 0:
                print("stamps: read failed")
                return
            t = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
            med = np.median(t[:, :3], axis=0)
            start = (t[:, 3] - t[:, 3].min()) / 100.0   # us (100 MHz clock)
            print("%-12s set-up %6.0f  loop %6.0f  epilogue %6.0f  period %6.0f cycles (median of %d tiles); last start %.0f us"
                  % (name, med[0], med[1], med[2], med.sum(), nwg, start.max()), flush=True)


if __name__ == "__main__":
    main()
