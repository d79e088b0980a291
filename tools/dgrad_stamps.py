#!/usr/bin/env python3
"""GPU box: the trunk data-gradient launches of the training step one by one at batch 32 (128 x 128 x 128): time per launch
(HIP events, interleaved rounds) and — with a -DACG_STAMP build loaded through ACGAN_HIP_LIB — where a tile's cycles go
(set-up, main loop, epilogue of consumer wave 0) and how the workgroups' start times spread.
    [ACGAN_HIP_LIB=build/lib_stamp.so] python tools/dgrad_stamps.py [--batch 32] [--iters 20]"""
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
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    N, H, W, C = a.batch, 128, 128, 128
    dev, st = torch.device("cuda"), ops._stream()
    d = ops.conv_desc(N, H, W, C, C, 3, 1, 1, 1, C, C)
    D = ctypes.byref(d)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda s=1.0: torch.randn((N, H, W, C), device=dev, generator=g) * s
    enc = lambda x: (lambda y: (_lib.call("acg_s16_encode", P(x), P(y), x.numel(), st), y)[1])(torch.empty_like(x))
    x, dy, skip, xn = torch.relu(rnd()), rnd(1e-3), rnd(1e-3), rnd()
    xs, dys = enc(x), enc(dy)
    w = torch.randn((C, C, 3, 3), device=dev, generator=g) * 0.05
    pk = ops.PackedConv(w, torch.zeros(C, device=dev), C, C)
    nb = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
    ws = ops.workspace(max(nb, 1))
    nwords = (x.numel() + 31) // 32
    mk = lambda: torch.randint(-2 ** 31, 2 ** 31 - 1, (nwords,), device=dev, dtype=torch.int32, generator=g)
    m1, m2, m3 = mk(), mk(), mk()
    mean, rstd = torch.randn(N * C, device=dev, generator=g), torch.rand(N * C, device=dev, generator=g) + 0.5
    dx = torch.empty_like(x)
    part = torch.empty((N, H * W // 128, 2, C), device=dev)
    y = torch.empty_like(x)
    spart = torch.empty_like(part)

    def sums_desc(mask):
        ns = _lib.NormSumsDesc()
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta, ns.gstride = P(xn), P(mean), P(rstd), None, None, 0
        ns.sign_mask, ns.act, ns.part = P(mask), ops.ACT_RELU, P(part)
        return ns
    ns_a, ns_b = sums_desc(m2), sums_desc(m2)
    runs = [
        ("fwd + stats", lambda: _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 0, P(spart), 0, st)),
        ("dgrad plain fp32", lambda: _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx), P(ws), nb, None, None, None, 0, st)),
        ("dgrad relu-bitmask S16 out", lambda: _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dx), P(ws), nb, P(m3), st)),
        ("dgrad + sums (no addend)", lambda: _lib.call("acg_conv2d_bwd_data_s16_sums", D, P(dys), P(pk.wb), P(dx), P(ws), nb, None, None, ctypes.byref(ns_b), st)),
        ("dgrad + addend/mask + sums", lambda: _lib.call("acg_conv2d_bwd_data_s16_sums", D, P(dys), P(pk.wb), P(dx), P(ws), nb, P(skip), P(m1), ctypes.byref(ns_a), st)),
    ]
    for _, f in runs:
        f()
    torch.cuda.synchronize()
    for rnd_i in range(3):
        line = "round %d:" % rnd_i
        for nm, f in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            line += "  %s %.4f ms" % (nm, e0.elapsed_time(e1) / a.iters)
        print(line, flush=True)
    lib = _lib.load()
    if hasattr(lib, "acg_debug_pre_tile"):
        nwg = N * H * W // 128
        for nm, f in runs:
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (nwg * 4))()
            assert lib.acg_debug_pre_tile(buf, nwg * 4) == 0
            r = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
            t0 = r[:, 3] - r[:, 3].min()       # 100 MHz ticks
            print("%-28s set-up %6.0f  loop %6.0f  epilogue %6.0f cycles (median over %d tiles; p10 / p90 of the epilogue %.0f / %.0f);"
                  " start times: first round within %.1f us, all within %.1f us"
                  % (nm, np.median(r[:, 0]), np.median(r[:, 1]), np.median(r[:, 2]), nwg, np.percentile(r[:, 2], 10),
                     np.percentile(r[:, 2], 90), np.sort(t0)[511] / 100.0, t0.max() / 100.0), flush=True)
            # phase spread: for every tile, how far into the previous tile's period did its slot's next tile start (mod period)
            per = np.median(r[:, 0] + r[:, 1] + r[:, 2])
            print("    tile period %.0f cycles; epilogue share %.1f %%" % (per, 100 * np.median(r[:, 2]) / per))


if __name__ == "__main__":
    main()
