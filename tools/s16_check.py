#!/usr/bin/env python3
"""GPU box: the pre-split (S16) convolution kernels against the fp32-operand kernels of the same arithmetic (they consume
the same hi / lo halves in the same order, so outputs must agree bit for bit), then interleaved timings of both.
    python tools/s16_check.py [--time] [--batch 32]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dtgan_amd  # noqa: E402
from dtgan_amd import _lib, ops  # noqa: E402

P = ops._ptr


def enc(x):
    y = torch.empty_like(x)
    _lib.call("acg_s16_encode", P(x), P(y), x.numel(), ops._stream())
    return y


def dec(x):
    y = torch.empty_like(x)
    _lib.call("acg_s16_decode", P(x), P(y), x.numel(), ops._stream())
    return y


def check(name, a, b, exact=True, tol=1e-5):
    d = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = d == 0.0 if exact else d <= tol * ref
    print("%-44s max|diff| %.3e (ref max %.3e) %s" % (name, d, ref, "OK" if ok else "MISMATCH"), flush=True)
    return ok


def run(N, H, W, C, time_it, iters):
    dev = torch.device("cuda")
    st = ops._stream()
    d = ops.conv_desc(N, H, W, C, C, 3, 1, 1, 1, C, C)
    assert _lib.query("acg_conv2d_s16_supported", ctypes.byref(d)), "layer not S16-capable"
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((N, H, W, C), device=dev, generator=g)
    x = torch.where(x > 0, x, torch.zeros_like(x))          # ReLU-like activations
    dy = torch.randn((N, H, W, C), device=dev, generator=g) * 1e-3
    w = torch.randn((C, C, 3, 3), device=dev, generator=g) * 0.05
    b = torch.randn(C, device=dev, generator=g)
    skip = torch.randn((N, H, W, C), device=dev, generator=g)
    pk = ops.PackedConv(w, b, C, C)
    xs, dys = enc(x), enc(dy)
    ok = check("decode(encode(x)) vs x (2^-16 rel)", dec(xs), x, exact=False)
    nb_d = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
    nb_w = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
    ws = ops.workspace(max(nb_d, nb_w, 1))
    D = ctypes.byref(d)
    y0, y1, y2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    part0 = torch.empty((N, H * W // 128, 2, C), device=dev)
    part1 = torch.empty_like(part0)

    def fwd_ref():
        _lib.call("acg_conv2d_fwd_stats", D, P(x), P(pk.wf), P(pk.bias), P(y0), P(part0), st)

    def fwd_s16():
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y1), 0, P(part1), 0, st)

    fwd_ref(); fwd_s16()
    ok &= check("fwd + stats: y", y1, y0)
    ok &= check("fwd + stats: tile statistics", part1, part0)
    _lib.call("acg_conv2d_fwd", D, P(x), P(pk.wf), P(pk.bias), P(y0), 1, st)
    _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y1), 1, None, 0, st)
    ok &= check("fwd + ReLU (fp32 out)", y1, y0)
    _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y2), 1, None, 1, st)
    ok &= check("fwd + ReLU (S16 out) vs encode(fp32 out)", y2.view(torch.int32), enc(y0).view(torch.int32))

    # conv + ReLU, pre-split output + the sign bitmask of that output (Co % 32 == 0, whole-row tiles)
    masked = W % 128 == 0 and H % 32 == 0 and H >= 64 and _lib.query("acg_conv2d_bwd_data_s16_sums_supported", ctypes.byref(d))
    if masked:
        y3 = torch.empty_like(x)
        bits = torch.zeros((x.numel() + 31) // 32, device=dev, dtype=torch.int32)
        _lib.call("acg_conv2d_fwd_s16_mask", D, P(xs), P(pk.wf), P(pk.bias), P(y3), P(bits), st)
        ok &= check("fwd + ReLU (S16 out) + sign bitmask: y", y3.view(torch.int32), y2.view(torch.int32))
        want = (y0.reshape(-1, 32) > 0).to(torch.int64)       # y0: fp32 conv + ReLU output
        want = (want << torch.arange(32, device=dev)).sum(1)
        got = bits.to(torch.int64) & 0xFFFFFFFF
        ok &= check("fwd + ReLU (S16 out) + sign bitmask: bits", got.double(), want.double())
    # data gradients
    dx0, dx1 = torch.empty_like(x), torch.empty_like(x)

    def dg_ref():
        _lib.call("acg_conv2d_bwd_data", D, P(dy), P(pk.wb), P(dx0), P(ws), nb_d, st)

    def dg_s16():
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, None, None, None, 0, st)

    # whole-row tiles (W % 128 == 0): the pre-split data gradient runs on the un-padded grid with summed weight slabs for the
    # mirrored rows and a separate column term — the same sums in another order, no longer bit-identical
    unpad = W % 128 == 0 and H % 32 == 0 and H >= 64 and not os.environ.get("ACG_NO_UNPAD")
    dg_ref(); dg_s16()
    ok &= check("dgrad (fp32 out)", dx1, dx0, exact=not unpad, tol=2e-5)
    mask = None
    if (H * W * (C // 4)) % 8 == 0:
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, ((N * H * W * C + 31) // 32,), device=dev, dtype=torch.int32, generator=g)
    _lib.call("acg_conv2d_bwd_data_add", D, P(dy), P(pk.wb), P(skip), P(mask), P(dx0), P(ws), nb_d, st)
    _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, P(skip), P(mask), None, 0, st)
    ok &= check("dgrad + masked skip addend (fp32 out)", dx1, dx0, exact=not unpad, tol=2e-5)
    _lib.call("acg_conv2d_bwd_data_relu", D, P(dy), P(pk.wb), P(x), P(dx0), P(ws), nb_d, st)
    _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, None, None, P(xs), 1, st)
    if masked:   # the same data gradient masked by the sign BITMASK of x instead of by x itself
        xbits = (x.reshape(-1, 32) > 0).to(torch.int64)
        xbits = (xbits << torch.arange(32, device=dev)).sum(1)
        xbits = torch.where(xbits >= 2 ** 31, xbits - 2 ** 32, xbits).to(torch.int32)
        dx2 = torch.empty_like(x)
        _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dx2), P(ws), nb_d, P(xbits), st)
        ok &= check("dgrad masked by the sign bitmask vs masked by x (S16 out)", dx2.view(torch.int32), dx1.view(torch.int32))
    if unpad:
        ok &= check("dgrad * (x > 0) (S16 out) vs ref", dec(dx1), dx0, exact=False, tol=3e-5)
    else:
        ok &= check("dgrad * (x > 0) (S16 out) vs encode(ref)", dx1.view(torch.int32), enc(dx0).view(torch.int32))

    # weight gradient
    dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
    db0, db1 = torch.empty_like(b), torch.empty_like(b)

    def wg_ref():
        _lib.call("acg_conv2d_bwd_weight", D, P(x), P(dy), P(dw0), P(db0), C, C, P(ws), nb_w, 0, st)

    def wg_s16():
        _lib.call("acg_conv2d_bwd_weight_s16", D, P(xs), P(dys), P(dw1), P(db1), C, C, P(ws), nb_w, 0, st)

    wg_ref(); k0 = _lib.query("acg_last_kernel").decode()
    wg_s16(); k1 = _lib.query("acg_last_kernel").decode()
    ok &= check("wgrad: dw (%s vs %s)" % (k1, k0), dw1, dw0)
    ok &= check("wgrad: db (hi + lo vs fp32 sums)", db1, db0, exact=False)
    torch.cuda.synchronize()
    if os.environ.get("ACG_STAMPS") and N == 32 and hasattr(_lib.load(), "acg_debug_pre_stamps"):
        import numpy as np
        for name, f, nwg in (("fwd", fwd_s16, 4096), ("dgrad", dg_s16, 4096)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (nwg * 24))()
            assert _lib.load().acg_debug_pre_stamps(buf, nwg * 24) == 0
            raw = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8, 3).astype(np.float64)
            wait, work = raw[:, :4, 0], raw[:, :4, 1]
            print("%s consumer stamps: loop cycles per wave %.0f, barrier wait share %.1f %% (by wave %s)"
                  % (name, (wait + work).mean(), 100 * wait.sum() / (wait + work).sum(), np.round(100 * (wait / (wait + work)).mean(0), 1)))
            ebuf = (ctypes.c_ulonglong * (nwg * 4))()   # acg_debug_pre_tile: (set-up, loop, epilogue, start time) per tile — see
            assert _lib.load().acg_debug_pre_tile(ebuf, nwg * 4) == 0   # tools/dgrad_stamps.py for the per-launch-kind table
            epi = np.frombuffer(ebuf, dtype=np.uint64).reshape(nwg, 4).astype(np.float64)
            print("%s tile: set-up %.0f cycles (entry -> first stage), loop %.0f, epilogue %.0f" % (name, np.median(epi[:, 0]), np.median(epi[:, 1]), np.median(epi[:, 2])))
            pr = raw[:, 4:, :]
            tot = pr.sum(-1).mean()
            print("%s producer stamps: loop cycles per wave %.0f: issuing %.1f %%, waiting for the pieces to land %.1f %%, at the barrier %.1f %%"
                  % (name, tot, 100 * pr[..., 0].mean() / tot, 100 * pr[..., 1].mean() / tot, 100 * pr[..., 2].mean() / tot))
    if os.environ.get("ACG_STAMPS"):   # diagnostic library (-DACG_STAMP): per-wave (barrier wait, rest) cycles of the main loop
        import numpy as np
        for _ in range(5):
            wg_s16()
        torch.cuda.synchronize()
        need = 85 * 9 * C * C * 4 if N == 32 else None
        if need is not None:
            off = (need + 255) // 256 * 256 + 49152
            raw = ws[off: off + 255 * 8 * 16].view(torch.int64).cpu().numpy().reshape(255, 8, 2)
            wait, work = raw[..., 0].astype(np.float64), raw[..., 1].astype(np.float64)
            print("stamps (cycles per wave over the whole loop): wait mean %.0f  work mean %.0f  -> wait share %.1f %%; per-wave wait share min %.1f max %.1f"
                  % (wait.mean(), work.mean(), 100 * wait.sum() / (wait.sum() + work.sum()),
                     100 * (wait / (wait + work)).min(), 100 * (wait / (wait + work)).max()))
            print("wait share by wave index:", np.round(100 * (wait / (wait + work)).mean(0), 1))
    if not time_it:
        return ok
    flops = 2.0 * N * H * W * C * C * 9

    def dg_s16_add():    # + masked skip addend (fp32 out): the block's first convolution
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, P(skip), P(mask), None, 0, st)

    def dg_s16_relu():   # pre-split output masked by the sign of the convolution's own input: conv + ReLU in front
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, None, None, P(xs), 1, st)

    def dg_s16_bits():   # ... by the sign bitmask of that input
        _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dx1), P(ws), nb_d, P(xbits), st)

    for rnd in range(3):
        line = "round %d:" % rnd
        for nm, f in (("fwd ref", fwd_ref), ("fwd s16", fwd_s16), ("dgrad ref", dg_ref), ("dgrad s16", dg_s16),
                      ("dgrad s16 + addend", dg_s16_add), ("dgrad s16 relu", dg_s16_relu), ("dgrad s16 bits", dg_s16_bits),
                      ("wgrad ref", wg_ref), ("wgrad s16", wg_s16)):
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += "  %s %.3f ms (%.0f TF)" % (nm, ms, flops / ms / 1e9)
        print(line, flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    ok = run(2, 64, 64, 128, False, 0)      # two segments per tile
    ok &= run(1, 64, 128, 128, False, 0)    # un-padded data gradient, the smallest map
    ok &= run(1, 32, 32, 128, False, 0)     # four segments per tile
    ok &= run(2, 128, 128, 128, False, 0)   # the config-3 trunk geometry
    if a.time:
        ok &= run(a.batch, 128, 128, 128, True, a.iters)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
