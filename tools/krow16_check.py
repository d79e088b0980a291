#!/usr/bin/env python3
"""GPU box: the trunk weight gradient on v_mfma_f32_16x16x32_bf16 (krow16, conv_wgrad_tr_s16.inc, opt-in through
ACG_KROW_M16) against the shipped 32x32x16 form (krow32) — the same bf16x3 products summed in another order, so dw / db agree
to fp32 summation noise (<= 1e-6 of the largest entry), not bit for bit.  The switch is read per call.
    ACG_DEBUG_SWITCHES=1 python tools/krow16_check.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ACG_DEBUG_SWITCHES", "1")
import torch  # noqa: E402

import dtgan_amd  # noqa: E402,F401
from dtgan_amd import _lib, ops  # noqa: E402

P = ops._ptr
TOL = 1e-6


def run(N, H, W, C, reflect):
    dev = torch.device("cuda")
    st = ops._stream()
    d = ops.conv_desc(N, H, W, C, C, 3, 1, 1, 1 if reflect else 0, C, C)
    D = ctypes.byref(d)
    assert _lib.query("acg_conv2d_s16_supported", D), "layer not S16-capable"
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((N, H, W, C), device=dev, generator=g).clamp_min(0)
    dy = torch.randn((N, H, W, C), device=dev, generator=g) * 1e-3
    xs, dys = torch.empty_like(x), torch.empty_like(dy)
    _lib.call("acg_s16_encode", P(x), P(xs), x.numel(), st)
    _lib.call("acg_s16_encode", P(dy), P(dys), dy.numel(), st)
    nb = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", D)
    ws = ops.workspace(nb)
    out, names = {}, {}
    for tag, on in (("krow32", False), ("krow16", True)):
        if on:
            os.environ["ACG_KROW_M16"] = "1"
        else:
            os.environ.pop("ACG_KROW_M16", None)
        dw = torch.full((C, C, 3, 3), float("nan"), device=dev)
        db = torch.full((C,), float("nan"), device=dev)
        _lib.call("acg_conv2d_bwd_weight_s16", D, P(xs), P(dys), P(dw), P(db), C, C, P(ws), nb, 0, st)
        names[tag] = _lib.query("acg_last_kernel").decode()
        torch.cuda.synchronize()
        out[tag] = (dw, db)
    os.environ.pop("ACG_KROW_M16", None)
    ok = names["krow32"] == "wgrad_x3_krow_s16" and names["krow16"] == "wgrad_x3_krow_s16<16x16x32>"
    if not ok:
        print("KERNEL SELECTION WRONG", names)
    for i, nm in enumerate(("dw", "db")):
        a, b = out["krow16"][i], out["krow32"][i]
        err = ((a - b).abs().max() / b.abs().max()).item()
        good = bool(torch.isfinite(a).all().item()) and err <= TOL
        print("N=%d %dx%d reflect=%d %s: max|diff| / max|ref| = %.2e %s" % (N, H, W, int(reflect), nm, err, "OK" if good else "MISMATCH"), flush=True)
        ok &= good
    return ok


def main():
    ops.set_precision("bf16x3")
    ok = run(1, 128, 128, 128, True)       # the N = 1 trunk geometry
    ok &= run(2, 64, 64, 128, True)        # two segments per grid row
    ok &= run(3, 32, 64, 128, True)        # odd image count, 32-pixel-row map
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
