#!/usr/bin/env python3
"""GPU box: the persistent trunk kernel (conv_x3_pp.hip) against the one-tile-per-workgroup kernel it replaces
(conv_x3_pre.hip), every launch kind, bit for bit, in one process (ACG_NO_PP is read per call), then interleaved timings.
    ACG_DEBUG_SWITCHES=1 python tools/pp_check.py [--time] [--batch 32] [--iters 20]"""
import argparse
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


def enc(x):
    y = torch.empty_like(x)
    _lib.call("acg_s16_encode", P(x), P(y), x.numel(), ops._stream())
    return y


def use_pp(on):
    os.environ["ACG_PP"] = "1"
    if on:
        os.environ.pop("ACG_NO_PP", None)
    else:
        os.environ["ACG_NO_PP"] = "1"


def same(name, a, b):
    a, b = a.view(torch.int32), b.view(torch.int32)
    bad = int((a != b).sum().item())
    print("%-52s %s" % (name, "identical" if bad == 0 else "MISMATCH in %d of %d words" % (bad, a.numel())), flush=True)
    return bad == 0


def run(N, H, W, C, time_it, iters):
    dev = torch.device("cuda")
    st = ops._stream()
    d = ops.conv_desc(N, H, W, C, C, 3, 1, 1, 1, C, C)
    D = ctypes.byref(d)
    assert _lib.query("acg_conv2d_s16_supported", D) and _lib.query("acg_conv2d_bwd_data_s16_sums_supported", D)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((N, H, W, C), device=dev, generator=g)
    x = torch.where(x > 0, x, torch.zeros_like(x))
    dy = torch.randn((N, H, W, C), device=dev, generator=g) * 1e-3
    w = torch.randn((C, C, 3, 3), device=dev, generator=g) * 0.05
    b = torch.randn(C, device=dev, generator=g)
    skip = torch.randn((N, H, W, C), device=dev, generator=g)
    xn = torch.randn((N, H, W, C), device=dev, generator=g)
    mean = torch.randn((N * C,), device=dev, generator=g) * 0.1
    rstd = torch.rand((N * C,), device=dev, generator=g) + 0.5
    nwords = (N * H * W * C + 31) // 32
    m_skip = torch.randint(-2 ** 31, 2 ** 31 - 1, (nwords,), device=dev, dtype=torch.int32, generator=g)
    m_norm = torch.randint(-2 ** 31, 2 ** 31 - 1, (nwords,), device=dev, dtype=torch.int32, generator=g)
    xbits = (x.reshape(-1, 32) > 0).to(torch.int64)
    xbits = (xbits << torch.arange(32, device=dev)).sum(1)
    xbits = torch.where(xbits >= 2 ** 31, xbits - 2 ** 32, xbits).to(torch.int32)
    pk = ops.PackedConv(w, b, C, C)
    xs, dys = enc(x), enc(dy)
    nb_d = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
    ws = ops.workspace(max(nb_d, 1))
    kernels = {}

    def launches(tag):
        """every launch kind of the trunk; returns {name: tensors}"""
        o = {}
        y = torch.full_like(x, float("nan")); part = torch.full((N, H * W // 128, 2, C), float("nan"), device=dev)
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 0, P(part), 0, st)
        kernels[tag + " fwd+stats"] = _lib.query("acg_last_kernel").decode()
        o["fwd + tile statistics: y"] = y; o["fwd + tile statistics: (mean, M2)"] = part
        y = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 1, None, 0, st)
        o["fwd + ReLU, fp32 out"] = y
        y = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 1, None, 1, st)
        o["fwd + ReLU, pre-split out"] = y
        y = torch.full_like(x, float("nan")); bits = torch.zeros(nwords, device=dev, dtype=torch.int32)
        _lib.call("acg_conv2d_fwd_s16_mask", D, P(xs), P(pk.wf), P(pk.bias), P(y), P(bits), st)
        kernels[tag + " fwd+mask"] = _lib.query("acg_last_kernel").decode()
        o["fwd + ReLU, pre-split out + sign bitmask: y"] = y; o["fwd + ReLU, pre-split out + sign bitmask: bits"] = bits
        dx = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, None, None, None, 0, st)
        o["dgrad, fp32 out"] = dx
        dx = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(skip), P(m_skip), None, 0, st)
        o["dgrad + masked skip addend"] = dx
        dx = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(skip), None, None, 0, st)
        o["dgrad + skip addend (no mask)"] = dx
        dx = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_bwd_data_s16", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, None, None, P(xs), 1, st)
        o["dgrad * (x > 0), pre-split out"] = dx
        dx = torch.full_like(x, float("nan"))
        _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(xbits), st)
        kernels[tag + " dgrad bits"] = _lib.query("acg_last_kernel").decode()
        o["dgrad masked by the sign bitmask, pre-split out"] = dx
        for nm, msk, add, act in (("dgrad + addend + norm sums (ReLU bitmask)", m_norm, True, ops.ACT_RELU),
                                  ("dgrad + norm sums (no activation)", None, False, ops.ACT_NONE)):
            dx = torch.full_like(x, float("nan")); psum = torch.full((N, H * W // 128, 2, C), float("nan"), device=dev)
            ns = _lib.NormSumsDesc()
            ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta, ns.gstride = P(xn), P(mean), P(rstd), None, None, 0
            ns.sign_mask, ns.act, ns.part = (P(msk) if msk is not None else None), act, P(psum)
            _lib.call("acg_conv2d_bwd_data_s16_sums", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(skip) if add else None,
                      P(m_skip) if add else None, ctypes.byref(ns), st)
            kernels[tag + " " + nm] = _lib.query("acg_last_kernel").decode()
            o[nm + ": dx"] = dx; o[nm + ": sums"] = psum
        torch.cuda.synchronize()
        return o

    use_pp(False); ref = launches("pre")
    use_pp(True); got = launches("pp")
    print("N=%d %dx%d:" % (N, H, W), {k: v for k, v in kernels.items()}, flush=True)
    ok = all("x3_pp" in v for k, v in kernels.items() if k.startswith("pp ")) and all("x3_pre" in v for k, v in kernels.items() if k.startswith("pre "))
    if not ok:
        print("KERNEL SELECTION WRONG")
    for k in ref:
        ok &= same(k, got[k], ref[k])
    if not time_it:
        return ok
    flops = 2.0 * N * H * W * C * C * 9
    y = torch.empty_like(x); part = torch.empty((N, H * W // 128, 2, C), device=dev); dx = torch.empty_like(x)
    bits = torch.zeros(nwords, device=dev, dtype=torch.int32); psum = torch.empty_like(part)
    ns = _lib.NormSumsDesc()
    ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta, ns.gstride = P(xn), P(mean), P(rstd), None, None, 0
    ns.sign_mask, ns.act, ns.part = P(m_norm), ops.ACT_RELU, P(psum)
    kinds = (
        ("fwd+stats", lambda: _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 0, P(part), 0, st)),
        ("fwd+mask", lambda: _lib.call("acg_conv2d_fwd_s16_mask", D, P(xs), P(pk.wf), P(pk.bias), P(y), P(bits), st)),
        ("dgrad+add+sums", lambda: _lib.call("acg_conv2d_bwd_data_s16_sums", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(skip), P(m_skip), ctypes.byref(ns), st)),
        ("dgrad bits", lambda: _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dx), P(ws), nb_d, P(xbits), st)),
    )
    for rnd in range(3):
        line = "round %d:" % rnd
        for nm, f in kinds:
            for on in (False, True):
                use_pp(on)
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                line += "  %s %s %.4f ms (%.0f TF)" % (nm, "pp" if on else "pre", ms, flops / ms / 1e9)
        print(line, flush=True)
    use_pp(True)
    if os.environ.get("PP_ABL_LIST"):   # timing-only ablations of the persistent kernel (ACG_PP_ABL bits, conv_x3_pp.hip)
        for abl in os.environ["PP_ABL_LIST"].split(","):
            os.environ["ACG_PP_ABL"] = abl
            line = "abl %3s:" % abl
            for nm, f in kinds:
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                line += "  %s %.4f ms" % (nm, e0.elapsed_time(e1) / iters)
                if os.environ.get("ACG_STAMPS") and hasattr(_lib.load(), "acg_debug_pp_hist"):
                    import numpy as np
                    hb = (ctypes.c_ulonglong * (256 * 36))()
                    assert _lib.load().acg_debug_pp_hist(hb, 256 * 36) == 0
                    hh = np.frombuffer(hb, dtype=np.uint64).reshape(256, 36).astype(np.float64).mean(0)
                    print("abl %s %s cycles per stage: %s" % (abl, nm, " ".join("%d" % (v / 16) for v in hh)), flush=True)
            print(line, flush=True)
        os.environ.pop("ACG_PP_ABL", None)
    lib = _lib.load()
    if os.environ.get("ACG_STAMPS") and hasattr(lib, "acg_debug_pp_stamps"):   # -DACG_STAMP build: barrier wait / rest per role
        import numpy as np
        for nm, f in kinds:
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (256 * 14 * 4))()
            assert lib.acg_debug_pp_stamps(buf, 256 * 14 * 4) == 0
            raw = np.frombuffer(buf, dtype=np.uint64).reshape(256, 14, 4).astype(np.float64)
            wait, work = raw[..., 0], raw[..., 1]
            tot = wait + work
            sh = 100 * wait / np.maximum(tot, 1)
            if hasattr(lib, "acg_debug_pp_hist"):
                hb = (ctypes.c_ulonglong * (256 * 36))()
                assert lib.acg_debug_pp_hist(hb, 256 * 36) == 0
                hh = np.frombuffer(hb, dtype=np.uint64).reshape(256, 36).astype(np.float64).mean(0)
                print("%s cycles per stage of a tile (barrier to barrier, mean over tiles x 16): %s" % (nm, " ".join("%d" % (v / 16) for v in hh)), flush=True)
            print("%s stamps: cycles per wave %.0f; share at the barrier: MFMA waves %.1f %% (by wave %s), drain waves %.1f %% (%s), A waves %.1f %%"
                  % (nm, tot[:, :8].mean(), sh[:, :8].mean(), np.round(sh[:, :8].mean(0), 1), sh[:, 8:12].mean(), np.round(sh[:, 8:12].mean(0), 1), sh[:, 12:].mean()), flush=True)
    return ok


def wide_stays_on_pre():
    """a 256 -> 128 channel layer (S = 72 stages per tile): the persistent kernel's drain is scheduled on the 36 stages of a
    128-channel tile, so acg_igemm_x3_pp_ok must leave this layer to igemm_conv_x3_pre whatever ACG_PP says"""
    dev = torch.device("cuda")
    st = ops._stream()
    N, H, W, Ci, Co = 1, 64, 128, 256, 128
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, 1, Ci, Co)
    D = ctypes.byref(d)
    if not _lib.query("acg_conv2d_s16_supported", D):
        print("256 input channels: not a pre-split layer (nothing to check)")
        return True
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((N, H, W, Ci), device=dev, generator=g).clamp_min(0)
    w = torch.randn((Co, Ci, 3, 3), device=dev, generator=g) * 0.05
    b = torch.randn(Co, device=dev, generator=g)
    pk = ops.PackedConv(w, b, Ci, Co)
    xs = enc(x)
    ys, names = [], []
    for on in (False, True):
        use_pp(on)
        y = torch.full((N, H, W, Co), float("nan"), device=dev)
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 1, None, 0, st)
        names.append(_lib.query("acg_last_kernel").decode())
        torch.cuda.synchronize()
        ys.append(y)
    ok = all("x3_pre" in n for n in names) and same("256 -> 128 forward, ACG_PP on vs off", ys[1], ys[0])
    print("256 input channels: %s %s" % ("stays on igemm_conv_x3_pre" if ok else "WRONG KERNEL", names), flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    ops.set_precision("bf16x3")
    ok = run(1, 64, 128, 128, False, 0)     # the smallest map of the un-padded data gradient: 64 tiles, one per workgroup
    ok &= run(3, 64, 256, 128, False, 0)    # two tiles per grid row; 384 tiles over 256 workgroups (uneven)
    ok &= run(5, 128, 128, 128, False, 0)   # 640 tiles: 2-3 per workgroup
    ok &= wide_stays_on_pre()
    if a.time:
        ok &= run(a.batch, 128, 128, 128, True, a.iters)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
