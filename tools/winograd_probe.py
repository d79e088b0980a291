#!/usr/bin/env python3
"""Numerics probe (CPU, NumPy; no GPU minutes): would Winograd F(2x2, 3x3) hold the 1e-3 parity bar for the residual-trunk
convolutions in the bf16x3 arithmetic?

The trunk (modules.py:139-235: 3x3 stride-1 reflection-padded C -> C convolutions) is matrix-pipe-bound at three bf16 MFMAs
per product; F(2x2, 3x3) needs 16 products per 2x2 output tile and channel pair instead of 36 (2.25x fewer).  What it costs
in accuracy with 16-bit operand mantissas is what this tool measures, by emulating the arithmetic a kernel would run:

  direct-x3    x = hi + lo (bf16, RNE), w = hi + lo; products hi*hi + hi*lo + lo*hi, fp32 accumulate      (today's kernels)
  wino-x3      V = B^T d B formed in fp32 from hi + lo, then split hi / lo; U = G g G^T in fp32, split hi / lo (once per
               optimiser step, on the host side of the kernel); M = sum_ci (Vh*Uh + Vh*Ul + Vl*Uh), fp32 accumulate;
               Y = A^T M A in fp32
  wino-f32     the same transforms with exact fp32 products (what the algorithm alone costs)

Part 1: one convolution against fp64 (rms / max error) on unit-normal, post-ReLU and 'init'-scale operands.
Part 2: the oracle's training step (oracle/step.py) with the trunk convolutions' forward and data gradient replaced by each
emulation, on the reference-generated step fixtures (tests/golden: 'init' = step_aug_cfg1_full / step_aug_small_s64_init,
'rich' = step_aug_small_s64): 13 losses, fake_A / fake_B, rec_A / rec_B against the reference's values.
Gate: losses / fake_* <= 1e-3 and rec_* no worse than direct-x3 on both flavours.

    python tools/winograd_probe.py [conv|step|all]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32 = np.float32
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], F32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], F32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], F32)


def bf16(x):
    """round-to-nearest-even to bf16, returned as float32"""
    u = np.ascontiguousarray(x, F32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(F32)


def split(x):
    x = np.ascontiguousarray(x, F32)
    hi = bf16(x)
    return hi, bf16(x - hi)


def mm3(a, b, x3):
    """[.., M, K] @ [.., K, N] in fp32; x3: both operands as hi + lo, the lo*lo term dropped"""
    if not x3:
        return np.matmul(a, b)
    ah, al = split(a)
    bh, bl = split(b)
    return np.matmul(al, bh) + np.matmul(ah, bl) + np.matmul(ah, bh)


def conv_direct(xp, w, x3=True):
    """valid 3x3 correlation of the padded input xp (N, C, H+2, W+2) with w (Co, Ci, 3, 3), fp32 accumulate"""
    N, C, Hp, Wp = xp.shape
    H, W = Hp - 2, Wp - 2
    Co = w.shape[0]
    xp = np.ascontiguousarray(xp, F32)
    if x3:   # the kernels see x as hi + lo: the stored tensor is already that sum
        h, l = split(xp)
    cols = np.empty((N, H, W, 9, C), F32)
    colsl = np.empty_like(cols) if x3 else None
    for t in range(9):
        dy, dx = divmod(t, 3)
        if x3:
            cols[:, :, :, t] = h[:, :, dy:dy + H, dx:dx + W].transpose(0, 2, 3, 1)
            colsl[:, :, :, t] = l[:, :, dy:dy + H, dx:dx + W].transpose(0, 2, 3, 1)
        else:
            cols[:, :, :, t] = xp[:, :, dy:dy + H, dx:dx + W].transpose(0, 2, 3, 1)
    wm = np.ascontiguousarray(w.astype(F32).transpose(2, 3, 1, 0)).reshape(9 * C, Co)
    A = cols.reshape(N * H * W, 9 * C)
    if x3:
        Al = colsl.reshape(N * H * W, 9 * C)
        wh, wl = split(wm)
        y = Al @ wh + A @ wl + A @ wh
    else:
        y = A @ wm
    return y.reshape(N, H, W, Co).transpose(0, 3, 1, 2)


def conv_wino(xp, w, x3=True):
    """F(2x2, 3x3) on the padded input (H, W even)"""
    N, C, Hp, Wp = xp.shape
    H, W = Hp - 2, Wp - 2
    assert H % 2 == 0 and W % 2 == 0
    Co = w.shape[0]
    xp = np.ascontiguousarray(xp, F32)
    if x3:   # the stored activation is hi + lo
        h, l = split(xp)
        xp = h + l
    th, tw = H // 2, W // 2
    d = np.empty((N, C, th, tw, 4, 4), F32)
    for i in range(4):
        for j in range(4):
            d[..., i, j] = xp[:, :, i:i + H:2, j:j + W:2]
    V = np.einsum("ai,ncyxij,bj->abnyxc", BT, d, BT, optimize=True).astype(F32)          # [4][4][n][ty][tx][c]
    U = np.einsum("ai,ocij,bj->abco", G, w.astype(F32), G, optimize=True).astype(F32)     # [4][4][ci][co]
    M = mm3(V.reshape(16, N * th * tw, C), U.reshape(16, C, Co), x3).reshape(4, 4, N, th, tw, Co)
    Y = np.einsum("ia,abnyxo,jb->noyixj", AT, M, AT, optimize=True).astype(F32)            # [n][co][ty][2][tx][2]
    return Y.reshape(N, Co, H, W)


def conv64(xp, w):
    xp, w = xp.astype(np.float64), w.astype(np.float64)
    N, C, Hp, Wp = xp.shape
    H, W = Hp - 2, Wp - 2
    y = np.zeros((N, w.shape[0], H, W))
    for t in range(9):
        dy, dx = divmod(t, 3)
        y += np.einsum("nchw,oc->nohw", xp[:, :, dy:dy + H, dx:dx + W], w[:, :, dy, dx], optimize=True)
    return y


def rms(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))


def mx(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def conv_part():
    rs = np.random.RandomState(0)
    print("== one 128 -> 128 3x3 convolution on a 2 x 32 x 32 map against fp64: rms / max relative error ==")
    for label, gen in (("unit-normal x, w ~ N(0, 0.05)", lambda: (rs.normal(0, 1, (2, 128, 34, 34)), rs.normal(0, 0.05, (128, 128, 3, 3)))),
                       ("post-ReLU x (half zeros), w ~ N(0, 0.02) ('init')", lambda: (np.maximum(rs.normal(0, 1, (2, 128, 34, 34)), 0), rs.normal(0, 0.02, (128, 128, 3, 3)))),
                       ("x with a large mean (mean 3, std 1), w ~ N(0, 0.03)", lambda: (rs.normal(3, 1, (2, 128, 34, 34)), rs.normal(0, 0.03, (128, 128, 3, 3))))):
        x, w = gen()
        ref = conv64(x, w)
        for name, fn in (("direct-f32", lambda: conv_direct(x, w, False)), ("direct-x3", lambda: conv_direct(x, w, True)),
                         ("wino-f32", lambda: conv_wino(x, w, False)), ("wino-x3", lambda: conv_wino(x, w, True))):
            y = fn()
            print("  %-52s %-10s rms %.2e  max %.2e" % (label, name, rms(y, ref), mx(y, ref)))


def step_part():
    from golden_util import load, rel_err
    from oracle import ops as oops, recipe, step

    raw_fwd, raw_dgrad = oops._conv_fwd_raw, oops._conv_dgrad_raw
    mode = {"m": None}

    def is_trunk(w, s):
        return mode["m"] is not None and s == 1 and w.shape[2] == 3 and w.shape[0] == w.shape[1] and w.shape[0] >= 16

    def fwd(xp, w, b, s):
        if not is_trunk(w, s) or (xp.shape[2] - 2) % 2 or (xp.shape[3] - 2) % 2:
            return raw_fwd(xp, w, b, s)
        kind, x3 = mode["m"]
        y = (conv_wino if kind == "wino" else conv_direct)(xp, w, x3)
        if b is not None:
            y = y + b.astype(F32)[None, :, None, None]
        return y.astype(xp.dtype)

    def dgrad(dy, w, s, Hp, Wp):
        # gradient w.r.t. the padded input = full correlation of dy with the flipped, transposed kernel: a valid 3x3
        # convolution of dy zero-padded by 2 (the layer the kernels run on the padded / un-padded grid)
        if not is_trunk(w, s) or Hp % 2 or Wp % 2:
            return raw_dgrad(dy, w, s, Hp, Wp)
        kind, x3 = mode["m"]
        wt = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
        dyp = np.pad(dy, ((0, 0), (0, 0), (2, 2), (2, 2)))
        return (conv_wino if kind == "wino" else conv_direct)(dyp, wt, x3).astype(dy.dtype)

    oops._conv_fwd_raw, oops._conv_dgrad_raw = fwd, dgrad
    try:
        for name in ("step_aug_cfg1_full", "step_aug_small_s64_init", "step_aug_small_s64"):
            arr, meta = load(name)
            print("== %s (flavour %s, %d steps; trunk %d channels) ==" % (name, meta["flavour"], meta["steps"], 4 * meta["opt"].get("ngf", 32)))
            for label, m_ in (("oracle fp32 (exact products)", None), ("direct-x3 (today's arithmetic)", ("direct", True)),
                              ("wino-f32", ("wino", False)), ("wino-x3", ("wino", True))):
                mode["m"] = m_
                opt = step.Opt(**meta["opt"])
                m = (step.AugStep if meta["aug"] else step.StochStep)(opt, dtype=np.float32)
                m.load({n: recipe.values_for(net.shapes, n, meta["seed"], meta["flavour"]) for n, net in m.nets().items()})
                t0 = time.time()
                for st in range(meta["steps"]):
                    A, B, z = arr["s%d/real_A" % st], arr["s%d/real_B" % st], arr["s%d/prior_z_B" % st]
                    losses, vis, gn = m.train_instance(A, B, z)
                    got, ref = np.array(list(losses.values())), arr["s%d/losses" % st]
                    le = float(np.max(np.abs(got - ref) / (np.abs(ref) + 2e-6 / 1e-3)))
                    ge = float(np.max(np.abs(np.array(list(gn.values())) - arr["s%d/gnorms" % st]) / (np.abs(arr["s%d/gnorms" % st]) + 1e-6)))
                    ims = {k: rel_err(vis[k], arr["s%d/%s" % (st, k)]) for k in ("fake_A", "fake_B", "rec_A", "rec_B")}
                    print("  %-32s step %d: losses %.2e  gnorms %.2e  %s   (%.0f s)" % (
                        label, st, le, ge, "  ".join("%s %.2e" % kv for kv in ims.items()), time.time() - t0))
    finally:
        oops._conv_fwd_raw, oops._conv_dgrad_raw = raw_fwd, raw_dgrad


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("conv", "all"):
        conv_part()
    if what in ("step", "all"):
        step_part()
