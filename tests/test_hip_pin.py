"""GPU parity: the kernels that engage only at the bench geometry, pinned to the fp64 oracle DIRECTLY (not through another
HIP kernel).

Round 3 added kernels that need whole 128-pixel row tiles (W % 128 == 0, H % 32 == 0): the un-padded reflect data gradient
with its summed weight slabs and column-term GEMM (`Geom.unpad`, `dgrad_colfix_kernel`), the norm-backward sums and the ReLU
sign bitmask out of `igemm_conv_x3_pre`'s epilogues, the pre-split kernel-row weight gradient, the four sub-pixel phases of
a stride-2 data gradient in one tile (`igemm_conv_ph4`) and the kernel-row weight gradient of the stride-2 pair (`wgrad_x3_krowg<NT=3,IS=2,BCI=64>`).  The
golden fixtures stop at 64 x 64 images, so here the oracle itself (oracle/ops.py, fp64, plain C loops) runs the layers at
the geometry bench.py times: a two-block residual trunk on a 128 x 128 x 128 map (N = 1), and single stride-2 /
ConvTranspose layers at W = 256.  Every test records which C-ABI entry points and which kernels ran and asserts the ones it
is about.

Reference semantics: /root/reference/augmented_cyclegan/modules.py:139-235 (ResnetBlock / CINResnetBlock),
networks.py:168, 178-179 (stride-2 downsample, ConvTranspose2d).
Bars: forward 1e-3 in bf16x3 (the north star; measured 1.2e-5), 1e-4 in f32 (measured 1e-6).  Gradients are compared
NORM-WISE in both arithmetics: the trunk holds five ReLU layers of 2 M units each, and a pre-activation within the operand
rounding of zero (2^-17 relative in bf16x3, 2^-24 in f32 against the fp64 oracle) lands on the other side of its ReLU and
changes the gradient discretely in that unit's receptive field — a flipped fraction p moves the gradient by ~sqrt(p)
norm-wise (p ~ 1e-5 -> 3e-3) whatever the kernels do, while an indexing or fusion error shows at the 1e-1 level.  bf16x3:
1e-2 on every gradient (measured 3.0e-3 .. 5.2e-3: the same level on the input, the stem and the first block, i.e. set by
the ReLU layers behind them, not by a kernel); f32: 1e-3 / 5e-4 (measured 1e-6 .. 2.3e-4: one flipped unit among 10 M shows
as 3e-3 max-abs on the input gradient).  Because ReLU flips blur the end-to-end gradients, the LINEAR maps of the backward
kernels are pinned separately and tightly below (test_presplit_kernels_at_bench_geometry_match_the_oracle: each C-ABI entry
point against the oracle's adjoints with the masks given, 2e-5 / 1e-4).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ops as oops  # noqa: E402
from oracle.tape import T, backward, leaf  # noqa: E402


from hip_util import Spy  # noqa: E402


# ---------------------------------------------------------------------------------------------------------------------
# two-block residual trunk at the bench geometry
# ---------------------------------------------------------------------------------------------------------------------
C, S, NL, CIN0 = 128, 128, 8, 16
_ORACLE = {}


def _trunk_values(kind):
    """O(1)-scale parameters (oracle.recipe's 'rich' flavour: every adjoint term is exercised)"""
    rs = np.random.RandomState(41 if kind == "plain" else 43)
    v = {}

    def conv(name, co, ci, k):
        v[name + ".weight"] = rs.normal(0, 1.0 / np.sqrt(ci * k * k), (co, ci, k, k))
        v[name + ".bias"] = rs.normal(0, 0.2, (co,))

    def inorm(name):
        v[name + ".scale"] = rs.normal(1.0, 0.3, (C,))
        v[name + ".shift"] = rs.normal(0.0, 0.2, (C,))

    def cnorm(name):
        for br, mu in (("shift_conv", 0.1), ("scale_conv", 0.8)):
            v["%s.%s.0.weight" % (name, br)] = rs.normal(0, 1.0 / np.sqrt(NL), (C, NL, 1, 1))
            v["%s.%s.0.bias" % (name, br)] = rs.normal(mu, 0.2, (C,))

    conv("stem", C, CIN0, 3)
    (inorm if kind == "plain" else cnorm)("stem_norm")
    for b in range(2):
        conv("b%d.c1" % b, C, C, 3)
        if kind == "cin":
            cnorm("b%d.n1" % b)
        conv("b%d.c2" % b, C, C, 3)
        inorm("b%d.n2" % b)
    x = rs.normal(0, 1, (1, CIN0, S, S))
    z = rs.normal(0, 1, (1, NL, 1, 1))
    r = rs.normal(0, 1, (1, C, S, S))
    return v, x, z, r


def _relu_given(x, m):
    """ReLU with the activation pattern GIVEN (m: bool array): forward x * m, adjoint g * m.  With m taken from the HIP
    forward the oracle differentiates the same piecewise-linear function the kernels did: no unit lands on the other side."""
    return T(np.where(m, x.v, 0).astype(x.v.dtype), (x,), lambda g: (g * m,))


def _oracle_trunk(kind, masks=None):
    """fp64 oracle of stem conv + norm + ReLU, then two (CIN)ResnetBlocks (modules.py:148-188, 199-235); cached per kind.
    masks: the activation patterns of the five ReLU layers (stem, block 0 inner / output, block 1 inner / output) to use
    instead of the oracle's own (not cached)"""
    if masks is None and kind in _ORACLE:
        return _ORACLE[kind]
    it = iter(masks) if masks is not None else None
    pats = []                                  # the oracle's own activation patterns (pre-activation > 0), layer by layer

    def relu(h):
        pats.append(h.v > 0)
        return _relu_given(h, next(it)) if masks is not None else oops.relu(h)
    v, x, z, r = _trunk_values(kind)
    P = {k: leaf(np.asarray(a, np.float64)) for k, a in v.items()}
    X, Z = leaf(np.asarray(x, np.float64)), leaf(np.asarray(z, np.float64))

    def cn(h, name):
        sh = oops.relu(oops.conv2d(Z, P[name + ".shift_conv.0.weight"], P[name + ".shift_conv.0.bias"]))
        sc = oops.relu(oops.conv2d(Z, P[name + ".scale_conv.0.weight"], P[name + ".scale_conv.0.bias"]))
        return oops.cond_instance_norm(h, sc, sh)

    def inn(h, name):
        return oops.instance_norm(h, P[name + ".scale"], P[name + ".shift"])

    h = oops.conv2d(X, P["stem.weight"], P["stem.bias"], pad=1)
    h = relu(inn(h, "stem_norm") if kind == "plain" else cn(h, "stem_norm"))
    for b in range(2):
        o = oops.conv2d(h, P["b%d.c1.weight" % b], P["b%d.c1.bias" % b], pad=1, pad_mode="reflect")
        if kind == "cin":
            o = cn(o, "b%d.n1" % b)
        o = relu(o)
        o = oops.conv2d(o, P["b%d.c2.weight" % b], P["b%d.c2.bias" % b], pad=1, pad_mode="reflect")
        o = inn(o, "b%d.n2" % b)
        h = relu(oops.add(h, o))
    backward(h, seed=np.asarray(r, np.float64))
    out = dict(y=h.v, gx=X.g, gz=Z.g, grads={k: p.g for k, p in P.items()}, acts=pats)
    if masks is None:
        _ORACLE[kind] = out
    return out


class _MaskTap(object):
    """collects, in order of first use, the sign-bitmask tensors (the only int32 tensors the ops hand to the library) while
    active: what the fused forward itself recorded about its ReLU layers (norm.hip layout: bit e % 32 of word e / 32, NHWC)"""

    def __enter__(self):
        from dtgan_amd import ops
        self.ops, self.real, self.seen, self.ids = ops, ops._ptr, [], set()

        def ptr(t):
            if t is not None and t.dtype == torch.int32 and t.data_ptr() not in self.ids:
                self.ids.add(t.data_ptr())
                self.seen.append(t)
            return self.real(t)
        ops._ptr = ptr
        return self

    def __exit__(self, *a):
        self.ops._ptr = self.real

    def patterns(self, shape):
        """bool NCHW arrays of the masks that cover a tensor of `shape` (N, C, H, W)"""
        N, Cn, H, W = shape
        out = []
        for t in self.seen:
            if t.numel() * 32 != N * Cn * H * W:
                continue
            w = t.detach().cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            b = ((w[:, None] >> np.arange(32)) & 1).astype(bool).reshape(N, H, W, Cn)
            out.append(np.ascontiguousarray(np.transpose(b, (0, 3, 1, 2))))
        return out


def _hip_relu_patterns(kind, net, xt, zt):
    """the activation patterns of the five ReLU layers as the HIP forward produced them: the net run piece by piece (stem +
    norm + ReLU, then per block its inner conv (+ CondIN) + ReLU and the block itself), outputs > 0.  Pieces run alone take
    the same kernels' arithmetic (tools/s16_check.py, tools/pp_check.py: fused and unfused outputs agree bit for bit)."""
    from dtgan_amd import modules as M
    from hip_util import n
    mods = list(net._modules.values())
    masks = []
    with torch.no_grad():
        if kind == "plain":
            h = M.Sequential(*mods[:3])(xt)
            blocks = mods[3:]
        else:
            h = M.TwoInputSequential(*mods[:2])(xt, zt)
            blocks = mods[2:]
        masks.append(n(h) > 0)
        for blk in blocks:
            inner = list(blk.conv_block._modules.values())[:3]       # pad, conv (or Merge(conv, CondIN)), ReLU
            a = M.TwoInputSequential(*inner)(h, zt) if kind == "cin" else M.Sequential(*inner)(h)
            masks.append(n(a) > 0)
            h = blk(h, zt) if kind == "cin" else blk(h)
            masks.append(n(h) > 0)
    return masks


def _hip_trunk(kind, v):
    """the same layers as product modules: modules.Sequential / TwoInputSequential over Conv2d, (Cond)InstanceNorm and the
    residual block classes; returns (net, {oracle parameter name: torch parameter})"""
    from dtgan_amd import modules as M
    names = {}
    if kind == "plain":
        norm = M.InstanceNorm2d
        stem, sn = M.Conv2d(CIN0, C, 3, padding=1, bias=True), M.InstanceNorm(C)
        blocks = [M.ResnetBlock(C, "reflect", norm, False, True) for _ in range(2)]
        net = M.Sequential(stem, sn, nn.ReLU(True), *blocks).cuda()
        names.update({"stem.weight": stem.weight, "stem.bias": stem.bias, "stem_norm.scale": sn.scale, "stem_norm.shift": sn.shift})
        for b, blk in enumerate(blocks):
            mods = list(blk.conv_block._modules.values())   # pad, conv, relu, pad, conv, norm
            c1, c2, n2 = mods[1], mods[4], mods[5]
            names.update({"b%d.c1.weight" % b: c1.weight, "b%d.c1.bias" % b: c1.bias, "b%d.c2.weight" % b: c2.weight,
                          "b%d.c2.bias" % b: c2.bias, "b%d.n2.scale" % b: n2.scale, "b%d.n2.shift" % b: n2.shift})
    else:
        stem, sn = M.Conv2d(CIN0, C, 3, padding=1, bias=True), M.CondInstanceNorm(C, NL)
        blocks = [M.CINResnetBlock(C, NL, "reflect", M.CondInstanceNorm, False, True) for _ in range(2)]
        net = M.TwoInputSequential(M.MergeModule(stem, sn), nn.ReLU(True), *blocks).cuda()

        def cnames(prefix, m):
            for br in ("shift_conv", "scale_conv"):
                names["%s.%s.0.weight" % (prefix, br)] = getattr(m, br)[0].weight
                names["%s.%s.0.bias" % (prefix, br)] = getattr(m, br)[0].bias
        names.update({"stem.weight": stem.weight, "stem.bias": stem.bias})
        cnames("stem_norm", sn)
        for b, blk in enumerate(blocks):
            mods = list(blk.conv_block._modules.values())   # pad, Merge(conv, cin), relu, pad, conv, norm
            mg, c2, n2 = mods[1], mods[4], mods[5]
            names.update({"b%d.c1.weight" % b: mg.module1.weight, "b%d.c1.bias" % b: mg.module1.bias,
                          "b%d.c2.weight" % b: c2.weight, "b%d.c2.bias" % b: c2.bias, "b%d.n2.scale" % b: n2.scale,
                          "b%d.n2.shift" % b: n2.shift})
            cnames("b%d.n1" % b, mg.module2)
    assert set(names) == set(v), (set(names) ^ set(v))
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(torch.from_numpy(np.ascontiguousarray(v[k], np.float32)).to(p.device))
    M.mark_dirty(net)
    return net, names


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("kind", ["plain", "cin"])
def test_trunk_at_bench_geometry_matches_the_oracle(kind, prec):
    from dtgan_amd import ops
    from hip_util import precision, t, n, rel, l2rel
    ref = _oracle_trunk(kind)
    v, x, z, r = _trunk_values(kind)
    with precision(prec):
        net, names = _hip_trunk(kind, v)
        xt, zt = t(x, grad=True), t(z, grad=True)
        used0 = ops.NORM_SUMS_USED
        with Spy() as spy:
            with _MaskTap() as tap:
                y = net(xt, zt) if kind == "cin" else net(xt)
                fused_masks = tap.patterns((1, C, S, S))   # (cloned to host before the backward frees them)
            y.backward(t(r))
        used = ops.NORM_SUMS_USED - used0
    x3 = prec == "bf16x3"
    e_y = rel(n(y), ref["y"])
    e_gx = l2rel(n(xt.grad), ref["gx"])
    print("%s/%s: forward %.2e, input gradient norm-wise %.2e (max-abs %.2e)" % (kind, prec, e_y, e_gx, rel(n(xt.grad), ref["gx"])))
    assert e_y < (1e-3 if x3 else 1e-4), e_y
    assert e_gx < (1e-2 if x3 else 1e-3), e_gx
    if kind == "cin":
        e_gz = l2rel(n(zt.grad), ref["gz"])
        assert e_gz < (1e-2 if x3 else 5e-4), e_gz
    gmax = max(float(np.max(np.abs(g))) for g in ref["grads"].values())
    worst = 0.0
    for k, p in names.items():
        g, go = n(p.grad), ref["grads"][k]
        assert g.shape == go.shape, k
        # (a convolution bias in front of a norm has the exact gradient 0: both sides hold summation noise there, allowed
        # for by the absolute term tied to the largest gradient of the net)
        tol = 1e-2 if x3 else 5e-4
        err = np.linalg.norm(g - go) / (np.linalg.norm(go) + 2e-5 * gmax * np.sqrt(go.size) / tol)
        assert err < tol, (k, err)
        worst = max(worst, err)
    print("%s/%s: worst parameter gradient %.2e" % (kind, prec, worst))
    # ---- the same gradients against the oracle differentiated AT THE HIP FORWARD'S ACTIVATION PATTERNS: with no unit on the
    # other side of its ReLU the comparison is between linear maps again, and the bars are the north-star ones
    with precision(prec):
        masks = _hip_relu_patterns(kind, net, xt.detach(), zt.detach())
    # (piece-by-piece patterns can differ from the fused run's in units within rounding of zero — the pieces take other
    # kernels for the norms; where the fused forward stored a layer's sign bitmask itself, that is the pattern: each recorded
    # bitmask replaces the piece-by-piece pattern it agrees with but for such units)
    taken = []
    for i, m in enumerate(masks):
        d = [float(np.mean(f != m)) for f in fused_masks]
        if d and min(d) < 1e-3:
            j = int(np.argmin(d))
            masks[i] = fused_masks[j]
            taken.append("%d<-%d (%.1e)" % (i, j, d[j]))
    print("%s/%s: sign bitmasks recorded by the fused forward: %d; used for layers %s" % (kind, prec, len(fused_masks), taken))
    if x3 and kind == "plain":   # the fused bf16x3 forward of a plain block DOES record bitmasks (acg_conv2d_fwd_s16_mask, the
        # block-output norm): the masked comparison below must be the one on those patterns, never silently the piece-wise one
        assert len(fused_masks) >= 1 and len(taken) >= 1, (len(fused_masks), taken)
    flipped = ["%.1e" % float(np.mean(m != a)) for m, a in zip(masks, ref["acts"])]   # units on the other side, per ReLU layer
    refm = _oracle_trunk(kind, masks)
    tolm = 1e-3 if x3 else 1e-4
    e_gxm = l2rel(n(xt.grad), refm["gx"])
    print("%s/%s: with the HIP activation patterns: input gradient norm-wise %.2e (free-running %.2e) %s" % (kind, prec, e_gxm, e_gx, flipped))
    assert rel(n(y), refm["y"]) < (1e-3 if x3 else 1e-4)
    assert e_gxm < tolm, e_gxm
    if kind == "cin":
        assert l2rel(n(zt.grad), refm["gz"]) < tolm
    gmaxm = max(float(np.max(np.abs(g))) for g in refm["grads"].values())
    for k, p in names.items():
        g, go = n(p.grad), refm["grads"][k]
        err = np.linalg.norm(g - go) / (np.linalg.norm(go) + 2e-5 * gmaxm * np.sqrt(go.size) / tolm)
        assert err < tolm, (k, err)
    ents, kerns = spy.entries(), spy.kernels()
    if x3:
        # the pre-split plan: S16 forward (+ sign bitmask in the plain block), un-padded data gradients that emit the norm
        # sums / take the bitmask, the pre-split kernel-row weight gradient, the column-term kernel's host (dgrad on W % 128)
        assert "acg_conv2d_fwd_s16" in ents and "acg_conv2d_bwd_weight_s16" in ents and "acg_conv2d_bwd_data_s16_sums" in ents, ents
        if kind == "plain":
            assert "acg_conv2d_fwd_s16_mask" in ents and "acg_conv2d_bwd_data_s16_mask" in ents, ents
        # (igemm_conv_x3_pre; its opt-in persistent form igemm_conv_x3_pp has its own test, tests/test_hip_parked.py)
        assert any(k.startswith("igemm_conv_x3_pre<REFLECT=1") for k in kerns), kerns
        assert any("SUMS" in k for k in kerns), kerns
        assert "wgrad_x3_krow_s16" in kerns, kerns
        # the norm in front of the first block, the first block's output norm, and in a CINResnetBlock its conditional norm
        assert used == (2 if kind == "plain" else 4), used
    else:
        assert not any(e.endswith("_s16") or "_s16_" in e for e in ents), ents


def _pack(w, b, Cn):
    from dtgan_amd import ops
    from hip_util import t
    return ops.PackedConv(t(w), t(b), Cn, Cn)


def test_presplit_kernels_at_bench_geometry_match_the_oracle():
    """Every pre-split (S16) entry point of the trunk layer at N = 1, 128 x 128 x 128 through the C ABI, against the oracle's
    convolution and its adjoints in fp64 (oracle/ops.py conv2d: ReflectionPad2d(1) + 3x3, modules.py:205-227) with the masks,
    the skip addend and the norm statistics GIVEN — linear maps, so the bars are the kernel bars of tests/test_hip_ops.py:
    2e-5 (forward, data gradient), 1e-4 (long pixel sums: weight gradient, bias gradient, per-tile norm sums).  Operands are
    what the step feeds these kernels: x = hi + lo of an fp32 tensor (decode(encode(.)) is what the oracle sees)."""
    import ctypes
    from dtgan_amd import ops, _lib
    from hip_util import precision, t, n, rel
    P, st = ops._ptr, None
    N, H, W, Cn = 1, 128, 128, 128
    rs = np.random.RandomState(5)
    w = rs.normal(0, 0.05, (Cn, Cn, 3, 3)); b = rs.normal(0, 0.5, (Cn,))
    x = np.maximum(rs.normal(0, 1, (N, Cn, H, W)), 0)          # a ReLU output
    dy = rs.normal(0, 1e-3, (N, Cn, H, W))
    skip = rs.normal(0, 1e-3, (N, Cn, H, W))
    xn = rs.normal(0.3, 1.2, (N, Cn, H, W))                     # the input of the norm in front of the layer
    skip_bits = rs.rand(N, Cn, H, W) > 0.4                      # sign bitmask of the block output the skip gradient passes
    norm_bits = rs.rand(N, Cn, H, W) > 0.5                      # activation mask of the norm's output
    nhwc = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))

    def bits(m):   # norm.hip layout: bit e % 32 of word e / 32 for NHWC float index e
        f = nhwc(m).reshape(-1, 32).astype(np.int64)
        v = (f << np.arange(32)).sum(1)
        return torch.from_numpy(np.where(v >= 2 ** 31, v - 2 ** 32, v).astype(np.int32)).cuda()

    with precision("bf16x3"):
        st = ops._stream()
        d = ops.conv_desc(N, H, W, Cn, Cn, 3, 1, 1, 1, Cn, Cn)
        D = ctypes.byref(d)
        assert _lib.query("acg_conv2d_s16_supported", D) and _lib.query("acg_conv2d_bwd_data_s16_sums_supported", D)
        pk = _pack(w, b, Cn)

        def enc(a):   # fp32 NHWC -> S16
            y = torch.empty_like(a)
            _lib.call("acg_s16_encode", P(a), P(y), a.numel(), st)
            return y

        def dec(a):
            y = torch.empty_like(a)
            _lib.call("acg_s16_decode", P(a), P(y), a.numel(), st)
            return y
        xs, dys = enc(t(nhwc(x))), enc(t(nhwc(dy)))
        x_hl, dy_hl = nchw(n(dec(xs))).astype(np.float64), nchw(n(dec(dys))).astype(np.float64)   # what the kernels consume
        assert rel(x_hl, x) < 1e-5 and rel(dy_hl, dy) < 1e-5
        # ---- oracle: forward, and the adjoints for the gradient dy
        X, Wt, Bt = leaf(x_hl), leaf(w), leaf(b)
        yo = oops.conv2d(X, Wt, Bt, pad=1, pad_mode="reflect")
        backward(yo, seed=dy_hl)
        # ---- forward: fp32 out + per-tile statistics; conv + ReLU with pre-split output + sign bitmask
        y = torch.empty((N, H, W, Cn), device="cuda")
        part = torch.empty((N, H * W // 128, 2, Cn), device="cuda")
        _lib.call("acg_conv2d_fwd_s16", D, P(xs), P(pk.wf), P(pk.bias), P(y), 0, P(part), 0, st)
        assert _lib.query("acg_last_kernel").decode() == "igemm_conv_x3_pre<REFLECT=1,STATS=1>"
        assert rel(nchw(n(y)), yo.v) < 2e-5, "forward"
        yt = nhwc(yo.v).reshape(N, H * W // 128, 128, Cn)
        assert rel(n(part)[:, :, 0], yt.mean(2)) < 1e-5 and rel(n(part)[:, :, 1], ((yt - yt.mean(2, keepdims=True)) ** 2).sum(2)) < 1e-4, "tile statistics"
        y2 = torch.empty_like(y)
        mk = torch.zeros((y.numel() + 31) // 32, device="cuda", dtype=torch.int32)
        _lib.call("acg_conv2d_fwd_s16_mask", D, P(xs), P(pk.wf), P(pk.bias), P(y2), P(mk), st)
        assert rel(nchw(n(dec(y2))), np.maximum(yo.v, 0)) < 2e-5, "conv + ReLU, pre-split output"
        # (the sign of an output within rounding of zero is the kernel's own: compare the bits where |y| is not tiny)
        got = ((n(mk).astype(np.int64)[:, None] >> np.arange(32)) & 1).reshape(N, H, W, Cn).astype(bool)
        sure = np.abs(nhwc(yo.v)) > 1e-4 * np.abs(yo.v).max()
        assert np.array_equal(got[sure], (nhwc(yo.v) > 0)[sure]), "sign bitmask of the conv + ReLU output"
        # ---- data gradient (un-padded grid + column-term kernel): fp32 out with the masked skip addend and the norm sums
        nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
        ws = ops.workspace(max(nbw, _lib.query("acg_conv2d_bwd_weight_workspace_bytes", D), 1))
        dx = torch.empty_like(y)
        mean = rs.normal(0.3, 0.1, (N, Cn)); rstd = rs.uniform(0.5, 1.5, (N, Cn))
        psum = torch.empty((N, H * W // 128, 2, Cn), device="cuda")
        ns = _lib.NormSumsDesc()
        xn_t, mean_t, rstd_t, nb_t, sb_t, skip_t = t(nhwc(xn)), t(mean.reshape(-1)), t(rstd.reshape(-1)), bits(norm_bits), bits(skip_bits), t(nhwc(skip))
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta, ns.gstride = P(xn_t), P(mean_t), P(rstd_t), None, None, 0
        ns.sign_mask, ns.act, ns.part = P(nb_t), ops.ACT_RELU, P(psum)
        _lib.call("acg_conv2d_bwd_data_s16_sums", D, P(dys), P(pk.wb), P(dx), P(ws), nbw, P(skip_t), P(sb_t), ctypes.byref(ns), st)
        assert _lib.query("acg_last_kernel").decode() == "igemm_conv_x3_pre<REFLECT=0,STATS=0,SUMS=1>"
        dx_o = X.g + skip * skip_bits
        assert rel(nchw(n(dx)), dx_o) < 2e-5, "data gradient + masked skip addend"
        gy = nhwc(dx_o * norm_bits).reshape(N, H * W // 128, 128, Cn)
        xh = nhwc((xn.astype(np.float32).astype(np.float64) - mean[:, :, None, None]) * rstd[:, :, None, None]).reshape(N, H * W // 128, 128, Cn)
        assert rel(n(psum)[:, :, 0], gy.sum(2)) < 1e-4 and rel(n(psum)[:, :, 1], (gy * xh).sum(2)) < 1e-4, "norm backward sums"
        # ---- data gradient with pre-split output masked by the sign bitmask of the layer's own input (conv + ReLU in front)
        dxs = torch.empty_like(y)
        _lib.call("acg_conv2d_bwd_data_s16_mask", D, P(dys), P(pk.wb), P(dxs), P(ws), nbw, P(bits(x > 0)), st)
        assert rel(nchw(n(dec(dxs))), X.g * (x > 0)) < 2e-5, "data gradient masked by the ReLU bitmask, pre-split output"
        # ---- weight gradient (kernel rows, pre-split operands)
        dw, db = torch.empty((Cn, Cn, 3, 3), device="cuda"), torch.empty(Cn, device="cuda")
        _lib.call("acg_conv2d_bwd_weight_s16", D, P(xs), P(dys), P(dw), P(db), Cn, Cn, P(ws),
                  _lib.query("acg_conv2d_bwd_weight_workspace_bytes", D), 0, st)
        assert _lib.query("acg_last_kernel").decode() == "wgrad_x3_krow_s16"
        assert rel(n(dw), Wt.g) < 1e-4 and rel(n(db), Bt.g) < 1e-4, "weight / bias gradient"


@pytest.mark.parametrize("case", ["relu_shared", "none_shared", "relu_per_sample"])
def test_row_pipeline_data_gradient_sums_match_the_oracle(case):
    """acg_conv2d_bwd_data_sums (conv_rows_x3: the fp32-operand twin of the trunk's ..._s16_sums) through the C ABI at N = 2,
    64 -> 32 channels, 21 x 256 (21 rows: two row chunks per (image, band), an odd last chunk; two 128-pixel bands): the data
    gradient against the oracle's convolution adjoint (2e-5) and the sums of the norm in front of the layer — S1 = sum gy,
    S2 = sum gy * xhat with gy = dx * [norm output > 0] (modules.py:83-97 behind networks.py:183) — against fp64 (1e-4), with
    `part` pre-filled with NaN (every chunk entry has to be written), ReLU and no activation, shared (gstride 0) and
    per-sample (gstride 64) affine parameters."""
    import ctypes
    from dtgan_amd import ops, _lib
    from hip_util import precision, t, n, rel
    P = ops._ptr
    N, H, W, Ci, Co = 2, 21, 256, 64, 32
    rs = np.random.RandomState(11)
    w = rs.normal(0, 0.06, (Co, Ci, 3, 3))
    dy = rs.normal(0, 1e-2, (N, Co, H, W))
    xn = rs.normal(0.2, 1.1, (N, Ci, H, W))                       # the input of the norm whose output this layer reads
    mean = rs.normal(0.2, 0.1, (N, Ci)); rstd = rs.uniform(0.6, 1.4, (N, Ci))
    per_sample = case == "relu_per_sample"
    act = ops.ACT_NONE if case == "none_shared" else ops.ACT_RELU
    gamma = rs.normal(1.0, 0.4, (N if per_sample else 1, Ci)); beta = rs.normal(0.0, 0.5, (N if per_sample else 1, Ci))
    nhwc = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))
    with precision("bf16x3"):
        st = ops._stream()
        d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, 0, Ci, Co)
        D = ctypes.byref(d)
        assert _lib.query("acg_conv2d_bwd_data_sums_supported", D)
        pk = ops.PackedConv(t(w), t(np.zeros(Co)), Ci, Co)
        # oracle: adjoint of the zero-padded 3x3 convolution for the gradient dy
        X, Wt = leaf(np.zeros((N, Ci, H, W))), leaf(w)
        backward(oops.conv2d(X, Wt, None, pad=1), seed=dy.astype(np.float32).astype(np.float64))
        xh = (xn.astype(np.float32).astype(np.float64) - mean[:, :, None, None]) * rstd[:, :, None, None]
        g_b = np.broadcast_to(gamma, (N, Ci))[:, :, None, None]; b_b = np.broadcast_to(beta, (N, Ci))[:, :, None, None]
        live = (xh * g_b + b_b > 0) if act == ops.ACT_RELU else np.ones_like(xh, dtype=bool)
        # (units whose norm output is within rounding of zero may fall on either side in fp32: leave them out of the sums' bar)
        edge = (np.abs(xh * g_b + b_b) < 1e-5) if act == ops.ACT_RELU else np.zeros_like(live)
        gy = X.g * live
        nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
        ws = ops.workspace(max(nbw, 1))
        dx = torch.full((N, H, W, Ci), float("nan"), device="cuda")
        nch = H * W // 128
        psum = torch.full((N, nch, 2, Ci), float("nan"), device="cuda")
        ns = _lib.NormSumsDesc()
        xn_t, mean_t, rstd_t, g_t, b_t = t(nhwc(xn)), t(mean.reshape(-1)), t(rstd.reshape(-1)), t(gamma.reshape(-1)), t(beta.reshape(-1))
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta = P(xn_t), P(mean_t), P(rstd_t), P(g_t), P(b_t)
        ns.gstride, ns.sign_mask, ns.act, ns.part = (Ci if per_sample else 0), None, act, P(psum)
        _lib.call("acg_conv2d_bwd_data_sums", D, P(t(nhwc(dy))), P(pk.wb), P(dx), P(ws), nbw, ctypes.byref(ns), st)
        assert _lib.query("acg_last_kernel").decode().startswith("conv_rows_x3<32,64>")
        assert rel(nchw(n(dx)), X.g) < 2e-5, "data gradient"
        got = n(psum)
        assert np.isfinite(got).all(), "a chunk entry of the sums was not written"
        s1, s2 = got[:, :, 0].sum(1), got[:, :, 1].sum(1)          # acg_norm_bwd_partials adds the chunk entries up
        assert edge.mean() < 1e-4
        gy_k = nchw(n(dx)).astype(np.float64) * live              # the kernel's own dx under the oracle's mask: isolates the sums
        assert rel(s1, gy_k.sum((2, 3))) < 1e-4 and rel(s2, (gy_k * xh).sum((2, 3))) < 1e-4, "norm backward sums (kernel dx)"
        assert rel(s1, gy.sum((2, 3))) < 2e-4 and rel(s2, (gy * xh).sum((2, 3))) < 2e-4, "norm backward sums (oracle dx)"


@pytest.mark.parametrize("case", ["relu_shared", "relu_per_sample", "none_shared"])
def test_stride2_data_gradient_sums_match_the_oracle(case):
    """acg_conv2d_bwd_data_sums on the stride-2 3x3 layer (igemm_conv_ph4<SUMS>, x staged through LDS-DMA, round 6: the data gradient of networks.py:168 is
    the gradient w.r.t. the output of the norm at networks.py:165-166) through the C ABI at N = 2, 64 -> 128 channels, 8 x 256
    -> 4 x 128: the data gradient against the oracle's convolution adjoint (2e-5), and S1 = sum gy, S2 = sum gy * xhat of that
    norm against fp64 (1e-4) with `part` pre-filled with NaN (every chunk entry has to be written — the tile's sums sit in its
    first chunk, the other three hold zeros), ReLU and no activation, shared and per-sample affine parameters."""
    import ctypes
    from dtgan_amd import ops, _lib
    from hip_util import precision, t, n, rel
    P = ops._ptr
    N, H, W, Ci, Co = 2, 8, 256, 64, 128
    rs = np.random.RandomState(13)
    w = rs.normal(0, 0.05, (Co, Ci, 3, 3))
    dy = rs.normal(0, 1e-2, (N, Co, H // 2, W // 2))
    xn = rs.normal(0.2, 1.1, (N, Ci, H, W))
    mean = rs.normal(0.2, 0.1, (N, Ci)); rstd = rs.uniform(0.6, 1.4, (N, Ci))
    per_sample = case == "relu_per_sample"
    act = ops.ACT_NONE if case == "none_shared" else ops.ACT_RELU
    gamma = rs.normal(1.0, 0.4, (N if per_sample else 1, Ci)); beta = rs.normal(0.0, 0.5, (N if per_sample else 1, Ci))
    nhwc = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))
    with precision("bf16x3"):
        st = ops._stream()
        d = ops.conv_desc(N, H, W, Ci, Co, 3, 2, 1, 0, Ci, Co)
        D = ctypes.byref(d)
        assert _lib.query("acg_conv2d_bwd_data_sums_supported", D)
        pk = ops.PackedConv(t(w), t(np.zeros(Co)), Ci, Co)
        X, Wt = leaf(np.zeros((N, Ci, H, W))), leaf(w)
        backward(oops.conv2d(X, Wt, None, stride=2, pad=1), seed=dy.astype(np.float32).astype(np.float64))
        xh = (xn.astype(np.float32).astype(np.float64) - mean[:, :, None, None]) * rstd[:, :, None, None]
        g_b = np.broadcast_to(gamma, (N, Ci))[:, :, None, None]; b_b = np.broadcast_to(beta, (N, Ci))[:, :, None, None]
        live = (xh * g_b + b_b > 0) if act == ops.ACT_RELU else np.ones_like(xh, dtype=bool)
        edge = (np.abs(xh * g_b + b_b) < 1e-5) if act == ops.ACT_RELU else np.zeros_like(live)
        gy = X.g * live
        nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
        ws = ops.workspace(max(nbw, 1))
        dx = torch.full((N, H, W, Ci), float("nan"), device="cuda")
        psum = torch.full((N, H * W // 128, 2, Ci), float("nan"), device="cuda")
        ns = _lib.NormSumsDesc()
        xn_t, mean_t, rstd_t, g_t, b_t = t(nhwc(xn)), t(mean.reshape(-1)), t(rstd.reshape(-1)), t(gamma.reshape(-1)), t(beta.reshape(-1))
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta = P(xn_t), P(mean_t), P(rstd_t), P(g_t), P(b_t)
        ns.gstride, ns.sign_mask, ns.act, ns.part = (Ci if per_sample else 0), None, act, P(psum)
        _lib.call("acg_conv2d_bwd_data_sums", D, P(t(nhwc(dy))), P(pk.wb), P(dx), P(ws), nbw, ctypes.byref(ns), st)
        assert _lib.query("acg_last_kernel").decode().startswith("igemm_conv_ph4<128,64,SUMS=1>")
        assert rel(nchw(n(dx)), X.g) < 2e-5, "data gradient"
        dx0 = torch.full_like(dx, float("nan"))                  # the plain entry point: the same dx, bit for bit
        _lib.call("acg_conv2d_bwd_data", D, P(t(nhwc(dy))), P(pk.wb), P(dx0), P(ws), nbw, st)
        assert _lib.query("acg_last_kernel").decode().startswith("igemm_conv_ph4<128,64> ")
        assert torch.equal(dx0, dx)
        got = n(psum)
        assert np.isfinite(got).all(), "a chunk entry of the sums was not written"
        s1, s2 = got[:, :, 0].sum(1), got[:, :, 1].sum(1)
        assert edge.mean() < 1e-4
        gy_k = nchw(n(dx)).astype(np.float64) * live
        assert rel(s1, gy_k.sum((2, 3))) < 1e-4 and rel(s2, (gy_k * xh).sum((2, 3))) < 1e-4, "norm backward sums (kernel dx)"
        assert rel(s1, gy.sum((2, 3))) < 2e-4 and rel(s2, (gy * xh).sum((2, 3))) < 2e-4, "norm backward sums (oracle dx)"


@pytest.mark.parametrize("case", ["relu_shared", "relu_per_sample", "none_shared"])
@pytest.mark.parametrize("layer", ["a2_generic_tile", "head_thinrow"])
def test_tile_data_gradient_sums_match_the_oracle(layer, case):
    """acg_conv2d_bwd_data_sums on the two remaining producers of round 6 — the generic 128-pixel tile (data gradient of the
    3x3 32 -> 64 layer, networks.py:164, the gradient w.r.t. the stem's norm output) and conv_thinrow_x3 (data gradient of the
    7x7 head, networks.py:187-188, w.r.t. the last norm's output) — through the C ABI at N = 2: dx against the oracle's
    convolution adjoint (2e-5) and bit-equal to the plain entry point's, S1 / S2 against fp64 (1e-4) with `part` pre-filled
    with NaN, ReLU and no activation, shared and per-sample affine parameters."""
    import ctypes
    from dtgan_amd import ops, _lib
    from hip_util import precision, t, n, rel
    P = ops._ptr
    if layer == "a2_generic_tile":
        N, H, W, Ci, Co, K, pad, Cis, Cos, kern = 2, 12, 128, 32, 64, 3, 1, 32, 64, "igemm_conv_bf16<128,32,"
    else:   # the head: 32 -> 3 channels, the image side stored C4
        N, H, W, Ci, Co, K, pad, Cis, Cos, kern = 2, 16, 48, 32, 3, 7, 3, 32, 4, "conv_thinrow_x3<REFLECT=0,SUMS=1>"
    rs = np.random.RandomState(17)
    w = rs.normal(0, 0.05, (Co, Ci, K, K))
    dy = rs.normal(0, 1e-2, (N, Co, H, W))
    xn = rs.normal(0.2, 1.1, (N, Ci, H, W))
    mean = rs.normal(0.2, 0.1, (N, Ci)); rstd = rs.uniform(0.6, 1.4, (N, Ci))
    per_sample = case == "relu_per_sample"
    act = ops.ACT_NONE if case == "none_shared" else ops.ACT_RELU
    gamma = rs.normal(1.0, 0.4, (N if per_sample else 1, Ci)); beta = rs.normal(0.0, 0.5, (N if per_sample else 1, Ci))
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))

    def nhwc(a, Cp):
        out = np.zeros((a.shape[0], a.shape[2], a.shape[3], Cp), np.float32)
        out[..., :a.shape[1]] = np.transpose(a, (0, 2, 3, 1))
        return out
    with precision("bf16x3"):
        st = ops._stream()
        d = ops.conv_desc(N, H, W, Cis, Cos, K, 1, pad, 0, Ci, Co)
        D = ctypes.byref(d)
        assert _lib.query("acg_conv2d_bwd_data_sums_supported", D)
        pk = ops.PackedConv(t(w), t(np.zeros(Co)), ops.cpad(Ci), ops.cpad(Co))   # (packed widths; the stored ones: pk.Cis / pk.Cos)
        assert (pk.Cis, pk.Cos) == (Cis, Cos)
        X, Wt = leaf(np.zeros((N, Ci, H, W))), leaf(w)
        backward(oops.conv2d(X, Wt, None, pad=pad), seed=dy.astype(np.float32).astype(np.float64))
        xh = (xn.astype(np.float32).astype(np.float64) - mean[:, :, None, None]) * rstd[:, :, None, None]
        g_b = np.broadcast_to(gamma, (N, Ci))[:, :, None, None]; b_b = np.broadcast_to(beta, (N, Ci))[:, :, None, None]
        live = (xh * g_b + b_b > 0) if act == ops.ACT_RELU else np.ones_like(xh, dtype=bool)
        edge = (np.abs(xh * g_b + b_b) < 1e-5) if act == ops.ACT_RELU else np.zeros_like(live)
        gy = X.g * live
        nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
        ws = ops.workspace(max(nbw, 1))
        dx = torch.full((N, H, W, Cis), float("nan"), device="cuda")
        psum = torch.full((N, H * W // 128, 2, Cis), float("nan"), device="cuda")
        ns = _lib.NormSumsDesc()
        xn_t, mean_t, rstd_t, g_t, b_t = t(nhwc(xn, Cis)), t(mean.reshape(-1)), t(rstd.reshape(-1)), t(gamma.reshape(-1)), t(beta.reshape(-1))
        dy_t = t(nhwc(dy, Cos))
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta = P(xn_t), P(mean_t), P(rstd_t), P(g_t), P(b_t)
        ns.gstride, ns.sign_mask, ns.act, ns.part = (Ci if per_sample else 0), None, act, P(psum)
        _lib.call("acg_conv2d_bwd_data_sums", D, P(dy_t), P(pk.wb), P(dx), P(ws), nbw, ctypes.byref(ns), st)
        assert _lib.query("acg_last_kernel").decode().startswith(kern), _lib.query("acg_last_kernel")
        assert rel(nchw(n(dx)), X.g) < 2e-5, "data gradient"
        dx0 = torch.full_like(dx, float("nan"))
        _lib.call("acg_conv2d_bwd_data", D, P(dy_t), P(pk.wb), P(dx0), P(ws), nbw, st)
        assert torch.equal(dx0, dx), "the sums epilogue must not change dx"
        got = n(psum)
        assert np.isfinite(got).all(), "a chunk entry of the sums was not written"
        s1, s2 = got[:, :, 0].sum(1), got[:, :, 1].sum(1)
        assert edge.mean() < 1e-4
        gy_k = nchw(n(dx)).astype(np.float64) * live
        assert rel(s1, gy_k.sum((2, 3))) < 1e-4 and rel(s2, (gy_k * xh).sum((2, 3))) < 1e-4, "norm backward sums (kernel dx)"
        assert rel(s1, gy.sum((2, 3))) < 2e-4 and rel(s2, (gy * xh).sum((2, 3))) < 2e-4, "norm backward sums (oracle dx)"


@pytest.mark.parametrize("case", ["relu_shared", "relu_per_sample", "none_shared"])
@pytest.mark.parametrize("layer", ["a2_generic_tile", "head_thinrow"])
def test_tile_data_gradient_sums_match_the_oracle(layer, case):
    """acg_conv2d_bwd_data_sums on the two remaining producers of round 6 — the generic 128-pixel tile (data gradient of the
    3x3 32 -> 64 layer, networks.py:164, the gradient w.r.t. the stem's norm output) and conv_thinrow_x3 (data gradient of the
    7x7 head, networks.py:187-188, w.r.t. the last norm's output) — through the C ABI at N = 2: dx against the oracle's
    convolution adjoint (2e-5) and bit-equal to the plain entry point's, S1 / S2 against fp64 (1e-4) with `part` pre-filled
    with NaN, ReLU and no activation, shared and per-sample affine parameters."""
    import ctypes
    from dtgan_amd import ops, _lib
    from hip_util import precision, t, n, rel
    P = ops._ptr
    if layer == "a2_generic_tile":
        N, H, W, Ci, Co, K, pad, Cis, Cos, kern = 2, 12, 128, 32, 64, 3, 1, 32, 64, "igemm_conv_bf16<128,32,"
    else:   # the head: 32 -> 3 channels, the image side stored C4
        N, H, W, Ci, Co, K, pad, Cis, Cos, kern = 2, 16, 48, 32, 3, 7, 3, 32, 4, "conv_thinrow_x3<REFLECT=0,SUMS=1>"
    rs = np.random.RandomState(17)
    w = rs.normal(0, 0.05, (Co, Ci, K, K))
    dy = rs.normal(0, 1e-2, (N, Co, H, W))
    xn = rs.normal(0.2, 1.1, (N, Ci, H, W))
    mean = rs.normal(0.2, 0.1, (N, Ci)); rstd = rs.uniform(0.6, 1.4, (N, Ci))
    per_sample = case == "relu_per_sample"
    act = ops.ACT_NONE if case == "none_shared" else ops.ACT_RELU
    gamma = rs.normal(1.0, 0.4, (N if per_sample else 1, Ci)); beta = rs.normal(0.0, 0.5, (N if per_sample else 1, Ci))
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))

    def nhwc(a, Cp):
        out = np.zeros((a.shape[0], a.shape[2], a.shape[3], Cp), np.float32)
        out[..., :a.shape[1]] = np.transpose(a, (0, 2, 3, 1))
        return out
    with precision("bf16x3"):
        st = ops._stream()
        d = ops.conv_desc(N, H, W, Cis, Cos, K, 1, pad, 0, Ci, Co)
        D = ctypes.byref(d)
        assert _lib.query("acg_conv2d_bwd_data_sums_supported", D)
        pk = ops.PackedConv(t(w), t(np.zeros(Co)), ops.cpad(Ci), ops.cpad(Co))   # (packed widths; the stored ones: pk.Cis / pk.Cos)
        assert (pk.Cis, pk.Cos) == (Cis, Cos)
        X, Wt = leaf(np.zeros((N, Ci, H, W))), leaf(w)
        backward(oops.conv2d(X, Wt, None, pad=pad), seed=dy.astype(np.float32).astype(np.float64))
        xh = (xn.astype(np.float32).astype(np.float64) - mean[:, :, None, None]) * rstd[:, :, None, None]
        g_b = np.broadcast_to(gamma, (N, Ci))[:, :, None, None]; b_b = np.broadcast_to(beta, (N, Ci))[:, :, None, None]
        live = (xh * g_b + b_b > 0) if act == ops.ACT_RELU else np.ones_like(xh, dtype=bool)
        edge = (np.abs(xh * g_b + b_b) < 1e-5) if act == ops.ACT_RELU else np.zeros_like(live)
        gy = X.g * live
        nbw = _lib.query("acg_conv2d_bwd_data_workspace_bytes", D)
        ws = ops.workspace(max(nbw, 1))
        dx = torch.full((N, H, W, Cis), float("nan"), device="cuda")
        psum = torch.full((N, H * W // 128, 2, Cis), float("nan"), device="cuda")
        ns = _lib.NormSumsDesc()
        xn_t, mean_t, rstd_t, g_t, b_t = t(nhwc(xn, Cis)), t(mean.reshape(-1)), t(rstd.reshape(-1)), t(gamma.reshape(-1)), t(beta.reshape(-1))
        dy_t = t(nhwc(dy, Cos))
        ns.x, ns.mean, ns.rstd, ns.gamma, ns.beta = P(xn_t), P(mean_t), P(rstd_t), P(g_t), P(b_t)
        ns.gstride, ns.sign_mask, ns.act, ns.part = (Ci if per_sample else 0), None, act, P(psum)
        _lib.call("acg_conv2d_bwd_data_sums", D, P(dy_t), P(pk.wb), P(dx), P(ws), nbw, ctypes.byref(ns), st)
        assert _lib.query("acg_last_kernel").decode().startswith(kern), _lib.query("acg_last_kernel")
        assert rel(nchw(n(dx)), X.g) < 2e-5, "data gradient"
        dx0 = torch.full_like(dx, float("nan"))
        _lib.call("acg_conv2d_bwd_data", D, P(dy_t), P(pk.wb), P(dx0), P(ws), nbw, st)
        assert torch.equal(dx0, dx), "the sums epilogue must not change dx"
        got = n(psum)
        assert np.isfinite(got).all(), "a chunk entry of the sums was not written"
        s1, s2 = got[:, :, 0].sum(1), got[:, :, 1].sum(1)
        assert edge.mean() < 1e-4
        gy_k = nchw(n(dx)).astype(np.float64) * live
        assert rel(s1, gy_k.sum((2, 3))) < 1e-4 and rel(s2, (gy_k * xh).sum((2, 3))) < 1e-4, "norm backward sums (kernel dx)"
        assert rel(s1, gy.sum((2, 3))) < 2e-4 and rel(s2, (gy * xh).sum((2, 3))) < 2e-4, "norm backward sums (oracle dx)"


# ---------------------------------------------------------------------------------------------------------------------
# single layers at W = 256: the four-phases-in-one-tile data gradient / ConvTranspose forward and the three-tap weight
# gradient (networks.py:168, 178-179), against the oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("N,H,W", [(1, 16, 256), (2, 8, 512)])
def test_stride2_downsample_at_full_width_matches_the_oracle(N, H, W, prec):
    from hip_util import precision, t, n, rel
    from dtgan_amd import modules as M
    rs = np.random.RandomState(N * 1000 + W)
    x = rs.normal(0, 1, (N, 64, H, W)); w = rs.normal(0, 0.1, (128, 64, 3, 3)); b = rs.normal(0, 0.5, (128,))
    with precision(prec):
        m = M.Sequential(M.Conv2d(64, 128, 3, stride=2, padding=1, bias=True)).cuda()
        conv = m[0]
        with torch.no_grad():
            conv.weight.copy_(t(w)); conv.bias.copy_(t(b))
        M.mark_dirty(m)
        xt = t(x, grad=True)
        with Spy() as spy:
            y = m(xt)
            X, Wt, Bt = leaf(x), leaf(w), leaf(b)
            yo = oops.conv2d(X, Wt, Bt, stride=2, pad=1)
            rr = rs.normal(0, 1, yo.v.shape)
            y.backward(t(rr))
        backward(yo, seed=rr)
    assert rel(n(y), yo.v) < 2e-5
    assert rel(n(xt.grad), X.g) < 2e-5, "dgrad"
    assert rel(n(conv.weight.grad), Wt.g) < 1e-4, "wgrad"
    assert rel(n(conv.bias.grad), Bt.g) < 1e-4, "bias grad"
    if prec == "bf16x3":
        assert any(k.startswith("igemm_conv_ph4") for k in spy.kernels("acg_conv2d_bwd_data")), spy.seen
        assert any(k == "wgrad_x3_krowg<NT=3,IS=2,BCI=64>" for k in spy.kernels("acg_conv2d_bwd_weight")), spy.seen


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_conv_transpose_at_full_width_matches_the_oracle(prec):
    from hip_util import precision, t, n, rel
    from dtgan_amd import modules as M
    Ci, Co, N, H, W = 128, 64, 1, 8, 128          # output 16 x 256
    rs = np.random.RandomState(77)
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.1, (Ci, Co, 3, 3)); b = rs.normal(0, 0.5, (Co,))
    with precision(prec):
        m = M.ConvTranspose2d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=True).cuda()
        with torch.no_grad():
            m.weight.copy_(t(w)); m.bias.copy_(t(b))
        M.mark_dirty(m)
        xt = t(x, grad=True)
        with Spy() as spy:
            y = m(xt)
            X, Wt, Bt = leaf(x), leaf(w), leaf(b)
            yo = oops.conv_transpose2d(X, Wt, Bt)
            rr = rs.normal(0, 1, yo.v.shape)
            y.backward(t(rr))
        backward(yo, seed=rr)
    assert y.shape == yo.v.shape == (N, Co, 2 * H, 2 * W)
    assert rel(n(y), yo.v) < 2e-5
    assert rel(n(xt.grad), X.g) < 2e-5
    assert rel(n(m.weight.grad), Wt.g) < 1e-4
    assert rel(n(m.bias.grad), Bt.g) < 1e-4
    if prec == "bf16x3":
        assert any(k.startswith("igemm_conv_ph4") for k in spy.kernels("acg_conv_transpose2d_fwd")), spy.seen
        assert any(k == "wgrad_x3_krowg<NT=3,IS=2,BCI=64>" for k in spy.kernels("acg_conv_transpose2d_bwd_weight")), spy.seen


# ---------------------------------------------------------------------------------------------------------------------
# D_B's deep 4x4 layers (networks.py:321-338) at the maps of the 256 x 256 step: 128 -> 256 on 64 x 64 (63-wide output rows),
# 256 -> 256 on 63-wide rows (62 out) — the kernel-row weight gradient wgrad_x3_krowg, plus other widths, stride 2 and the 64-channel
# input tile of D_B's second layer
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("case", [(1, 128, 256, 1, 64, 64), (1, 256, 256, 1, 20, 63), (3, 128, 128, 2, 10, 40),
                                  (1, 128, 128, 1, 5, 17), (5, 256, 128, 1, 7, 25), (2, 128, 128, 2, 9, 33),
                                  (2, 128, 128, 1, 6, 49), (2, 128, 256, 2, 12, 128), (2, 64, 128, 2, 16, 128), (1, 64, 128, 2, 10, 48)],
                         ids=lambda c: "n%d_%dto%d_s%d_%dx%d" % c)
def test_discriminator_4x4_layers_match_the_oracle(case, prec):
    from hip_util import precision, t, n, rel
    from dtgan_amd import modules as M
    N, Ci, Co, s, H, W = case
    rs = np.random.RandomState(N * 100 + W + s)
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.05, (Co, Ci, 4, 4)); b = rs.normal(0, 0.5, (Co,))
    with precision(prec):
        m = M.Sequential(M.Conv2d(Ci, Co, 4, stride=s, padding=1, bias=True)).cuda()
        conv = m[0]
        with torch.no_grad():
            conv.weight.copy_(t(w)); conv.bias.copy_(t(b))
        M.mark_dirty(m)
        xt = t(x, grad=True)
        with Spy() as spy:
            y = m(xt)
            X, Wt, Bt = leaf(x), leaf(w), leaf(b)
            yo = oops.conv2d(X, Wt, Bt, stride=s, pad=1)
            rr = rs.normal(0, 1, yo.v.shape)
            y.backward(t(rr))
        backward(yo, seed=rr)
    assert rel(n(y), yo.v) < 2e-5
    assert rel(n(xt.grad), X.g) < 2e-5, "dgrad"
    assert rel(n(conv.weight.grad), Wt.g) < 1e-4, "wgrad"
    assert rel(n(conv.bias.grad), Bt.g) < 1e-4, "bias grad"
    if prec == "bf16x3":
        assert any(k.startswith("wgrad_x3_krowg<NT=4,IS=%d" % s) for k in spy.kernels("acg_conv2d_bwd_weight")), spy.seen
