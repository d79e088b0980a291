"""GPU tests of the bf16 throughput mode (`ops.set_precision('bf16')`): conv operands are rounded to bf16 on their
way into LDS, products accumulate in fp32 (v_mfma_f32_32x32x16_bf16); tensors in HBM stay fp32.

This is NOT a parity path (those are bf16x3 and f32: tests/test_hip_ops.py ... test_hip_step.py, 1e-3 bar).  Tolerances here
follow the error model: two 2^-9 roundings per product, fp32 sums -> ~4e-3 of the output scale per conv; a dozen
layers deep (whole generator) a few 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ops as oops  # noqa: E402
from oracle.tape import backward, leaf  # noqa: E402
from test_hip_ops import CONV_CASES, _conv_module  # noqa: E402
from golden_util import load  # noqa: E402


@pytest.fixture(autouse=True)
def bf16_mode():
    from dtgan_amd import ops
    before = ops.get_precision()
    ops.set_precision("bf16")
    yield
    ops.set_precision(before)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k%ds%dp%d%s_%dto%d_%dx%dx%d" % c)
def test_conv2d_bf16(case):
    from hip_util import t, n, rel
    K, stride, pad, mode, Ci, Co, N, H, W = case
    rs = np.random.RandomState(sum(c if isinstance(c, int) else len(c) for c in case))
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.3, (Co, Ci, K, K)); b = rs.normal(0, 0.5, (Co,))
    m = _conv_module(K, stride, pad, mode, Ci, Co)
    conv = [c for c in m.modules() if c.__class__.__name__ == "Conv2d"][0]
    with torch.no_grad():
        conv.weight.copy_(t(w)); conv.bias.copy_(t(b))
    xt = t(x, grad=True)
    y = m(xt)
    X, Wt, Bt = leaf(x), leaf(w), leaf(b)
    yo = oops.conv2d(X, Wt, Bt, stride=stride, pad=pad, pad_mode=mode)
    assert rel(n(y), yo.v) < 1e-2
    # the kernel must be EXACT on bf16-representable data (products of bf16 values are exact in fp32):
    xq = t(x).bfloat16().float(); wq = t(w).bfloat16().float()
    with torch.no_grad():
        conv.weight.copy_(wq)
    from dtgan_amd.modules import mark_dirty
    mark_dirty(m)
    yq = m(xq)
    yoq = oops.conv2d(leaf(n(xq).astype(np.float64)), leaf(n(wq).astype(np.float64)), Bt, stride=stride, pad=pad, pad_mode=mode)
    assert rel(n(yq), yoq.v) < 2e-5, "bf16-exact inputs must reproduce the fp64 result to fp32 rounding"
    # same exactness requirement for the data- and weight-gradient kernels: bf16-representable x, w AND dy
    xq2 = xq.clone().requires_grad_(True)
    yq = m(xq2)
    rq = t(rs.normal(0, 1, yo.v.shape)).bfloat16().float()
    conv.weight.grad = None
    yq.backward(rq)
    Xq, Wq = leaf(n(xq).astype(np.float64)), leaf(n(wq).astype(np.float64))
    yoq = oops.conv2d(Xq, Wq, Bt, stride=stride, pad=pad, pad_mode=mode)
    backward(yoq, seed=n(rq).astype(np.float64))
    assert rel(n(xq2.grad), Xq.g) < 2e-5, "dgrad must be exact on bf16-representable data"
    assert rel(n(conv.weight.grad), Wq.g) < 1e-4, "wgrad must be exact on bf16-representable data"
    # and on generic data the error stays at the bf16 rounding level
    with torch.no_grad():
        conv.weight.copy_(t(w))
    mark_dirty(m)
    conv.weight.grad = None
    y = m(xt)
    r = rs.normal(0, 1, yo.v.shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-2, "dgrad"
    assert rel(n(conv.weight.grad), Wt.g) < 1e-2, "wgrad"
    assert rel(n(conv.bias.grad), Bt.g) < 1e-4, "bias grad (fp32 path)"


def test_conv_transpose_bf16():
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    Ci, Co, N, H, W = 128, 64, 1, 6, 6
    rs = np.random.RandomState(3)
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.3, (Ci, Co, 3, 3)); b = rs.normal(0, 0.5, (Co,))
    m = M.ConvTranspose2d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=True).cuda()
    with torch.no_grad():
        m.weight.copy_(t(w)); m.bias.copy_(t(b))
    xt = t(x, grad=True)
    y = m(xt)
    X, Wt, Bt = leaf(x), leaf(w), leaf(b)
    yo = oops.conv_transpose2d(X, Wt, Bt)
    assert rel(n(y), yo.v) < 1e-2
    r = rs.normal(0, 1, yo.v.shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-2 and rel(n(m.weight.grad), Wt.g) < 1e-2


@pytest.mark.parametrize("name", ["G_B_A_s16_nb3", "G_A_B_s16_nb9", "D_B_s40", "E_B_s64"])
def test_nets_bf16_close_to_reference(name):
    from hip_util import t, n, rel, load_recipe
    from test_hip_nets import build
    arr, meta = load(name)
    net = load_recipe(build(meta), meta["net"], meta["seed"], meta["flavour"])
    net.train()
    ins, i = [], 0
    while "in%d" % i in arr:
        ins.append(t(arr["in%d" % i], grad=True)); i += 1
    out = net.forward(*ins)
    outs = list(out) if isinstance(out, tuple) else [out]
    def l2rel(a, b):  # whole-tensor relative L2 error: robust to the few ReLU-mask flips bf16 rounding causes
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    # measured on MI355X: forward 0.7-2e-2, input gradients 0.1-0.33 relative L2.  The gradient figure is the
    # ReLU/LeakyReLU mask-flip discontinuity (~1% of units flip under 1e-2 forward noise -> sqrt(0.01) = 10%),
    # inherent to reduced precision on these random high-gain networks; kernel correctness is pinned by the
    # exactness checks in test_conv2d_bf16, so these are sanity bounds only.
    for j, o in enumerate(outs):
        assert l2rel(n(o), arr["out%d" % j]) < 4e-2, "forward out%d" % j
    loss = sum((o * t(arr["R%d" % j])).sum() for j, o in enumerate(outs))
    loss.backward()
    for j, x in enumerate(ins):
        assert l2rel(n(x.grad), arr["gin%d" % j]) < 0.5, "input grad %d" % j


def test_step_bf16_close_to_reference():
    from hip_util import t
    from test_hip_step import build_model
    arr, meta = load("step_aug_small_s64")
    m = build_model(meta)
    A, B, z = (t(arr["s0/%s" % k]) for k in ("real_A", "real_B", "prior_z_B"))
    losses, visuals, gnorms = m.train_instance(A, B, z)
    got, ref = np.array(list(losses.values())), arr["s0/losses"]
    assert np.allclose(got, ref, rtol=3e-2, atol=1e-3), dict(zip(meta["loss_keys"], zip(got, ref)))
    gg, gr = np.array(list(gnorms.values())), arr["s0/gnorms"]
    # mask flips, see above; E_B additionally sits behind a BatchNorm over a 4-sample batch of 1x1 maps, which amplifies
    # whatever rounding reaches it (the bf16x3 / f32 parity tests hold it to 3e-3 / 1e-3)
    tol = np.array([1.0 if k == "gnorm_E_B" else 0.5 for k in meta["gnorm_keys"]])
    assert np.all(np.abs(gg - gr) <= tol * np.abs(gr) + 1e-3), dict(zip(meta["gnorm_keys"], zip(gg, gr)))
