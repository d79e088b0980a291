"""Randomised convolution geometries: the MFMA path (whatever tile / kernel the dispatcher picks: wave-specialised, generic,
patch, thin, frame fold ...) against the naive `direct` kernels, which take their geometry straight from the descriptor and
share no tiling code with it.  Forward, data gradient, weight gradient and bias gradient, both parity arithmetics.
Seeded: the same 160 cases every run."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _cases():
    rs = np.random.RandomState(20260)
    out = []
    chans = [1, 3, 8, 16, 32, 48, 64, 128, 160, 256]
    while len(out) < 160:
        K = int(rs.choice([1, 3, 4, 7]))
        stride = int(rs.choice([1, 1, 1, 2]))
        mode = str(rs.choice(["zero", "reflect"]))
        if stride == 2:
            if K not in (3, 4):
                continue
            mode = "zero"
        pad = int(rs.choice([0, K // 2, min(1, K - 1)]))  # the ABI requires pad < K
        if mode == "reflect" and pad == 0:
            mode = "zero"
        Ci, Co = int(rs.choice(chans)), int(rs.choice(chans))
        if Ci <= 4 and Co <= 4:
            continue
        if stride == 2 and Co <= 4:
            continue  # not in the reference; dgrad raises (documented)
        N = int(rs.choice([1, 2, 3]))
        H, W = int(rs.randint(max(K, 2 * pad + 2, 5), 41)), int(rs.randint(max(K, 2 * pad + 2, 5), 41))
        if Ci * Co >= 128 * 128:
            H, W = min(H, 20), min(W, 24)
        out.append((K, stride, pad, mode, Ci, Co, N, H, W))
    return out


def _run(case, impl):
    from hip_util import t, n
    from dtgan_amd import modules as M, ops
    K, stride, pad, mode, Ci, Co, N, H, W = case
    rs = np.random.RandomState(abs(hash(case)) % (2 ** 31))
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.2, (Co, Ci, K, K)); b = rs.normal(0, 0.5, (Co,))
    ops.set_conv_impl(impl)
    try:
        if mode == "reflect":
            m = M.Sequential(nn.ReflectionPad2d(pad), M.Conv2d(Ci, Co, K, stride=stride, padding=0, bias=True)).cuda()
        else:
            m = M.Sequential(M.Conv2d(Ci, Co, K, stride=stride, padding=pad, bias=True)).cuda()
        conv = [c for c in m.modules() if c.__class__.__name__ == "Conv2d"][0]
        with torch.no_grad():
            conv.weight.copy_(t(w)); conv.bias.copy_(t(b))
        xt = t(x, grad=True)
        y = m(xt)
        r = np.random.RandomState(7).normal(0, 1, tuple(y.shape))
        y.backward(t(r))
        return n(y), n(xt.grad), n(conv.weight.grad), n(conv.bias.grad)
    finally:
        ops.set_conv_impl("mfma")


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("case", _cases(), ids=lambda c: "k%ds%dp%d%s_%dto%d_%dx%dx%d" % c)
def test_conv_mfma_equals_direct(case, prec):
    from hip_util import rel, precision
    with precision(prec):
        a = _run(case, "mfma")
        d = _run(case, "direct")
    tol = 3e-5 if prec == "bf16x3" else 1e-5
    for got, ref, name in zip(a, d, ("fwd", "dgrad", "wgrad", "bias")):
        assert got.shape == ref.shape, name
        assert rel(got, ref) < (tol * 4 if name == "wgrad" else tol), name


def _norm_cases():
    rs = np.random.RandomState(777)
    out = []
    for _ in range(24):
        C = int(rs.choice([3, 8, 16, 24, 48, 64, 128, 130, 256]))
        N = int(rs.choice([1, 2, 5]))
        H, W = int(rs.randint(2, 50)), int(rs.randint(2, 50))
        out.append((N, C, H, W))
    return out


@pytest.mark.parametrize("shape", _norm_cases(), ids=lambda s: "%dx%dx%dx%d" % s)
def test_instance_norm_random_shapes(shape):
    """InstanceNorm (modules.py:64-97: biased variance, eps 1e-5, affine) forward / backward on random shapes — channel
    counts that are not multiples of 16, pixel counts around the 256-pixel reduction chunks — against the same formula in
    fp64 torch on the device."""
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    N, C, H, W = shape
    rs = np.random.RandomState(N * 1000 + C + H * 7 + W)
    x = rs.normal(0.3, 1.2, shape); sc = rs.normal(1, 0.3, C); sh = rs.normal(0, 0.3, C); r = rs.normal(0, 1, shape)
    m = M.InstanceNorm(C).cuda()
    with torch.no_grad():
        m.scale.copy_(t(sc)); m.shift.copy_(t(sh))
    xt = t(x, grad=True)
    y = m(xt)
    y.backward(t(r))
    xd = torch.tensor(x, dtype=torch.float64, device="cuda", requires_grad=True)
    scd = torch.tensor(sc, dtype=torch.float64, device="cuda", requires_grad=True)
    shd = torch.tensor(sh, dtype=torch.float64, device="cuda", requires_grad=True)
    mu = xd.mean(dim=(2, 3), keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    yd = (xd - mu) / torch.sqrt(var + 1e-5) * scd.view(1, C, 1, 1) + shd.view(1, C, 1, 1)
    yd.backward(torch.tensor(r, dtype=torch.float64, device="cuda"))
    assert rel(n(y), yd.detach().cpu().numpy()) < 2e-5
    assert rel(n(xt.grad), xd.grad.cpu().numpy()) < 2e-4
    assert rel(n(m.scale.grad), scd.grad.cpu().numpy()) < 2e-4
    assert rel(n(m.shift.grad), shd.grad.cpu().numpy()) < 2e-4


def _ct_cases():
    rs = np.random.RandomState(4242)
    out = []
    for _ in range(20):
        Ci, Co = int(rs.choice([8, 16, 32, 48, 64, 128, 256])), int(rs.choice([8, 16, 32, 64, 128]))
        N, H, W = int(rs.choice([1, 2, 3])), int(rs.randint(2, 24)), int(rs.randint(2, 24))
        out.append((Ci, Co, N, H, W))
    return out


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("dims", _ct_cases(), ids=lambda c: "%dto%d_%dx%dx%d" % c)
def test_conv_transpose_mfma_equals_direct(dims, prec):
    """nn.ConvTranspose2d(k3, s2, p1, op1) (networks.py:176-181, 228-233): sub-pixel-phase MFMA path vs the naive kernels"""
    from hip_util import t, n, rel, precision
    from dtgan_amd import modules as M, ops
    Ci, Co, N, H, W = dims
    rs = np.random.RandomState(Ci * 31 + Co + H)
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.3, (Ci, Co, 3, 3)); b = rs.normal(0, 0.5, (Co,))
    r = rs.normal(0, 1, (N, Co, 2 * H, 2 * W))
    res = {}
    with precision(prec):
        for impl in ("mfma", "direct"):
            ops.set_conv_impl(impl)
            try:
                m = M.ConvTranspose2d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=True).cuda()
                with torch.no_grad():
                    m.weight.copy_(t(w)); m.bias.copy_(t(b))
                xt = t(x, grad=True)
                y = m(xt)
                y.backward(t(r))
                res[impl] = (n(y), n(xt.grad), n(m.weight.grad), n(m.bias.grad))
            finally:
                ops.set_conv_impl("mfma")
    tol = 3e-5 if prec == "bf16x3" else 1e-5
    for got, ref, name in zip(res["mfma"], res["direct"], ("fwd", "dgrad", "wgrad", "bias")):
        assert rel(got, ref) < (tol * 4 if name == "wgrad" else tol), name
