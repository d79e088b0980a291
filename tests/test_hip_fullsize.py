"""GPU tests at the headline geometry (BASELINE.json configs[2]: 256x256x3, ngf 32, 9 resblocks, E_B + D_z_B; batch
kept small so the file runs in seconds).  No oracle can run at this size, so these are size-independent PROPERTIES:

* the two parity arithmetics agree with each other: the exact-fp32 kernels (conv_igemm / conv_wgrad, full reflect fold,
  separate statistics passes) and the bf16x3 kernels (wave-specialised tile, frame fold, epilogue statistics, patch
  kernel, fused skip gradient) are different code paths all the way down;
* the MFMA convolution equals the naive `direct` kernels (geometry straight from the descriptor) on the 128x128x128
  resblock shape;
* the data gradient is linear in its argument;
* InstanceNorm networks are per-sample: a batch of two equals two batches of one;
* the step is deterministic (fixed-order reductions, no float atomics): same state + same inputs -> bit-identical
  losses, gradient norms and images."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from test_hip_step import make_opt  # noqa: E402

FULL = dict(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, n_blocks=9)
S, N = 256, 4  # >= 3: the latent discriminator / encoder end in BatchNorm over the batch


def _model(seed=0):
    from hip_util import load_recipe
    from dtgan_amd import model as M
    m = M.AugmentedCycleGAN(make_opt(**FULL), testing=True)
    for k, net in m._net_dict().items():
        load_recipe(net, k, seed, "init")
    return m


def _inputs(seed):
    from oracle import recipe
    return recipe.inputs(seed, N, 3, 3, S, 16)


def _step(prec, steps=1):
    from hip_util import t, n, precision
    with precision(prec):
        m = _model()
        out = []
        for s in range(steps):
            A, B, z = _inputs(40 + s)
            losses, visuals, gnorms = m.train_instance(t(A), t(B), t(z))
            out.append((dict(losses), {k: n(v) for k, v in visuals.items()}, dict(gnorms)))
    return out


def test_fullsize_step_f32_and_bf16x3_agree():
    from hip_util import rel
    (l32, v32, g32), = _step("f32")
    (lx3, vx3, gx3), = _step("bf16x3")
    assert list(l32.keys()) == list(lx3.keys())
    a, b = np.array(list(lx3.values())), np.array(list(l32.values()))
    bad = {k: (x, y) for k, x, y in zip(l32.keys(), a, b) if not np.isclose(x, y, rtol=1e-3, atol=2e-6)}
    assert not bad, bad                                                                       # the north-star bar
    for k in ("fake_A", "fake_B", "rec_A", "rec_B"):
        assert rel(vx3[k], v32[k]) < 1e-3, k
    a, b = np.array(list(gx3.values())), np.array(list(g32.values()))
    assert np.allclose(a, b, rtol=3e-3, atol=1e-6), dict(zip(g32.keys(), zip(a, b)))


def test_fullsize_step_is_deterministic():
    r1 = _step("bf16x3", steps=2)
    r2 = _step("bf16x3", steps=2)
    for (l1, v1, g1), (l2, v2, g2) in zip(r1, r2):
        assert l1 == l2 and g1 == g2
        for k in v1:
            assert np.array_equal(v1[k], v2[k]), k


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_resblock_conv_mfma_equals_direct_kernels(prec):
    """3x3 reflect 128->128 on 128x128 (the dominant layer): forward, data gradient, weight gradient"""
    from hip_util import t, n, rel, precision
    from dtgan_amd import modules as M, ops
    rs = np.random.RandomState(2)
    x = rs.normal(0, 1, (2, 128, 128, 128)).astype(np.float32)
    w = (rs.normal(0, 1, (128, 128, 3, 3)) * 0.03).astype(np.float32)
    b = rs.normal(0, 0.5, 128).astype(np.float32)
    r = rs.normal(0, 1, (2, 128, 128, 128)).astype(np.float32)
    res = {}
    with precision(prec):
        for impl in ("mfma", "direct"):
            ops.set_conv_impl(impl)
            try:
                m = M.Sequential(nn.ReflectionPad2d(1), M.Conv2d(128, 128, 3, padding=0, bias=True)).cuda()
                with torch.no_grad():
                    m[1].weight.copy_(t(w)); m[1].bias.copy_(t(b))
                xt = t(x, grad=True)
                y = m(xt)
                y.backward(t(r))
                res[impl] = (n(y), n(xt.grad), n(m[1].weight.grad), n(m[1].bias.grad))
            finally:
                ops.set_conv_impl("mfma")
    tol = 2e-5 if prec == "bf16x3" else 5e-6
    for a, d, name in zip(res["mfma"], res["direct"], ("fwd", "dgrad", "wgrad", "bias")):
        assert rel(a, d) < (2e-4 if name == "wgrad" and prec == "bf16x3" else tol * (10 if name == "wgrad" else 1)), name


def test_data_gradient_is_linear():
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    rs = np.random.RandomState(3)
    m = M.Sequential(nn.ReflectionPad2d(1), M.Conv2d(128, 128, 3, padding=0, bias=True)).cuda()
    x = t(rs.normal(0, 1, (1, 128, 128, 128)), grad=True)
    y = m(x)
    g1, g2 = t(rs.normal(0, 1, y.shape)), t(rs.normal(0, 1, y.shape))
    d1, = torch.autograd.grad(y, x, g1, retain_graph=True)
    d2, = torch.autograd.grad(y, x, g2, retain_graph=True)
    d12, = torch.autograd.grad(y, x, 0.5 * g1 - 2.0 * g2)
    assert rel(n(d12), n(0.5 * d1 - 2.0 * d2)) < 1e-4  # bf16x3 splits each operand: linear up to the 2^-17 operand rounding


def test_instance_norm_generator_is_per_sample():
    from hip_util import t, n, rel, load_recipe
    from dtgan_amd import networks as Nw
    net = Nw.define_G(3, 3, 32, gpu_ids=[0], n_blocks=9)
    load_recipe(net, "netG_B_A", 0, "init")
    net.train()
    B = _inputs(50)[1]
    with torch.no_grad():
        both = n(net.forward(t(B)))
        one = n(net.forward(t(B[:1])))
    assert rel(both[:1], one) < 1e-5
