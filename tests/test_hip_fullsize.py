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
    # gradient norms: 3e-3, except G_A_B's, the smallest by three orders of magnitude at initialisation (its CondInstanceNorm
    # scales start near zero): 4.3e-3 measured between the two arithmetics, allowed 6e-3
    a, b = np.array(list(gx3.values())), np.array(list(g32.values()))
    tol = np.array([6e-3 if k == "gnorm_G_A_B" else 3e-3 for k in g32.keys()])
    assert np.all(np.abs(a - b) <= tol * np.abs(b) + 1e-6), dict(zip(g32.keys(), zip(a, b)))


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


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] (128x128x3, 6 resblocks, batch 16, fp32) and configs[4] (512x512x1, 9 resblocks) at full widths
# ----------------------------------------------------------------------------------------------------------------------
CFG2 = dict(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, n_blocks=6)
CFG5 = dict(input_nc=1, output_nc=1, ngf=32, nef=32, ndf=64, nlatent=16, n_blocks=9)


def _run_cfg(kw, S, Nb, prec, steps=1, seed=0, in_seed=60):
    from hip_util import t, n, precision, load_recipe
    from dtgan_amd import model as M
    from oracle import recipe
    with precision(prec):
        m = M.AugmentedCycleGAN(make_opt(**kw), testing=True)
        for k, net in m._net_dict().items():
            load_recipe(net, k, seed, "init")
        out = []
        for s in range(steps):
            A, B, z = recipe.inputs(in_seed + s, Nb, kw["input_nc"], kw["output_nc"], S, 16)
            losses, visuals, gnorms = m.train_instance(t(A), t(B), t(z))
            out.append((dict(losses), {k: n(v) for k, v in visuals.items()}, dict(gnorms)))
    return out


def _agree(rx3, r32):
    from hip_util import rel
    (lx3, vx3, gx3), (l32, v32, g32) = rx3, r32
    a, b = np.array(list(lx3.values())), np.array(list(l32.values()))
    bad = {k: (x, y) for k, x, y in zip(l32.keys(), a, b) if not np.isclose(x, y, rtol=1e-3, atol=2e-6)}
    assert not bad, bad
    for k in ("fake_A", "fake_B", "rec_A", "rec_B"):
        assert rel(vx3[k], v32[k]) < 1e-3, (k, rel(vx3[k], v32[k]))
    a, b = np.array(list(gx3.values())), np.array(list(g32.values()))
    assert np.allclose(a, b, rtol=3e-3, atol=1e-6), dict(zip(g32.keys(), zip(a, b)))


def test_config2_full_batch_step_fp32():
    """configs[1] exactly as named: 128x128x3, 6-resblock generators, batch 16, exact-fp32 arithmetic — two steps (the second
    runs on Adam-updated weights); the bf16x3 arithmetic agrees with it at the north-star bar on step 0."""
    r32 = _run_cfg(CFG2, 128, 16, "f32", steps=2)
    for losses, visuals, gnorms in r32:
        assert all(np.isfinite(v) for v in losses.values()) and all(np.isfinite(v) for v in gnorms.values())
        assert visuals["rec_B"].shape == (16, 3, 128, 128) and np.abs(visuals["fake_A"]).max() <= 1.0
    assert r32[0][0] != r32[1][0]                      # the update was applied
    _agree(_run_cfg(CFG2, 128, 16, "bf16x3")[0], r32[0])


_ORACLE_CACHE = {}


def _oracle_cfg2_step():
    """(computed once for both precisions: ~15 s of host time)"""
    if "cfg2" not in _ORACLE_CACHE:
        from oracle import recipe, step
        o = step.AugStep(step.Opt(**CFG2))
        o.load({k: recipe.values_for(net.shapes, k, 2, "init") for k, net in o.nets().items()})
        _ORACLE_CACHE["cfg2"] = o.train_instance(*recipe.inputs(70, 4, 3, 3, 128, 16))
    return _ORACLE_CACHE["cfg2"]


def _oracle_generators(cfg, S, nc, nb):
    if ("gen", cfg) not in _ORACLE_CACHE:
        from oracle import nets, recipe
        from oracle.tape import T
        A, B, z = recipe.inputs(80, 1, nc, nc, S, 16)
        oB = nets.ResnetGenerator(nc, nc, 32, nb, np.float32); recipe.fill(oB, "netG_B_A", 4, "init")
        oA = nets.CINResnetGenerator(16, nc, nc, 32, nb, np.float32); recipe.fill(oA, "netG_A_B", 4, "init")
        _ORACLE_CACHE[("gen", cfg)] = (oB.forward(T(B)).v, oA.forward(T(A), T(z)).v)
    return _ORACLE_CACHE[("gen", cfg)]


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_config2_step_matches_oracle(prec):
    """configs[1] geometry at full widths, batch 4 (the oracle's C loops take a few seconds per pair here): the whole
    Augmented CycleGAN step — 13 losses, 6 gradient norms, 4 images — against the fp32 oracle.  The encoder runs on a 5x5
    map (S = 128): both sides use the spatial-mean extension."""
    from hip_util import rel
    (l1, v1, g1), = _run_cfg(CFG2, 128, 4, prec, seed=2, in_seed=70)
    l0, v0, g0 = _oracle_cfg2_step()
    lt, gt, vt = (2e-4, 1e-3, 1e-4) if prec == "f32" else (1e-3, 3e-3, 1e-3)
    assert list(l1.keys()) == list(l0.keys())
    assert np.allclose(list(l1.values()), list(l0.values()), rtol=lt, atol=2e-6), (l1, l0)
    assert np.allclose(list(g1.values()), list(g0.values()), rtol=gt, atol=1e-6), (g1, g0)
    for k in ("fake_A", "fake_B", "rec_A", "rec_B"):
        assert rel(v1[k], v0[k]) < vt, (k, rel(v1[k], v0[k]))


def test_config5_step_512x512x1():
    """configs[4]: 512x512x1 Livneh-shaped fields, 9 resblocks (batch 4 of the 16 per GPU): the two parity arithmetics —
    different kernels all the way down — agree at the north-star bar, on 1-channel images, 256x256 resblock maps
    (largest operand of the full 16-image batch: 16 x 512 x 512 x 64 x 4 B = 1.07 GB, inside the 4 GiB buffer-addressing
    limit of the conv launchers)."""
    rx3 = _run_cfg(CFG5, 512, 4, "bf16x3")
    r32 = _run_cfg(CFG5, 512, 4, "f32")
    assert rx3[0][1]["rec_A"].shape == (4, 1, 512, 512)
    _agree(rx3[0], r32[0])


def test_config3_full_batch_step():
    """configs[2] exactly as named — 256x256x3, 9 resblocks, E_B + D_z_B, batch 32 — the step the bench times: the bf16x3 step
    (pre-split trunk storage, every fused path) is deterministic bit for bit and agrees with the exact-fp32 arithmetic, whose
    kernels differ all the way down, at the north-star bar on all 13 losses and the four image tensors."""
    full = dict(FULL)
    rx3 = _run_cfg(full, 256, 32, "bf16x3", seed=0, in_seed=90)
    rx3b = _run_cfg(full, 256, 32, "bf16x3", seed=0, in_seed=90)
    assert rx3[0][0] == rx3b[0][0] and rx3[0][2] == rx3b[0][2]
    assert np.array_equal(rx3[0][1]["rec_B"], rx3b[0][1]["rec_B"])
    del rx3b
    assert rx3[0][1]["fake_B"].shape == (32, 3, 256, 256)
    r32 = _run_cfg(full, 256, 32, "f32", seed=0, in_seed=90)
    _agree(rx3[0], r32[0])


def test_config3_step_takes_the_fused_paths():
    """The hand-off protocols between autograd nodes (ops.ConvStats / SkipGrad / ReluLink / NormSums, the S16 plan) fall back
    silently when a tensor identity does not match: the step of configs[2] must take every fused path the bench line reports
    (`fused_paths`) — counts per step, independent of the batch (here 4; 2 generators x 2 passes x 9 blocks x 2 convolutions = 72 trunk
    convolutions; 54 of their data gradients feed a norm backward, 18 a ReLU bitmask link; the four norms of a generator pass that
    sit in front of a full-resolution convolution — the stem's, the one in front of the stride-2 layer, the one behind the
    ConvTranspose, the one in front of the head — take their sums from that convolution's fp32-operand data gradient:
    4 x 4 = 16, round 6 (round 5: the row pipeline's 4))."""
    from dtgan_amd import ops
    ops.FUSED.clear()
    _run_cfg(dict(FULL), 256, 4, "bf16x3", seed=0, in_seed=91)
    want = {"conv_fwd_s16": 54, "conv_fwd_s16_relu_bitmask": 18, "wgrad_s16": 72, "dgrad_s16_norm_sums": 54,
            "dgrad_s16_relu_bitmask": 18, "dgrad_s16_lazy_skip": 36, "norm_bwd_sums_from_dgrad": 70, "dgrad_f32_norm_sums": 16,
            "norm_bwd_sign_bitmask": 36,
            "norm_stats_from_conv_epilogue": 86, "conv_fwd_tile_stats": 28}
    got = {k: ops.FUSED.get(k, 0) for k in want}
    assert got == want, (got, dict(ops.FUSED))


def test_config5_full_per_gpu_batch_runs():
    """the full per-GPU batch of configs[4] (16 x 512x512x1): one bf16x3 step, finite, deterministic operand sizes"""
    (losses, visuals, gnorms), = _run_cfg(CFG5, 512, 16, "bf16x3")
    assert all(np.isfinite(v) for v in losses.values()) and all(np.isfinite(v) for v in gnorms.values())
    assert visuals["fake_B"].shape == (16, 1, 512, 512)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg5"])
def test_generator_forward_matches_oracle_at_full_size(cfg, prec):
    """Both generators at the FULL geometry of configs[1], [2] and [4] (one image; the oracle's forward takes seconds)
    against the fp32 oracle: generator activations within the north-star bar."""
    from hip_util import t, n, rel, precision, load_recipe
    from dtgan_amd import networks as Nw
    from oracle import recipe
    S, nc, nb = {"cfg2": (128, 3, 6), "cfg3": (256, 3, 9), "cfg5": (512, 1, 9)}[cfg]
    A, B, z = recipe.inputs(80, 1, nc, nc, S, 16)
    ref_A, ref_B = _oracle_generators(cfg, S, nc, nb)
    with precision(prec):
        gB = load_recipe(Nw.define_G(nc, nc, 32, gpu_ids=[0], n_blocks=nb), "netG_B_A", 4, "init")
        gA = load_recipe(Nw.define_stochastic_G(16, nc, nc, 32, gpu_ids=[0], n_blocks=nb), "netG_A_B", 4, "init")
        with torch.no_grad():
            fake_A, fake_B = n(gB(t(B))), n(gA(t(A), t(z)))
    tol = 1e-4 if prec == "f32" else 1e-3
    assert rel(fake_A, ref_A) < tol, rel(fake_A, ref_A)
    assert rel(fake_B, ref_B) < tol, rel(fake_B, ref_B)
