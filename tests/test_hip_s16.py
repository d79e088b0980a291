"""Pre-split ("S16") activation storage of the residual trunk (include/acgan_hip.h acg_*_s16, ops.S16Plan): the kernels
against their fp32-operand twins through the C ABI (they consume the same bf16 hi / lo halves in the same order: bit for
bit), and whole generators with the trunk pre-split against the same generators with fp32 storage (what changes is the
skip connection, which now carries hi + lo instead of the fp32 value: 2^-17 relative per block).
Reference semantics: /root/reference/augmented_cyclegan/modules.py:139-235, networks.py:149-252."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_s16_kernels_equal_the_fp32_operand_kernels_bit_for_bit():
    """Forward, weight gradient and the padded-grid data gradient: bit for bit.  Maps whose rows are whole 128-pixel tiles
    take the un-padded data gradient (Geom.unpad: summed weight slabs for the two mirrored rows, the mirrored columns as a
    separate small GEMM — modules.py:205-227's ReflectionPad2d(1) adjoint in another summation order): 2e-5 of the largest
    element against the fp32-operand kernel, which test_conv2d_fwd_bwd holds to the oracle."""
    import s16_check
    from hip_util import precision
    with precision("bf16x3"):
        assert s16_check.run(2, 64, 64, 128, False, 0)      # two row segments per tile
        assert s16_check.run(1, 32, 32, 128, False, 0)      # four
        assert s16_check.run(1, 64, 128, 128, False, 0)     # un-padded data gradient: the smallest map (two column-term tiles)
        assert s16_check.run(2, 128, 128, 128, False, 0)    # the config-3 trunk geometry
        assert s16_check.run(1, 96, 256, 128, False, 0)     # two tiles per row


def _gen(kind, n_blocks, seed):
    from dtgan_amd import networks
    from hip_util import load_recipe
    if kind == "cin":
        net = networks.CINResnetGenerator(8, 3, 3, ngf=32, n_blocks=n_blocks, gpu_ids=[0]).cuda()
        return load_recipe(net, "G_A_B", seed, "init")
    net = networks.ResnetGenerator(3, 3, ngf=32, n_blocks=n_blocks, gpu_ids=[0]).cuda()
    return load_recipe(net, "G_B_A", seed, "init")


@pytest.mark.parametrize("kind", ["plain", "cin"])
def test_generator_with_presplit_trunk_matches_fp32_storage(kind):
    """64 x 64 input -> 32 x 32 x 128 trunk.  Forward images, input gradient and every parameter gradient, with the exact-fp32
    arithmetic as the neutral reference: gradients of these networks move at the 1e-3 level (norm-wise) under ANY 1e-5
    perturbation of the activations because a few ReLU masks flip (tools/conditioning_probe.py), so the pre-split run is
    held to the deviation the fp32-storage bf16x3 run itself shows against exact fp32; an indexing or fusion error would
    show at the 1e-1 level."""
    from dtgan_amd import ops
    from hip_util import precision, l2rel, rel
    g = torch.Generator().manual_seed(3)
    x = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1).cuda()
    z = torch.randn((2, 8, 1, 1), generator=g).cuda()
    w = torch.randn((2, 3, 64, 64), generator=g).cuda()
    out = {}
    for prec, on in (("f32", False), ("bf16x3", False), ("bf16x3", True)):
        with precision(prec):
            net = _gen(kind, 2, 5)
            ops.S16_ENABLED = on
            used = []
            real = ops._lib.call

            def spy(name, *a):
                used.append(name)
                return real(name, *a)
            ops._lib.call = spy
            try:
                xi = x.clone().requires_grad_(True)
                zi = z.clone().requires_grad_(True)
                y = net(xi, zi) if kind == "cin" else net(xi)
                (y * w).sum().backward()
            finally:
                ops._lib.call = real
                ops.S16_ENABLED = True
            grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
            out[(prec, on)] = (y.detach().cpu().numpy(), xi.grad.cpu().numpy(), grads, used)
    ref, plain, pre = out[("f32", False)], out[("bf16x3", False)], out[("bf16x3", True)]
    assert "acg_conv2d_fwd_s16" in pre[3] and "acg_conv2d_bwd_weight_s16" in pre[3] and "acg_conv2d_bwd_data_s16" in pre[3]
    assert "acg_s16_decode" not in pre[3]     # the last block's output norm writes fp32 for the layer behind the trunk
    assert not any(n.endswith("_s16") for n in plain[3])
    assert rel(pre[0], plain[0]) < 2e-5 and rel(pre[0], ref[0]) < 1e-3           # images: storage rounding only
    e_plain, e_pre = l2rel(plain[1], ref[1]), l2rel(pre[1], ref[1])
    print("input gradient vs exact fp32: fp32 storage %.2e, pre-split %.2e" % (e_plain, e_pre))
    assert e_pre < max(3 * e_plain, 2e-2), (e_plain, e_pre)
    worst = 0.0
    for k in ref[2]:
        a, b, c = pre[2][k], plain[2][k], ref[2][k]
        assert a.shape == c.shape
        # bias gradients of convolutions in front of an InstanceNorm are rounding noise around zero in every run
        if np.linalg.norm(c) > 1e-4 * np.sqrt(c.size):
            ea, eb = l2rel(a, c), l2rel(b, c)
            worst = max(worst, ea)
            assert ea < max(3 * eb, 2e-2), (k, ea, eb)
    print("worst parameter gradient vs exact fp32, pre-split: %.2e" % worst)


@pytest.mark.parametrize("kind", ["plain", "cin"])
def test_norm_backward_sums_from_the_data_gradient_epilogue(kind):
    """256 x 256 input -> 128 x 128 x 128 trunk: whole-row tiles, so the trunk's data gradients run on the un-padded grid and
    emit the backward sums of the norm in front of them (ops.NormSums, acg_conv2d_bwd_data_s16_sums).  Same generator, same
    input, with and without that fusion: the sums are the same numbers added in another order (modules.py:83-97, 121-131)."""
    from dtgan_amd import ops
    from hip_util import precision
    torch.manual_seed(3)
    xin = torch.randn(1, 3, 256, 256, device="cuda")
    z = torch.randn(1, 8, 1, 1, device="cuda")
    res = {}
    with precision("bf16x3"):
        for fused in (True, False):
            net = _gen(kind, 3, 5)
            x = xin.clone().requires_grad_(True)
            old, ops.NORM_SUMS = ops.NORM_SUMS, fused
            used0 = ops.NORM_SUMS_USED
            try:
                y = net(x, z) if kind == "cin" else net(x)
                (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
            finally:
                ops.NORM_SUMS = old
            used = ops.NORM_SUMS_USED - used0
            # the norm in front of the first block + per block: its output norm (all but the last block's) and, in a
            # CINResnetBlock, the conditional norm between its two convolutions; + the four norms in front of a full-resolution
            # convolution whose fp32-operand data gradient carries the sums (acg_conv2d_bwd_data_sums): the stem's (generic tile
            # of the 32 -> 64 layer), the one in front of the stride-2 layer (four-phase tile), the one behind the ConvTranspose
            # (row pipeline), the one in front of the head (thin-row kernel) — all but the row pipeline's from round 6
            assert used == ((1 + 2 + 3 + 4 if kind == "cin" else 1 + 2 + 4) if fused else 0), used
            res[fused] = (y.detach(), x.grad.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    ya, ga, pa = res[True]
    yb, gb, pb = res[False]
    assert torch.equal(ya, yb)
    assert (ga - gb).abs().max().item() <= 2e-5 * gb.abs().max().item()
    for k in pb:
        # (a convolution bias in front of a norm has the exact gradient 0: what both runs hold there is the rounding noise of
        # sums over 16 k pixels, allowed for against the weight gradient of the same layer)
        noise = 0.0
        if k.endswith(".bias") and k.replace(".bias", ".weight") in pb:
            # (measured up to 5.1e-4 of the layer's largest weight gradient: 1.22e-3 absolute on model.10.conv_block.4.bias,
            # a sum over 16 k pixels whose exact value is 0 — round 6, three more norms take their sums from a data gradient)
            noise = 7e-4 * pb[k.replace(".bias", ".weight")].abs().max().item()
        # (the scale / shift layers of a CondInstanceNorm sum ReLU-gated per-sample terms over a whole plane: the two runs'
        # differently ordered sums reach them with 2e-5 .. 3e-5; measured 2.0003e-5 on model.2.shift_conv)
        tol = 4e-5 if ("shift_conv" in k or "scale_conv" in k) else 2e-5
        assert (pa[k] - pb[k]).abs().max().item() <= tol * pb[k].abs().max().item() + noise + 1e-12, k


def test_cond_bank_matches_per_norm_layers():
    """CINResnetGenerator: the scale / shift layers of all CondInstanceNorms as one dense layer (ops.CondBankFn,
    modules.cond_bank; modules.py:104-132) against two small layers per norm.  The same dot products in the same order: the
    image and every parameter gradient agree bit for bit, the gradient w.r.t. z up to the order of its sum over layers."""
    from dtgan_amd import ops
    torch.manual_seed(11)
    xin = torch.randn(2, 3, 64, 64, device="cuda")
    zin = torch.randn(2, 8, 1, 1, device="cuda")
    res = {}
    for bank in (True, False):
        net = _gen("cin", 2, 7)
        x, z = xin.clone().requires_grad_(True), zin.clone().requires_grad_(True)
        old, ops.COND_BANK = ops.COND_BANK, bank
        try:
            y = net(x, z)
            (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        finally:
            ops.COND_BANK = old
        res[bank] = (y.detach(), x.grad.detach(), z.grad.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    ya, xa, za, pa = res[True]
    yb, xb, zb, pb = res[False]
    assert torch.equal(ya, yb) and torch.equal(xa, xb)
    assert (za - zb).abs().max().item() <= 1e-5 * zb.abs().max().item()
    for k in pb:
        assert torch.equal(pa[k], pb[k]), k


def test_presplit_trunk_in_eval_and_no_grad():
    from dtgan_amd import ops
    from hip_util import precision, rel
    x = (torch.rand((1, 3, 64, 64)) * 2 - 1).cuda()
    with precision("bf16x3"):
        net = _gen("plain", 1, 7)
        with torch.no_grad():
            a = net(x)
            ops.S16_ENABLED = False
            try:
                b = net(x)
            finally:
                ops.S16_ENABLED = True
    assert rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
