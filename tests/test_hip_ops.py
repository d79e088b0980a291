"""GPU parity, op level: every HIP kernel family reached through the product's module API and the
C ABI, compared with the fp64 oracle (oracle/ops.py) on the same seeded inputs.

Arithmetic: the convolution tests run each case three times — `mfma-bf16x3` (the product default: split-bf16 products,
16-bit operand mantissas, ~4e-6 rms per convolution), `mfma-f32` (the strict mode: an exact fp32 FMA chain on
v_mfma_f32_32x32x2_f32) and `direct` (the naive cross-check kernels).  All three are held to the same bounds, far inside
the north star's 1e-3 on activations: 2e-5 relative to the fp64 oracle (max-abs error / max-abs value), 1e-4 on long
pixel reductions (weight gradients, norm statistics).  The other tests run in the process default (bf16x3); norms,
losses and Adam are fp32 kernels in every mode.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import ops as oops  # noqa: E402
from oracle.tape import T, backward, leaf  # noqa: E402

CONV_CASES = [
    # K, stride, pad, mode, Ci, Co, N, H, W
    (3, 1, 1, "zero", 8, 8, 2, 12, 10),
    (3, 1, 1, "reflect", 32, 32, 2, 9, 11),
    (7, 1, 3, "reflect", 3, 8, 2, 16, 16),
    (7, 1, 3, "zero", 8, 3, 2, 16, 16),
    (3, 2, 1, "zero", 16, 32, 2, 16, 16),
    (3, 2, 1, "zero", 16, 32, 1, 15, 13),
    (4, 2, 1, "zero", 3, 16, 2, 16, 16),
    (4, 1, 1, "zero", 32, 64, 2, 9, 9),
    (4, 1, 0, "zero", 32, 1, 3, 6, 6),
    (1, 1, 0, "zero", 64, 16, 2, 3, 3),
    (3, 1, 1, "reflect", 128, 128, 1, 8, 8),
    (4, 1, 1, "zero", 256, 256, 1, 6, 6),
    (3, 2, 1, "zero", 64, 128, 1, 12, 12),
    (3, 1, 1, "zero", 6, 40, 1, 20, 20),
    # 3x3 stride-1 128-wide layers: the row-taps weight-gradient kernel (32-pixel row segments, partial tails)
    (3, 1, 1, "zero", 128, 128, 2, 9, 37),
    (3, 1, 1, "reflect", 64, 128, 1, 6, 40),
    (3, 1, 0, "zero", 128, 256, 1, 10, 34),
    # 32 <-> 3 channel layers: the patch kernel (8x16 output tiles, partial tiles, reflect / zero halo), forward and
    # as the data gradient of the 3 -> 32 stem
    (7, 1, 3, "reflect", 32, 3, 2, 20, 37),
    (7, 1, 3, "zero", 32, 3, 1, 9, 18),
    (7, 1, 3, "reflect", 3, 32, 2, 18, 21),
    (3, 1, 1, "zero", 32, 3, 1, 10, 10),
    # row-patch variant of the wave-specialised kernel: output width 16 / 32 / 64 / 128 (R = 8 / 4 / 2 / 1 rows per tile)
    (3, 1, 1, "reflect", 128, 128, 1, 16, 16),
    (3, 1, 1, "zero", 128, 128, 2, 8, 32),
    (3, 1, 1, "reflect", 64, 128, 1, 4, 64),
    (4, 1, 1, "zero", 128, 128, 1, 17, 17),
    (3, 1, 1, "reflect", 128, 256, 1, 2, 128),
    (4, 1, 1, "zero", 128, 128, 2, 16, 32),   # its DATA gradient (grid = the 16x32 input, taps with descending dx)
    (3, 1, 1, "zero", 128, 128, 3, 9, 13),    # 13-wide rows: ~11 segments per tile, tiles straddling images
    (3, 1, 1, "reflect", 160, 128, 2, 11, 130),  # 130-wide rows (the padded data-gradient width), Cin = 5 x 32
    # kernel-row weight gradient (conv_wgrad_tr.hip): 128-multiple channels, width a multiple of the 32-pixel run
    (3, 1, 1, "reflect", 128, 128, 2, 6, 32),
    (3, 1, 1, "zero", 128, 256, 1, 5, 64),
    (3, 1, 1, "zero", 256, 128, 3, 4, 32),
    (3, 1, 1, "reflect", 128, 128, 1, 40, 96),   # several splits, runs that cross image rows within a split
    # its 32 <-> 64 channel variant (128-pixel runs, waves split the pixels)
    (3, 1, 1, "zero", 32, 64, 1, 3, 128),
    (3, 1, 1, "reflect", 64, 32, 2, 4, 128),
    (3, 1, 1, "zero", 64, 32, 2, 20, 128),       # several splits
    (3, 1, 1, "reflect", 32, 64, 1, 5, 256),
    # persistent row pipeline (conv_rows.hip): zero-padded 3x3, 32 gathered -> 64 written channels, 128-pixel bands; forward of a
    # 32 -> 64 layer (two bands, two row chunks per band, an odd chunk) and the data gradient of a 64 -> 32 layer
    (3, 1, 1, "zero", 32, 64, 2, 21, 256),
    (3, 1, 1, "zero", 64, 32, 1, 17, 128),
]


def _conv_module(K, stride, pad, mode, Ci, Co):
    from dtgan_amd import modules as M
    if mode == "reflect":
        return M.Sequential(nn.ReflectionPad2d(pad), M.Conv2d(Ci, Co, K, stride=stride, padding=0, bias=True)).cuda()
    return M.Sequential(M.Conv2d(Ci, Co, K, stride=stride, padding=pad, bias=True)).cuda()


class _conv_mode(object):
    """'mfma-bf16x3' | 'mfma-f32' | 'direct' -> implementation + arithmetic for the duration of a test"""

    def __init__(self, mode):
        self.impl, _, self.prec = mode.partition("-")

    def __enter__(self):
        from dtgan_amd import ops
        self.before = ops.get_precision()
        if self.prec:
            ops.set_precision(self.prec)
        ops.set_conv_impl(self.impl)

    def __exit__(self, *a):
        from dtgan_amd import ops
        ops.set_conv_impl("mfma")
        ops.set_precision(self.before)


CONV_MODES = ["mfma-bf16x3", "mfma-f32", "direct"]


@pytest.mark.parametrize("impl", CONV_MODES)
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k%ds%dp%d%s_%dto%d_%dx%dx%d" % c)
def test_conv2d_fwd_bwd(case, impl):
    with _conv_mode(impl):
        _conv2d_fwd_bwd(case, check_kernels=impl == "mfma-bf16x3")


# Kernels the rows above are ABOUT (bf16x3 MFMA mode): {case: (forward, data gradient, weight gradient)}, each a substring of
# the kernel name the dispatcher reports (acg_last_kernel) or None.  A row whose dispatch predicate silently stops matching
# would otherwise turn into a second test of the generic tile.
CONV_KERNELS = {
    # thin layers (C4 tensors): N-packed / row-packed patch kernels, persistent thin weight gradient
    (4, 1, 0, "zero", 32, 1, 3, 6, 6): ("conv_patchn_x3", "conv_thinrow_x3", None),
    (7, 1, 3, "reflect", 32, 3, 2, 20, 37): ("conv_patchn_x3<REFLECT=1> (7x7", "conv_thinrow_x3", None),
    (7, 1, 3, "zero", 32, 3, 1, 9, 18): ("conv_patchn_x3<REFLECT=0> (7x7", "conv_thinrow_x3", "wgrad_thin_patch_x3<K=7,flip=1>"),
    (7, 1, 3, "reflect", 3, 32, 2, 18, 21): ("conv_thinrow_x3<REFLECT=1> (7x7", "conv_patchn_x3", "wgrad_thin_patch_x3<K=7,flip=0>"),
    (3, 1, 1, "zero", 32, 3, 1, 10, 10): ("conv_patchn_x3<REFLECT=0> (3x3", "conv_thinrow_x3", "wgrad_thin_patch_x3<K=3,flip=1>"),
    # wave-specialised kernel, row-patch stages
    (3, 1, 1, "reflect", 128, 128, 1, 16, 16): ("igemm_conv_x3_ws<REFLECT=1,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", None),
    (3, 1, 1, "zero", 128, 128, 2, 8, 32): ("igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "wgrad_x3_krow"),
    (3, 1, 1, "reflect", 64, 128, 1, 4, 64): ("igemm_conv_x3_ws<REFLECT=1,STATS=0,ROWP=1>", None, "wgrad_bf16<64,128,SPLIT=1,NT=3>"),
    (4, 1, 1, "zero", 128, 128, 1, 17, 17): ("igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "wgrad_x3_krowg<NT=4,IS=1,BCI=128>"),
    (3, 1, 1, "reflect", 128, 256, 1, 2, 128): ("igemm_conv_x3_ws<REFLECT=1,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "wgrad_x3_krow"),
    (4, 1, 1, "zero", 128, 128, 2, 16, 32): ("igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "wgrad_x3_krowg<NT=4,IS=1,BCI=128>"),
    (3, 1, 1, "zero", 128, 128, 3, 9, 13): ("igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", None),
    (3, 1, 1, "reflect", 160, 128, 2, 11, 130): ("igemm_conv_x3_ws<REFLECT=1,STATS=0,ROWP=1>", "igemm_conv_x3_ws<REFLECT=0,STATS=0,ROWP=1>", None),
    # kernel-row weight gradients
    (3, 1, 1, "reflect", 128, 128, 2, 6, 32): (None, None, "wgrad_x3_krow"),
    (3, 1, 1, "zero", 128, 256, 1, 5, 64): (None, None, "wgrad_x3_krow"),
    (3, 1, 1, "zero", 256, 128, 3, 4, 32): (None, None, "wgrad_x3_krow"),
    (3, 1, 1, "reflect", 128, 128, 1, 40, 96): (None, None, "wgrad_x3_krow"),
    (3, 1, 1, "reflect", 64, 32, 2, 4, 128): ("RP=1", None, "wgrad_x3_krow_s<64,32>"),
    (3, 1, 1, "reflect", 32, 64, 1, 5, 256): ("RP=1", None, "wgrad_x3_krow_s<32,64>"),
    # persistent row pipeline (32 gathered -> 64 written channels) and the generic row-patch tile on its mirror shape
    (3, 1, 1, "zero", 32, 64, 1, 3, 128): ("conv_rows_x3<32,64>", "RP=1", "wgrad_x3_krow_s<32,64>"),
    (3, 1, 1, "zero", 64, 32, 2, 20, 128): ("RP=1", "conv_rows_x3<32,64>", "wgrad_x3_krow_s<64,32>"),
    (3, 1, 1, "zero", 32, 64, 2, 21, 256): ("conv_rows_x3<32,64>", "RP=1", "wgrad_x3_krow_s<32,64>"),
    (3, 1, 1, "zero", 64, 32, 1, 17, 128): ("RP=1", "conv_rows_x3<32,64>", "wgrad_x3_krow_s<64,32>"),
}
assert all(c in CONV_CASES for c in CONV_KERNELS)


def _conv2d_fwd_bwd(case, check_kernels=False):
    from hip_util import t, n, rel, Spy
    import os
    K, stride, pad, mode, Ci, Co, N, H, W = case
    rs = np.random.RandomState(sum(c if isinstance(c, int) else len(c) for c in case))
    x = rs.normal(0, 1, (N, Ci, H, W))
    w = rs.normal(0, 0.3, (Co, Ci, K, K))
    b = rs.normal(0, 0.5, (Co,))
    m = _conv_module(K, stride, pad, mode, Ci, Co)
    conv = [c for c in m.modules() if c.__class__.__name__ == "Conv2d"][0]
    with torch.no_grad():
        conv.weight.copy_(t(w)); conv.bias.copy_(t(b))
    xt = t(x, grad=True)
    X, Wt, Bt = leaf(x), leaf(w), leaf(b)
    yo = oops.conv2d(X, Wt, Bt, stride=stride, pad=pad, pad_mode=mode)
    r = rs.normal(0, 1, yo.v.shape)
    with Spy() as spy:
        y = m(xt)
        y.backward(t(r))
    assert y.shape == yo.v.shape
    assert rel(n(y), yo.v) < 2e-5
    backward(yo, seed=r)
    if check_kernels:
        got = (spy.kernels("acg_conv2d_fwd"), spy.kernels("acg_conv2d_bwd_data"), spy.kernels("acg_conv2d_bwd_weight"))
        if os.environ.get("ACG_PRINT_KERNELS"):
            print("KERNELS", case, got)
        for want, have, what in zip(CONV_KERNELS.get(case, (None, None, None)), got, ("forward", "data gradient", "weight gradient")):
            assert want is None or any(want in k for k in have), (what, want, have)
    assert rel(n(xt.grad), X.g) < 2e-5, "dgrad"
    assert rel(n(conv.weight.grad), Wt.g) < 1e-4, "wgrad"
    assert rel(n(conv.bias.grad), Bt.g) < 1e-4, "bias grad"


@pytest.mark.parametrize("impl", CONV_MODES)
@pytest.mark.parametrize("dims", [(32, 16, 2, 5, 7), (128, 64, 1, 6, 6), (8, 8, 2, 4, 4), (128, 64, 1, 8, 64)])
def test_conv_transpose2d(dims, impl):
    with _conv_mode(impl):
        _conv_transpose2d(dims)


def _conv_transpose2d(dims):
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    Ci, Co, N, H, W = dims
    rs = np.random.RandomState(7 + Ci)
    x = rs.normal(0, 1, (N, Ci, H, W)); w = rs.normal(0, 0.3, (Ci, Co, 3, 3)); b = rs.normal(0, 0.5, (Co,))
    m = M.ConvTranspose2d(Ci, Co, 3, stride=2, padding=1, output_padding=1, bias=True).cuda()
    with torch.no_grad():
        m.weight.copy_(t(w)); m.bias.copy_(t(b))
    xt = t(x, grad=True)
    y = m(xt)
    X, Wt, Bt = leaf(x), leaf(w), leaf(b)
    yo = oops.conv_transpose2d(X, Wt, Bt)
    assert y.shape == yo.v.shape == (N, Co, 2 * H, 2 * W)
    assert rel(n(y), yo.v) < 2e-5
    r = rs.normal(0, 1, yo.v.shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 2e-5
    assert rel(n(m.weight.grad), Wt.g) < 1e-4
    assert rel(n(m.bias.grad), Bt.g) < 1e-4


@pytest.mark.parametrize("shape", [(2, 8, 9, 7), (3, 32, 16, 16), (1, 128, 40, 40), (2, 20, 5, 5)])
def test_instance_norm(shape):
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    N, C, H, W = shape
    rs = np.random.RandomState(C)
    x = rs.normal(0.7, 1.5, shape); sc = rs.normal(1, 0.3, C); sh = rs.normal(0, 0.3, C)
    m = M.InstanceNorm(C).cuda()
    with torch.no_grad():
        m.scale.copy_(t(sc)); m.shift.copy_(t(sh))
    xt = t(x, grad=True)
    y = m(xt)
    X, S, B = leaf(x), leaf(sc), leaf(sh)
    yo = oops.instance_norm(X, S, B)
    assert rel(n(y), yo.v) < 2e-5
    r = rs.normal(0, 1, shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-4
    assert rel(n(m.scale.grad), S.g) < 1e-4
    assert rel(n(m.shift.grad), B.g) < 1e-4


def test_instance_norm_large_mean_is_stable():
    """|mean| >> std: the one-pass statistics (Chan merge) must still match the two-pass oracle"""
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    rs = np.random.RandomState(3)
    x = rs.normal(50.0, 1.0, (2, 16, 64, 64))
    m = M.InstanceNorm(16).cuda()
    with torch.no_grad():
        m.scale.fill_(1.0); m.shift.fill_(0.0)
    y = m(t(x))
    yo = oops.instance_norm(leaf(x), leaf(np.ones(16)), leaf(np.zeros(16)))
    assert rel(n(y), yo.v) < 1e-3


@pytest.mark.parametrize("shape", [(2, 8, 9, 7), (3, 32, 12, 12)])
def test_cond_instance_norm(shape):
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    N, C, H, W = shape
    nl = 4
    rs = np.random.RandomState(C + 1)
    x = rs.normal(0.2, 1.3, shape); z = rs.normal(0, 1, (N, nl, 1, 1))
    ws, bs_ = rs.normal(0, 0.5, (C, nl, 1, 1)), rs.normal(0.3, 0.3, C)
    wc, bc = rs.normal(0, 0.5, (C, nl, 1, 1)), rs.normal(0.5, 0.3, C)
    m = M.CondInstanceNorm(C, nl).cuda()
    with torch.no_grad():
        m.shift_conv[0].weight.copy_(t(ws)); m.shift_conv[0].bias.copy_(t(bs_))
        m.scale_conv[0].weight.copy_(t(wc)); m.scale_conv[0].bias.copy_(t(bc))
    xt, zt = t(x, grad=True), t(z, grad=True)
    y = m(xt, zt)
    X, Z = leaf(x), leaf(z)
    Ws, Bs, Wc, Bc = leaf(ws), leaf(bs_), leaf(wc), leaf(bc)
    sh = oops.relu(oops.conv2d(Z, Ws, Bs)); sc = oops.relu(oops.conv2d(Z, Wc, Bc))
    yo = oops.cond_instance_norm(X, sc, sh)
    assert rel(n(y), yo.v) < 2e-5
    r = rs.normal(0, 1, shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-4
    assert rel(n(zt.grad), Z.g) < 1e-4
    assert rel(n(m.shift_conv[0].weight.grad), Ws.g) < 1e-4
    assert rel(n(m.scale_conv[0].weight.grad), Wc.g) < 1e-4
    assert rel(n(m.scale_conv[0].bias.grad), Bc.g) < 1e-4


def test_batch_norm2d_train():
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    shape = (3, 24, 7, 5)
    rs = np.random.RandomState(5)
    x = rs.normal(0.4, 1.2, shape); w = rs.normal(1, 0.3, 24); b = rs.normal(0, 0.3, 24)
    m = M.BatchNorm2d(24).cuda()
    with torch.no_grad():
        m.weight.copy_(t(w)); m.bias.copy_(t(b))
    xt = t(x, grad=True)
    y = m(xt)
    X, Wt, Bt = leaf(x), leaf(w), leaf(b)
    stats = dict(running_mean=np.zeros(24), running_var=np.ones(24), num_batches_tracked=np.zeros((), np.int64))
    yo = oops.batch_norm(X, Wt, Bt, stats, True)
    assert rel(n(y), yo.v) < 2e-5
    assert rel(n(m.running_mean), stats["running_mean"]) < 1e-5
    assert rel(n(m.running_var), stats["running_var"]) < 1e-5
    assert int(m.num_batches_tracked) == 1
    r = rs.normal(0, 1, shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-4
    assert rel(n(m.weight.grad), Wt.g) < 1e-4
    assert rel(n(m.bias.grad), Bt.g) < 1e-4


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_conv_epilogue_statistics_feed_instance_norm(prec):
    """128-wide 3x3 reflect conv -> InstanceNorm -> ReLU (the resblock pattern at bench scale): in bf16x3 the conv
    epilogue emits the norm's per-128-pixel-tile (mean, M2) and the norm only merges them — same result as the oracle,
    and the fused entry point must really have been taken (strict f32 has no such kernel: separate statistics pass)."""
    from hip_util import t, n, rel, precision
    from dtgan_amd import modules as M, ops, _lib
    N, C, H, W = 2, 128, 16, 24  # 384 pixels per image = 3 tiles
    rs = np.random.RandomState(11)
    x = rs.normal(0.3, 1.0, (N, C, H, W)); w = rs.normal(0, 0.05, (C, C, 3, 3)); b = rs.normal(0.5, 0.5, C)
    sc = rs.normal(1, 0.3, C); sh = rs.normal(0, 0.3, C)
    with precision(prec):
        m = M.Sequential(nn.ReflectionPad2d(1), M.Conv2d(C, C, 3, padding=0, bias=True), M.InstanceNorm(C), nn.ReLU(True)).cuda()
        with torch.no_grad():
            m[1].weight.copy_(t(w)); m[1].bias.copy_(t(b)); m[2].scale.copy_(t(sc)); m[2].shift.copy_(t(sh))
        calls = []
        real = _lib.call
        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        _lib.call = spy
        try:
            xt = t(x, grad=True)
            y = m(xt)
        finally:
            _lib.call = real
        fused = "acg_conv2d_fwd_stats" in calls and "acg_norm_stats_from_partials" in calls
        assert fused == (prec == "bf16x3"), calls
        X, Wt, Bt, S, Sh = leaf(x), leaf(w), leaf(b), leaf(sc), leaf(sh)
        yo = oops.relu(oops.instance_norm(oops.conv2d(X, Wt, Bt, stride=1, pad=1, pad_mode="reflect"), S, Sh))
        assert rel(n(y), yo.v) < 5e-5
        r = rs.normal(0, 1, yo.v.shape)
        y.backward(t(r)); backward(yo, seed=r)
        assert rel(n(xt.grad), X.g) < 2e-4
        assert rel(n(m[1].weight.grad), Wt.g) < 2e-4
        assert rel(n(m[2].scale.grad), S.g) < 2e-4


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_resnet_block_128_fused_paths(prec):
    """A 128-wide ResnetBlock (the bench-scale block) against the oracle.  In bf16x3 its backward must take the fused
    routes: the skip gradient is added inside the first convolution's data-gradient epilogue
    (acg_conv2d_bwd_data_add: frame path of the reflect fold) and the second convolution feeds the InstanceNorm's
    statistics from its epilogue; strict f32 takes the separate passes.  Same numbers either way."""
    from hip_util import t, n, rel, precision
    from dtgan_amd import modules as M, _lib
    N, C, H, W = 2, 128, 16, 24
    rs = np.random.RandomState(5)
    x = rs.normal(0.1, 1.0, (N, C, H, W))
    w1 = rs.normal(0, 0.04, (C, C, 3, 3)); b1 = rs.normal(0, 0.3, C)
    w2 = rs.normal(0, 0.04, (C, C, 3, 3)); b2 = rs.normal(0, 0.3, C)
    sc = rs.normal(1, 0.3, C); sh = rs.normal(0, 0.3, C)
    with precision(prec):
        blk = M.ResnetBlock(C, "reflect", M.InstanceNorm, False, True).cuda()
        cb = blk.conv_block
        with torch.no_grad():
            cb[1].weight.copy_(t(w1)); cb[1].bias.copy_(t(b1)); cb[4].weight.copy_(t(w2)); cb[4].bias.copy_(t(b2))
            cb[5].scale.copy_(t(sc)); cb[5].shift.copy_(t(sh))
        calls = []
        real = _lib.call
        def spy(name, *a):
            calls.append(name)
            return real(name, *a)
        _lib.call = spy
        try:
            xt = t(x, grad=True)
            y = blk(xt * 1.0)  # a non-leaf input, as inside a generator
            r = rs.normal(0, 1, y.shape)
            y.backward(t(r))
        finally:
            _lib.call = real
        fused = "acg_conv2d_bwd_data_add" in calls and "acg_conv2d_fwd_stats" in calls
        assert fused == (prec == "bf16x3"), sorted(set(calls))
        # bf16x3 also: the first convolution's ReLU is undone in the second one's data-gradient epilogue (no act_bwd pass)
        assert ("acg_conv2d_bwd_data_relu" in calls and "acg_act_bwd" not in calls) == (prec == "bf16x3"), sorted(set(calls))
        X, W1, B1, W2, B2, S, Sh = (leaf(a) for a in (x, w1, b1, w2, b2, sc, sh))
        h = oops.relu(oops.conv2d(X, W1, B1, stride=1, pad=1, pad_mode="reflect"))
        yo = oops.relu(oops.add(X, oops.instance_norm(oops.conv2d(h, W2, B2, stride=1, pad=1, pad_mode="reflect"), S, Sh)))
        assert rel(n(y), yo.v) < 5e-5
        backward(yo, seed=r)
        assert rel(n(xt.grad), X.g) < 2e-4
        assert rel(n(cb[1].weight.grad), W1.g) < 2e-4
        assert rel(n(cb[4].weight.grad), W2.g) < 2e-4


@pytest.mark.parametrize("switch", ["RELU_LINK", "LAZY_DRES", "NORM_SIGN_MASK", "DIRECT_GRAD"])
def test_backward_fusions_are_bit_identical_to_the_separate_passes(switch):
    """Each fused route of a ResnetBlock's backward — ReLU mask in the data-gradient epilogue, the skip gradient formed from
    dy and the sign bitmask inside the first convolution's epilogue, the bitmask itself, parameter gradients added straight
    into .grad — computes the same values in the same order as the separate pass it replaces: bit-identical results."""
    from hip_util import t, n
    from dtgan_amd import modules as M, ops
    from dtgan_amd.model import FlatNet
    N, C, H, W = 2, 128, 16, 32
    rs = np.random.RandomState(9)
    x, r = rs.normal(0.1, 1.0, (N, C, H, W)).astype(np.float32), rs.normal(0, 1, (N, C, H, W)).astype(np.float32)
    torch.manual_seed(3)
    blks = [M.ResnetBlock(C, "reflect", M.InstanceNorm, False, True).cuda() for _ in range(2)]
    net = torch.nn.Sequential(*blks)
    with torch.no_grad():
        for b in blks:
            b.conv_block[5].scale.normal_(1.0, 0.3); b.conv_block[5].shift.normal_(0, 0.3)
    flat = FlatNet(net)   # parameters / .grad as views of flat buffers, as inside the model
    res = {}
    for on in (True, False):
        setattr(ops, switch, on)
        try:
            flat.zero_grad()
            xt = t(x, grad=True)
            h = ops.ToNHWC.apply(xt * 1.0)
            for b in blks:
                h = b.forward_nhwc(h)
            ops.ToNCHW.apply(h, C).backward(t(r))
            res[on] = [n(xt.grad), n(flat.gv)]
        finally:
            setattr(ops, switch, True)
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    assert np.abs(res[True][1]).max() > 0


@pytest.mark.parametrize("N,nlatent,ndf,flat", [(4, 16, 64, False), (32, 16, 64, True), (64, 16, 64, True), (96, 16, 64, True), (7, 8, 32, False),
                                                (1024, 16, 64, False)])
def test_fused_latent_discriminator_equals_the_layer_chain(N, nlatent, ndf, flat):
    """DiscriminatorLatent (networks.py:396-433) as one launch per direction against the Linear / BatchNorm1d / LeakyReLU
    launches it replaces: output, input gradient, every parameter gradient (as autograd tensors and added into a FlatNet's
    .grad), running statistics and num_batches_tracked.  fp32 sums in a different order: 2e-5.  N = 96 and 1024 are beyond
    the fused kernel's 4096 activations per layer and must take the layer chain by themselves."""
    from hip_util import t, n, rel
    from dtgan_amd import networks, ops, _lib
    from dtgan_amd.model import FlatNet
    rs = np.random.RandomState(N)
    z, r = rs.normal(0, 1, (N, nlatent)).astype(np.float32), rs.normal(0, 1, (N, 1)).astype(np.float32)
    torch.manual_seed(5)
    net = networks.DiscriminatorLatent(nlatent, ndf).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.normal_(1.0 if p.dim() == 1 and "weight" in k else 0.0, 0.4)
    fn = FlatNet(net) if flat else None
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    res, used = {}, {}
    for fused in (True, False):
        ops.LATENT_MLP = fused
        try:
            net.load_state_dict(state0)
            networks.mark_dirty(net)
            net.zero_grad() if fn is None else fn.zero_grad()
            zt = t(z, grad=True)
            calls = []
            orig = ops._lib.call
            ops._lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
            try:
                for _ in range(2):      # twice: gradients accumulate, running buffers move twice
                    y = net(zt)
                    y.backward(t(r))
            finally:
                ops._lib.call = orig
            used[fused] = set(calls)
            grads = n(fn.gv) if fn is not None else np.concatenate([n(p.grad).ravel() for p in net.parameters()])
            res[fused] = [n(y), n(zt.grad), grads] + [n(v.float()) for k, v in net.state_dict().items() if "running" in k or "num_batches" in k]
        finally:
            ops.LATENT_MLP = True
    fits = bool(_lib.query("acg_latent_mlp_supported", N, nlatent, ndf))   # 256 % H == 0 and N * H <= 4096
    assert fits == (256 % ndf == 0 and N * ndf <= 4096)
    assert ("acg_latent_mlp_fwd" in used[True]) == fits and ("acg_latent_mlp_bwd" in used[True]) == fits
    assert "acg_latent_mlp_fwd" not in used[False]
    if fits:
        assert "acg_linear_fwd" not in used[True] and "acg_norm_bwd" not in used[True]
    assert tuple(res[True][0].shape) == (N, 1)
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape and rel(a, b) < 2e-5, rel(a, b)
    assert np.abs(res[True][2]).max() > 0


def test_multi_tensor_clip_adam_is_bit_identical_to_the_per_network_calls():
    """acg_clip_adam_multi (one partial-sums, one final, one update launch for all networks of a phase; model.py:447-452,
    510-515) against acg_sumsq + acg_adam_step per network: same block counts and order of sums -> identical bits, over
    sizes from below one block to beyond the 1024-block cap, clipped and unclipped, three steps."""
    from hip_util import n
    from dtgan_amd import ops
    torch.manual_seed(2)
    sizes = [7, 2048 * 3 + 5, 2_500_000, 40_000]
    scales = [1.0, 30.0, 0.3, 1e-3]         # gradient norms on both sides of max_norm
    def make():
        bufs = []
        g = torch.Generator(device="cuda").manual_seed(4)
        for sz, sc in zip(sizes, scales):
            p = torch.randn(sz, device="cuda", generator=g)
            gr = torch.randn(sz, device="cuda", generator=g) * sc
            bufs.append([p, gr, torch.zeros(sz, device="cuda"), torch.zeros(sz, device="cuda"), torch.zeros(1, device="cuda")])
        return bufs
    A, B = make(), make()
    for step in (1, 2, 3):
        for (p, gr, m, v, ss) in A:
            ops.sumsq(gr, ss)
        for (p, gr, m, v, ss) in A:
            ops.adam_step(p, gr, m, v, ss, 50.0, 2e-4, 0.5, 0.999, 1e-8, step)
        ops.clip_adam_multi([tuple(b) for b in B], 50.0, 2e-4, 0.5, 0.999, 1e-8, step)
        for a, b in zip(A, B):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        if step == 1:
            assert float(A[1][4]) > 50.0 ** 2 > float(A[3][4])      # one network clipped, one not
    with pytest.raises(Exception):
        ops.clip_adam_multi([tuple(B[0])] * 9, 50.0, 2e-4, 0.5, 0.999, 1e-8, 1)


def test_residual_norm_relu_fusion():
    """ResnetBlock tail: y = ReLU(x + IN(conv(...))) with the add + ReLU fused into the norm pass"""
    from hip_util import t, n, rel
    from dtgan_amd import modules as M
    import functools
    C, shape = 16, (2, 16, 10, 10)
    rs = np.random.RandomState(11)
    blk = M.ResnetBlock(C, "reflect", functools.partial(M.InstanceNorm2d, affine=True), False, True).cuda()
    vals = {}
    with torch.no_grad():
        for k, p in blk.named_parameters():
            vals[k] = rs.normal(1.0 if k.endswith("scale") else 0.0, 0.3, tuple(p.shape))
            p.copy_(t(vals[k]))
    x = rs.normal(0, 1, shape)
    xt = t(x, grad=True)
    y = blk(xt)
    X = leaf(x)
    P = {k: leaf(v) for k, v in vals.items()}
    o = oops.relu(oops.conv2d(X, P["conv_block.1.weight"], P["conv_block.1.bias"], pad=1, pad_mode="reflect"))
    o = oops.conv2d(o, P["conv_block.4.weight"], P["conv_block.4.bias"], pad=1, pad_mode="reflect")
    o = oops.instance_norm(o, P["conv_block.5.scale"], P["conv_block.5.shift"])
    yo = oops.relu(oops.add(X, o))
    assert rel(n(y), yo.v) < 2e-5
    r = rs.normal(0, 1, shape)
    y.backward(t(r)); backward(yo, seed=r)
    assert rel(n(xt.grad), X.g) < 1e-4
    gmax = max(float(np.max(np.abs(v.g))) for v in P.values())
    for k, p in blk.named_parameters():
        # conv_block.4.bias feeds the InstanceNorm: analytically zero gradient -> absolute floor
        assert np.max(np.abs(n(p.grad) - P[k].g)) < 2e-4 * np.max(np.abs(P[k].g)) + 2e-6 * gmax, k


@pytest.mark.parametrize("shape", [(2, 10, 10, 16), (3, 16, 24, 128), (2, 9, 7, 16), (1, 5, 5, 48)])
def test_norm_sign_bitmask_equals_reading_y(shape):
    """The backward of ReLU(res + IN(x)) needs only sign(y): the apply pass stores it as one bit per element and the two
    backward passes read that instead of y.  Same bits -> bit-identical gradients to the path that reads y; shapes whose
    float4 count per image is not a multiple of 8 (9x7x16, 5x5x48: 252, 300) fall back to reading y."""
    from hip_util import t, n
    from dtgan_amd import ops
    N, H, W, C = shape
    rs = np.random.RandomState(H * W + C)
    x, res, r = (rs.normal(0, 1, shape).astype(np.float32) for _ in range(3))
    gam, bet = rs.normal(1, 0.3, C).astype(np.float32), rs.normal(0, 0.3, C).astype(np.float32)
    out = {}
    for use_mask in (True, False):
        ops.NORM_SIGN_MASK = use_mask
        try:
            xt, rt, gt, bt = t(x, grad=True), t(res, grad=True), t(gam, grad=True), t(bet, grad=True)
            y = ops.NormAct.apply(xt, gt, bt, rt, "in", ops.ACT_RELU, 1e-5, gt.detach(), bt.detach(), None, None, 0.0)
            y.backward(t(r))
            out[use_mask] = [n(v) for v in (y, xt.grad, rt.grad, gt.grad, bt.grad)]
        finally:
            ops.NORM_SIGN_MASK = True
    for a, b in zip(out[True], out[False]):
        assert np.array_equal(a, b)
    # and against the formula
    mu = x.mean(axis=(1, 2), keepdims=True); var = x.var(axis=(1, 2), keepdims=True)
    yy = np.maximum(res + (x - mu) / np.sqrt(var + 1e-5) * gam + bet, 0)
    assert np.allclose(out[True][0], yy, rtol=1e-4, atol=1e-5)
    assert np.allclose(out[True][2], r * (yy > 0), rtol=0, atol=0)


def test_losses_and_optimizer():
    from hip_util import t, n
    from dtgan_amd import ops
    rs = np.random.RandomState(2)
    # LSGAN + L1 on C16 tensors with 3 valid channels
    a = np.zeros((2, 9, 9, 16), np.float32); b = np.zeros_like(a)
    a[..., :3] = rs.normal(0, 1, (2, 9, 9, 3)); b[..., :3] = rs.normal(0, 1, (2, 9, 9, 3))
    at, bt = t(a, grad=True), t(b, grad=True)
    l = ops.MseConst.apply(at, 3, 1.0)
    ref = ((a[..., :3] - 1.0) ** 2).mean()
    assert abs(float(l) - ref) < 1e-5 * abs(ref)
    (l * 0.5).backward()
    g = np.zeros_like(a); g[..., :3] = 0.5 * 2 * (a[..., :3] - 1.0) / a[..., :3].size
    assert np.allclose(n(at.grad), g, rtol=1e-5, atol=1e-9)
    at.grad = None
    l1 = ops.L1.apply(at, bt, 3)
    ref = np.abs(a[..., :3] - b[..., :3]).mean()
    assert abs(float(l1) - ref) < 1e-5 * ref
    l1.backward()
    g = np.zeros_like(a); g[..., :3] = np.sign(a[..., :3] - b[..., :3]) / a[..., :3].size
    assert np.allclose(n(at.grad), g, atol=1e-9) and np.allclose(n(bt.grad), -g, atol=1e-9)
    assert abs(float(ops.mean_valid(at, 3)) - a[..., :3].mean()) < 1e-6
    # clip + Adam, two steps, against the oracle's Adam
    from oracle.step import Adam, clip_grad_norm
    from oracle.tape import leaf
    p0 = rs.normal(0, 1, 1000).astype(np.float32)
    P = leaf(p0.copy()); opt = Adam([P], 2e-4, 0.5)
    pt = t(p0); gt = torch.zeros_like(pt); m = torch.zeros_like(pt); v = torch.zeros_like(pt)
    ss = torch.zeros((), device="cuda")
    for step in (1, 2):
        g = rs.normal(0, 30 if step == 1 else 0.1, 1000).astype(np.float32)
        P.g = g.copy(); norm = clip_grad_norm([P], 500.0); opt.step()
        gt.copy_(t(g)); ops.sumsq(gt, ss)
        ops.adam_step(pt, gt, m, v, ss, 500.0, 2e-4, 0.5, 0.999, 1e-8, step)
        assert abs(float(ss) ** 0.5 - norm) < 1e-4 * norm
        assert np.allclose(n(pt), P.v, rtol=0, atol=2e-7)
        assert np.allclose(n(gt), P.g, rtol=1e-5, atol=1e-8)


def test_linear_and_spatial_mean():
    from hip_util import t, n, rel
    from dtgan_amd import ops
    rs = np.random.RandomState(9)
    x = rs.normal(0, 1, (5, 16)); w = rs.normal(0, 0.5, (8, 6)); b = rs.normal(0, 0.5, 8)
    xt, wt, bt = t(x, True), t(w, True), t(b, True)
    y = ops.LinearFn.apply(xt, wt, bt, ops.ACT_LRELU, 8)
    pre = x[:, :6] @ w.T + b
    ref = np.where(pre > 0, pre, 0.2 * pre)
    assert rel(n(y), ref) < 1e-5
    r = rs.normal(0, 1, ref.shape)
    y.backward(t(r))
    g = r * np.where(pre > 0, 1, 0.2)
    gx = np.zeros_like(x); gx[:, :6] = g @ w
    assert rel(n(xt.grad), gx) < 1e-5 and rel(n(wt.grad), g.T @ x[:, :6]) < 1e-5 and rel(n(bt.grad), g.sum(0)) < 1e-5
    # wide layers (CondInstanceNorm's latent -> 128 channels, the latent MLP): the workgroup-parallel dx kernel
    for N_, I_, ld_, O_, act in ((32, 16, 16, 128, ops.ACT_RELU), (7, 64, 64, 64, ops.ACT_LRELU), (3, 5, 16, 200, ops.ACT_NONE)):
        x = rs.normal(0, 1, (N_, ld_)); w = rs.normal(0, 0.5, (O_, I_)); b = rs.normal(0, 0.5, O_)
        xt, wt, bt = t(x, True), t(w, True), t(b, True)
        y = ops.LinearFn.apply(xt, wt, bt, act, O_)
        pre = x[:, :I_] @ w.T + b
        slope = {ops.ACT_RELU: 0.0, ops.ACT_LRELU: 0.2, ops.ACT_NONE: 1.0}[act]
        ref = np.where(pre > 0, pre, slope * pre)
        assert rel(n(y)[:, :O_], ref) < 1e-5
        r = rs.normal(0, 1, n(y).shape); r[:, O_:] = 0
        y.backward(t(r))
        g = r[:, :O_] * np.where(pre > 0, 1, slope)
        gx = np.zeros_like(x); gx[:, :I_] = g @ w
        assert rel(n(xt.grad), gx) < 1e-5 and rel(n(wt.grad), g.T @ x[:, :I_]) < 1e-5 and rel(n(bt.grad), g.sum(0)) < 1e-5
    a = rs.normal(0, 1, (2, 5, 3, 16))
    at = t(a, True)
    sm = ops.SpatialMean.apply(at)
    assert rel(n(sm), a.mean(axis=(1, 2))) < 1e-5
    sm.backward(t(np.ones((2, 16))))
    assert np.allclose(n(at.grad), 1.0 / 15)


def test_c_abi_error_convention():
    """negative status + thread-local message instead of a fault: bad descriptor, unpadded channels, short workspace,
    unsupported geometry (reflect + stride 2)"""
    import ctypes
    from dtgan_amd import _lib, ops
    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 16, device="cuda"); y = torch.zeros(1, 8, 8, 16, device="cuda")
    w = torch.zeros(16 * 16 * 9 * 2, device="cuda")
    P, st = ops._ptr, ops._stream()
    bad = _lib.ConvDesc(1, 8, 8, 12, 8, 8, 16, 3, 1, 1, 0, 0, 0)          # Ci not padded to 16
    assert lib.acg_conv2d_fwd(ctypes.byref(bad), P(x), P(w), None, P(y), 0, st) == -1
    assert b"padded to 16" in lib.acg_last_error()
    bad = _lib.ConvDesc(1, 8, 8, 16, 7, 7, 16, 3, 1, 1, 0, 0, 0)          # inconsistent output size
    assert lib.acg_conv2d_fwd(ctypes.byref(bad), P(x), P(w), None, P(y), 0, st) == -1
    assert b"inconsistent" in lib.acg_last_error()
    bad = _lib.ConvDesc(1, 8, 8, 16, 4, 4, 16, 3, 2, 1, 1, 0, 0)          # reflect + stride 2
    assert lib.acg_conv2d_fwd(ctypes.byref(bad), P(x), P(w), None, P(y), 0, st) == -1
    ok = _lib.ConvDesc(1, 8, 8, 16, 8, 8, 16, 3, 1, 1, 1, 0, 0)           # reflect dgrad needs a workspace
    assert lib.acg_conv2d_bwd_data(ctypes.byref(ok), P(y), P(w), P(x), None, 0, st) == -2
    assert b"workspace" in lib.acg_last_error()
    assert lib.acg_conv2d_bwd_weight(ctypes.byref(ok), P(x), P(y), P(w), None, 16, 16, None, 0, 0, st) == -1
    assert lib.acg_norm_stats(P(x), 1, 64, 18, 1e-5, 0, P(y), P(y), None, None, 0.0, P(w), 1 << 20, st) == -1   # C % 4
    assert lib.acg_set_conv_precision(7) == -1 and lib.acg_set_conv_impl(9) == -1
    with pytest.raises(_lib.AcgError):
        _lib.call("acg_adam_step", P(x), P(x), P(x), P(x), 16, None, 1.0, 1e-3, 0.5, 0.999, 1e-8, 0, 0, st)   # step < 1
    torch.cuda.synchronize()


def test_stride2_phases_in_one_tile_match_the_fp32_kernels():
    """networks.py:168, 178-179: the stride-2 64 -> 128 convolution's data gradient and the 128 -> 64 ConvTranspose2d forward
    on maps whose phase-grid rows are whole tiles run all four sub-pixel phases in one tile (conv_ph4.hip, bf16x3); the strict
    fp32 kernels (four launches of the generic tile) are the reference: border rows / columns, bias and activation included."""
    import ctypes
    from hip_util import t, n, rel, precision
    from dtgan_amd import ops, _lib
    P = ops._ptr
    rs = np.random.RandomState(5)
    NB, H, W, Cs, Cl = 2, 6, 128, 128, 64         # small side H x W x 128, large side 2H x 2W x 64
    xs = t(rs.normal(0.1, 1, (NB, H, W, Cs)))     # small-side tensor (ConvTranspose input / gradient of the conv output)
    w = t(rs.normal(0, 0.1, (Cs, Cl, 3, 3))); b = t(rs.normal(0, 1, Cl))
    out = {}
    for prec in ("f32", "bf16x3"):
        with precision(prec):
            st = ops._stream()
            d = ops.conv_desc(NB, 2 * H, 2 * W, Cl, Cs, 3, 2, 1, 0, Cl, Cs)   # the Conv2d 64 -> 128, stride 2
            pk = ops.PackedConv(w, b, Cl, Cs)
            y = torch.empty((NB, 2 * H, 2 * W, Cl), device="cuda")
            _lib.call("acg_conv_transpose2d_fwd", ctypes.byref(d), P(xs), P(pk.wb), P(pk.bias), P(y), 1, st)   # + ReLU
            kern_fwd = _lib.query("acg_last_kernel").decode()
            dx = torch.empty((NB, 2 * H, 2 * W, Cl), device="cuda")
            nb = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
            ws = ops.workspace(max(nb, 1))
            _lib.call("acg_conv2d_bwd_data", ctypes.byref(d), P(xs), P(pk.wb), P(dx), P(ws), nb, st)
            out[prec] = (n(y), n(dx), kern_fwd, _lib.query("acg_last_kernel").decode())
    assert "ph4" in out["bf16x3"][2] and "ph4" in out["bf16x3"][3], out["bf16x3"][2:]
    assert rel(out["bf16x3"][0], out["f32"][0]) < 2e-5 and rel(out["bf16x3"][1], out["f32"][1]) < 2e-5


@pytest.mark.parametrize("case", [("conv", 32, 64, 3, 1, 32, 32), ("conv", 64, 32, 3, 1, 16, 64), ("conv", 16, 32, 3, 2, 32, 32),
                                  ("conv", 32, 32, 3, 1, 32, 32), ("conv", 16, 32, 3, 2, 64, 64), ("conv", 32, 32, 3, 1, 8, 16),
                                  ("conv", 64, 128, 3, 2, 32, 32), ("convT", 128, 64, 3, 2, 32, 32), ("convT", 64, 32, 3, 2, 16, 32),
                                  ("convT", 128, 64, 3, 2, 8, 128),   # four phases in one tile (conv_ph4.hip)
                                  ("stem", 3, 32, 7, 1, 24, 48),        # C4 image -> 32 channels, 8 x 16 pixel tiles (conv_thinrow_x3)
                                  ("conv0", 32, 64, 3, 1, 21, 256)])    # zero padding: the persistent row pipeline (conv_rows_x3)
def test_conv_epilogue_statistics_equal_the_statistics_pass(case):
    """Per-tile (mean, M2) from the convolution epilogues (generic bf16 tile, wave-specialised tile, the four phase launches
    of ConvTranspose2d) merged by acg_norm_stats_from_partials against acg_norm_stats on the stored output: the mean / rstd
    an InstanceNorm behind the convolution would use (modules.py:83-97)."""
    import ctypes
    from hip_util import t, n, rel, precision
    from dtgan_amd import ops, _lib
    kind, Ci, Co, K, stride, H, W = case
    P = ops._ptr
    with precision("bf16x3"):
        st = ops._stream()
        rs = np.random.RandomState(Ci + Co)
        NB = 4
        if kind == "stem":   # networks.py:159-160: ReflectionPad2d(3) + 7x7 on an image stored C4
            d = ops.conv_desc(NB, H, W, 4, Co, K, stride, 3, 1, Ci, Co)
            x = t(np.concatenate([rs.normal(0.3, 1, (NB, H, W, Ci)), np.zeros((NB, H, W, 4 - Ci))], -1))
            w = t(rs.normal(0, 0.2, (Co, Ci, K, K))); b = t(rs.normal(0, 1, Co))
            pk = ops.PackedConv(w, b, 16, Co)
            y = torch.empty((NB, d.Ho, d.Wo, Co), device="cuda")
            assert _lib.query("acg_conv2d_fwd_stats_supported", ctypes.byref(d))
            part = torch.zeros((NB, d.Ho * d.Wo // 128, 2, Co), device="cuda")
            _lib.call("acg_conv2d_fwd_stats", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y), P(part), st)
            assert _lib.query("acg_last_kernel").decode().startswith("conv_thinrow_x3")
            y2 = torch.empty_like(y)
            _lib.call("acg_conv2d_fwd", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y2), 0, st)
            C = Co
        elif kind in ("conv", "conv0"):
            d = ops.conv_desc(NB, H, W, Ci, Co, K, stride, 1, 1 if (stride == 1 and kind == "conv") else 0, Ci, Co)
            x = t(rs.normal(0.3, 1, (NB, H, W, Ci)))
            w = t(rs.normal(0, 0.2, (Co, Ci, K, K))); b = t(rs.normal(0, 1, Co))
            pk = ops.PackedConv(w, b, Ci, Co)
            y = torch.empty((NB, d.Ho, d.Wo, Co), device="cuda")
            assert _lib.query("acg_conv2d_fwd_stats_supported", ctypes.byref(d))
            part = torch.zeros((NB, d.Ho * d.Wo // 128, 2, Co), device="cuda")
            _lib.call("acg_conv2d_fwd_stats", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y), P(part), st)
            if kind == "conv0":
                assert _lib.query("acg_last_kernel").decode().startswith("conv_rows_x3")
            y2 = torch.empty_like(y)
            _lib.call("acg_conv2d_fwd", ctypes.byref(d), P(x), P(pk.wf), P(pk.bias), P(y2), 0, st)
            C = Co
        else:   # descriptor = the Conv2d (Co <- Ci ... ) the transposed convolution is the adjoint of: its input side is the output
            Cs, Cl = Ci, Co          # small-side (input of convT) channels, large-side (output) channels
            d = ops.conv_desc(NB, 2 * H, 2 * W, Cl, Cs, K, 2, 1, 0, Cl, Cs)
            x = t(rs.normal(0.3, 1, (NB, H, W, Cs)))
            w = t(rs.normal(0, 0.2, (Cs, Cl, K, K))); b = t(rs.normal(0, 1, Cl))
            pk = ops.PackedConv(w, b, Cl, Cs)
            y = torch.empty((NB, 2 * H, 2 * W, Cl), device="cuda")
            assert _lib.query("acg_conv_transpose2d_fwd_stats_supported", ctypes.byref(d))
            part = torch.zeros((NB, 4 * H * W // 128, 2, Cl), device="cuda")
            _lib.call("acg_conv_transpose2d_fwd_stats", ctypes.byref(d), P(x), P(pk.wb), P(pk.bias), P(y), P(part), st)
            y2 = torch.empty_like(y)
            _lib.call("acg_conv_transpose2d_fwd", ctypes.byref(d), P(x), P(pk.wb), P(pk.bias), P(y2), 0, st)
            C = Cl
        assert torch.equal(y, y2)
        Pn = y.shape[1] * y.shape[2]
        for unbiased in (0, 1):   # InstanceNorm / CondInstanceNorm
            m1, r1 = torch.empty(NB * C, device="cuda"), torch.empty(NB * C, device="cuda")
            m2, r2 = torch.empty(NB * C, device="cuda"), torch.empty(NB * C, device="cuda")
            _lib.call("acg_norm_stats_from_partials", P(part), NB, Pn, C, 128, 1e-5, unbiased, P(m1), P(r1), st)
            nb = _lib.query("acg_norm_workspace_bytes", NB, Pn, C)
            ws = ops.workspace(nb)
            _lib.call("acg_norm_stats", P(y), NB, Pn, C, 1e-5, unbiased, P(m2), P(r2), None, None, 0.0, P(ws), nb, st)
            yy = n(y).reshape(NB, Pn, C).astype(np.float64)
            assert rel(n(m2), yy.mean(1).ravel()) < 1e-5
            var = yy.var(1, ddof=unbiased).ravel()
            assert np.max(np.abs(n(r2) * np.sqrt(var + 1e-5) - 1)) < 1e-5
            sd = np.sqrt(var)
            print("stats vs fp64 (%s, unbiased %d): mean error / sigma: epilogue %.1e, pass %.1e; rstd rel error: epilogue %.1e, pass %.1e"
                  % (str(case), unbiased, np.max(np.abs(n(m1) - yy.mean(1).ravel()) / sd), np.max(np.abs(n(m2) - yy.mean(1).ravel()) / sd),
                     np.max(np.abs(n(r1) * np.sqrt(var + 1e-5) - 1)), np.max(np.abs(n(r2) * np.sqrt(var + 1e-5) - 1))))
            assert np.max(np.abs(n(m1) - n(m2))) < 1e-5 * np.max(np.abs(n(m2))), np.max(np.abs(n(m1) - n(m2)))
            assert np.max(np.abs(n(r1) / n(r2) - 1)) < 1e-5, np.max(np.abs(n(r1) / n(r2) - 1))
