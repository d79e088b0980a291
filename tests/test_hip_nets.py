"""GPU parity, network level: each of the six networks (HIP path, public NCHW API) against the golden
fixtures produced from the reference itself (tests/golden/, tools/make_goldens.py): forward outputs,
input gradients, every parameter gradient, BatchNorm running statistics.

Bar: 1e-3 relative (BASELINE.json north star) on activations.  Exact-fp32 arithmetic ('f32'): 1e-4 on outputs /
input grads and 5e-4 (+ a noise floor for analytically-zero bias gradients) on parameter gradients.  Split-bf16
arithmetic ('bf16x3', operands carry 16 mantissa bits): outputs at the 1e-3 bar; gradients of these deliberately
ill-conditioned 'rich' fixtures are compared norm-wise, because a pre-activation within 1e-5 of zero can land on
the other side of a ReLU and change individual gradient entries discretely (the reference on another GPU, whose
cuDNN picks Winograd/FFT algorithms with ~1e-5 error, differs from its own CPU path in the same way).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, names  # noqa: E402


def build(meta):
    from dtgan_amd import networks as N
    c, n = meta["cfg"], meta["net"]
    g = [0]
    if n == "netG_B_A":   # --norm batch / --use_dropout fixtures carry the option in their cfg (options.py:64-65)
        return N.define_G(c["input_nc"], c["output_nc"], c["ngf"], norm=c.get("norm", "instance"),
                          use_dropout=c.get("use_dropout", False), gpu_ids=g, n_blocks=c["n_blocks"])
    if n == "netG_A_B":
        return N.define_stochastic_G(c["nlatent"], c["input_nc"], c["output_nc"], c["ngf"],
                                     use_dropout=c.get("use_dropout", False), gpu_ids=g, n_blocks=c["n_blocks"])
    if n == "netD_B":
        return N.define_D_B(c["input_nc"], c["ndf"], "basic", "instance", gpu_ids=g)
    if n == "netD_A":
        return N.define_D_A(c["input_nc"], c["ndf"], "basic", "instance", gpu_ids=g)
    if n == "netE_B":
        return N.define_E(c["nlatent"], c["input_nc"], c["nef"], "batch", gpu_ids=g)
    if n == "netD_z_B":
        return N.define_LAT_D(c["nlatent"], c["ndf"], gpu_ids=g)
    raise KeyError(n)


@pytest.mark.parametrize("impl", ["mfma", "direct", "mfma-bf16x3"])
@pytest.mark.parametrize("name", names("net"))
def test_net_matches_reference_golden(name, impl):
    from hip_util import precision, injected_dropout
    from dtgan_amd import ops
    arr, meta = load(name)
    if impl == "direct" and name not in ("G_A_B_s16_nb3", "D_B_s40", "E_B_s64"):
        pytest.skip("direct cross-check on a subset")
    x3 = impl.endswith("bf16x3")
    impl = impl.split("-")[0]
    ops.set_conv_impl(impl)
    try:
        with precision("bf16x3" if x3 else "f32"), injected_dropout(meta.get("drop_seed")):
            _check_net(arr, meta, x3, name)
    finally:
        ops.set_conv_impl("mfma")


def _check_net(arr, meta, x3, meta_name):
    from hip_util import t, n, rel, l2rel, load_recipe
    out_tol = 1e-3 if x3 else 1e-4
    net = load_recipe(build(meta), meta["net"], meta["seed"], meta["flavour"])
    net.train()
    ins, i = [], 0
    while "in%d" % i in arr:
        ins.append(t(arr["in%d" % i], grad=True)); i += 1
    out = net.forward(*ins)
    outs = list(out) if isinstance(out, tuple) else [out]
    for j, o in enumerate(outs):
        assert tuple(o.shape) == arr["out%d" % j].shape
        assert rel(n(o), arr["out%d" % j]) < out_tol, "forward out%d" % j
    loss = sum((o * t(arr["R%d" % j])).sum() for j, o in enumerate(outs))
    loss.backward()
    # tools/conditioning_probe.py: these two fixtures hold a ReLU input within 4e-6 of zero — the exact-fp32 path
    # itself moves its input gradient by 1.2e-2 (norm-wise) when the INPUT is perturbed by 4e-6 relative.  With 16-bit
    # operand mantissas that unit lands on the other side, so their gradients are only compared as one vector.
    kinked = x3 and meta_name in ("G_A_B_s32_nc1_nb3", "G_B_A_s32_nc1_nb3")
    for j, x in enumerate(ins):
        if x3:
            assert l2rel(n(x.grad), arr["gin%d" % j]) < (3e-2 if kinked else 5e-3), "input grad %d" % j
        else:
            assert rel(n(x.grad), arr["gin%d" % j]) < 1e-4, "input grad %d" % j
    gmax = max(float(np.max(np.abs(arr[k]))) for k in arr if k.startswith("grad/"))
    allgot, allref = [], []
    for k, p in dict(net.named_parameters()).items():
        ref = arr["grad/" + k]
        got = n(p.grad) if p.grad is not None else np.zeros_like(ref)
        allgot.append(got.ravel()); allref.append(ref.ravel())
        if kinked:
            continue
        if x3:
            assert np.linalg.norm(got - ref) < 5e-3 * np.linalg.norm(ref) + 2e-5 * gmax * np.sqrt(ref.size), "param grad %s" % k
        else:
            assert np.max(np.abs(got - ref)) < 5e-4 * np.max(np.abs(ref)) + 2e-6 * gmax, "param grad %s" % k
    assert l2rel(np.concatenate(allgot), np.concatenate(allref)) < (3e-2 if kinked else 5e-3), "all parameter gradients"
    for k, b in net.named_buffers():
        if "buf/" + k in arr and not k.endswith("num_batches_tracked"):
            assert rel(n(b), arr["buf/" + k]) < out_tol, k

