"""GPU parity, network level: each of the six networks (HIP path, public NCHW API) against the golden
fixtures produced from the reference itself (tests/golden/, tools/make_goldens.py): forward outputs,
input gradients, every parameter gradient, BatchNorm running statistics.

Bar: 1e-3 relative (BASELINE.json north star) on activations; we assert 1e-4 on outputs/input grads
and 5e-4 (+ a noise floor for analytically-zero bias gradients) on parameter gradients.
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
    if n == "netG_B_A":
        return N.define_G(c["input_nc"], c["output_nc"], c["ngf"], gpu_ids=g, n_blocks=c["n_blocks"])
    if n == "netG_A_B":
        return N.define_stochastic_G(c["nlatent"], c["input_nc"], c["output_nc"], c["ngf"], gpu_ids=g, n_blocks=c["n_blocks"])
    if n == "netD_B":
        return N.define_D_B(c["input_nc"], c["ndf"], "basic", "instance", gpu_ids=g)
    if n == "netD_A":
        return N.define_D_A(c["input_nc"], c["ndf"], "basic", "instance", gpu_ids=g)
    if n == "netE_B":
        return N.define_E(c["nlatent"], c["input_nc"], c["nef"], "batch", gpu_ids=g)
    if n == "netD_z_B":
        return N.define_LAT_D(c["nlatent"], c["ndf"], gpu_ids=g)
    raise KeyError(n)


@pytest.mark.parametrize("impl", ["mfma", "direct"])
@pytest.mark.parametrize("name", names("net"))
def test_net_matches_reference_golden(name, impl):
    from hip_util import t, n, rel, load_recipe
    from dtgan_amd import ops
    arr, meta = load(name)
    if impl == "direct" and name not in ("G_A_B_s16_nb3", "D_B_s40", "E_B_s64"):
        pytest.skip("direct cross-check on a subset")
    ops.set_conv_impl(impl)
    try:
        net = load_recipe(build(meta), meta["net"], meta["seed"], meta["flavour"])
        net.train()
        ins, i = [], 0
        while "in%d" % i in arr:
            ins.append(t(arr["in%d" % i], grad=True)); i += 1
        out = net.forward(*ins)
        outs = list(out) if isinstance(out, tuple) else [out]
        for j, o in enumerate(outs):
            assert tuple(o.shape) == arr["out%d" % j].shape
            assert rel(n(o), arr["out%d" % j]) < 1e-4, "forward out%d" % j
        loss = sum((o * t(arr["R%d" % j])).sum() for j, o in enumerate(outs))
        loss.backward()
        for j, x in enumerate(ins):
            assert rel(n(x.grad), arr["gin%d" % j]) < 1e-4, "input grad %d" % j
        gmax = max(float(np.max(np.abs(arr[k]))) for k in arr if k.startswith("grad/"))
        for k, p in dict(net.named_parameters()).items():
            ref = arr["grad/" + k]
            got = n(p.grad) if p.grad is not None else np.zeros_like(ref)
            assert np.max(np.abs(got - ref)) < 5e-4 * np.max(np.abs(ref)) + 2e-6 * gmax, "param grad %s" % k
        for k, b in net.named_buffers():
            if "buf/" + k in arr and not k.endswith("num_batches_tracked"):
                assert rel(n(b), arr["buf/" + k]) < 1e-4, k
    finally:
        ops.set_conv_impl("mfma")
