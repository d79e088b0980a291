"""Pins the oracle (oracle/) against outputs of the reference itself (tests/golden/, produced
by tools/make_goldens.py importing /root/reference).  CPU only.

Tolerances: the oracle is an independent fp32 (and fp64) restatement; against the reference's
fp32 torch-CPU numbers we require <= 2e-5 relative on activations/gradients (fp32) —
well inside the 1e-3 bar of BASELINE.json — and the fp64 oracle must be at least as close.
"""
import numpy as np
import pytest

from oracle import nets, recipe, step
from oracle.tape import T, backward, leaf

from golden_util import load, names, rel_err, digest

NET_TOL = 2e-5


def seeded_dropout(seed):
    """mask source of a --use_dropout fixture: the k-th Dropout forward takes oracle.ops.dropout_keep(seed, k, shape)"""
    from oracle import ops
    k = [0]

    def src(shape):
        m = ops.dropout_keep(seed, k[0], shape)
        k[0] += 1
        return m
    return src


def build_net(meta, dtype):
    c, n = meta["cfg"], meta["net"]
    if n == "netG_B_A":
        net = nets.ResnetGenerator(c["input_nc"], c["output_nc"], c["ngf"], c["n_blocks"], dtype, norm=c.get("norm", "instance"),
                                   use_dropout=c.get("use_dropout", False))
    elif n == "netG_A_B":
        net = nets.CINResnetGenerator(c["nlatent"], c["input_nc"], c["output_nc"], c["ngf"], c["n_blocks"], dtype,
                                      use_dropout=c.get("use_dropout", False))
    else:
        net = None
    if net is not None:
        if "drop_seed" in meta:
            net.drop = seeded_dropout(meta["drop_seed"])
        return net
    if n == "netD_B":
        return nets.Discriminator(c["input_nc"], c["ndf"], dtype)
    if n == "netD_A":
        return nets.Discriminator_edges(c["input_nc"], c["ndf"], dtype)
    if n == "netE_B":
        return nets.LatentEncoder(c["nlatent"], c["input_nc"], c["nef"], dtype)
    if n == "netD_z_B":
        return nets.DiscriminatorLatent(c["nlatent"], c["ndf"], dtype)
    raise KeyError(n)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name", names("net"))
def test_net_forward_backward_matches_reference(name, dtype):
    arr, meta = load(name)
    net = build_net(meta, dtype)
    recipe.fill(net, meta["net"], meta["seed"], meta["flavour"])
    ins = []
    i = 0
    while "in%d" % i in arr:
        ins.append(leaf(arr["in%d" % i].astype(dtype)))
        i += 1
    out = net.forward(*ins)
    outs = list(out) if isinstance(out, tuple) else [out]
    for j, o in enumerate(outs):
        assert rel_err(o.v, arr["out%d" % j]) < NET_TOL, "forward out%d" % j
    # L = sum_j sum(out_j * R_j): seed each output with R_j
    from oracle import ops
    loss = None
    for j, o in enumerate(outs):
        term = T(np.asarray((o.v * arr["R%d" % j]).sum(), dtype), (o,), (lambda R: (lambda g: (g * R.astype(dtype),)))(arr["R%d" % j]))
        loss = term if loss is None else ops.add(loss, term)
    backward(loss)
    for j, x in enumerate(ins):
        assert rel_err(x.g, arr["gin%d" % j]) < 5 * NET_TOL, "input grad %d" % j
    # a conv bias that feeds an InstanceNorm has an analytically ZERO gradient (the norm removes
    # the mean); both sides then hold rounding noise, hence the absolute term tied to the largest
    # gradient in the network.
    gmax = max(float(np.max(np.abs(arr[k]))) for k in arr if k.startswith("grad/"))
    for k, p in net.params.items():
        ref = arr["grad/" + k]
        got = p.g if p.g is not None else np.zeros_like(ref)
        bound = 2e-4 * np.max(np.abs(ref)) + 2e-6 * gmax
        assert np.max(np.abs(got - ref)) < bound, "param grad %s" % k
    for k in arr:
        if k.startswith("buf/") and not k.endswith("num_batches_tracked"):
            assert rel_err(net.buffers[k[4:]], arr[k]) < NET_TOL, k


def _run_step(name, dtype):
    arr, meta = load(name)
    opt = step.Opt(**meta["opt"])
    m = (step.AugStep if meta["aug"] else step.StochStep)(opt, dtype=dtype)
    vals = {n: recipe.values_for(net.shapes, n, meta["seed"], meta["flavour"]) for n, net in m.nets().items()}
    m.load(vals)
    if "drop_seed" in meta:
        m.set_dropout_seed(meta["drop_seed"])
    return arr, meta, m


@pytest.mark.parametrize("name", names("step"))
def test_step_matches_reference(name):
    dtype = np.float32
    arr, meta, m = _run_step(name, dtype)
    pre = {n: {k: p.v.copy() for k, p in net.params.items()} for n, net in m.nets().items()}
    for st in range(meta["steps"]):
        A, B, z = arr["s%d/real_A" % st], arr["s%d/real_B" % st], arr["s%d/prior_z_B" % st]
        if "s%d/eps" % st in arr:   # --stoch_enc fixture: the reparametrisation noise the reference was given
            losses, visuals, gnorms = m.train_instance(A, B, z, eps=arr["s%d/eps" % st])
        else:
            losses, visuals, gnorms = m.train_instance(A, B, z)
        assert list(losses.keys()) == meta["loss_keys"]
        assert list(gnorms.keys()) == meta["gnorm_keys"]
        got = np.array(list(losses.values()))
        ref = arr["s%d/losses" % st]
        # step 0 is a pure function of the inputs: tight.  From step 1 on, the weights carry the
        # first Adam update, which turns fp32 rounding noise on ~zero-gradient tensors into +-lr
        # moves (see the note on Adam below) — in the reference as here — so later steps are looser.
        lt, gt = (2e-4, 5e-4) if st == 0 else (1e-2, 6e-2)
        assert np.allclose(got, ref, rtol=lt, atol=2e-6), (st, dict(zip(meta["loss_keys"], zip(got, ref))))
        gg = np.array(list(gnorms.values()))
        gr = arr["s%d/gnorms" % st]
        assert np.allclose(gg, gr, rtol=gt, atol=1e-6), (st, dict(zip(meta["gnorm_keys"], zip(gg, gr))))
        assert rel_err(visuals["fake_B"], arr["s%d/fake_B" % st]) < (1e-4 if st == 0 else 5e-3)
        assert rel_err(visuals["fake_A"], arr["s%d/fake_A" % st]) < (1e-4 if st == 0 else 5e-3)
        # the cycle reconstructions (model.py:467, 493) chain two generators (and the encoder for rec_B)
        # (measured: <= 3.7e-5 at step 0 in fp32 AND fp64; after the first Adam update the 'rich' fixtures reach 1.9e-2
        # in fp32 and 1.3e-2 in fp64 — the noise amplification described below, passed through two high-gain generators)
        assert rel_err(visuals["rec_A"], arr["s%d/rec_A" % st]) < (1e-4 if st == 0 else 3e-2)
        assert rel_err(visuals["rec_B"], arr["s%d/rec_B" % st]) < (1e-4 if st == 0 else 3e-2)
        # gradient digests (abs-sum, L2 per tensor) and post-step Adam-update digests.
        # Adam's update is g/(|g|+1e-8): where |g| is at fp32-noise level (conv biases feeding a
        # mean-removing norm have analytically zero gradient; 'init'-flavour inner-block tensors see
        # |g|~1e-8) rounding noise is amplified to O(lr), in the reference as much as here.  Updates
        # are therefore compared only on well-conditioned tensors (all |g| > 1e-5 * net-wide max);
        # every tensor is still covered by the gradient digests with a noise-floor term.
        # Per-tensor digests are checked on step 0 only; later steps are chaotic at per-tensor level
        # (noise-driven +-lr moves) and are covered by the losses / norms / images above.
        bad = []
        for n, net in (m.nets().items() if st == 0 else []):
            # (--stoch_enc: D_z_B receives no gradient at all in the D phase, model.py:438-439)
            gmax = max([float(np.max(np.abs(p.g))) for p in net.params.values() if p.g is not None] + [0.0])
            for k, p in net.params.items():
                g = p.g if p.g is not None else np.zeros_like(p.v)
                dg, rg = digest(g), arr["s%d/grad/%s/%s" % (st, n, k)]
                # noise floor: analytically-zero gradients (conv bias in front of a mean-removing norm) hold pure summation
                # noise on both sides; the reference's reaches 1e-4 of the net-wide max per entry at N=4 (stoch_enc fixture)
                floor = 3e-5 * gmax * np.array([g.size, np.sqrt(g.size)])
                # 3e-3: at full width / 'init' flavour the REFERENCE's own fp32 gradients sit up to
                # ~1.5e-3 from the fp64 oracle (L1-sign / ReLU-mask flips; e.g. netG_B_A model.19.bias:
                # ref 7.1622, fp32 oracle 7.1742, fp64 oracle 7.1730), so fp32-vs-fp32 cannot be tighter.
                if not np.all(np.abs(dg[1:3] - rg[1:3]) <= (3e-3 if st == 0 else 2e-2) * np.abs(rg[1:3]) + floor):
                    bad.append(("grad", n, k, dg[:3], rg[:3]))
                d = digest(p.v.astype(np.float64) - pre[n][k].astype(np.float64))
                r = arr["s%d/upd/%s/%s" % (st, n, k)]
                if np.min(np.abs(g)) > 1e-5 * gmax and np.min(np.abs(g)) > 1e-6:
                    ut = 5e-3 if st == 0 else 3e-2
                    if not (abs(d[1] - r[1]) <= ut * r[1] + 1e-12 and abs(d[2] - r[2]) <= ut * r[2] + 1e-12):
                        bad.append(("upd", n, k, d[:3], r[:3]))
        assert not bad, bad[:5]
    if meta["aug"]:
        for n in ("netE_B", "netD_z_B"):
            net = m.nets()[n]
            for k, b in net.buffers.items():
                if k.endswith("num_batches_tracked"):
                    assert int(b) == int(arr["final/buf/%s/%s" % (n, k)])
                else:
                    assert rel_err(b, arr["final/buf/%s/%s" % (n, k)]) < 2e-3, (n, k)


def test_supervised_step_matches_reference():
    """oracle AugStep.supervised_train_instance vs the reference's (model.py:541-604)"""
    arr, meta = load("sup_aug_small_s64")
    opt = step.Opt(**meta["opt"])
    m = step.AugStep(opt, dtype=np.float32)
    m.load({n: recipe.values_for(net.shapes, n, meta["seed"], meta["flavour"]) for n, net in m.nets().items()})
    vals = m.supervised_train_instance(arr["real_A"], arr["real_B"], arr["prior_z_B"])
    assert list(vals.keys()) == meta["keys"]
    assert np.allclose(list(vals.values()), arr["values"], rtol=5e-4, atol=2e-6), dict(zip(meta["keys"], zip(vals.values(), arr["values"])))
    fb = m.netG_A_B.forward(T(arr["probe_A"]), T(arr["probe_z"])).v
    assert rel_err(fb, arr["probe_fake_B_after"]) < 5e-3  # weights after the paired step
