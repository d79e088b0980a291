"""GPU parity, step level: AugmentedCycleGAN / StochCycleGAN .train_instance on the HIP path against
(a) the golden fixtures captured from the reference itself and (b) the fp32 oracle run on this box.

Bar (BASELINE.json north star): generator activations and losses within 1e-3 relative of the reference
CPU path.  Step 0 is a pure function of the inputs.  Exact-fp32 arithmetic holds it to 2e-4 (losses) / 1e-3
(gradient norms) / 1e-4 (images); the split-bf16 arithmetic (16-bit operand mantissas, hip_util.PRECISIONS) to
the bar itself: 1e-3 losses and images, 3e-3 gradient norms.  Later steps inherit Adam's amplification of rounding
noise on ~zero-gradient tensors (see tests/test_oracle_golden.py): 1e-2 / 6e-2 / 5e-3, and 2e-2 / 6e-2 / 2e-2.
"""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, names, digest  # noqa: E402


def make_opt(**kw):
    d = dict(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, lr=2e-4, beta1=0.5, max_gnorm=500.0,
             lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025, lambda_sup_A=0.1, lambda_sup_B=0.1, stoch_enc=False, z_gan=1,
             enc_A_B=1, no_lsgan=False, norm="instance", use_dropout=False, which_model_netG="resnet",
             which_model_netD="basic", gpu_ids=[0], monitor_gnorm=True, niter_decay=25, expr_dir="/tmp", n_blocks=3)
    d.update(kw)
    return argparse.Namespace(**d)


def build_model(meta):
    from hip_util import load_recipe
    from dtgan_amd import model as M
    opt = make_opt(**meta["opt"])
    m = M.AugmentedCycleGAN(opt, testing=True) if meta["aug"] else M.StochCycleGAN(opt, testing=True)
    for k, net in m._net_dict().items():
        load_recipe(net, k, meta["seed"], meta["flavour"])
    return m


STEP_TOL = {  # precision -> ((loss, gnorm, image) at step 0, the same after Adam updates)
    "f32": ((2e-4, 1e-3, 1e-4), (1e-2, 6e-2, 5e-3)),
    "bf16x3": ((1e-3, 3e-3, 1e-3), (2e-2, 6e-2, 2e-2)),
}


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("name", names("step"))
def test_train_instance_matches_reference_golden(name, prec):
    from hip_util import precision, injected_dropout
    with precision(prec), injected_dropout(load(name)[1].get("drop_seed")):   # one mask counter over all steps of the fixture
        _check_steps(name, prec)


# cycle reconstructions rec_A / rec_B (model.py:467, 493) chain two generators: (step 0, later steps) per precision and
# fixture flavour.  'init' = the reference's own initialisation statistics (what training starts from): the north-star
# 1e-3 bar holds in bf16x3.  'rich' = O(1) InstanceNorm gains everywhere, deliberately ill-conditioned
# (tools/conditioning_probe.py: the EXACT-fp32 path itself moves rec_* by 4e-4..9e-4 under a 4e-6 input perturbation):
# bf16x3 lands at 1.8-7.9e-4 on the reference goldens (0.8-1.4e-3 in the 6-block oracle case, allowed 3e-3 there).  After an Adam update the fp32 oracle itself is 1.9e-2 from the
# reference on 'rich' (tests/test_oracle_golden.py).
# (after the first Adam update the exact-fp32 HIP path sits 8e-3 and bf16x3 2.5e-2 from the reference on the 'init' stoch_enc fixture: Adam turns
# summation-order noise on ~zero gradients into +-lr moves, and rec_* passes them through two generators)
# Round 6: the measured step-0 errors are recorded every run (ACG_REC_ERR_LOG; profiles/r06_rec_errors.txt) — 'rich' in bf16x3
# lands at 1.8e-4 .. 7.9e-4, so the reference goldens are all held to the north-star 1e-3 at step 0 now (the 3e-3 allowance
# of rounds 2-5 remains only in the build's own 6-block oracle case below, which measures 0.8 / 1.4e-3).
REC_TOL = {("f32", "init"): (2e-4, 2e-2), ("f32", "rich"): (3e-4, 4e-2),
           ("bf16x3", "init"): (1e-3, 4e-2), ("bf16x3", "rich"): (1e-3, 6e-2)}


# What the fixtures hold beyond losses and images (tools/make_goldens.py:292-304, taken from the reference after
# `optimizer.step()`, model.py:447-452, 510-515): per-tensor digests of .grad and of the applied update (post - pre), and
# the BatchNorm buffers after the last step.  (grad, update) tolerances on the abs-sum / L2 digests, step 0 only — later
# steps are chaotic per tensor (tests/test_oracle_golden.py).  bf16x3 gradients are norm-wise quantities: a ReLU mask that
# flips under the 2^-17 operand rounding changes single entries discretely (tools/conditioning_probe.py).
# Measured worst over the five fixtures: f32 inside (3e-3, 5e-3); bf16x3 on the 'init' flavour (the reference's own
# initialisation statistics) 1.2e-2 on one CondInstanceNorm shift-conv weight of the full-width config-1 fixture (a sum over
# ReLU-gated per-sample shifts), all other tensors < 1e-2.  On the deliberately ill-conditioned 'rich' flavour some bf16x3
# digests are not a pin: switching the InstanceNorm statistics between two equally accurate fp32 methods (both 1e-7 from fp64,
# test_conv_epilogue_statistics_equal_the_statistics_pass) moves the forward by 3e-5 and the CondInstanceNorm shift / scale
# convolution gradients by up to 17 % there (tools/debug_stats_ab.py).  Those tensors are SKIPPED BY NAME in bf16x3 on 'rich'
# (RICH_X3_SKIP: the f32 mode pins them on the same fixtures); every other tensor is held to the bf16x3 tolerance.
DIGEST_TOL = {"f32": (3e-3, 5e-3), "bf16x3": (2e-2, 2e-2)}
# (network, substring of the parameter name): the latent-conditioned SHIFT layers (modules.py:111-118: ReLU(1x1 conv(z)),
# a sum over ReLU-gated per-sample shifts) of the two full-resolution CondInstanceNorms of G_A_B whose planes are largest —
# the only tensors of the 'rich' fixtures outside the bf16x3 tolerance (measured 3e-2 .. 0.16; all others <= 2e-2)
RICH_X3_SKIP = (("netG_A_B", "model.2.shift_conv"), ("netG_A_B", "model.14.shift_conv"))
# The --norm batch --use_dropout fixture: BatchNorm gains are ~N(1, 0.02) at initialisation (networks.py:19-21) where the
# InstanceNorm gains are ~N(0, 0.02), so G_B_A is a high-gain network there although the flavour is 'init', and D_A ends in
# BatchNorms over 4 samples x (2x2 | 1x1) maps.  tools/step_grad_conditioning.py: the EXACT-fp32 path moves G_B_A's gradients
# by 2e-4 .. 8e-3 (discretely: one LeakyReLU / ReLU unit on the other side) when its inputs are perturbed by 4e-6, G_A_B's by
# 2e-5; bf16x3 lands 8e-2 off on the head bias (a sum over all pixels of the image gradient), 2-4e-2 on two more tensors
# whose digests stay inside the tolerance.  That one tensor is skipped by name in bf16x3; f32 pins it on the same fixture.
BN_DROPOUT_X3_SKIP = (("netG_B_A", "model.19.bias"),)
# Networks whose .grad after the step is comparable: the reference lets loss_G.backward() pile the (unused) G-phase
# gradients on top of the discriminators' D-phase .grad (model.py:509, no zero_grad for them); the HIP path skips those
# weight gradients, so the discriminators are pinned by their UPDATE digests (which only see the D-phase gradient).
GRAD_NETS = ("netG_A_B", "netG_B_A", "netE_B")


def _check_digests(m, arr, pre, prec, flavour="init", bn_dropout=False):
    gt, ut = DIGEST_TOL[prec]
    skip = RICH_X3_SKIP if (prec == "bf16x3" and flavour == "rich") else ()
    if prec == "bf16x3" and bn_dropout:
        skip = skip + BN_DROPOUT_X3_SKIP
    bad, seen, skipped = [], 0, []
    for nname, net in m._net_dict().items():
        params = dict(net.named_parameters())
        grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
                 for k, p in params.items()}
        gmax = max([float(np.max(np.abs(g))) for g in grads.values()] + [0.0])
        for k, p in params.items():
            key = "s0/grad/%s/%s" % (nname, k)
            assert key in arr, key
            g = grads[k]
            is_skip = any(nname == sn and sub in k for sn, sub in skip)
            if nname in GRAD_NETS:
                dg, rg = digest(g), arr[key]
                floor = 3e-5 * gmax * np.array([g.size, np.sqrt(g.size)])    # summation noise on analytically-zero gradients
                if not np.all(np.abs(dg[1:3] - rg[1:3]) <= gt * np.abs(rg[1:3]) + floor):
                    (skipped if is_skip else bad).append(("grad", nname, k, float(np.max(np.abs(dg[1:3] - rg[1:3]) / (np.abs(rg[1:3]) + 1e-30)))))
                seen += 0 if is_skip else 1
            d = digest(p.detach().cpu().numpy().astype(np.float64) - pre[nname][k].astype(np.float64))
            r = arr["s0/upd/%s/%s" % (nname, k)]
            # Adam's g / (|g| + eps) amplifies rounding noise on ~zero gradients to O(lr): well-conditioned tensors only (for
            # the discriminators the criterion uses this side's D-phase gradient, which is what their update was made from)
            if np.min(np.abs(g)) > 1e-5 * gmax and np.min(np.abs(g)) > 1e-6:
                if not (abs(d[1] - r[1]) <= ut * r[1] + 1e-12 and abs(d[2] - r[2]) <= ut * r[2] + 1e-12):
                    (skipped if is_skip else bad).append(("upd", nname, k, float(max(abs(d[1] - r[1]) / (r[1] + 1e-30), abs(d[2] - r[2]) / (r[2] + 1e-30)))))
                seen += 0 if is_skip else 1
    if skipped:
        print("digests outside the tolerance on tensors skipped by name (%s, %s):" % (prec, flavour), skipped)
    assert seen > 50 and not bad, (len(bad), bad)


def _check_steps(name, prec):
    from hip_util import t, n, rel
    from dtgan_amd import model as M
    arr, meta = load(name)
    m = build_model(meta)
    pre = {nn_: {k: p.detach().cpu().numpy().copy() for k, p in net.named_parameters()} for nn_, net in m._net_dict().items()}
    orig_reparam = M.gauss_reparametrize
    for st in range(meta["steps"]):
        A, B, z = (t(arr["s%d/%s" % (st, k)]) for k in ("real_A", "real_B", "prior_z_B"))
        if "s%d/eps" % st in arr:   # --stoch_enc fixture: the reference ran with this reparametrisation noise injected
            eps = t(arr["s%d/eps" % st])

            def fixed_reparametrize(mu, logvar, n_sample=1, eps=eps):  # model.py:15-22 with the draw replaced by `eps`
                zz = eps.mul(logvar.mul(0.5).exp()[:, None, :]).add(mu[:, None, :]).clamp(-4.0, 4.0)
                return zz.view(zz.size(0) * zz.size(1), zz.size(2), 1, 1)
            M.gauss_reparametrize = fixed_reparametrize
        try:
            losses, visuals, gnorms = m.train_instance(A, B, z)
        finally:
            M.gauss_reparametrize = orig_reparam
        assert list(losses.keys()) == meta["loss_keys"]
        assert list(gnorms.keys()) == meta["gnorm_keys"]
        lt, gt, vt = STEP_TOL[prec][0 if st == 0 else 1]
        got, ref = np.array(list(losses.values())), arr["s%d/losses" % st]
        assert np.allclose(got, ref, rtol=lt, atol=2e-6), (st, dict(zip(meta["loss_keys"], zip(got, ref))))
        gg, gr = np.array(list(gnorms.values())), arr["s%d/gnorms" % st]
        assert np.allclose(gg, gr, rtol=gt, atol=1e-6), (st, dict(zip(meta["gnorm_keys"], zip(gg, gr))))
        assert rel(n(visuals["fake_B"]), arr["s%d/fake_B" % st]) < vt
        assert rel(n(visuals["fake_A"]), arr["s%d/fake_A" % st]) < vt
        rt = REC_TOL[(prec, meta["flavour"])][0 if st == 0 else 1]
        for k in ("rec_A", "rec_B"):
            e = rel(n(visuals[k]), arr["s%d/%s" % (st, k)])
            # the measurement beside the allowance, every run: ACG_REC_ERR_LOG=<file> collects the lines (profiles/r06_rec_errors.txt)
            line = "rec_err fixture=%s flavour=%s prec=%s step=%d %s=%.3e allowed=%.1e" % (name, meta["flavour"], prec, st, k, e, rt)
            print(line)
            if os.environ.get("ACG_REC_ERR_LOG"):
                with open(os.environ["ACG_REC_ERR_LOG"], "a") as f:
                    f.write(line + "\n")
            assert e < rt, (k, st, e)
        for k in ("real_A", "real_B"):
            assert np.array_equal(n(visuals[k]), arr["s%d/%s" % (st, k)])
        if st == 0:
            _check_digests(m, arr, pre, prec, meta["flavour"], bn_dropout=meta["opt"].get("norm") == "batch")
    if meta["aug"]:   # BatchNorm running buffers after the last step (networks.py:407-415, 450-462)
        for nname in ("netE_B", "netD_z_B"):
            for k, b in m._net_dict()[nname].named_buffers():
                ref = arr["final/buf/%s/%s" % (nname, k)]
                if k.endswith("num_batches_tracked"):
                    assert int(b) == int(ref), (nname, k)
                else:
                    assert rel(n(b), ref) < (2e-3 if prec == "f32" else 5e-3), (nname, k, rel(n(b), ref))


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_step_against_oracle_with_6_blocks(prec):
    """BASELINE config-1 shape with the north star's n_blocks=6 (the reference itself only builds 3):
    HIP path vs the oracle run here, incl. post-step generator output (i.e. the applied update)."""
    from hip_util import precision
    with precision(prec):
        _check_6_blocks(prec)


def _check_6_blocks(prec):
    from hip_util import t, n, rel, load_recipe
    from dtgan_amd import model as M
    from oracle import recipe, step
    kw = dict(input_nc=1, output_nc=1, ngf=8, nef=8, ndf=8, nlatent=4, n_blocks=6)
    opt = make_opt(**kw)
    m = M.AugmentedCycleGAN(opt, testing=True)
    for k, net in m._net_dict().items():
        load_recipe(net, k, 3, "rich")
    o = step.AugStep(step.Opt(**kw))
    o.load({k: recipe.values_for(net.shapes, k, 3, "rich") for k, net in o.nets().items()})
    A, B, z = recipe.inputs(5, 8, 1, 1, 64, 4)  # E ends in a BatchNorm over (batch x 1x1): 8 samples keep it well conditioned
    l1, v1, g1 = m.train_instance(t(A), t(B), t(z))
    l0, v0, g0 = o.train_instance(A, B, z)
    (lt, gt, vt), (_, _, vt1) = STEP_TOL[prec]
    assert np.allclose(list(l1.values()), list(l0.values()), rtol=lt, atol=2e-6), (l1, l0)
    assert np.allclose(list(g1.values()), list(g0.values()), rtol=gt, atol=1e-6), (g1, g0)
    for k in ("fake_A", "fake_B"):
        assert rel(n(v1[k]), v0[k]) < vt, k
    # the cycle reconstructions chain two of these high-gain generators: tools/conditioning_probe.py (`step`) shows the
    # EXACT-fp32 path moving rec_A / rec_B by 4e-4 .. 9e-4 when its inputs are perturbed by 4e-6 relative, the
    # operand rounding of bf16x3 (which lands at 0.8 / 1.4e-3 here; single-pass images at 1.1-1.3e-4)
    for k in ("rec_A", "rec_B"):
        assert rel(n(v1[k]), v0[k]) < (vt if prec == "f32" else 3e-3), k
    # weights after the step: compare the generators' outputs on a fresh batch
    A2, B2, z2 = recipe.inputs(6, 8, 1, 1, 64, 4)
    from oracle.tape import T
    fb = n(m.predict_B(t(A2), t(z2)))
    fbo = o.netG_A_B.forward(T(A2), T(z2)).v
    assert rel(fb, fbo) < vt1


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_stoch_enc_branch_against_oracle(prec, monkeypatch):
    """--stoch_enc (model.py:414-419, 501-502, 519-522): post_z is a clamped reparametrised sample, the KLD term joins
    loss_G and the latent GAN terms drop out.  Oracle-side variant at batch 8 with the same eps injected into both sides;
    the reference-derived pin of this branch is the golden `step_aug_small_s64_stoch_enc` (tools/make_goldens.py patches the
    normal_() draw inside the reference's gauss_reparametrize), run by test_train_instance_matches_reference_golden."""
    from hip_util import t, n, rel, load_recipe, precision
    from dtgan_amd import model as M
    from oracle import recipe, step
    kw = dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4, n_blocks=3, stoch_enc=True)
    N = 8
    A, B, z = recipe.inputs(21, N, 3, 3, 64, 4)
    eps = np.random.RandomState(77).normal(0, 1, (N, 1, 4)).astype(np.float32)

    def fixed_reparametrize(mu, logvar, n_sample=1):  # model.py:15-22 with the draw replaced by `eps`
        assert n_sample == 1
        zz = t(eps).mul(logvar.mul(0.5).exp()[:, None, :]).add(mu[:, None, :]).clamp(-4.0, 4.0)
        return zz.view(zz.size(0) * zz.size(1), zz.size(2), 1, 1)

    monkeypatch.setattr(M, "gauss_reparametrize", fixed_reparametrize)
    with precision(prec):
        m = M.AugmentedCycleGAN(make_opt(**kw), testing=True)
        for k, net in m._net_dict().items():
            load_recipe(net, k, 9, "init")
        o = step.AugStep(step.Opt(**kw))
        o.load({k: recipe.values_for(net.shapes, k, 9, "init") for k, net in o.nets().items()})
        l1, v1, g1 = m.train_instance(t(A), t(B), t(z))
        l0, v0, g0 = o.train_instance(A, B, z, eps=eps)
        (lt, gt, vt), _ = STEP_TOL[prec]
        assert list(l1.keys()) == list(l0.keys())
        assert np.allclose(list(l1.values()), list(l0.values()), rtol=lt, atol=2e-6), (l1, l0)
        assert np.allclose(list(g1.values()), list(g0.values()), rtol=gt, atol=1e-6), (g1, g0)
        for k in ("fake_A", "fake_B", "rec_A", "rec_B"):
            assert rel(n(v1[k]), v0[k]) < vt, k
        assert l1["KLD_z_B"] != 0.0


@pytest.mark.parametrize("aug", [True, False])
def test_step_as_one_captured_graph_matches_the_eager_step(aug):
    """enable_step_graph(): train_instance replayed as one HIP graph (inputs, Adam step number and reported scalars kept on
    the device) against the eager step on an identically initialised model, over seven steps (two eager warm-up calls, the
    capture, more replays) with a learning-rate change in between (re-capture).  Not bit-identical: the graph's Adam
    kernel forms the bias corrections on the device."""
    meta = dict(opt=dict(input_nc=1, output_nc=1, n_blocks=2), aug=aug, seed=5, flavour="init")
    ref, gr = build_model(meta), build_model(meta)
    gr.enable_step_graph()
    g = torch.Generator(device="cuda").manual_seed(3)
    for step in range(7):
        if step == 5:
            ref.update_learning_rate(); gr.update_learning_rate()
        a = torch.randn(4, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1)
        b = torch.randn(4, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1)
        z = torch.randn(4, 16, 1, 1, device="cuda", generator=g)
        lr_, vr, gn_r = ref.train_instance(a, b, z)
        lg, vg, gn_g = gr.train_instance(a, b, z)
        assert list(lr_.keys()) == list(lg.keys()) and list(gn_r.keys()) == list(gn_g.keys())
        # the first three steps run the same kernels on the same numbers (two eager warm-up calls, then the capture replayed
        # with the device-side bias correction, which differs from the host's by an ulp of pow()): equal to rounding.
        # Later steps inherit Adam's amplification of that ulp on ~zero gradients (cf. STEP_TOL's second row).
        lt, gt, vt = (1e-5, 1e-4, 1e-5) if step < 3 else (2e-3, 2e-2, 2e-2)
        for k in lr_:
            assert abs(lr_[k] - lg[k]) <= lt * max(1.0, abs(lr_[k])), (step, k, lr_[k], lg[k])
        for k in gn_r:
            assert abs(gn_r[k] - gn_g[k]) <= gt * max(1e-3, abs(gn_r[k])), (step, k, gn_r[k], gn_g[k])
        assert float((vr["fake_B"] - vg["fake_B"]).abs().max()) < vt
        assert vg["real_A"].shape == a.shape and torch.equal(vg["real_A"], a)
    assert gr._step_graph.graph is not None
    for o_r, o_g in zip(ref._optimizers().values(), gr._optimizers().values()):
        assert o_r.t == o_g.t == 7


def test_step_graph_deferred_scalars_and_interleaved_eager_steps():
    """Round 6: (a) enable_step_graph(defer_scalars=True) — a replayed step returns a DeferredStep whose scalars travel to pinned
    memory behind the replay; read one step late they equal the synchronous graph's, bit for bit.  (b) eager steps between
    replays (how bench.py samples its event brackets) and replays share the layers' packed weights, which every optimiser
    step refreshes in place: a model that alternates graph and eager steps equals one that only replays (same kernels; the
    eager Adam forms its bias corrections on the host, hence rounding-level differences), and the graph is not re-captured."""
    from dtgan_amd import model as M
    meta = dict(opt=dict(input_nc=1, output_nc=1, n_blocks=2), aug=True, seed=5, flavour="init")
    sync, lazy, mixed = build_model(meta), build_model(meta), build_model(meta)
    sync.enable_step_graph(); lazy.enable_step_graph(defer_scalars=True); mixed.enable_step_graph()
    g = torch.Generator(device="cuda").manual_seed(6)
    prev, captures = None, 0
    for step in range(8):
        a = torch.randn(4, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1)
        b = torch.randn(4, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1)
        z = torch.randn(4, 16, 1, 1, device="cuda", generator=g)
        ls, vs, gs = sync.train_instance(a, b, z)
        out = lazy.train_instance(a, b, z)
        if prev is not None:                     # the PREVIOUS step's deferred scalars, read while this step is in flight
            assert prev[0].result()[0] == prev[1] and prev[0].result()[2] == prev[2]
            prev = None
        if isinstance(out, M.DeferredStep):
            prev = (out, ls, gs)
        else:                                    # the two eager warm-up calls return the tuple itself
            assert step < 2 and out[0] == ls
        graph_before = mixed._step_graph.graph
        if step in (4, 6):                       # an eagerly enqueued step between replays
            sg, mixed._step_graph = mixed._step_graph, None
            try:
                lm, vm, gm = mixed.train_instance(a, b, z)
            finally:
                mixed._step_graph = sg
        else:
            lm, vm, gm = mixed.train_instance(a, b, z)
        if graph_before is not None:
            assert mixed._step_graph.graph is graph_before, "the graph was captured again (step %d)" % step
        lt, gt, vt = (1e-5, 1e-4, 1e-5) if step < 5 else (2e-3, 2e-2, 2e-2)
        for k in ls:
            assert abs(ls[k] - lm[k]) <= lt * max(1.0, abs(ls[k])), (step, k, ls[k], lm[k])
        for k in gs:
            assert abs(gs[k] - gm[k]) <= gt * max(1e-3, abs(gs[k])), (step, k, gs[k], gm[k])
        assert float((vs["fake_B"] - vm["fake_B"]).abs().max()) < vt
    assert prev is not None and prev[0].result()[0] == prev[1]


def test_step_graph_owns_its_scratch():
    """The captured kernels keep the POINTERS of the scratch buffers ops.workspace() handed out during the capture.  An eager
    op that needs more scratch afterwards (a larger evaluation batch backpropagating through G_A_B, train.py's eval_ubo_B)
    replaces the eager buffer; the graph must not write into the block that went back to the allocator."""
    from dtgan_amd import ops
    meta = dict(opt=dict(input_nc=1, output_nc=1, n_blocks=2), aug=True, seed=5, flavour="init")
    gr = build_model(meta)
    gr.enable_step_graph()
    g = torch.Generator(device="cuda").manual_seed(4)

    def batch(nb):
        return (torch.randn(nb, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1),
                torch.randn(nb, 1, 64, 64, device="cuda", generator=g).clamp_(-1, 1),
                torch.randn(nb, 16, 1, 1, device="cuda", generator=g))
    for _ in range(4):                                   # two eager warm-up calls, the capture, one replay
        gr.train_instance(*batch(4))
    sg = gr._step_graph
    assert sg.graph is not None and sg.ws, "the capture keeps its own workspace table"
    graph_ptrs = {k: v.data_ptr() for k, v in sg.ws.items()}
    eager_before = {k: v.data_ptr() for k, v in ops._WS.items()}
    assert not set(graph_ptrs.values()) & set(eager_before.values())
    # an eager backward at three times the batch: grows / replaces the eager workspaces
    a, b, z = batch(12)
    zz = z.clone().requires_grad_(True)
    gr.predict_B(a, zz).sum().backward()
    torch.cuda.synchronize()
    # whatever the allocator hands out now must survive further replays
    canary = [torch.full((1 << 18,), 7.0, device="cuda") for _ in range(8)]
    for _ in range(2):
        losses, _, _ = gr.train_instance(*batch(4))
        assert all(np.isfinite(v) for v in losses.values())
    torch.cuda.synchronize()
    assert all(bool((c == 7.0).all()) for c in canary)
    assert {k: v.data_ptr() for k, v in sg.ws.items()} == graph_ptrs
