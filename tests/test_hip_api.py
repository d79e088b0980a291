"""GPU tests of the remaining model-level API the reference's drivers call (SURVEY.md §8b Face 1):
supervised_train_instance, inference helpers, predict_B differentiable w.r.t. z (evaluate.py:70-71,120-126),
eval-mode BatchNorm, save/load, update_learning_rate, ignore_noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from test_hip_step import make_opt  # noqa: E402


def _model(aug=True, **kw):
    from hip_util import load_recipe
    from dtgan_amd import model as M
    d = dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4)
    d.update(kw)
    opt = make_opt(**d)
    m = (M.AugmentedCycleGAN if aug else M.StochCycleGAN)(opt, testing=True)
    for k, net in m._net_dict().items():
        load_recipe(net, k, 0, "rich")
    return m


def test_supervised_train_instance_matches_reference_golden():
    from hip_util import t, n, rel
    arr, meta = load("sup_aug_small_s64")
    m = _model(**meta["opt"])
    vals = m.supervised_train_instance(t(arr["real_A"]), t(arr["real_B"]), t(arr["prior_z_B"]))
    assert list(vals.keys()) == meta["keys"]
    assert np.allclose(list(vals.values()), arr["values"], rtol=1e-3, atol=2e-6), dict(zip(meta["keys"], zip(vals.values(), arr["values"])))
    fb = n(m.predict_B(t(arr["probe_A"]), t(arr["probe_z"])))
    assert rel(fb, arr["probe_fake_B_after"]) < 5e-3


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_predict_B_is_differentiable_wrt_noise(prec):
    """evaluate.py's variational bound optimises (mu, logvar) THROUGH predict_B: d fake_B / d z must match the oracle"""
    from hip_util import precision
    with precision(prec):
        _check_dz(prec)


def _check_dz(prec):
    from hip_util import t, n, rel
    from oracle import nets, recipe
    from oracle.tape import T, backward, leaf
    m = _model()
    A, B, z = recipe.inputs(3, 2, 3, 3, 32, 4)
    zt = t(z, grad=True)
    fb = m.predict_B(t(A), zt)
    r = np.random.RandomState(0).normal(0, 1, fb.shape)
    (fb * t(r)).sum().backward()
    og = nets.CINResnetGenerator(4, 3, 3, 8, 3, np.float64)
    recipe.fill(og, "netG_A_B", 0, "rich")
    Z = leaf(z.astype(np.float64))
    out = og.forward(T(A.astype(np.float64)), Z)
    backward(out, seed=r)
    assert rel(n(fb), out.v) < (1e-4 if prec == "f32" else 1e-3)
    assert rel(n(zt.grad), Z.g) < (1e-3 if prec == "f32" else 5e-3)


def test_inference_helpers_shapes_and_eval_mode():
    from hip_util import t, n
    from oracle import recipe
    m = _model()
    A, B, z = recipe.inputs(4, 3, 3, 3, 64, 4)
    tA, tB, tz = t(A), t(B), t(z)
    m.train_instance(tA, tB, tz)          # populate BatchNorm running statistics
    m.eval()
    vis = m.generate_cycle(tA, tB, tz)
    assert list(vis.keys()) == ["real_A", "fake_B", "rec_A", "real_B", "fake_A", "rec_B"]
    assert all(v.shape == tA.shape for v in vis.values())
    multi_z = t(np.random.RandomState(1).normal(0, 1, (6, 4, 1, 1)))
    assert m.generate_multi(tA, multi_z).shape == (6, 3, 64, 64)
    assert m.inference_multi(tA, tB).shape == (9, 3, 64, 64)
    (mu,) = m.predict_enc_params(tA, tB)
    assert mu.shape == (3, 4)
    # eval-mode BatchNorm uses the running buffers: the same sample gives the same code whatever the batch
    (mu1,) = m.predict_enc_params(tA[:1], tB[:1])
    assert np.allclose(n(mu1), n(mu[:1]), rtol=1e-4, atol=1e-5)
    assert m.predict_A(tB).shape == tA.shape
    fa, mfb = m.generate_cycle_B_multi(tB, multi_z)
    assert fa.shape == tA.shape and mfb.shape == (6, 3, 64, 64)
    imgs = m.generate_multi_cycle(tB, 2)
    assert len(imgs) == 5
    m.train()


def test_save_load_roundtrip_and_lr_schedule(tmp_path):
    from hip_util import t, n
    from oracle import recipe
    m = _model()
    m.opt.expr_dir = str(tmp_path)
    A, B, z = recipe.inputs(5, 3, 3, 3, 64, 4)
    m.train_instance(t(A), t(B), t(z))
    m.save("latest")
    ck = torch.load(str(tmp_path / "latest"), map_location="cpu")
    assert set(ck.keys()) == {"netG_A_B", "netG_B_A", "netD_A", "netD_B", "netD_z_B", "netE_B", "optimizer_D_A",
                              "optimizer_G_A", "optimizer_D_B", "optimizer_G_B"}                 # model.py:752-763
    ref_next = m.train_instance(t(A), t(B), t(z))[0]
    m2 = _model()
    m2.load(str(tmp_path / "latest"))
    got_next = m2.train_instance(t(A), t(B), t(z))[0]
    # identical weights + Adam state -> identical next step (BatchNorm buffers travel in the state_dicts)
    assert np.allclose(list(got_next.values()), list(ref_next.values()), rtol=1e-5, atol=1e-7)
    lr0 = m.optimizer_G_A.param_groups[0]["lr"]
    m.update_learning_rate()
    assert abs(m.optimizer_G_A.param_groups[0]["lr"] - (lr0 - m.opt.lr / m.opt.niter_decay)) < 1e-12
    assert m.optimizer_D_A.param_groups[0]["lr"] == m.optimizer_G_A.param_groups[0]["lr"]       # model.py:735-745 quirk


def test_stoch_cyclegan_ignore_noise_and_aliases():
    from hip_util import t
    from oracle import recipe
    from dtgan_amd import model as M
    m = _model(aug=False, output_nc=1)
    m.ignore_noise = True                                                                        # model.py:128-129
    A, B, z = recipe.inputs(6, 2, 3, 1, 64, 4)
    l1, _, _ = m.train_instance(t(A), t(B), t(z))
    m2 = _model(aug=False, output_nc=1)
    m2.ignore_noise = True
    l2, _, _ = m2.train_instance(t(A), t(B), t(z * 0 + 7.0))  # z is replaced by ones: the value must not matter
    assert np.allclose(list(l1.values()), list(l2.values()), rtol=1e-6)
    assert M.AugmentedCycleGAN_Model is M.AugmentedCycleGAN
    m3 = _model()
    m3.set_input({"A": t(recipe.inputs(7, 3, 3, 3, 64, 4)[0]), "B": t(recipe.inputs(7, 3, 3, 3, 64, 4)[1])})
    out = m3.optimize_parameters()
    assert len(out) == 3 and "G_A" in out[0]


def _oracle_twin(seed=0, flavour="rich", **kw):
    from oracle import recipe, step
    d = dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4, n_blocks=3)
    d.update(kw)
    o = step.AugStep(step.Opt(**d))
    o.load({k: recipe.values_for(net.shapes, k, seed, flavour) for k, net in o.nets().items()})
    return o


def test_public_discriminate_and_criterion_gan_match_oracle():
    """model.py:327-334 `discriminate(net, crit, fake, real)` and model.py:56-72 `criterion_GAN` on PUBLIC (NCHW) tensors —
    the form evaluate.py / a user script calls them in (the step itself uses the fused NHWC path)."""
    from hip_util import t, n, rel
    from dtgan_amd import model as M
    from oracle import recipe, step
    from oracle.tape import T
    m, o = _model(), _oracle_twin()
    A, B, z = recipe.inputs(12, 3, 3, 3, 64, 4)
    for net, onet, fake, real in ((m.netD_A, o.netD_A, B, A), (m.netD_B, o.netD_B, A, B)):
        lf, lt, pf, pt = M.discriminate(net, m.criterionGAN, t(fake), t(real))
        olf, olt, opf, opt_ = step.discriminate(onet, T(fake), T(real))
        assert pf.shape == opf.v.shape and rel(n(pf), opf.v) < 1e-3 and rel(n(pt), opt_.v) < 1e-3
        assert abs(float(lf) - float(olf.v)) < 1e-3 * abs(float(olf.v)) and abs(float(lt) - float(olt.v)) < 1e-3 * abs(float(olt.v))
    # latent discriminator: (N, nl, 1, 1) input, (N, 1) prediction
    lf, lt, pf, pt = M.discriminate(m.netD_z_B, m.criterionGAN, t(z), t(z[::-1].copy()))
    olf, olt, opf, opt_ = step.discriminate(o.netD_z_B, T(z), T(z[::-1].copy()))
    assert pf.shape == (3, 1) and rel(n(pf), opf.v.reshape(3, 1)) < 1e-3
    assert abs(float(lt) - float(olt.v)) < 1e-3 * abs(float(olt.v))
    # gradients flow through the public form too
    x = t(A, grad=True)
    M.criterion_GAN(m.netD_A(x), True, use_sigmoid=False).backward()
    assert x.grad is not None and float(x.grad.abs().max()) > 0
    with pytest.raises(NotImplementedError):
        M.criterion_GAN(m.netD_A(t(A)), True, use_sigmoid=True)


def test_generate_multi_and_inference_multi_values_match_oracle():
    """model.py:687-733: values, not only shapes — each A repeated `num` times against `num` latent codes
    (generate_multi), and every A against the posterior codes of every B (inference_multi)."""
    from hip_util import t, n, rel
    from oracle import ops as oops, recipe
    from oracle.tape import T
    m, o = _model(), _oracle_twin()
    A, B, z = recipe.inputs(13, 3, 3, 3, 32, 4)
    num = 2
    mz = np.random.RandomState(5).normal(0, 1, (3 * num, 4, 1, 1)).astype(np.float32)
    got = n(m.generate_multi(t(A), t(mz)))
    ref = o.netG_A_B.forward(T(np.repeat(A, num, axis=0)), T(mz)).v
    assert got.shape == ref.shape and rel(got, ref) < 1e-3
    # inference_multi needs S = 64 for the encoder's 1x1 map
    A, B, z = recipe.inputs(14, 3, 3, 3, 64, 4)
    got = n(m.inference_multi(t(A), t(B)))
    fake_A = o.netG_B_A.forward(T(B))
    mu, _ = o.netE_B.forward(oops.cat_channels(fake_A, T(B)))
    post = mu.v.reshape(3, 4, 1, 1)
    ref = o.netG_A_B.forward(T(np.repeat(A, 3, axis=0)), T(np.tile(post, (3, 1, 1, 1)))).v
    # a chain of G_B_A, the encoder and G_A_B on 'rich' weights: the rec_B conditioning (test_hip_step.REC_TOL), measured 1.8e-3
    assert got.shape == ref.shape == (9, 3, 64, 64) and rel(got, ref) < 3e-3
    # generate_cycle: all six images
    A, B, z = recipe.inputs(15, 3, 3, 3, 64, 4)
    vis = m.generate_cycle(t(A), t(B), t(z))
    fB = o.netG_A_B.forward(T(A), T(z)); fA = o.netG_B_A.forward(T(B))
    mu, _ = o.netE_B.forward(oops.cat_channels(fA, T(B)))
    rA = o.netG_B_A.forward(fB); rB = o.netG_A_B.forward(fA, T(mu.v.reshape(3, 4, 1, 1)))
    for k, r in (("fake_B", fB), ("fake_A", fA), ("rec_A", rA), ("rec_B", rB)):
        assert rel(n(vis[k]), r.v) < (1e-3 if k.startswith("fake") else 3e-3), k     # 'rich' weights: see test_hip_step.REC_TOL


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_evaluation_numbers_match_reference_golden(prec):
    """evaluate.py:10-19 (eval_mse_A) and evaluate.py:39-148 (variational_ubo: RMSprop on (mu, logvar) THROUGH predict_B):
    the golden was computed by the reference's own model and model.py helpers with the same noise draws."""
    from hip_util import t, precision
    from dtgan_amd import evaluate as E
    arr, meta = load("eval_aug_small_s64")
    with precision(prec):
        m = _model(**meta["opt"])
        A, B = t(arr["real_A"]), t(arr["real_B"])
        mse = E.eval_mse_A([{"A": A, "B": B}], m, use_gpu=True)
        assert abs(mse - float(arr["mse_A"])) < 1e-3 * float(arr["mse_A"])
        trace = []
        eps = [t(e) for e in arr["eps"]]
        ubo, kld, bpp = E.variational_ubo(m, A, B, meta["steps"], dequant=t(arr["dequant"]), eps_seq=eps, trace=trace)
        ref = arr["trace"]
        tol = 1e-4 if prec == "f32" else 1e-3      # sums over 12288 pixels of |x - fake_B| / sd: relative on the bound
        assert np.allclose(np.array(trace), ref, rtol=tol), (trace, ref)
        assert (ubo, kld, bpp) == trace[-1]
