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
