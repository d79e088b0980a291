#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported, unmodified,
from /root/reference/augmented_cyclegan) on CPU with torch.  Test infrastructure:
runs only in the build container (the reference does not travel to the GPU box);
the fixtures it writes are data (inputs + expected outputs), never reference source.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [fixture names ...]     (no names: all of them)

Parameters come from oracle/recipe.py (regenerated from seeds on every box), so the
fixtures hold only inputs, outputs, gradients and digests.
"""
import json
import os
import sys
import warnings
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/augmented_cyclegan"
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import networks as rnet  # noqa: E402  (reference)
import modules as rmod  # noqa: E402  (reference)
import model as rmodel  # noqa: E402  (reference)

from oracle import recipe  # noqa: E402
from oracle import ops as oracle_ops  # noqa: E402  (dropout_keep: the seeded keep masks both sides regenerate)

torch.set_num_threads(8)
ONLY = set(a for a in sys.argv[1:] if not a.startswith("-"))   # regenerate just these fixtures
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
META = dict(torch=torch.__version__, numpy=np.__version__, generator="tools/make_goldens.py",
            reference="adrianalbert/domain-transfer-GAN @ /root/reference (imported unmodified)")


def unique_named_parameters(net):
    return OrderedDict(net.named_parameters())  # first name wins for aliased tensors


def load_recipe(net, net_name, seed, flavour, double=False):
    for k, p in unique_named_parameters(net).items():
        v = recipe.param(seed, net_name, k, tuple(p.shape), flavour)
        p.data.copy_(torch.from_numpy(v).to(p.dtype))
    return net


def with_blocks(gen, n_blocks, cin, ngf, nlatent=None):
    """Re-compose a reference generator with `n_blocks` residual blocks using the reference's
    own block classes (the constructor ignores n_blocks: networks.py:173,225 — SURVEY D3)."""
    mods = list(gen.model.children())
    stem, tail = mods[:10], mods[13:]
    blocks = []
    for _ in range(n_blocks):
        if cin:
            blocks.append(rmod.CINResnetBlock(x_dim=4 * ngf, z_dim=nlatent, padding_type="reflect",
                                              norm_layer=rmod.CondInstanceNorm, use_dropout=False, use_bias=True))
        else:
            import functools
            blocks.append(rmod.ResnetBlock(4 * ngf, padding_type="reflect",
                                           norm_layer=functools.partial(rmod.InstanceNorm2d, affine=True),
                                           use_dropout=False, use_bias=True))
    seq = rmod.TwoInputSequential if cin else nn.Sequential
    gen.model = seq(*(stem + blocks + tail))
    return gen


def grads_of(net):
    return {("grad/" + k): (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
            for k, p in unique_named_parameters(net).items()}


def save(name, arrays, **meta):
    m = dict(META); m.update(meta)
    arrays = dict(arrays)
    arrays["__meta__"] = np.frombuffer(json.dumps(m).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def rnd(seed, shape):
    return np.random.RandomState(seed).normal(0, 1, shape).astype(np.float32)


class InjectedDropout(object):
    """--use_dropout fixtures: while active, torch.nn.Dropout.forward (the reference's blocks call nn.Dropout(0.5),
    modules.py:167-168, 214-215) multiplies by the SEEDED keep mask oracle.ops.dropout_keep(seed, k, shape) of its k-th
    call instead of drawing from torch's RNG stream — torch's own module is patched, never the reference's source — so
    the branch is a pure function of the fixture's inputs on every side."""

    def __init__(self, seed):
        self.seed, self.k = seed, 0

    def __enter__(self):
        self.orig = nn.Dropout.forward
        me = self

        def forward(mod, x):
            if not mod.training:
                return x
            keep = oracle_ops.dropout_keep(me.seed, me.k, tuple(x.shape), mod.p)
            me.k += 1
            return x * torch.from_numpy(keep.astype(np.float32) / (1.0 - mod.p)).to(x.dtype)
        nn.Dropout.forward = forward
        return self

    def __exit__(self, *a):
        nn.Dropout.forward = self.orig


# ---------------------------------------------------------------- network-level goldens
def net_case(name, build, net_name, inputs, seed=0, flavour="rich", buffers=False, cfg=None, drop_seed=None):
    if ONLY and name not in ONLY:
        return
    net = load_recipe(build(), net_name, seed, flavour)
    net.train()
    tin = [torch.from_numpy(a.copy()).requires_grad_(True) for a in inputs]
    if drop_seed is not None:
        with InjectedDropout(drop_seed):
            out = net.forward(*tin)
    else:
        out = net.forward(*tin)
    outs = list(out) if isinstance(out, tuple) else [out]
    Rs = [rnd(seed + 900 + i, tuple(o.shape)) for i, o in enumerate(outs)]
    loss = sum((o * torch.from_numpy(R)).sum() for o, R in zip(outs, Rs))
    loss.backward()
    arr = {}
    for i, a in enumerate(inputs):
        arr["in%d" % i] = a
        arr["gin%d" % i] = tin[i].grad.numpy().copy()
    for i, (o, R) in enumerate(zip(outs, Rs)):
        arr["out%d" % i] = o.detach().numpy().copy()
        arr["R%d" % i] = R
    arr.update(grads_of(net))
    if buffers:
        for k, b in net.named_buffers():
            arr["buf/" + k] = b.detach().numpy().copy()
    extra = {} if drop_seed is None else {"drop_seed": drop_seed}
    save(name, arr, kind="net", net=net_name, seed=seed, flavour=flavour, cfg=cfg or {}, **extra)


def make_net_goldens():
    ngf, nl, nc = 8, 4, 3
    x16 = np.random.RandomState(11).uniform(-1, 1, (2, nc, 16, 16)).astype(np.float32)
    x32 = np.random.RandomState(12).uniform(-1, 1, (2, 1, 32, 32)).astype(np.float32)
    z2 = rnd(13, (2, nl, 1, 1))
    net_case("G_B_A_s16_nb3", lambda: rnet.define_G(nc, nc, ngf), "netG_B_A", [x16],
             cfg=dict(input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=3))
    net_case("G_B_A_s32_nc1_nb3", lambda: rnet.define_G(1, 1, ngf), "netG_B_A", [x32],
             cfg=dict(input_nc=1, output_nc=1, ngf=ngf, n_blocks=3))
    net_case("G_B_A_s16_nb6", lambda: with_blocks(rnet.define_G(nc, nc, ngf), 6, False, ngf), "netG_B_A", [x16],
             cfg=dict(input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=6))
    net_case("G_A_B_s16_nb3", lambda: rnet.define_stochastic_G(nl, nc, nc, ngf), "netG_A_B", [x16, z2],
             cfg=dict(nlatent=nl, input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=3))
    net_case("G_A_B_s32_nc1_nb3", lambda: rnet.define_stochastic_G(nl, 1, 1, ngf), "netG_A_B", [x32, z2],
             cfg=dict(nlatent=nl, input_nc=1, output_nc=1, ngf=ngf, n_blocks=3))
    net_case("G_A_B_s16_nb9", lambda: with_blocks(rnet.define_stochastic_G(nl, nc, nc, ngf), 9, True, ngf, nl),
             "netG_A_B", [x16, z2], cfg=dict(nlatent=nl, input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=9))
    # --norm batch (options.py:64 -> networks.py:23-31: BatchNorm2d in the stem, the tail AND inside ResnetBlock) and
    # --use_dropout (options.py:65 -> modules.py:167-168, 214-215: Dropout(0.5) behind the first ReLU of both block types)
    x16b = np.random.RandomState(18).uniform(-1, 1, (3, nc, 16, 16)).astype(np.float32)
    net_case("G_B_A_s16_nb3_batchnorm", lambda: rnet.define_G(nc, nc, ngf, norm="batch"), "netG_B_A", [x16b], buffers=True,
             cfg=dict(input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=3, norm="batch"))
    net_case("G_B_A_s16_nb3_dropout", lambda: rnet.define_G(nc, nc, ngf, use_dropout=True), "netG_B_A", [x16], drop_seed=31,
             cfg=dict(input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=3, use_dropout=True))
    net_case("G_A_B_s16_nb3_dropout", lambda: rnet.define_stochastic_G(nl, nc, nc, ngf, use_dropout=True), "netG_A_B",
             [x16, z2], drop_seed=32, cfg=dict(nlatent=nl, input_nc=nc, output_nc=nc, ngf=ngf, n_blocks=3, use_dropout=True))
    x64 = np.random.RandomState(14).uniform(-1, 1, (2, nc, 64, 64)).astype(np.float32)
    x40 = np.random.RandomState(15).uniform(-1, 1, (2, nc, 40, 40)).astype(np.float32)
    net_case("D_B_s40", lambda: rnet.define_D_B(nc, 8, "basic", "instance"), "netD_B", [x40],
             cfg=dict(input_nc=nc, ndf=8))
    net_case("D_A_s64", lambda: rnet.define_D_A(nc, 8, "basic", "instance"), "netD_A", [x64],
             cfg=dict(input_nc=nc, ndf=8))
    x64e = np.random.RandomState(16).uniform(-1, 1, (3, 2 * nc, 64, 64)).astype(np.float32)
    net_case("E_B_s64", lambda: rnet.define_E(nl, 2 * nc, 8, "batch"), "netE_B", [x64e], buffers=True,
             cfg=dict(nlatent=nl, input_nc=2 * nc, nef=8))
    z4 = rnd(17, (4, nl, 1, 1))
    net_case("D_z_B_n4", lambda: rnet.define_LAT_D(nl, 8), "netD_z_B", [z4], buffers=True,
             cfg=dict(nlatent=nl, ndf=8))


# ---------------------------------------------------------------- step-level goldens
class Namespace(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def ref_opt(**kw):
    d = dict(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, lr=2e-4, beta1=0.5, max_gnorm=500.0,
             lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025, lambda_sup_A=0.1, lambda_sup_B=0.1,
             stoch_enc=False, z_gan=1, enc_A_B=1, no_lsgan=False, norm="instance", use_dropout=False,
             which_model_netG="resnet", which_model_netD="basic", gpu_ids=[], monitor_gnorm=True,
             niter_decay=25, expr_dir="/tmp")
    d.update(kw)
    return Namespace(**d)


def digest(a):
    a = np.asarray(a, np.float64).ravel()
    idx = (np.arange(8) * 2654435761 % max(a.size, 1)).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[idx]])


class Recorder(object):
    def __init__(self):
        self.gan, self.l1, self.gn, self.predA, self.predB, self.enc = [], [], [], [], [], []
        self.gAB, self.gBA, self.cycz = [], [], []   # generator outputs in call order; -mean log-prob (stoch_enc Cyc_z_B)


def run_ref_step(m, rec, A, B, z, aug):
    """Run the reference's train_instance; it completes ALL compute then raises IndexError at
    its PyTorch-0.3 reporting line (`loss.data[0]`, model.py:518 / 193) — caught here."""
    try:
        m.train_instance(torch.from_numpy(A.copy()), torch.from_numpy(B.copy()), torch.from_numpy(z.copy()))
        raise RuntimeError("reference unexpectedly returned")
    except IndexError:
        pass


def step_case(name, aug, opt_kw, N, S, steps=2, seed=0, flavour="rich", eps_seed=None, drop_seed=None):
    """eps_seed: for --stoch_enc cases — the N(0,1) draw inside the reference's gauss_reparametrize
    (`std.data.new(N, 1, nl).normal_()`, model.py:19) is replaced by a fixed, recorded eps (torch.Tensor.normal_ is
    patched for tensors of exactly that shape while the step runs), so the branch becomes a pure function of the inputs."""
    if ONLY and name not in ONLY:
        return
    opt = ref_opt(**opt_kw)
    rec = Recorder()
    # recorders around the reference's own loss / clip functions
    orig_l1, orig_clip = rmodel.F.l1_loss, torch.nn.utils.clip_grad_norm
    orig_crit = rmodel.criterion_GAN
    orig_lpg, orig_normal = rmodel.log_prob_gaussian, torch.Tensor.normal_
    cur_eps = [None]

    def lpg(z, mu, lv):
        v = orig_lpg(z, mu, lv); rec.cycz.append(float(-1.0 * v.mean(1).mean(0))); return v

    def normal_(self, *a, **kw):
        e = cur_eps[0]
        if e is not None and tuple(self.shape) == e.shape:
            self.copy_(torch.from_numpy(e)); return self
        return orig_normal(self, *a, **kw)

    def l1(a, b, *k, **kw):
        v = orig_l1(a, b, *k, **kw); rec.l1.append(float(v)); return v

    def crit(pred, real, use_sigmoid=True):
        v = orig_crit(pred, real, use_sigmoid=use_sigmoid); rec.gan.append(float(v)); return v

    def clip(params, max_norm, *k, **kw):
        v = orig_clip(params, max_norm, *k, **kw); rec.gn.append(float(v)); return v

    rmodel.F.l1_loss, rmodel.criterion_GAN, torch.nn.utils.clip_grad_norm = l1, crit, clip
    rmodel.log_prob_gaussian = lpg
    m = rmodel.AugmentedCycleGAN(opt, testing=True) if aug else rmodel.StochCycleGAN(opt, testing=True)
    names = ["netG_A_B", "netG_B_A", "netD_A", "netD_B"] + (["netE_B", "netD_z_B"] if aug else [])
    for n in names:
        load_recipe(getattr(m, n), n, seed, flavour)
    pre = {n: {k: p.detach().numpy().copy() for k, p in unique_named_parameters(getattr(m, n)).items()} for n in names}

    def wrap_forward(net, sink, fn):
        f = net.forward

        def g(*a):
            o = f(*a); sink.append(fn(o)); return o
        net.forward = g

    wrap_forward(m.netG_A_B, rec.gAB, lambda o: o.detach().numpy().copy())   # call 0 = fake_B, call 1 = rec_B
    wrap_forward(m.netG_B_A, rec.gBA, lambda o: o.detach().numpy().copy())   # call 0 = fake_A, call 1 = rec_A
    wrap_forward(m.netD_A, rec.predA, lambda o: float(o.mean()))
    wrap_forward(m.netD_B, rec.predB, lambda o: float(o.mean()))
    if aug:
        wrap_forward(m.netE_B, rec.enc, lambda o: (o[0].detach().numpy().copy(), o[1].detach().numpy().copy()))

    arr = {}
    drop = InjectedDropout(drop_seed) if drop_seed is not None else None   # one mask counter over all steps
    if drop is not None:
        drop.__enter__()
    try:
        for st in range(steps):
            A, B, z = recipe.inputs(seed + st, N, opt.input_nc, opt.output_nc, S, opt.nlatent)
            arr["s%d/real_A" % st], arr["s%d/real_B" % st], arr["s%d/prior_z_B" % st] = A, B, z
            for lst in (rec.gan, rec.l1, rec.gn, rec.predA, rec.predB, rec.enc, rec.gAB, rec.gBA, rec.cycz):
                del lst[:]
            if eps_seed is not None:
                cur_eps[0] = np.random.RandomState(eps_seed + st).normal(0, 1, (N, 1, opt.nlatent)).astype(np.float32)
                arr["s%d/eps" % st] = cur_eps[0]
                torch.Tensor.normal_ = normal_
            try:
                run_ref_step(m, rec, A, B, z, aug)
            finally:
                torch.Tensor.normal_ = orig_normal
            # visuals (model.py:524-525 / 199-200): the tensors the reference's own forward calls returned in this step
            arr["s%d/fake_B" % st], arr["s%d/rec_B" % st] = rec.gAB[0], rec.gAB[1]
            arr["s%d/fake_A" % st], arr["s%d/rec_A" % st] = rec.gBA[0], rec.gBA[1]
            if aug:
                # call order (model.py:423-464): D_A f/t, D_B f/t, D_z post/prior, G_A, G_B, G_z
                g = rec.gan
                mu, lv = rec.enc[0]
                if opt.stoch_enc:   # l1 calls: Cyc_A, Cyc_B; Cyc_z_B is the Gaussian NLL (model.py:478-484)
                    cyc_z, cyc_B = rec.cycz[0], rec.l1[1]
                    mu64, lv64 = mu.astype(np.float64), lv.astype(np.float64)
                    kld = float((-0.5 * (lv64 + 1.0 - mu64 ** 2 - np.exp(lv64)).sum(1)).mean())   # model.py:45-53
                else:               # l1 calls: Cyc_A, Cyc_z_B, Cyc_B; logvar is zeroed (model.py:419)
                    cyc_z, cyc_B = rec.l1[1], rec.l1[2]
                    lv = lv * 0.0
                    kld = float((0.5 * (mu.astype(np.float64) ** 2).sum(1)).mean())
                losses = OrderedDict([("D_A", 0.5 * (g[0] + g[1])), ("G_A", g[6]), ("Cyc_A", rec.l1[0]),
                                      ("Cyc_z_B", cyc_z), ("KLD_z_B", kld),
                                      ("D_B", 0.5 * (g[2] + g[3])), ("G_B", g[7]), ("Cyc_B", cyc_B),
                                      ("D_z_B", 0.5 * (g[4] + g[5])),
                                      ("P_t_A", rec.predA[1]), ("P_f_A", rec.predA[2]),
                                      ("P_t_B", rec.predB[1]), ("P_f_B", rec.predB[2])])
                # clip order model.py:447-449, 510-512
                gn = OrderedDict([("gnorm_G_A_B", rec.gn[3]), ("gnorm_G_B_A", rec.gn[4]), ("gnorm_E_B", rec.gn[5]),
                                  ("gnorm_D_B", rec.gn[1]), ("gnorm_D_z_B", rec.gn[2]), ("gnorm_D_A", rec.gn[0]),
                                  ("mu_min", float(mu.min())), ("mu_max", float(mu.max())),
                                  ("logvar_min", float(lv.min())), ("logvar_max", float(lv.max()))])
                arr["s%d/mu_z_realB" % st] = mu
                if opt.stoch_enc:
                    arr["s%d/logvar_z_realB" % st] = lv
            else:
                g = rec.gan  # D_A f/t, D_B f/t, G_A, G_B (model.py:139-171)
                losses = OrderedDict([("D_A", 0.5 * (g[0] + g[1])), ("G_A", g[4]), ("Cyc_A", rec.l1[0]),
                                      ("D_B", 0.5 * (g[2] + g[3])), ("G_B", g[5]), ("Cyc_B", rec.l1[1]),
                                      ("P_t_A", rec.predA[1]), ("P_f_A", rec.predA[2]),
                                      ("P_t_B", rec.predB[1]), ("P_f_B", rec.predB[2])])
                gn = OrderedDict([("gnorm_G_A_B", rec.gn[2]), ("gnorm_G_B_A", rec.gn[3]),
                                  ("gnorm_D_B", rec.gn[1]), ("gnorm_D_A", rec.gn[0])])
            arr["s%d/losses" % st] = np.array(list(losses.values()), np.float64)
            arr["s%d/gnorms" % st] = np.array(list(gn.values()), np.float64)
            loss_keys, gn_keys = list(losses.keys()), list(gn.keys())
            # post-step update digests (post - pre), per tensor, and refresh `pre`
            for n in names:
                for k, p in unique_named_parameters(getattr(m, n)).items():
                    post = p.detach().numpy()
                    # .grad after the step: G nets = clipped G-phase grads; D nets = clipped D-phase grads
                    # plus the (unused) G-phase accumulation on top (model.py:509 — no zero_grad for D there)
                    arr["s%d/grad/%s/%s" % (st, n, k)] = digest(p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape)))
                    arr["s%d/upd/%s/%s" % (st, n, k)] = digest(post.astype(np.float64) - pre[n][k].astype(np.float64))
                    pre[n][k] = post.copy()
        if aug:
            for n in ("netE_B", "netD_z_B"):
                for k, b in getattr(m, n).named_buffers():
                    arr["final/buf/%s/%s" % (n, k)] = b.detach().numpy().copy()
    finally:
        rmodel.F.l1_loss, rmodel.criterion_GAN, torch.nn.utils.clip_grad_norm = orig_l1, orig_crit, orig_clip
        rmodel.log_prob_gaussian = orig_lpg
        if drop is not None:
            drop.__exit__()
    extra = {} if drop_seed is None else {"drop_seed": drop_seed}
    save(name, arr, kind="step", aug=bool(aug), seed=seed, flavour=flavour, N=N, S=S, steps=steps,
         opt={k: v for k, v in opt_kw.items()}, loss_keys=loss_keys, gnorm_keys=gn_keys, **extra)


def make_step_goldens():
    small = dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4)
    step_case("step_aug_small_s64", True, small, N=4, S=64, steps=2, flavour="rich")  # N>=3: BatchNorm over the 1x1 map needs >2 samples to be well conditioned
    step_case("step_aug_small_s64_init", True, small, N=3, S=64, steps=2, flavour="init")
    step_case("step_stoch_small_s64", False, dict(small, input_nc=3, output_nc=1), N=2, S=64, steps=2, flavour="rich")
    # BASELINE config 1 at full reference widths (64x64x1, batch 4, reference-faithful 3 blocks)
    step_case("step_aug_cfg1_full", True, dict(input_nc=1, output_nc=1), N=4, S=64, steps=1, flavour="init")
    # --stoch_enc branch (model.py:15-22, 414-419, 478-484, 501-502) with the reparametrisation noise injected
    step_case("step_aug_small_s64_stoch_enc", True, dict(small, stoch_enc=True), N=4, S=64, steps=2, flavour="init",
              eps_seed=77)
    # --norm batch --use_dropout: BatchNorm2d in G_B_A / D_A / D_B, Dropout(0.5) in every residual block of both generators.
    # One step: BatchNorm gains ~N(1, 0.02) make G_B_A a high-gain network (tests/test_hip_step.py BN_DROPOUT_X3_SKIP), and a
    # second step behind Adam's noise amplification pins nothing there (bf16x3 10 % on gnorm_E_B; step 0 is a pure function)
    step_case("step_aug_small_s64_bn_dropout", True, dict(small, norm="batch", use_dropout=True), N=4, S=64, steps=1,
              flavour="init", drop_seed=55)


if __name__ == "__main__":
    make_net_goldens()
    make_step_goldens()


def sup_case(name, opt_kw, N, S, seed=0, flavour="rich"):
    """AugmentedCycleGAN.supervised_train_instance (model.py:541-604); it too raises IndexError at its reporting
    line after all compute, so values come from recorders (call order: criterionGAN D_z fake/true, then G_z;
    l1 sup_A, sup_B; clips D_z, G_A_B, G_B_A, E)."""
    opt = ref_opt(**opt_kw)
    gan, l1s, gns, encs = [], [], [], []
    orig_l1, orig_clip, orig_crit = rmodel.F.l1_loss, torch.nn.utils.clip_grad_norm, rmodel.criterion_GAN
    rmodel.F.l1_loss = lambda a, b, *k, **kw: (lambda v: (l1s.append(float(v)), v)[1])(orig_l1(a, b, *k, **kw))
    rmodel.criterion_GAN = lambda p, r, use_sigmoid=True: (lambda v: (gan.append(float(v)), v)[1])(orig_crit(p, r, use_sigmoid=use_sigmoid))
    torch.nn.utils.clip_grad_norm = lambda ps, mx, *k, **kw: (lambda v: (gns.append(float(v)), v)[1])(orig_clip(ps, mx, *k, **kw))
    try:
        m = rmodel.AugmentedCycleGAN(opt, testing=True)
        names = ["netG_A_B", "netG_B_A", "netD_A", "netD_B", "netE_B", "netD_z_B"]
        for n in names:
            load_recipe(getattr(m, n), n, seed, flavour)
        f = m.netE_B.forward
        m.netE_B.forward = lambda *a: (lambda o: (encs.append(o[0].detach().numpy().copy()), o)[1])(f(*a))
        A, B, z = recipe.inputs(seed + 40, N, opt.input_nc, opt.output_nc, S, opt.nlatent)
        try:
            m.supervised_train_instance(torch.from_numpy(A.copy()), torch.from_numpy(B.copy()), torch.from_numpy(z.copy()))
            raise RuntimeError("reference unexpectedly returned")
        except IndexError:
            pass
        mu = encs[0].astype(np.float64)
        vals = OrderedDict([("S_A", l1s[0]), ("S_B", l1s[1]), ("KLD_z_B", float((0.5 * (mu ** 2).sum(1)).mean())),
                            ("D_z_B", 0.5 * (gan[0] + gan[1])), ("gnorm_G_A_B", gns[1]), ("gnorm_G_B_A", gns[2]),
                            ("gnorm_E_B", gns[3]), ("gnorm_D_z_B", gns[0])])
        with torch.no_grad():
            probe = recipe.inputs(seed + 41, 2, opt.input_nc, opt.output_nc, S, opt.nlatent)
            post = m.netG_A_B.model(torch.from_numpy(probe[0]), torch.from_numpy(probe[2])).numpy().copy()
        save(name, dict(real_A=A, real_B=B, prior_z_B=z, values=np.array(list(vals.values()), np.float64),
                        probe_A=probe[0], probe_z=probe[2], probe_fake_B_after=post),
             kind="sup", seed=seed, flavour=flavour, N=N, S=S, opt=dict(opt_kw), keys=list(vals.keys()))
    finally:
        rmodel.F.l1_loss, rmodel.criterion_GAN, torch.nn.utils.clip_grad_norm = orig_l1, orig_crit, orig_clip


def make_key_fixture():
    """state_dict keys + shapes of the reference's six networks at default widths."""
    nets = dict(netG_A_B=rnet.define_stochastic_G(16, 3, 3, 32), netG_B_A=rnet.define_G(3, 3, 32),
                netD_A=rnet.define_D_A(3, 32, "basic", "instance"), netD_B=rnet.define_D_B(3, 64, "basic", "instance"),
                netD_z_B=rnet.define_LAT_D(16, 64), netE_B=rnet.define_E(16, 6, 32, "batch"))
    keys = {k: [(a, list(b.shape)) for a, b in n.state_dict().items()] for k, n in nets.items()}
    save("statedict_keys", dict(keys_json=np.frombuffer(json.dumps(keys).encode(), dtype=np.uint8)), kind="keys")


def make_init_fixture(seed=1234):
    """Seed-for-seed initialisation (networks.py:13-21, 47; modules.py:78-81, 145-146): the reference's six networks built
    on the CPU after torch.manual_seed(seed).  `net.apply(weights_init)` visits the convolutions of every CINResnetBlock
    TWICE (the block re-registers its children under numeric names), so the draw sequence depends on the exact module tree;
    the fixture stores a digest of every tensor."""
    arr = {}
    builders = dict(netG_A_B=lambda: rnet.define_stochastic_G(16, 3, 3, 32), netG_B_A=lambda: rnet.define_G(3, 3, 32),
                    netD_A=lambda: rnet.define_D_A(3, 32, "basic", "instance"),
                    netD_B=lambda: rnet.define_D_B(3, 64, "basic", "instance"),
                    netD_z_B=lambda: rnet.define_LAT_D(16, 64), netE_B=lambda: rnet.define_E(16, 6, 32, "batch"))
    for k, mk in builders.items():
        torch.manual_seed(seed)
        for name, v in mk().state_dict().items():
            arr["%s/%s" % (k, name)] = digest(v.detach().numpy())
    save("init_seeded", arr, kind="init", seed=seed)


def eval_case(name, opt_kw, N, S, steps=3, seed=0, flavour="rich"):
    """Evaluation numbers (evaluate.py:10-19 eval_mse_A, evaluate.py:39-148 variational_ubo).  evaluate.py itself is
    Python 2 and hard-codes .cuda(), so it cannot be imported here; the harness walks through its algorithm with the
    reference's OWN model (predict_A / predict_B / predict_enc_params) and model.py helpers (gauss_reparametrize,
    log_prob_laplace, kld_std_guss) on the CPU, with the two random draws — the dequantisation noise and the
    reparametrisation noise of every iterate — fixed and recorded."""
    import math
    opt = ref_opt(**opt_kw)
    m = rmodel.AugmentedCycleGAN(opt, testing=True)
    for n in ["netG_A_B", "netG_B_A", "netD_A", "netD_B", "netE_B", "netD_z_B"]:
        load_recipe(getattr(m, n), n, seed, flavour)
    A, B, _ = recipe.inputs(seed + 50, N, opt.input_nc, opt.output_nc, S, opt.nlatent)
    rs = np.random.RandomState(seed + 51)
    dequant = rs.uniform(0, 1. / 127.5, B.shape).astype(np.float32)
    eps = rs.normal(0, 1, (steps + 1, N, 1, opt.nlatent)).astype(np.float32)
    tA, tB = torch.from_numpy(A), torch.from_numpy(B)
    mse_A = float(torch.nn.functional.mse_loss(m.predict_A(tB), tA))                    # evaluate.py:17-18
    orig_normal, cur = torch.Tensor.normal_, [None]

    def normal_(self, *a, **kw):
        if cur[0] is not None and tuple(self.shape) == cur[0].shape:
            self.copy_(torch.from_numpy(cur[0])); return self
        return orig_normal(self, *a, **kw)
    torch.Tensor.normal_ = normal_
    try:
        npx = opt.output_nc * S * S                                                     # 64*64*3 in evaluate.py:104
        params = m.predict_enc_params(tA, tB)                                           # evaluate.py:56-62
        mu = params[0].detach().clone().requires_grad_(True)
        logvar = torch.full((N, opt.nlatent), math.log(0.01)).requires_grad_(True)
        if len(params) == 2:
            logvar = params[1].detach().clone().requires_grad_(True)
        logvar_B = torch.full((1, opt.output_nc, S, S), math.log(0.01))
        it = torch.optim.RMSprop([mu, logvar], lr=1e-2)                                 # evaluate.py:65
        rB = tB + torch.from_numpy(dequant)
        cur[0] = eps[0]
        z_B = rmodel.gauss_reparametrize(mu, logvar)
        fake_B = m.predict_B(tA, z_B)
        trace = []
        for i in range(steps):                                                          # evaluate.py:93-126
            log_prob = rmodel.log_prob_laplace(rB, fake_B, logvar_B).view(N, -1).sum(1)
            kld = rmodel.kld_std_guss(mu, logvar)
            ubo = (-log_prob + kld) + npx * math.log(127.5)
            trace.append([float(ubo.mean(0)), float(kld.mean(0)), float(ubo.mean(0)) / (npx * math.log(2.))])
            it.zero_grad()
            ubo.mean(0).backward()
            it.step()
            cur[0] = eps[i + 1]
            z_B = rmodel.gauss_reparametrize(mu, logvar)
            fake_B = m.predict_B(tA, z_B)
    finally:
        torch.Tensor.normal_ = orig_normal
    save(name, dict(real_A=A, real_B=B, dequant=dequant, eps=eps, mse_A=np.array(mse_A), trace=np.array(trace, np.float64),
                    mu_final=mu.detach().numpy().copy(), logvar_final=logvar.detach().numpy().copy()),
         kind="eval", seed=seed, flavour=flavour, N=N, S=S, steps=steps, opt=dict(opt_kw))


def _py3_source(path):
    """Read a Python-2 reference file as TEXT and make it executable under Python 3 without changing its arithmetic:
    print statements -> print(), and the integer division of the iterators' batch counts (`num_samples / batch_size` on
    ints is floor division in Python 2) -> //.  The transformed text exists only in this process."""
    import re
    src = open(path).read()
    src = re.sub(r'^(\s*)print (.+)$', r'\1print(\2)', src, flags=re.M)
    src = src.replace("self.num_samples / batch_size", "self.num_samples // batch_size")
    src = src.replace("self.num_samples / self.batch_size", "self.num_samples // self.batch_size")
    return src


def data_case(name="data_pipeline"):
    """Data path (dataloader.py:13-59 load_numpy_data, 61-156 Aligned/UnalignedIterator).  The reference file is Python 2
    and imports skimage / torchvision (absent): it is executed from its text through _py3_source with stub modules for
    the two imports.  Pinned: NaN handling, channel selection, per-sample/channel min-max, layout, dev split, iterator
    batch counts and the last-batch wrap.  NOT pinnable here: skimage.transform.resize (not installed) and Python 2's
    random.shuffle permutation for seed 123 (Python 3's differs) — the fixture runs with grid_size=None, shuffle=False."""
    import tempfile
    import types
    for modname in ("torchvision", "torchvision.transforms", "skimage", "skimage.transform"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules["skimage.transform"].resize = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("skimage is not installed"))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    ns = {"__name__": "ref_dataloader"}
    exec(compile(_py3_source(os.path.join(REF, "dataloader.py")), "dataloader.py(py3 view)", "exec"), ns)
    rs = np.random.RandomState(3)
    n_tr, n_te = 206, 5                                      # DEV_SIZE = 200 samples go to the dev split
    raw = dict(trainA=rs.normal(3, 2, (n_tr, 6, 5, 4)), trainB=rs.gamma(2.0, 1.5, (n_tr, 6, 5, 1)),
               testA=rs.normal(-1, 4, (n_te, 6, 5, 4)), testB=rs.gamma(1.0, 1.0, (n_te, 6, 5, 1)))
    raw["trainA"][0, 0, 0, 0] = np.nan
    raw["trainA"][1, :, :, 2] = 7.0                          # constant plane: 0/0 in the min-max
    raw["testB"][2] = 0.0
    root = tempfile.mkdtemp()
    for k, v in raw.items():
        np.savez(os.path.join(root, k + ".npz"), data=v)
    out = ns["load_numpy_data"](root, shuffle=False, grid_size=None)
    arr = {"raw/" + k: v for k, v in raw.items()}
    for k, v in zip(("trainA", "trainB", "devA", "devB", "testA", "testB"), out):
        arr["out/" + k] = v
    # iterators: numpy's global RNG decides the permutations
    A = np.arange(10, dtype=np.float32).reshape(10, 1, 1, 1)
    np.random.seed(11)
    un = ns["UnalignedIterator"](A, -A, batch_size=4)
    batches = []
    for _ in range(2):                                       # two epochs: StopIteration resets
        while True:
            try:
                b = un.next()
            except StopIteration:
                break
            batches.append(np.stack([b["A"].numpy().ravel(), b["B"].numpy().ravel()]))
    arr["unaligned_batches"] = np.stack(batches)
    al = ns["AlignedIterator"](A, -A, batch_size=4)
    sizes = []
    while True:
        try:
            sizes.append(al.next()["A"].shape[0])
        except StopIteration:
            break
    arr["aligned_sizes"] = np.array(sizes)
    save(name, arr, kind="data")


if __name__ == "__main__" and not ONLY:
    make_key_fixture()
    make_init_fixture()
    eval_case("eval_aug_small_s64", dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4), N=4, S=64)
    data_case()
    sup_case("sup_aug_small_s64", dict(input_nc=3, output_nc=3, ngf=8, nef=8, ndf=8, nlatent=4), N=4, S=64)
