"""ORACLE (test infrastructure): the training step, restated.

AugStep.train_instance follows /root/reference/augmented_cyclegan/model.py:402-539
line by line; StochStep.train_instance follows model.py:126-208.  Adam and
clip_grad_norm restate the torch semantics the reference calls at
model.py:379-389, 447-452, 510-515.
"""
from collections import OrderedDict

import numpy as np

from . import nets, ops
from .tape import T, backward, zero_grad


class Opt(object):
    """Reference defaults — options.py:28-85 as recorded in SURVEY.md §5."""

    def __init__(self, **kw):
        d = dict(input_nc=3, output_nc=3, ngf=32, nef=32, ndf=64, nlatent=16, n_blocks=3,
                 lr=2e-4, beta1=0.5, max_gnorm=500.0, lambda_A=1.0, lambda_B=1.0, lambda_z_B=0.025,
                 lambda_sup_A=0.1, lambda_sup_B=0.1,
                 stoch_enc=False, z_gan=1, enc_A_B=1, niter_decay=25, norm="instance", use_dropout=False)
        d.update(kw)
        self.__dict__.update(d)


class Adam(object):
    """torch.optim.Adam(lr, betas=(beta1, 0.999), eps=1e-8), no weight decay / amsgrad:
         m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
         p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)"""

    def __init__(self, params, lr, beta1, beta2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, beta1, beta2, eps
        self.t = 0
        self.m = [np.zeros_like(p.v) for p in self.params]
        self.s = [np.zeros_like(p.v) for p in self.params]

    def step(self):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for i, p in enumerate(self.params):
            if p.g is None:
                continue
            g = p.g
            self.m[i] = (self.b1 * self.m[i] + (1 - self.b1) * g).astype(p.v.dtype)
            self.s[i] = (self.b2 * self.s[i] + (1 - self.b2) * g * g).astype(p.v.dtype)
            denom = np.sqrt(self.s[i]) / np.sqrt(bc2) + self.eps
            p.v = (p.v - (self.lr / bc1) * self.m[i] / denom).astype(p.v.dtype)


def clip_grad_norm(params, max_norm):
    """torch.nn.utils.clip_grad_norm: total L2 norm over all grads; if it exceeds
    max_norm scale every grad by max_norm/(norm+1e-6).  Returns the norm."""
    tot = 0.0
    for p in params:
        if p.g is not None:
            tot += float(np.sum(p.g.astype(np.float64) ** 2))
    norm = float(np.sqrt(tot))
    coef = max_norm / (norm + 1e-6)
    if coef < 1.0:
        for p in params:
            if p.g is not None:
                p.g = (p.g * coef).astype(p.g.dtype)
    return norm


def criterion_gan(pred, target_is_real):
    """model.py:56-72, LSGAN branch (use_sigmoid False is the default and the only
    branch that runs on modern torch)."""
    return ops.mse_to_const(pred, 1.0 if target_is_real else 0.0)


def discriminate(net, fake, real):
    """model.py:327-334."""
    pf = net.forward(fake)
    lf = criterion_gan(pf, False)
    pt = net.forward(real)
    lt = criterion_gan(pt, True)
    return lf, lt, pf, pt


def _f(t):
    return float(t.v)


def _half_sum(a, b):
    return ops.scale(ops.add(a, b), 0.5)


class AugStep(object):
    """model.py:337-539 AugmentedCycleGAN (constructor + train_instance)."""

    def __init__(self, opt, dtype=np.float32):
        self.opt, self.dtype = opt, np.dtype(dtype)
        o = opt
        # model.py:343-371: opt.norm reaches G_B_A and both image discriminators (G_A_B is always CondInstanceNorm,
        # E_B always 'batch'); opt.use_dropout both generators
        self.netG_A_B = nets.CINResnetGenerator(o.nlatent, o.input_nc, o.output_nc, o.ngf, o.n_blocks, dtype,
                                                use_dropout=o.use_dropout)
        self.netG_B_A = nets.ResnetGenerator(o.output_nc, o.input_nc, o.ngf, o.n_blocks, dtype, norm=o.norm,
                                             use_dropout=o.use_dropout)
        enc_nc = o.output_nc + (o.input_nc if o.enc_A_B else 0)                       # model.py:360-362
        self.netE_B = nets.LatentEncoder(o.nlatent, enc_nc, o.nef, dtype)
        self.netD_A = nets.Discriminator_edges(o.input_nc, 32, dtype, norm=o.norm)    # model.py:366-367: ndf=32
        self.netD_B = nets.Discriminator(o.output_nc, o.ndf, dtype, norm=o.norm)
        self.netD_z_B = nets.DiscriminatorLatent(o.nlatent, o.ndf, dtype)
        self._make_optimizers()

    def _make_optimizers(self):
        o = self.opt
        self.optimizer_G_A = Adam(self.netG_B_A.parameters(), o.lr, o.beta1)          # model.py:379-380
        self.optimizer_G_B = Adam(self.netG_A_B.parameters() + self.netE_B.parameters(), o.lr, o.beta1)
        self.optimizer_D_A = Adam(self.netD_A.parameters(), o.lr / 5.0, o.beta1)
        self.optimizer_D_B = Adam(self.netD_B.parameters() + self.netD_z_B.parameters(), o.lr / 5.0, o.beta1)

    def set_dropout_seed(self, seed):
        """--use_dropout fixtures: the k-th Dropout forward of the step sequence (both generators share the counter, in
        the reference's call order model.py:404, 407, 467, 493) takes ops.dropout_keep(seed, k, shape)"""
        k = [0]

        def src(shape):
            m = ops.dropout_keep(seed, k[0], shape)
            k[0] += 1
            return m
        self.netG_A_B.drop = self.netG_B_A.drop = src

    def nets(self):
        return OrderedDict([("netG_A_B", self.netG_A_B), ("netG_B_A", self.netG_B_A), ("netE_B", self.netE_B),
                            ("netD_A", self.netD_A), ("netD_B", self.netD_B), ("netD_z_B", self.netD_z_B)])

    def load(self, values_by_net):
        for k, n in self.nets().items():
            n.load(values_by_net[k])
        self._make_optimizers()

    def train_instance(self, real_A, real_B, prior_z_B, eps=None):
        o = self.opt
        A = T(np.asarray(real_A, self.dtype))
        B = T(np.asarray(real_B, self.dtype))
        Z = T(np.asarray(prior_z_B, self.dtype))
        bs = A.v.shape[0]

        fake_B = self.netG_A_B.forward(A, Z)                                          # model.py:404
        fake_A = self.netG_B_A.forward(B)                                             # model.py:407
        enc_in = ops.cat_channels(fake_A, B) if o.enc_A_B else B                      # model.py:409-413
        mu_rB, lv_rB = self.netE_B.forward(enc_in)
        if o.stoch_enc:
            post_z = ops.gauss_reparametrize(mu_rB, lv_rB, eps)                       # model.py:416
        else:
            post_z = ops.reshape(mu_rB, (bs, mu_rB.v.shape[1], 1, 1))                 # model.py:418
            lv_rB = ops.scale(lv_rB, 0.0)                                             # model.py:419

        # ---- D phase: model.py:423-452
        lfA, ltA, pfA, ptA = discriminate(self.netD_A, fake_A.detach(), A)
        lfB, ltB, pfB, ptB = discriminate(self.netD_B, fake_B.detach(), B)
        lfz, ltz, pfz, ptz = discriminate(self.netD_z_B, post_z.detach(), Z)
        loss_D_A, loss_D_B, loss_D_z_B = _half_sum(lfA, ltA), _half_sum(lfB, ltB), _half_sum(lfz, ltz)
        loss_D = ops.add(loss_D_A, loss_D_B)
        if o.z_gan and not o.stoch_enc:
            loss_D = ops.add(loss_D, loss_D_z_B)
        zero_grad(self.optimizer_D_A.params); zero_grad(self.optimizer_D_B.params)
        backward(loss_D)
        gn_D_A = clip_grad_norm(self.netD_A.parameters(), o.max_gnorm)
        gn_D_B = clip_grad_norm(self.netD_B.parameters(), o.max_gnorm)
        gn_D_z = clip_grad_norm(self.netD_z_B.parameters(), o.max_gnorm)
        self.optimizer_D_A.step(); self.optimizer_D_B.step()

        # ---- G phase (updated D weights): model.py:457-515
        pfA = self.netD_A.forward(fake_A); loss_G_A = criterion_gan(pfA, True)
        pfB = self.netD_B.forward(fake_B); loss_G_B = criterion_gan(pfB, True)
        ppz = self.netD_z_B.forward(post_z); loss_G_z = criterion_gan(ppz, True)
        rec_A = self.netG_B_A.forward(fake_B)
        loss_cyc_A = ops.l1_loss(rec_A, A)
        enc_in2 = ops.cat_channels(A, fake_B) if o.enc_A_B else fake_B                # model.py:471-475
        mu_fB, lv_fB = self.netE_B.forward(enc_in2)
        zflat = ops.reshape(Z, (bs, o.nlatent))
        if o.stoch_enc:
            lp = ops.log_prob_gaussian(zflat, mu_fB, lv_fB)
            loss_cyc_z = ops.scale(ops.mean_all(lp), -1.0)                            # mean(1).mean(0) == mean over all
        else:
            loss_cyc_z = ops.l1_loss(mu_fB, zflat)                                    # model.py:486-487
        kld = ops.mean0(ops.kld_std_gauss(mu_rB, lv_rB))                              # model.py:490
        rec_B = self.netG_A_B.forward(fake_A, post_z)                                 # model.py:493
        loss_cyc_B = ops.l1_loss(rec_B, B)
        loss_G = ops.add(ops.add(loss_G_A, loss_G_B),
                         ops.add(ops.add(ops.scale(loss_cyc_A, o.lambda_A), ops.scale(loss_cyc_B, o.lambda_B)),
                                 ops.scale(loss_cyc_z, o.lambda_z_B)))
        if o.stoch_enc:
            loss_G = ops.add(loss_G, ops.scale(kld, o.lambda_z_B))
        if o.z_gan and not o.stoch_enc:
            loss_G = ops.add(loss_G, loss_G_z)
        zero_grad(self.optimizer_G_A.params); zero_grad(self.optimizer_G_B.params)
        # model.py:509: autograd also deposits (unused) grads in the discriminators here;
        # they are zeroed at the next step (model.py:442-443) before use, so the oracle
        # simply lets them accumulate on the D leaves too — same observable behaviour.
        backward(loss_G)
        gn_G_A_B = clip_grad_norm(self.netG_A_B.parameters(), o.max_gnorm)
        gn_G_B_A = clip_grad_norm(self.netG_B_A.parameters(), o.max_gnorm)
        gn_E = clip_grad_norm(self.netE_B.parameters(), o.max_gnorm)
        self.optimizer_G_A.step(); self.optimizer_G_B.step()

        losses = OrderedDict([("D_A", _f(loss_D_A)), ("G_A", _f(loss_G_A)), ("Cyc_A", _f(loss_cyc_A)),
                              ("Cyc_z_B", _f(loss_cyc_z)), ("KLD_z_B", _f(kld)),
                              ("D_B", _f(loss_D_B)), ("G_B", _f(loss_G_B)), ("Cyc_B", _f(loss_cyc_B)),
                              ("D_z_B", _f(loss_D_z_B)),
                              ("P_t_A", float(ptA.v.mean())), ("P_f_A", float(pfA.v.mean())),
                              ("P_t_B", float(ptB.v.mean())), ("P_f_B", float(pfB.v.mean()))])   # model.py:518-523
        visuals = OrderedDict([("real_A", A.v), ("fake_B", fake_B.v), ("rec_A", rec_A.v),
                               ("real_B", B.v), ("fake_A", fake_A.v), ("rec_B", rec_B.v)])
        gnorms = OrderedDict([("gnorm_G_A_B", gn_G_A_B), ("gnorm_G_B_A", gn_G_B_A), ("gnorm_E_B", gn_E),
                              ("gnorm_D_B", gn_D_B), ("gnorm_D_z_B", gn_D_z), ("gnorm_D_A", gn_D_A),
                              ("mu_min", float(mu_rB.v.min())), ("mu_max", float(mu_rB.v.max())),
                              ("logvar_min", float(lv_rB.v.min())), ("logvar_max", float(lv_rB.v.max()))])
        return losses, visuals, gnorms


def _sup(self, real_A, real_B, prior_z_B, eps=None):
    """model.py:541-604 supervised_train_instance (paired step, --supervised)."""
    o = self.opt
    A = T(np.asarray(real_A, self.dtype)); B = T(np.asarray(real_B, self.dtype)); Z = T(np.asarray(prior_z_B, self.dtype))
    bs = A.v.shape[0]
    mu, lv = self.netE_B.forward(ops.cat_channels(A, B) if o.enc_A_B else B)                 # model.py:543-547
    if o.stoch_enc:
        post_z = ops.gauss_reparametrize(mu, lv, eps)
    else:
        post_z = ops.reshape(mu, (bs, mu.v.shape[1], 1, 1))
        lv = ops.scale(lv, 0.0)
    lf, lt, _, _ = discriminate(self.netD_z_B, post_z.detach(), Z)                            # model.py:555-557
    loss_D_z_B = _half_sum(lf, lt)
    zero_grad(self.optimizer_D_B.params)
    backward(loss_D_z_B)
    gn_D_z = clip_grad_norm(self.netD_z_B.parameters(), o.max_gnorm)
    self.optimizer_D_B.step()
    pred_B = self.netG_A_B.forward(A, post_z)
    pred_A = self.netG_B_A.forward(B)
    loss_sup_A = ops.l1_loss(pred_A, A); loss_sup_B = ops.l1_loss(pred_B, B)
    loss_G_z = criterion_gan(self.netD_z_B.forward(post_z), True)
    kld = ops.mean0(ops.kld_std_gauss(mu, lv))
    loss_G = ops.add(ops.scale(loss_sup_A, o.lambda_sup_A), ops.scale(loss_sup_B, o.lambda_sup_B))
    if o.stoch_enc:
        loss_G = ops.add(loss_G, ops.scale(kld, o.lambda_z_B))
    if o.z_gan and not o.stoch_enc:
        loss_G = ops.add(loss_G, loss_G_z)
    zero_grad(self.optimizer_G_A.params); zero_grad(self.optimizer_G_B.params)
    backward(loss_G)
    gn_G_A_B = clip_grad_norm(self.netG_A_B.parameters(), o.max_gnorm)
    gn_G_B_A = clip_grad_norm(self.netG_B_A.parameters(), o.max_gnorm)
    gn_E = clip_grad_norm(self.netE_B.parameters(), o.max_gnorm)
    self.optimizer_G_A.step(); self.optimizer_G_B.step()
    return OrderedDict([("S_A", _f(loss_sup_A)), ("S_B", _f(loss_sup_B)), ("KLD_z_B", _f(kld)), ("D_z_B", _f(loss_D_z_B)),
                        ("gnorm_G_A_B", gn_G_A_B), ("gnorm_G_B_A", gn_G_B_A), ("gnorm_E_B", gn_E),
                        ("gnorm_D_z_B", gn_D_z)])                                              # model.py:596-604


AugStep.supervised_train_instance = _sup


class StochStep(object):
    """model.py:75-208 StochCycleGAN (no encoder / latent discriminator; works at any S)."""

    def __init__(self, opt, ignore_noise=False, dtype=np.float32):
        self.opt, self.dtype, self.ignore_noise = opt, np.dtype(dtype), ignore_noise
        o = opt
        self.netG_A_B = nets.CINResnetGenerator(o.nlatent, o.input_nc, o.output_nc, o.ngf, o.n_blocks, dtype)
        self.netG_B_A = nets.ResnetGenerator(o.output_nc, o.input_nc, o.ngf, o.n_blocks, dtype)
        self.netD_A = nets.Discriminator_edges(o.input_nc, 32, dtype)
        self.netD_B = nets.Discriminator(o.output_nc, o.ndf, dtype)
        self._make_optimizers()

    def _make_optimizers(self):
        o = self.opt
        self.optimizer_G = Adam(self.netG_A_B.parameters() + self.netG_B_A.parameters(), o.lr, o.beta1)
        self.optimizer_D = Adam(self.netD_A.parameters() + self.netD_B.parameters(), o.lr / 5.0, o.beta1)

    def nets(self):
        return OrderedDict([("netG_A_B", self.netG_A_B), ("netG_B_A", self.netG_B_A),
                            ("netD_A", self.netD_A), ("netD_B", self.netD_B)])

    def load(self, values_by_net):
        for k, n in self.nets().items():
            n.load(values_by_net[k])
        self._make_optimizers()

    def train_instance(self, real_A, real_B, prior_z_B):
        o = self.opt
        A = T(np.asarray(real_A, self.dtype)); B = T(np.asarray(real_B, self.dtype))
        Z = T(np.asarray(prior_z_B, self.dtype))
        if self.ignore_noise:
            Z = T(np.ones_like(Z.v))                                                  # model.py:128-129
        fake_B = self.netG_A_B.forward(A, Z)
        fake_A = self.netG_B_A.forward(B)
        lfA, ltA, pfA, ptA = discriminate(self.netD_A, fake_A.detach(), A)
        lfB, ltB, pfB, ptB = discriminate(self.netD_B, fake_B.detach(), B)
        loss_D_A, loss_D_B = _half_sum(lfA, ltA), _half_sum(lfB, ltB)
        loss_D = ops.add(loss_D_A, loss_D_B)
        zero_grad(self.optimizer_D.params)
        backward(loss_D)
        gn_D_A = clip_grad_norm(self.netD_A.parameters(), o.max_gnorm)
        gn_D_B = clip_grad_norm(self.netD_B.parameters(), o.max_gnorm)
        self.optimizer_D.step()
        pfA = self.netD_A.forward(fake_A); loss_G_A = criterion_gan(pfA, True)
        pfB = self.netD_B.forward(fake_B); loss_G_B = criterion_gan(pfB, True)
        rec_A = self.netG_B_A.forward(fake_B); loss_cyc_A = ops.l1_loss(rec_A, A)
        rec_B = self.netG_A_B.forward(fake_A, Z); loss_cyc_B = ops.l1_loss(rec_B, B)
        loss_G = ops.add(ops.add(loss_G_A, loss_G_B),
                         ops.add(ops.scale(loss_cyc_A, o.lambda_A), ops.scale(loss_cyc_B, o.lambda_B)))
        zero_grad(self.optimizer_G.params)
        backward(loss_G)
        gn_G_A_B = clip_grad_norm(self.netG_A_B.parameters(), o.max_gnorm)
        gn_G_B_A = clip_grad_norm(self.netG_B_A.parameters(), o.max_gnorm)
        self.optimizer_G.step()
        losses = OrderedDict([("D_A", _f(loss_D_A)), ("G_A", _f(loss_G_A)), ("Cyc_A", _f(loss_cyc_A)),
                              ("D_B", _f(loss_D_B)), ("G_B", _f(loss_G_B)), ("Cyc_B", _f(loss_cyc_B)),
                              ("P_t_A", float(ptA.v.mean())), ("P_f_A", float(pfA.v.mean())),
                              ("P_t_B", float(ptB.v.mean())), ("P_f_B", float(pfB.v.mean()))])
        visuals = OrderedDict([("real_A", A.v), ("fake_B", fake_B.v), ("rec_A", rec_A.v),
                               ("real_B", B.v), ("fake_A", fake_A.v), ("rec_B", rec_B.v)])
        gnorms = OrderedDict([("gnorm_G_A_B", gn_G_A_B), ("gnorm_G_B_A", gn_G_B_A),
                              ("gnorm_D_B", gn_D_B), ("gnorm_D_A", gn_D_A)])
        return losses, visuals, gnorms
