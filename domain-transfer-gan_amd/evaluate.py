"""Evaluation — Py3 counterpart of /root/reference/augmented_cyclegan/evaluate.py:10-148: B->A MSE and the
variational upper bound / bits-per-pixel on B (RMSprop on per-sample (mu, logvar) THROUGH model.predict_B).
The reference hard-codes 64*64*3 (evaluate.py:52,104,107); here it is C*H*W of the batch.  Generator forwards and
the gradient w.r.t. the latent run on the HIP kernels; the scalar bookkeeping of the bound is plain torch."""
import math

import numpy as np
import torch

from .model import gauss_reparametrize, kld_std_guss, log_prob_laplace


def eval_mse_A(dataset, model, use_gpu=True):
    """evaluate.py:10-19"""
    mse_A = []
    for batch in dataset:
        real_A, real_B = batch['A'], batch['B']
        if use_gpu:
            real_A, real_B = real_A.cuda(), real_B.cuda()
        with torch.no_grad():
            pred_A = model.predict_A(real_B)
        mse_A.append(float(((pred_A - real_A) ** 2).mean()))
    return float(np.mean(mse_A))


class _frozen(object):
    """no weight gradients while optimising the latent (the reference lets autograd compute and discard them)"""

    def __init__(self, net):
        self.params = [p for p in net.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p in self.params:
            p.requires_grad_(True)


def variational_ubo(model, real_A, real_B, steps, logvar_B=None, verbose=False, dequant=None, eps_seq=None, trace=None):
    """dequant / eps_seq / trace are test hooks (not in the reference): the dequantisation noise, the reparametrisation
    noise of iterate i (eps_seq[i], shape (N, 1, nlatent)) and a list receiving (ubo, kld, bpp) of every iterate — the
    golden `eval_aug_small_s64` was produced by the reference's model with exactly these draws."""
    with _frozen(model.netG_A_B):
        return _variational_ubo(model, real_A, real_B, steps, logvar_B, verbose, dequant, eps_seq, trace)


def _reparametrize(mu, logvar, eps):
    if eps is None:
        return gauss_reparametrize(mu, logvar)
    z = eps.mul(logvar.mul(0.5).exp()[:, None, :]).add(mu[:, None, :]).clamp(-4., 4.)      # model.py:15-22 with a given eps
    return z.view(z.size(0) * z.size(1), z.size(2), 1, 1)


def _variational_ubo(model, real_A, real_B, steps, logvar_B=None, verbose=False, dequant=None, eps_seq=None, trace=None):
    """evaluate.py:39-148 without the PNG dumps.  Returns (ubo, kld, bpp) of the LAST evaluated iterate."""
    size = real_A.size()
    nl = model.opt.nlatent
    npx = real_B[0].numel()
    dev = real_A.device
    if dequant is None:
        dequant = torch.zeros_like(real_B).uniform_(0, 1. / 127.5)
    mu = torch.zeros(size[0], nl, device=dev, requires_grad=True)
    logvar = torch.full((size[0], nl), math.log(0.01), device=dev, requires_grad=True)
    if logvar_B is None:
        logvar_B = torch.full((1,) + tuple(real_B.shape[1:]), math.log(0.01), device=dev)
    if hasattr(model, 'netE_B'):
        with torch.no_grad():
            params = model.predict_enc_params(real_A, real_B)
        mu = params[0].detach().clone().requires_grad_(True)
        if len(params) == 2:
            logvar = params[1].detach().clone().requires_grad_(True)
    opt = torch.optim.RMSprop([mu, logvar], lr=1e-2)
    real_B = real_B + dequant
    ubo_val = kld_val = bpp = float('nan')
    for i in range(steps):
        z_B = _reparametrize(mu, logvar, None if eps_seq is None else eps_seq[i])
        fake_B = model.predict_B(real_A, z_B)
        log_prob = log_prob_laplace(real_B, fake_B, logvar_B).view(size[0], -1).sum(1)
        kld = kld_std_guss(mu, logvar)
        ubo = (-log_prob + kld) + npx * math.log(127.5)
        ubo_val, kld_val = float(ubo.detach().mean(0)), float(kld.detach().mean(0))
        bpp = ubo_val / (npx * math.log(2.))
        if trace is not None:
            trace.append((ubo_val, kld_val, bpp))
        if verbose:
            print('[%d] UBO: %.4f, KLD: %.4f, BPP: %.4f' % (i, ubo_val, kld_val, bpp))
        opt.zero_grad()
        ubo.mean(0).backward()
        opt.step()
    return ubo_val, kld_val, bpp


def eval_ubo_B(dataset, model, steps=500, use_gpu=True, logvar_B=None, verbose=False):
    """evaluate.py:21-37 -> (mean ubo, mean bpp, mean kld)"""
    ubo_B, bpp_B, kld_B = [], [], []
    for batch in dataset:
        real_A, real_B = batch['A'], batch['B']
        if use_gpu:
            real_A, real_B = real_A.cuda(), real_B.cuda()
        ubo, kld, bpp = variational_ubo(model, real_A, real_B, steps, logvar_B, verbose)
        ubo_B.append(ubo); bpp_B.append(bpp); kld_B.append(kld)
    return float(np.mean(ubo_B)), float(np.mean(bpp_B)), float(np.mean(kld_B))


def one_to_three_channels(img):
    """evaluate.py:155-161"""
    if img.size(1) == 1:
        z = torch.zeros_like(img)
        return torch.cat((img.float(), z, z), dim=1)
    return img
