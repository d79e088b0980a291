"""Training-step orchestration — the API surface of /root/reference/augmented_cyclegan/model.py
(StochCycleGAN, AugmentedCycleGAN with train_instance / supervised_train_instance / generate_* /
predict_* / save / load / update_learning_rate / eval / train, plus the loss helpers), running on
the HIP kernels of libacgan_hip.so.

What differs from the reference, by design:
  * train_instance returns python floats exactly like the reference, but gathers all 13 losses and
    the monitors with ONE device->host copy per step instead of 19 syncs;
  * per-network clip_grad_norm + Adam are one fused kernel per network on flat buffers, the clip
    coefficient is read on the device;
  * in the G phase the discriminators' (discarded) weight gradients are not computed;
  * multi-GPU = one process per GPU, gradients averaged by RCCL all-reduce before clipping (dist.py).
Aliases for the north-star names: AugmentedCycleGAN_Model, .optimize_parameters(), .set_input().
"""
import functools  # noqa: F401
import math
import os
from collections import OrderedDict

import numpy as np  # noqa: F401
import torch

from . import dist as acg_dist
from . import networks, ops
from .modules import as_latent, mark_dirty, packed_of, repack
from .ops import cpad


# ------------------------------------------------------------------------------------------------
# loss helpers (model.py:15-72) — tensor functions, autograd-capable, device-agnostic
# ------------------------------------------------------------------------------------------------
def gauss_reparametrize(mu, logvar, n_sample=1):
    """model.py:15-22"""
    std = logvar.mul(0.5).exp()
    size = std.size()
    eps = std.new_empty((size[0], n_sample, size[1])).normal_()
    z = eps.mul(std[:, None, :]).add(mu[:, None, :])
    z = torch.clamp(z, -4., 4.)
    return z.view(z.size(0) * z.size(1), z.size(2), 1, 1)


def log_prob_laplace(z, mu, log_var):
    """model.py:24-28"""
    sd = torch.exp(0.5 * log_var)
    res = - 0.5 * log_var - (torch.abs(z - mu) / sd)
    return res + (-math.log(2))


def log_prob_gaussian(z, mu, log_var):
    """model.py:31-34"""
    res = - 0.5 * log_var - ((z - mu) ** 2.0 / (2.0 * torch.exp(log_var)))
    return res - 0.5 * math.log(2 * math.pi)


def kld_std_guss(mu, log_var):
    """model.py:45-53"""
    return -0.5 * torch.sum(log_var + 1. - mu ** 2 - torch.exp(log_var), dim=1)


def criterion_GAN(pred, target_is_real, use_sigmoid=True):
    """model.py:56-72.  `pred` is an NCHW / (N,1) tensor (public API form)."""
    if use_sigmoid:
        raise NotImplementedError("use_sigmoid (--no_lsgan): the reference's BCE branch builds a Long target and fails "
                                  "on modern torch (model.py:59-63); only LSGAN is implemented")
    t = 1.0 if target_is_real else 0.0
    p = pred.reshape(-1, 1).contiguous()
    return ops.MseConst.apply(_pad_cols(p, 4), 1, t)


def _pad_cols(p, Cp):
    out = p.new_zeros((p.shape[0], Cp))
    out[:, :p.shape[1]] = p
    return out


def _gan_loss(pred_c16, target_is_real):
    """LSGAN on an internal C16 prediction map (1 valid channel)."""
    return ops.MseConst.apply(pred_c16, 1, 1.0 if target_is_real else 0.0)


# ------------------------------------------------------------------------------------------------
# flat parameter storage + fused clip/Adam
# ------------------------------------------------------------------------------------------------
class FlatNet(object):
    """All parameters (and their .grad) of one network as views into single flat fp32 buffers —
    what the fused l2-norm / clip / Adam kernels and the RCCL all-reduce operate on."""

    def __init__(self, net):
        self.net = net
        self.params = [p for p in net.parameters()]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.n = n
        self.p = torch.zeros(n, device=dev, dtype=torch.float32)
        # the gradient buffer carries dist.SCALAR_TAIL extra floats: the all-reduce of the buffer also averages the
        # step's reported scalars written there (no separate collective, no host sync)
        self.g = torch.zeros(n + acg_dist.SCALAR_TAIL, device=dev, dtype=torch.float32)
        self.gv, self.gtail = self.g[:n], self.g[n:]
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offs = offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                self.p[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = self.p[o:o + p.numel()].view(p.shape)
                p.grad = self.g[o:o + p.numel()].view(p.shape)
                p._acg_direct_grad = True    # ops: weight-gradient kernels add straight into this view
        self.sumsq = torch.zeros((), device=dev, dtype=torch.float32)
        mark_dirty(net)
        self._exchange = None
        if acg_dist.exchange_on():
            acg_dist.hook_params(self)

    def check(self):
        p0 = self.params[0]
        if p0.data_ptr() != self.p.data_ptr() or p0.grad is None or p0.grad.data_ptr() != self.g.data_ptr():
            raise RuntimeError("network parameters were re-allocated after the model was built (e.g. .cuda()/.to()); "
                               "build the model on its final device")

    def zero_grad(self):
        self.g.zero_()

    def set_requires_grad(self, flag):
        for p in self.params:
            p.requires_grad_(flag)


class FusedAdam(object):
    """torch.optim.Adam(lr, betas=(beta1, 0.999)) over one or more FlatNets, with the reference's
    per-network clip_grad_norm folded in (model.py:379-389, 447-452, 510-515).
    Exposes `param_groups` (update_learning_rate mutates 'lr') and torch-compatible state dicts."""

    def __init__(self, flats, lr, betas, eps=1e-8):
        self.flats = list(flats)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)]
        self.t = 0
        self.t_dev = None      # int32 device copy of t, read by the kernels of a captured step graph (StepGraph)
        self.dev_step = False  # True only while StepGraph captures: the recorded launches take the step from t_dev

    def zero_grad(self):
        for f in self.flats:
            f.zero_grad()

    def clip_and_step(self, max_norm):
        """per network: sumsq -> (device) clip coefficient -> Adam.  Returns the sumsq scalars."""
        g = self.param_groups[0]
        self.t += 1
        ops.clip_adam_multi([(f.p, f.gv, f.m, f.v, f.sumsq) for f in self.flats], max_norm, g['lr'], g['betas'][0],
                            g['betas'][1], g['eps'], self.t, self.t_dev if self.dev_step else None)
        for f in self.flats:   # the packed copies follow the parameters: one launch per network (modules.repack)
            repack(f.net)
        return [f.sumsq for f in self.flats]

    def state_dict(self):
        state, idx = {}, 0
        for f in self.flats:
            for p, o in zip(f.params, f.offs):
                n = p.numel()
                state[idx] = dict(step=torch.tensor(float(self.t)), exp_avg=f.m[o:o + n].view(p.shape).clone(),
                                  exp_avg_sq=f.v[o:o + n].view(p.shape).clone())
                idx += 1
        pg = dict(self.param_groups[0])
        pg['params'] = list(range(idx))
        return dict(state=state, param_groups=[pg])

    def load_state_dict(self, sd):
        idx = 0
        for f in self.flats:
            for p, o in zip(f.params, f.offs):
                st = sd['state'].get(idx)
                if st is not None:
                    n = p.numel()
                    f.m[o:o + n].copy_(st['exp_avg'].reshape(-1))
                    f.v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                    self.t = int(float(st['step']))
                idx += 1
        self.param_groups[0]['lr'] = sd['param_groups'][0]['lr']


class _PendingVals(object):
    """the step's scalars still on the device (graph capture): names, the stacked tensor and the closure that turns the
    host-side values into what train_instance returns"""

    def __init__(self, names, dev):
        self.names, self.dev, self.finish = names, dev, None

    def resolve(self):
        return self.finish(OrderedDict(zip(self.names, self.dev.tolist())))


class DeferredStep(object):
    """What `train_instance` returns under `enable_step_graph(defer_scalars=True)`: the step's reported scalars are on their way
    to pinned host memory (an asynchronous copy enqueued behind the replay); `result()` waits for that copy and builds the
    usual (losses, visuals[, gnorms]) — so the host can enqueue step k + 1 while step k runs and read step k's losses later
    (the reference's loop reads them every step, train.py:198-243; a loop that logs every n-th step need not wait every step).
    The tensors in `visuals` are the graph's static buffers: overwritten by the next step."""

    def __init__(self, pending, host, event):
        self._pending, self._host, self._event, self._out = pending, host, event, None

    def result(self):
        if self._out is None:
            self._event.synchronize()
            self._out = self._pending.finish(OrderedDict(zip(self._pending.names, self._host.tolist())))
        return self._out


class StepGraph(object):
    """train_instance captured into a HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture / hipGraphLaunch; the
    library's ctypes launches go to the capturing stream like torch's own kernels).  What a replay cannot take from the
    host is kept on the device: the inputs (static buffers copied into), the Adam step number (FusedAdam.t_dev, read by
    acg_clip_adam_multi) and the reported scalars (one .tolist() after the replay).  The learning rates are launch
    arguments: a change re-captures."""
    WARMUP = 2

    def __init__(self, model):
        self.model, self.graph, self.key, self.calls, self.ws, self.packed = model, None, None, 0, None, None
        self.defer_scalars = False

    def _key(self, a, b, z):
        m = self.model
        lrs = tuple(opt.param_groups[0]['lr'] for opt in m._optimizers().values())
        # everything a captured launch bakes in as an argument: shapes, learning rates, train/eval mode, the kernel
        # configuration (precision / implementation switch) and the scalar options of the step
        baked = tuple(getattr(m.opt, k, None) for k in ("max_gnorm", "lambda_A", "lambda_B", "lambda_z_B", "lambda_sup_A",
                                                        "lambda_sup_B", "stoch_enc", "z_gan", "beta1"))
        return (tuple(a.shape), tuple(b.shape), tuple(z.shape), lrs, m.netG_A_B.training, ops.CONFIG_EPOCH, baked)

    def _capture(self, key, real_A, real_B, prior_z_B):
        m = self.model
        opts = list(m._optimizers().values())
        inputs = [real_A.clone(), real_B.clone(), prior_z_B.clone()]
        for opt in opts:
            if opt.t_dev is None:
                opt.t_dev = torch.zeros(1, dtype=torch.int32, device=real_A.device)
        graph = torch.cuda.CUDAGraph()
        # every derived tensor the step uses must be current at every replay.  The packed convolution weights are: each
        # optimiser step refreshes them IN PLACE (modules.repack) — eagerly and inside the graph alike — so the objects the
        # layers hold now are baked in as they are (round 6: rebuilding them inside the graph cost every replay 68 pack
        # launches), kept alive by this object and checked before every replay.  The other derived forms (padded norm
        # vectors) are copies a kernel made once: those are rebuilt INSIDE the graph.
        for net in m._nets():
            mark_dirty(net, keep_packed=True)
        # ... and so must the scratch buffers: the captured kernels keep the workspace POINTERS, and ops.workspace() replaces
        # an eager buffer as soon as a later eager op (a larger evaluation batch) needs more.  The capture therefore starts
        # from an empty workspace table, so its buffers come from the graph's private pool, and this object keeps them alive;
        # the eager table is put back afterwards.
        eager_ws, ops._WS = ops._WS, {}
        t_before = [opt.t for opt in opts]
        m._capturing = True
        for opt in opts:
            opt.dev_step = True
        # Python's cyclic collector must not run inside the capture: what it finds may own device objects of an EARLIER graph
        # (another model's StepGraph in a reference cycle: its hipGraph and private pool), and destroying those while a stream is
        # capturing is an error raised in a destructor, i.e. an abort.  torch.cuda.graph() collects once before it starts.
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            with torch.cuda.graph(graph), _in_train_step():
                pending = m._train_instance(*inputs)
            graph_ws = ops._WS
        finally:
            if gc_was_on:
                gc.enable()
            ops._WS = eager_ws
            m._capturing = False
            for opt, t in zip(opts, t_before):     # the capture ran clip_and_step's host side without executing it
                opt.t = t
                opt.dev_step = False
        # only a capture that completed is kept: a failed one leaves no half-built graph behind for the next call to replay
        self.graph, self.key, self.inputs, self.pending, self.ws = graph, key, inputs, pending, graph_ws
        self.packed = [packed_of(net) for net in m._nets()]     # (strong references: the graph holds their device pointers)

    def __call__(self, real_A, real_B, prior_z_B):
        m = self.model
        key = self._key(real_A, real_B, prior_z_B)
        self.calls += 1
        if self.calls <= self.WARMUP:                 # lazily built state (packed weights, workspaces) settles eagerly
            with _in_train_step():
                return m._train_instance(real_A, real_B, prior_z_B)
        opts = list(m._optimizers().values())
        if self.graph is not None and key == self.key:
            # the layers must still hold the packed weights the graph was captured with (a checkpoint load, a precision switch or
            # mark_dirty() in between replaces them): otherwise capture again
            now = [packed_of(net) for net in m._nets()]
            if any(a is not b for pa, pb in zip(now, self.packed) for a, b in zip(pa, pb)):
                self.graph = None
        if self.graph is None or key != self.key:
            self.graph = self.key = None
            self._capture(key, real_A, real_B, prior_z_B)
        for dst, src in zip(self.inputs, (real_A, real_B, prior_z_B)):
            dst.copy_(src)
        for opt in opts:
            opt.t_dev.fill_(opt.t)
        self.graph.replay()
        for opt in opts:
            opt.t += 1
        deferred = None
        if self.defer_scalars:   # the scalars leave for pinned memory behind the replay; nobody waits here
            host = torch.empty(self.pending.dev.shape, dtype=self.pending.dev.dtype, pin_memory=True)
            host.copy_(self.pending.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            deferred = DeferredStep(self.pending, host, ev)
        for net in m._nets():    # the replay updated the weights behind Python's caches (and repacked the convolutions' in place)
            mark_dirty(net, keep_packed=True)
        return deferred if deferred is not None else self.pending.resolve()


class _Base(object):
    def _dev(self):
        return next(self.netG_A_B.parameters()).device

    def _nchw(self, x, C):
        return ops.ToNCHW.apply(x, C).detach()

    def _backward(self, loss, phase, order, tail=None):
        """loss.backward() with the gradient exchange of this optimiser phase overlapped (dist.PhaseExchange): `order` =
        the FlatNets in the order their gradients complete; tail = (carrier FlatNet, sum scalars, min/max monitors) rides
        behind the carrier's gradients.  Returns the exchange (wait(flat) before that network's clip) or None."""
        if not acg_dist.exchange_on():
            loss.backward()
            return None
        ex = self._exchanges.setdefault(phase, acg_dist.PhaseExchange(phase))
        if tail is not None:
            acg_dist.write_scalar_tail(tail[0].gtail, tail[1], tail[2] if len(tail) > 2 else None)
        ex.arm(order)
        try:
            with _in_train_step():
                loss.backward()
        finally:
            ex.flush()
        return ex

    @staticmethod
    def _wait(ex, *flats):
        if ex is not None:
            for f in flats:
                ex.wait(f)

    def _scalars(self, ex, carrier, names, sums, mins=(), maxs=(), local=()):
        """ONE device->host copy for every reported scalar.  `sums` are rank-averaged, `mins`/`maxs` reduced over ranks
        (both through the carrier's gradient tail when the exchange is on), `local` = values that are already identical
        on every rank (norms of the averaged gradients)."""
        sums = [t.detach().reshape(()).float() for t in sums]
        mm = [t.detach().reshape(()).float() for t in list(mins) + list(maxs)]
        if ex is not None:
            avg, per_rank = acg_dist.read_scalar_tail(carrier.gtail, len(sums), len(mm))
            sums = list(avg.unbind(0))
            if mm:
                mm = [per_rank[:, i].min() for i in range(len(mins))] + \
                     [per_rank[:, len(mins) + i].max() for i in range(len(maxs))]
        dev = torch.stack(sums + [t.detach().reshape(()).float() for t in local] + mm)
        if self._capturing:        # StepGraph: the copy to the host happens after the replay
            return _PendingVals(names, dev)
        return OrderedDict(zip(names, dev.tolist()))

    _capturing = False
    _step_graph = None

    def _report(self, vals, finish):
        """finish(vals) builds what the step returns from the host-side scalars; deferred while a graph is being captured"""
        if isinstance(vals, _PendingVals):
            vals.finish = finish
            return vals
        return finish(vals)

    def enable_step_graph(self, on=True, defer_scalars=False):
        """Run train_instance as ONE captured HIP graph per (shapes, learning rates): the whole step — about 3 000 kernel
        launches — is replayed by a single host call.  Worth it where the step is launch-bound (small images / batches:
        64 x 64 x 4 runs 28 ms eager against the 33 ms of Python it takes to enqueue); at 256 x 256 x 32 the GPU is the
        bound either way.  The first calls run eagerly (warm-up), the tensors in the returned `visuals` are overwritten by the
        next call, and the data-parallel exchange keeps the eager path.  defer_scalars: a replayed step returns a DeferredStep
        (`.result()` gives the usual tuple) instead of waiting for its scalars — the host then enqueues the next step while
        this one runs."""
        self._step_graph = StepGraph(self) if on else None
        if self._step_graph is not None:
            self._step_graph.defer_scalars = bool(defer_scalars)

    # ---- forward-only helpers shared by both models (model.py:210-280, 606-733): compositions of the two generators.
    # Subclass hooks: _z (noise transform), _cycle_code (the latent the B -> A -> B cycle is closed with).
    def _draw_prior(self, like):
        return like.new_empty((like.size(0), self.opt.nlatent, 1, 1)).normal_(0, 1)

    def predict_A(self, real_B):
        return self.netG_B_A.forward(real_B)

    def predict_B(self, real_A, z_B):
        return self.netG_A_B.forward(real_A, self._z(z_B))

    def generate_cycle(self, real_A, real_B, prior_z_B):
        fake_B, fake_A = self.predict_B(real_A, prior_z_B), self.predict_A(real_B)
        rec_A = self.predict_A(fake_B)
        rec_B = self.netG_A_B.forward(fake_A, self._cycle_code(fake_A, real_B, prior_z_B))
        return OrderedDict([('real_A', real_A.data), ('fake_B', fake_B.data), ('rec_A', rec_A.data),
                            ('real_B', real_B.data), ('fake_A', fake_A.data), ('rec_B', rec_B.data)])

    def generate_multi(self, real_A, multi_prior_z_B):
        """each A against multi_prior_z_B.size(0) / |A| consecutive codes (train.py:66)"""
        return self.predict_B(_each_n_times(real_A, multi_prior_z_B.size(0) // real_A.size(0)), multi_prior_z_B)

    def generate_cycle_B_multi(self, real_B, multi_prior_z_B):
        fake_A = self.predict_A(real_B)
        return fake_A, self.netG_A_B.forward(_each_n_times(fake_A, multi_prior_z_B.size(0) // real_B.size(0)), multi_prior_z_B)

    def generate_noisy_cycle(self, real_B, std):
        """B -> A, perturb A by N(0, std/127.5) (clamped to the image range), -> B"""
        fake_A = self.predict_A(real_B)
        perturb = lambda: torch.clamp(fake_A + torch.empty_like(fake_A).normal_(0, std / 127.5), -1, 1)
        if self._noise_before_code:     # order of the random draws as in the reference (model.py:626-645 vs 257-266)
            noisy, code = perturb(), self._cycle_code(fake_A, real_B, None)
        else:
            code, noisy = self._cycle_code(fake_A, real_B, None), perturb()
        return self.netG_A_B.forward(noisy, code)

    def generate_multi_cycle(self, real_B, steps):
        images, B = [real_B.data], real_B
        for _ in range(steps):
            A = self.predict_A(B)
            B = self.netG_A_B.forward(A, self._z(self._draw_prior(real_B)))
            images += [A.data, B.data]
        return images

    # ---- north-star aliases -----------------------------------------------------
    def set_input(self, data, prior_z_B=None):
        self._input = (data['A'], data['B'], prior_z_B)

    def optimize_parameters(self):
        A, B, z = self._input
        if z is None:
            z = torch.randn(A.size(0), self.opt.nlatent, 1, 1, device=A.device)
        self._last = self.train_instance(A, B, z)
        return self._last

    def eval(self):
        for n in self._nets():
            n.eval()

    def train(self):
        for n in self._nets():
            n.train()


import contextlib  # noqa: E402


@contextlib.contextmanager
def _in_train_step():
    """SyncBN collectives are issued only inside a training step, where every rank runs the same forward/backward; the
    rank-0-only forwards of train.py (visualisation, evaluation) then use local statistics instead of posting
    collectives the other ranks never join."""
    from . import modules
    prev, modules.IN_TRAIN_STEP = modules.IN_TRAIN_STEP, True
    try:
        yield
    finally:
        modules.IN_TRAIN_STEP = prev


def _each_n_times(x, n):
    """(N, ...) -> (N*n, ...): every sample n times in a row"""
    return x.repeat_interleave(n, dim=0)


def _n_blocks(opt):
    from . import modules
    modules.SYNC_BN = bool(getattr(opt, 'sync_bn', False))   # extension: BatchNorm statistics over all ranks
    return int(getattr(opt, 'n_blocks', 3))


class StochCycleGAN(_Base):
    """Stochastic cycle gan — model.py:75-325"""

    def __init__(self, opt, ignore_noise=False, testing=False):
        self.ignore_noise = ignore_noise
        self._exchanges = {}
        self.old_lr = opt.lr
        opt.use_sigmoid = opt.no_lsgan
        self.opt = opt
        nb = _n_blocks(opt)
        self.netG_A_B = networks.define_stochastic_G(nlatent=opt.nlatent, input_nc=opt.input_nc, output_nc=opt.output_nc,
                                                     ngf=opt.ngf, which_model_netG=opt.which_model_netG, norm=opt.norm,
                                                     use_dropout=opt.use_dropout, gpu_ids=opt.gpu_ids, n_blocks=nb)
        self.netG_B_A = networks.define_G(input_nc=opt.output_nc, output_nc=opt.input_nc, ngf=opt.ngf,
                                          which_model_netG=opt.which_model_netG, norm=opt.norm,
                                          use_dropout=opt.use_dropout, gpu_ids=opt.gpu_ids, n_blocks=nb)
        self.netD_A = networks.define_D_A(input_nc=opt.input_nc, ndf=32, which_model_netD=opt.which_model_netD,
                                          norm=opt.norm, use_sigmoid=opt.use_sigmoid, gpu_ids=opt.gpu_ids)
        self.netD_B = networks.define_D_B(input_nc=opt.output_nc, ndf=opt.ndf, which_model_netD=opt.which_model_netD,
                                          norm=opt.norm, use_sigmoid=opt.use_sigmoid, gpu_ids=opt.gpu_ids)
        self._build_optimizers()
        self.criterionGAN = functools.partial(criterion_GAN, use_sigmoid=opt.use_sigmoid)
        self.criterionCycle = lambda a, b: ops.L1.apply(_as2d(a), _as2d(b), a.shape[1])
        if not testing:
            with open("%s/nets.txt" % opt.expr_dir, 'w') as nets_f:
                for n in self._nets():
                    networks.print_network(n, nets_f)

    def _nets(self):
        return [self.netG_A_B, self.netG_B_A, self.netD_A, self.netD_B]

    def _build_optimizers(self):
        o = self.opt
        acg_dist.broadcast_params_(self._nets())
        self.f_G_A_B, self.f_G_B_A = FlatNet(self.netG_A_B), FlatNet(self.netG_B_A)
        self.f_D_A, self.f_D_B = FlatNet(self.netD_A), FlatNet(self.netD_B)
        self.optimizer_G = FusedAdam([self.f_G_A_B, self.f_G_B_A], o.lr, (o.beta1, 0.999))      # model.py:109-111
        self.optimizer_D = FusedAdam([self.f_D_A, self.f_D_B], o.lr / 5., (o.beta1, 0.999))     # model.py:112-114

    def train_instance(self, real_A, real_B, prior_z_B):
        if self._step_graph is not None and not acg_dist.exchange_on():
            return self._step_graph(real_A, real_B, prior_z_B)
        with _in_train_step():
            return self._train_instance(real_A, real_B, prior_z_B)

    def _train_instance(self, real_A, real_B, prior_z_B):
        o = self.opt
        nA, nB = o.input_nc, o.output_nc
        for f in (self.f_G_A_B, self.f_G_B_A, self.f_D_A, self.f_D_B):
            f.check()
        if self.ignore_noise:
            prior_z_B = prior_z_B.mul(0.).add(1.)                                               # model.py:128-129
        A, B, z = ops.ToNHWC.apply(real_A, True), ops.ToNHWC.apply(real_B, True), as_latent(prior_z_B)
        fake_B = self.netG_A_B.forward_nhwc(A, z)
        fake_A = self.netG_B_A.forward_nhwc(B)

        # ---- D phase (model.py:139-162)
        p_fA = self.netD_A.forward_nhwc(fake_A.detach()); l_fA = _gan_loss(p_fA, False)
        p_tA = self.netD_A.forward_nhwc(A); l_tA = _gan_loss(p_tA, True)
        p_fB = self.netD_B.forward_nhwc(fake_B.detach()); l_fB = _gan_loss(p_fB, False)
        p_tB = self.netD_B.forward_nhwc(B); l_tB = _gan_loss(p_tB, True)
        loss_D_A, loss_D_B = 0.5 * (l_fA + l_tA), 0.5 * (l_fB + l_tB)
        loss_D = loss_D_A + loss_D_B
        self.optimizer_D.zero_grad()
        ex_D = self._backward(loss_D, "stoch.D", [self.f_D_B, self.f_D_A])    # D_B was built last: its backward runs first
        m_tA, m_tB = ops.mean_valid(p_tA, 1), ops.mean_valid(p_tB, 1)

        # ---- G phase (model.py:167-190).  The cycle forwards do not depend on the discriminators: they run while the
        # D gradients are being all-reduced; the D forwards wait for the UPDATED discriminators (model.py:164-166)
        rec_A = self.netG_B_A.forward_nhwc(fake_B); loss_cycle_A = ops.L1.apply(rec_A, A, nA)
        rec_B = self.netG_A_B.forward_nhwc(fake_A, z); loss_cycle_B = ops.L1.apply(rec_B, B, nB)
        self._wait(ex_D, self.f_D_A, self.f_D_B)
        ss_D_A, ss_D_B = self.optimizer_D.clip_and_step(o.max_gnorm)
        ss_D_A, ss_D_B = ss_D_A.clone(), ss_D_B.clone()
        self.f_D_A.set_requires_grad(False); self.f_D_B.set_requires_grad(False)   # D weight grads are not needed
        try:
            p_fA = self.netD_A.forward_nhwc(fake_A); loss_G_A = _gan_loss(p_fA, True)
            p_fB = self.netD_B.forward_nhwc(fake_B); loss_G_B = _gan_loss(p_fB, True)
            loss_G = loss_G_A + loss_G_B + loss_cycle_A * o.lambda_A + loss_cycle_B * o.lambda_B
            self.optimizer_G.zero_grad()
            sums = [loss_D_A, loss_G_A, loss_cycle_A, loss_D_B, loss_G_B, loss_cycle_B,
                    m_tA, ops.mean_valid(p_fA, 1), m_tB, ops.mean_valid(p_fB, 1)]
            ex_G = self._backward(loss_G, "stoch.G", [self.f_G_B_A, self.f_G_A_B], tail=(self.f_G_A_B, sums))
        finally:
            self.f_D_A.set_requires_grad(True); self.f_D_B.set_requires_grad(True)
        self._wait(ex_G, self.f_G_A_B, self.f_G_B_A)
        ss_G_A_B, ss_G_B_A = self.optimizer_G.clip_and_step(o.max_gnorm)

        names = ['D_A', 'G_A', 'Cyc_A', 'D_B', 'G_B', 'Cyc_B', 'P_t_A', 'P_f_A', 'P_t_B', 'P_f_B',
                 'gnorm_G_A_B', 'gnorm_G_B_A', 'gnorm_D_B', 'gnorm_D_A']
        vals = self._scalars(ex_G, self.f_G_A_B, names, sums, local=[ss_G_A_B, ss_G_B_A, ss_D_B, ss_D_A])
        visuals = OrderedDict([('real_A', real_A.detach()), ('fake_B', self._nchw(fake_B, nB)),
                               ('rec_A', self._nchw(rec_A, nA)), ('real_B', real_B.detach()),
                               ('fake_A', self._nchw(fake_A, nA)), ('rec_B', self._nchw(rec_B, nB))])

        def finish(vals):
            losses = OrderedDict((k, vals[k]) for k in names[:10])                               # model.py:193-196
            if o.monitor_gnorm:
                gnorms = OrderedDict((k, math.sqrt(max(vals[k], 0.0))) for k in names[10:])      # model.py:202-205
                return losses, visuals, gnorms
            return losses, visuals
        return self._report(vals, finish)

    # ---- hooks of the shared forward-only helpers (_Base; model.py:210-280) -------------------
    _noise_before_code = False

    def _z(self, z):
        return z.mul(0.).add(1.) if self.ignore_noise else z                                     # model.py:128-129

    def _cycle_code(self, fake_A, real_B, prior_z_B):
        """the code the B -> A -> B cycle is closed with: no encoder here, the given prior"""
        return self._z(prior_z_B) if prior_z_B is not None else self._z(self._draw_prior(real_B))

    def _optimizers(self):
        return OrderedDict([('optimizer_D', self.optimizer_D), ('optimizer_G', self.optimizer_G)])

    def _net_dict(self):
        return OrderedDict([('netG_A_B', self.netG_A_B), ('netG_B_A', self.netG_B_A), ('netD_A', self.netD_A),
                            ('netD_B', self.netD_B)])

    def update_learning_rate(self):
        """model.py:282-291 (also overwrites the discriminators' lr/5 — reference behaviour)"""
        lrd = self.opt.lr / self.opt.niter_decay
        lr = self.old_lr - lrd
        for opt in self._optimizers().values():
            for param_group in opt.param_groups:
                param_group['lr'] = lr
        print('update learning rate: %f -> %f' % (self.old_lr, lr))
        self.old_lr = lr

    def save(self, chk_name):
        """model.py:293-303 / 750-764: same checkpoint keys, torch.load-able"""
        chk_path = os.path.join(self.opt.expr_dir, chk_name)
        checkpoint = {k: n.state_dict() for k, n in self._net_dict().items()}
        checkpoint.update({k: o.state_dict() for k, o in self._optimizers().items()})
        torch.save(checkpoint, chk_path)

    def load(self, chk_path):
        checkpoint = torch.load(chk_path, map_location=self._dev())
        for k, n in self._net_dict().items():
            n.load_state_dict(checkpoint[k])
            mark_dirty(n)
        for k, o in self._optimizers().items():
            o.load_state_dict(checkpoint[k])


def _as2d(t):
    """public-API tensors (NCHW / (N,C)) -> (rows, Cp) C16 view for the loss kernels"""
    if t.dim() == 4:
        return ops.ToNHWC.apply(t).reshape(-1, cpad(t.shape[1]))
    return t.reshape(t.shape[0], -1)


def discriminate(net, crit, fake, real):
    """model.py:327-334 (public-API form)"""
    pred_fake = net(fake)
    loss_fake = crit(pred_fake, False)
    pred_true = net(real)
    loss_true = crit(pred_true, True)
    return loss_fake, loss_true, pred_fake, pred_true


class AugmentedCycleGAN(_Base):
    """Augmented cycle gan — model.py:337-794"""

    def __init__(self, opt, testing=False):
        self._exchanges = {}
        self.old_lr = opt.lr
        opt.use_sigmoid = opt.no_lsgan
        self.opt = opt
        nb = _n_blocks(opt)
        self.netG_A_B = networks.define_stochastic_G(nlatent=opt.nlatent, input_nc=opt.input_nc, output_nc=opt.output_nc,
                                                     ngf=opt.ngf, which_model_netG=opt.which_model_netG, norm=opt.norm,
                                                     use_dropout=opt.use_dropout, gpu_ids=opt.gpu_ids, n_blocks=nb)
        self.netG_B_A = networks.define_G(input_nc=opt.output_nc, output_nc=opt.input_nc, ngf=opt.ngf,
                                          which_model_netG=opt.which_model_netG, norm=opt.norm,
                                          use_dropout=opt.use_dropout, gpu_ids=opt.gpu_ids, n_blocks=nb)
        enc_input_nc = opt.output_nc
        if opt.enc_A_B:
            enc_input_nc += opt.input_nc
        self.netE_B = networks.define_E(nlatent=opt.nlatent, input_nc=enc_input_nc, nef=opt.nef, norm='batch',
                                        gpu_ids=opt.gpu_ids)
        self.netD_A = networks.define_D_A(input_nc=opt.input_nc, ndf=32, which_model_netD=opt.which_model_netD,
                                          norm=opt.norm, use_sigmoid=opt.use_sigmoid, gpu_ids=opt.gpu_ids)
        self.netD_B = networks.define_D_B(input_nc=opt.output_nc, ndf=opt.ndf, which_model_netD=opt.which_model_netD,
                                          norm=opt.norm, use_sigmoid=opt.use_sigmoid, gpu_ids=opt.gpu_ids)
        self.netD_z_B = networks.define_LAT_D(nlatent=opt.nlatent, ndf=opt.ndf, use_sigmoid=opt.use_sigmoid,
                                              gpu_ids=opt.gpu_ids)
        self._build_optimizers()
        self.criterionGAN = functools.partial(criterion_GAN, use_sigmoid=opt.use_sigmoid)
        self.criterionCycle = lambda a, b: ops.L1.apply(_as2d(a), _as2d(b), a.shape[1])
        if not testing:
            with open("%s/nets.txt" % opt.expr_dir, 'w') as nets_f:
                for n in (self.netG_A_B, self.netG_B_A, self.netD_A, self.netD_B, self.netD_z_B, self.netE_B):
                    networks.print_network(n, nets_f)                                           # model.py:393-400

    def _nets(self):
        return [self.netG_A_B, self.netG_B_A, self.netE_B, self.netD_A, self.netD_B, self.netD_z_B]

    def _build_optimizers(self):
        o = self.opt
        acg_dist.broadcast_params_(self._nets())
        self.f_G_A_B, self.f_G_B_A, self.f_E_B = FlatNet(self.netG_A_B), FlatNet(self.netG_B_A), FlatNet(self.netE_B)
        self.f_D_A, self.f_D_B, self.f_D_z_B = FlatNet(self.netD_A), FlatNet(self.netD_B), FlatNet(self.netD_z_B)
        b = (o.beta1, 0.999)
        self.optimizer_G_A = FusedAdam([self.f_G_B_A], o.lr, b)                                 # model.py:379-380
        self.optimizer_G_B = FusedAdam([self.f_G_A_B, self.f_E_B], o.lr, b)                     # model.py:381-383
        self.optimizer_D_A = FusedAdam([self.f_D_A], o.lr / 5., b)                              # model.py:384-385
        self.optimizer_D_B = FusedAdam([self.f_D_B, self.f_D_z_B], o.lr / 5., b)                # model.py:386-389

    def _encode(self, a_or_fake_a, b):
        """E_B on cat((A-side, B-side), 1) (A first: model.py:410, 472) -> (mu, logvar) (N, cpad(nl))"""
        o = self.opt
        x = ops.Concat.apply(a_or_fake_a, b, o.input_nc, o.output_nc) if o.enc_A_B else b
        return self.netE_B.forward_nhwc(x)

    def train_instance(self, real_A, real_B, prior_z_B):
        if self._step_graph is not None and not acg_dist.exchange_on():
            return self._step_graph(real_A, real_B, prior_z_B)
        with _in_train_step():
            return self._train_instance(real_A, real_B, prior_z_B)

    def _train_instance(self, real_A, real_B, prior_z_B):
        o = self.opt
        nA, nB, nl = o.input_nc, o.output_nc, o.nlatent
        flats_D = [self.f_D_A, self.f_D_B, self.f_D_z_B]
        for f in flats_D + [self.f_G_A_B, self.f_G_B_A, self.f_E_B]:
            f.check()
        A, B, z = ops.ToNHWC.apply(real_A, True), ops.ToNHWC.apply(real_B, True), as_latent(prior_z_B)
        bs = z.shape[0]

        fake_B = self.netG_A_B.forward_nhwc(A, z)                                               # model.py:404
        fake_A = self.netG_B_A.forward_nhwc(B)                                                  # model.py:407
        mu_rB, lv_rB = self._encode(fake_A, B)                                                  # model.py:409-413
        if o.stoch_enc:
            post_z = gauss_reparametrize(mu_rB[:, :nl], lv_rB[:, :nl]).view(bs, nl)             # model.py:416
        else:
            post_z = mu_rB                                                                      # model.py:418
            lv_rB = lv_rB * 0.0                                                                 # model.py:419

        # ---- D phase (model.py:423-452)
        p_fA = self.netD_A.forward_nhwc(fake_A.detach()); l_fA = _gan_loss(p_fA, False)
        p_tA = self.netD_A.forward_nhwc(A); l_tA = _gan_loss(p_tA, True)
        p_fB = self.netD_B.forward_nhwc(fake_B.detach()); l_fB = _gan_loss(p_fB, False)
        p_tB = self.netD_B.forward_nhwc(B); l_tB = _gan_loss(p_tB, True)
        l_pz = _gan_loss(self.netD_z_B.forward_dense(post_z.detach()), False)
        l_rz = _gan_loss(self.netD_z_B.forward_dense(z), True)
        loss_D_A, loss_D_B, loss_D_z_B = 0.5 * (l_fA + l_tA), 0.5 * (l_fB + l_tB), 0.5 * (l_pz + l_rz)
        loss_D = loss_D_A + loss_D_B
        z_gan = bool(o.z_gan and not o.stoch_enc)
        if z_gan:
            loss_D = loss_D + loss_D_z_B
        self.optimizer_D_A.zero_grad(); self.optimizer_D_B.zero_grad()
        # completion order of the D backward = reverse build order: D_z_B, D_B, D_A (without the latent GAN term D_z_B
        # receives no gradient at all, model.py:438-439, and goes last)
        order_D = [self.f_D_z_B, self.f_D_B, self.f_D_A] if z_gan else [self.f_D_B, self.f_D_A, self.f_D_z_B]
        ex_D = self._backward(loss_D, "aug.D", order_D)
        m_tA, m_tB = ops.mean_valid(p_tA, 1), ops.mean_valid(p_tB, 1)

        # ---- G phase (model.py:457-515).  The cycle / encoder forwards (model.py:467-494) do not depend on the
        # discriminators: they are enqueued first and run while the D gradients are being all-reduced ...
        rec_A = self.netG_B_A.forward_nhwc(fake_B); loss_cycle_A = ops.L1.apply(rec_A, A, nA)
        mu_fB, lv_fB = self._encode(A, fake_B)                                                  # model.py:471-475
        if o.stoch_enc:
            lp = log_prob_gaussian(z, mu_fB[:, :nl], lv_fB[:, :nl])
            loss_cycle_z_B = -1.0 * lp.mean(1).mean(0)                                          # model.py:480-484
        else:
            loss_cycle_z_B = ops.L1.apply(mu_fB, _pad_cols(z, mu_fB.shape[1]), nl)              # model.py:486-487
        rec_B = self.netG_A_B.forward_nhwc(fake_A, post_z); loss_cycle_B = ops.L1.apply(rec_B, B, nB)
        kld_z_B = kld_std_guss(mu_rB[:, :nl], lv_rB[:, :nl]).mean(0)                            # model.py:490
        # ... the discriminator forwards need the UPDATED discriminators (model.py:455-457)
        self._wait(ex_D, self.f_D_A)
        (ss_D_A,) = self.optimizer_D_A.clip_and_step(o.max_gnorm)
        self._wait(ex_D, self.f_D_B, self.f_D_z_B)
        ss_D_B, ss_D_z = self.optimizer_D_B.clip_and_step(o.max_gnorm)
        ss_D_A, ss_D_B, ss_D_z = ss_D_A.clone(), ss_D_B.clone(), ss_D_z.clone()
        for f in flats_D:                                   # D weight gradients are not needed in the G phase
            f.set_requires_grad(False)
        try:
            p_fA = self.netD_A.forward_nhwc(fake_A); loss_G_A = _gan_loss(p_fA, True)
            p_fB = self.netD_B.forward_nhwc(fake_B); loss_G_B = _gan_loss(p_fB, True)
            loss_G_z_B = _gan_loss(self.netD_z_B.forward_dense(post_z), True)
            loss_G = loss_G_A + loss_G_B + loss_cycle_A * o.lambda_A + loss_cycle_B * o.lambda_B \
                + loss_cycle_z_B * o.lambda_z_B
            if o.stoch_enc:
                loss_G = loss_G + kld_z_B * o.lambda_z_B
            if z_gan:
                loss_G = loss_G + loss_G_z_B
            self.optimizer_G_A.zero_grad(); self.optimizer_G_B.zero_grad()
            mu_v, lv_v = mu_rB.detach()[:, :nl], lv_rB.detach()[:, :nl]
            sums = [loss_D_A, loss_G_A, loss_cycle_A, loss_cycle_z_B, kld_z_B, loss_D_B, loss_G_B, loss_cycle_B,
                    loss_D_z_B, m_tA, ops.mean_valid(p_fA, 1), m_tB, ops.mean_valid(p_fB, 1)]
            mins, maxs = [mu_v.min(), lv_v.min()], [mu_v.max(), lv_v.max()]
            # completion order of the G backward: E_B (its first call is the last of the three first-pass networks to have
            # been built), then G_B_A, then G_A_B — which therefore carries the scalar tail
            ex_G = self._backward(loss_G, "aug.G", [self.f_E_B, self.f_G_B_A, self.f_G_A_B],
                                  tail=(self.f_G_A_B, sums, mins + maxs))
        finally:
            for f in flats_D:
                f.set_requires_grad(True)
        self._wait(ex_G, self.f_G_B_A)
        (ss_G_B_A,) = self.optimizer_G_A.clip_and_step(o.max_gnorm)
        self._wait(ex_G, self.f_G_A_B, self.f_E_B)
        ss_G_A_B, ss_E = self.optimizer_G_B.clip_and_step(o.max_gnorm)

        names = ['D_A', 'G_A', 'Cyc_A', 'Cyc_z_B', 'KLD_z_B', 'D_B', 'G_B', 'Cyc_B', 'D_z_B',
                 'P_t_A', 'P_f_A', 'P_t_B', 'P_f_B',
                 'gnorm_G_A_B', 'gnorm_G_B_A', 'gnorm_E_B', 'gnorm_D_B', 'gnorm_D_z_B', 'gnorm_D_A',
                 'mu_min', 'logvar_min', 'mu_max', 'logvar_max']
        vals = self._scalars(ex_G, self.f_G_A_B, names, sums, mins, maxs,
                             local=[ss_G_A_B, ss_G_B_A, ss_E, ss_D_B, ss_D_z, ss_D_A])
        visuals = OrderedDict([('real_A', real_A.detach()), ('fake_B', self._nchw(fake_B, nB)),
                               ('rec_A', self._nchw(rec_A, nA)), ('real_B', real_B.detach()),
                               ('fake_A', self._nchw(fake_A, nA)), ('rec_B', self._nchw(rec_B, nB))])

        def finish(vals):
            losses = OrderedDict((k, vals[k]) for k in names[:13])                              # model.py:518-523
            if o.monitor_gnorm:
                gnorms = OrderedDict((k, math.sqrt(max(vals[k], 0.0))) for k in names[13:19])   # model.py:527-533
                for k in ('mu_min', 'mu_max', 'logvar_min', 'logvar_max'):
                    gnorms[k] = vals[k]
                return losses, visuals, gnorms
            return losses, visuals
        return self._report(vals, finish)

    def supervised_train_instance(self, real_A, real_B, prior_z_B):
        """model.py:541-604 (paired step; off by default, --supervised)"""
        with _in_train_step():
            return self._supervised_train_instance(real_A, real_B, prior_z_B)

    def _supervised_train_instance(self, real_A, real_B, prior_z_B):
        o = self.opt
        nA, nB, nl = o.input_nc, o.output_nc, o.nlatent
        A, B, z = ops.ToNHWC.apply(real_A, True), ops.ToNHWC.apply(real_B, True), as_latent(prior_z_B)
        bs = z.shape[0]
        mu, logvar = self._encode(A, B)
        if o.stoch_enc:
            post_z = gauss_reparametrize(mu[:, :nl], logvar[:, :nl]).view(bs, nl)
        else:
            post_z = mu
            logvar = logvar * 0.0
        l_pz = _gan_loss(self.netD_z_B.forward_dense(post_z.detach()), False)
        l_rz = _gan_loss(self.netD_z_B.forward_dense(z), True)
        loss_D_z_B = 0.5 * (l_pz + l_rz)
        self.optimizer_D_B.zero_grad()
        ex_D = self._backward(loss_D_z_B, "sup.D", [self.f_D_z_B, self.f_D_B])
        self._wait(ex_D, self.f_D_B, self.f_D_z_B)
        _, ss_D_z = self.optimizer_D_B.clip_and_step(o.max_gnorm)
        ss_D_z = ss_D_z.clone()
        self.f_D_z_B.set_requires_grad(False)
        try:
            pred_B = self.netG_A_B.forward_nhwc(A, post_z)
            pred_A = self.netG_B_A.forward_nhwc(B)
            loss_sup_A = ops.L1.apply(pred_A, A, nA)
            loss_sup_B = ops.L1.apply(pred_B, B, nB)
            loss_G_z_B = _gan_loss(self.netD_z_B.forward_dense(post_z), True)
            kld_z_B = kld_std_guss(mu[:, :nl], logvar[:, :nl]).mean(0)
            loss_G = loss_sup_A * o.lambda_sup_A + loss_sup_B * o.lambda_sup_B
            if o.stoch_enc:
                loss_G = loss_G + kld_z_B * o.lambda_z_B
            if o.z_gan and not o.stoch_enc:
                loss_G = loss_G + loss_G_z_B
            self.optimizer_G_A.zero_grad(); self.optimizer_G_B.zero_grad()
            sums = [loss_sup_A, loss_sup_B, kld_z_B, loss_D_z_B]
            # backward order: G_B_A (built last), G_A_B, then the encoder behind post_z
            ex_G = self._backward(loss_G, "sup.G", [self.f_G_B_A, self.f_G_A_B, self.f_E_B], tail=(self.f_E_B, sums))
        finally:
            self.f_D_z_B.set_requires_grad(True)
        self._wait(ex_G, self.f_G_B_A)
        (ss_G_B_A,) = self.optimizer_G_A.clip_and_step(o.max_gnorm)
        self._wait(ex_G, self.f_G_A_B, self.f_E_B)
        ss_G_A_B, ss_E = self.optimizer_G_B.clip_and_step(o.max_gnorm)
        names = ['S_A', 'S_B', 'KLD_z_B', 'D_z_B', 'gnorm_G_A_B', 'gnorm_G_B_A', 'gnorm_E_B', 'gnorm_D_z_B']
        vals = self._scalars(ex_G, self.f_E_B, names, sums, local=[ss_G_A_B, ss_G_B_A, ss_E, ss_D_z])
        for k in names[4:]:
            vals[k] = math.sqrt(max(vals[k], 0.0))
        return vals                                                                             # model.py:596-604

    # ---- hooks of the shared forward-only helpers (_Base) + the encoder-specific ones (model.py:606-733) ----
    _noise_before_code = True

    def _z(self, z):
        return z

    def _enc_public(self, a, b):
        x = torch.cat((a, b), 1) if self.opt.enc_A_B else b
        return self.netE_B.forward(x)

    def _post_z(self, mu, logvar):
        if self.opt.stoch_enc:
            return gauss_reparametrize(mu, logvar)
        return mu.reshape(mu.size(0), mu.size(1), 1, 1)

    def _cycle_code(self, fake_A, real_B, prior_z_B):
        """the code the B -> A -> B cycle is closed with: the encoder's posterior for (fake_A, real_B)"""
        return self._post_z(*self._enc_public(fake_A, real_B))

    def predict_enc_params(self, real_A, real_B):
        """model.py:653-662"""
        mu, logvar = self._enc_public(real_A, real_B)
        return (mu, logvar) if self.opt.stoch_enc else (mu,)

    def generate_multi_cycle(self, real_B, steps, from_prior=True):
        """model.py:664-685: alternate B -> A -> B `steps` times, re-drawing (or re-encoding) the code every round"""
        images, B = [real_B.data], real_B
        for _ in range(steps):
            A = self.predict_A(B)
            code = self._draw_prior(real_B) if from_prior else self._cycle_code(A, B, None)
            B = self.netG_A_B.forward(A, code)
            images += [A.data, B.data]
        return images

    def inference_multi(self, real_A, real_B):
        """model.py:710-733: every A against the posterior code of every B — (|A| * |B|) images, A-major"""
        codes = self._cycle_code(self.predict_A(real_B) if self.opt.enc_A_B else real_B, real_B, None)
        return self.netG_A_B.forward(_each_n_times(real_A, real_B.size(0)), codes.data.repeat(real_A.size(0), 1, 1, 1))

    def _optimizers(self):
        return OrderedDict([('optimizer_D_A', self.optimizer_D_A), ('optimizer_G_A', self.optimizer_G_A),
                            ('optimizer_D_B', self.optimizer_D_B), ('optimizer_G_B', self.optimizer_G_B)])

    def _net_dict(self):
        return OrderedDict([('netG_A_B', self.netG_A_B), ('netG_B_A', self.netG_B_A), ('netD_A', self.netD_A),
                            ('netD_B', self.netD_B), ('netD_z_B', self.netD_z_B), ('netE_B', self.netE_B)])

    update_learning_rate = StochCycleGAN.update_learning_rate                                    # model.py:735-748
    save = StochCycleGAN.save                                                                    # model.py:750-764
    load = StochCycleGAN.load                                                                    # model.py:766-778


# north-star alias (BASELINE.json names the class AugmentedCycleGAN_Model)
AugmentedCycleGAN_Model = AugmentedCycleGAN
