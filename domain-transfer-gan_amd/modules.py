"""Building blocks — same names, constructor signatures and state_dict keys as the reference's
/root/reference/augmented_cyclegan/modules.py, executed by the HIP kernels in libacgan_hip.so.

Every layer class keeps the torch.nn parameter layout (OIHW weights, etc.) so reference
checkpoints load unchanged; `forward` on NCHW tensors is provided for drop-in use of a single
module, while the networks run the fused NHWC pipeline in `run_sequence` below (conv + bias +
activation in one kernel; norm + activation (+ residual add) in one pass).
"""
import functools  # noqa: F401  (kept: reference modules export it implicitly via networks)

import ctypes

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, cpad

USE_PYTORCH_IN = False  # modules.py:9
SYNC_BN = False         # set by model.py from opt.sync_bn: BatchNorm statistics across all data-parallel ranks
IN_TRAIN_STEP = False   # set by model.py around (supervised_)train_instance: the only place where every rank runs the same
                        # BatchNorm forwards, i.e. where SyncBN may post collectives (train.py's visualisation and
                        # evaluation forwards run on rank 0 only and use local statistics)


def ops_dist_on():
    from . import dist
    return dist.exchange_on()


def sync_bn_active():
    """True where a train-mode BatchNorm forward would exchange its statistics with the other ranks"""
    return SYNC_BN and IN_TRAIN_STEP and ops_dist_on()


def mark_dirty(net, keep_packed=False):
    """Parameters of `net` changed outside torch's version counter (fused Adam kernel):
    drop the cached packed weights / padded vectors of all its layers.  keep_packed: the packed convolution weights are
    known to be current (`repack` refreshed them in place behind the optimiser step) and stay; only the other derived forms go."""
    for m in net.modules():
        if hasattr(m, "_acg_cache"):
            c = m._acg_cache
            if keep_packed and c is not None and isinstance(c[1], ops.PackedConv) and c[0] == m._cache_key():
                continue
            m._acg_cache = None


def packed_of(net):
    """the PackedConv objects the convolution layers of `net` currently hold (None where a layer has none): what a captured
    graph bakes in as pointers"""
    out = []
    for m in net.modules():
        if isinstance(m, (Conv2d, ConvTranspose2d)):
            c = getattr(m, "_acg_cache", None)
            out.append(c[1] if (c is not None and isinstance(c[1], ops.PackedConv) and c[0] == m._cache_key()) else None)
    return out


def repack(net):
    """after the fused Adam kernel changed `net`'s parameters: refresh the packed weights of its convolution layers in place, all
    regular layers in ONE launch (ops.repack_many; one launch per layer was 68 small kernels per training step), and drop
    every other cached derived form (padded norm vectors).  A layer that has not run yet has nothing to refresh."""
    entries = []
    for m in net.modules():
        c = getattr(m, "_acg_cache", None)
        if c is None:
            continue
        if isinstance(m, (Conv2d, ConvTranspose2d)) and isinstance(c[1], ops.PackedConv) and c[0] == m._cache_key():
            entries.append((c[1], m.weight, m.bias))
        else:
            m._acg_cache = None
    if entries:
        ops.repack_many(entries)


class _Cached(object):
    """Mixin: per-layer cache of device-side derived forms, keyed on the parameters' versions."""
    _acg_cache = None

    def _cache_key(self):
        return (ops.CONFIG_EPOCH,) + tuple((p.data_ptr(), p._version) for p in self.parameters(recurse=False))

    def _cached(self, build):
        key = self._cache_key()
        c = self._acg_cache
        if c is None or c[0] != key:
            c = (key, build())
            self._acg_cache = c
        return c[1]


def _padded_vec(v, npad):
    if v.numel() == npad and v.is_contiguous() and v.dtype == torch.float32:
        return v.detach()   # already a multiple of 16 channels: the kernels read the parameter itself (no copy, never stale)
    out = torch.empty(npad, device=v.device, dtype=torch.float32)
    ops._lib.call("acg_pad_vector", ops._ptr(v.detach().contiguous()), v.numel(), ops._ptr(out), npad, ops._stream())
    return out


# ------------------------------------------------------------------------------------------------
# parameter-holding layers (torch.nn layouts / reprs / state_dict keys; HIP forward)
# ------------------------------------------------------------------------------------------------
class Conv2d(nn.Conv2d, _Cached):
    """nn.Conv2d executed by acg_conv2d_fwd (implicit-GEMM MFMA)."""

    def packed(self):
        return self._cached(lambda: ops.PackedConv(self.weight, self.bias, cpad(self.in_channels), cpad(self.out_channels)))

    def forward_nhwc(self, x, act=ACT_NONE, reflect_pad=0, want_stats=None, want_identity=False, link_out=None,
                     link_in=None, skip_grad=None, s16=None, norm_sums=None):
        if reflect_pad:
            pad, mode = reflect_pad, PAD_REFLECT
        else:
            pad, mode = self.padding[0], PAD_ZERO
        x = restore_width(x, self.packed().Cis)
        out = ops.Conv2dFn.apply(x, self.weight, self.bias, self.packed(), self.stride[0], pad, mode, act, want_stats,
                                 want_identity, link_out, link_in, skip_grad, s16, norm_sums)
        if s16 is not None and s16.x:   # tag what left the convolution pre-split: its output (conv + ReLU) and the alias of x
            if want_identity:
                ops.tag_s16(out[1])
            if s16.y:
                ops.tag_s16(out[0] if want_identity else out)
        return out

    def dgrad_sums_ok(self, x):
        """does this (zero-padded) convolution's data gradient, for an NHWC input of x's shape, emit the backward sums of a norm
        in front of it (acg_conv2d_bwd_data_sums: the persistent row pipeline)?"""
        pk = self.packed()
        if x.dim() != 4 or x.shape[-1] != pk.Cis:
            return False
        d = ops.conv_desc(x.shape[0], x.shape[1], x.shape[2], pk.Cis, pk.Cos, pk.K, self.stride[0], self.padding[0], PAD_ZERO, pk.Ir, pk.Or)
        return bool(ops._lib.query("acg_conv2d_bwd_data_sums_supported", ctypes.byref(d)))

    def forward(self, input):
        return ops.ToNCHW.apply(self.forward_nhwc(ops.ToNHWC.apply(input, True)), self.out_channels)


def restore_width(x, C):
    """An image tensor stored C4 (ops.cimg) that reaches a layer which is not its thin-channel consumer (a 1x1 or a
    3 -> 3 convolution, a ConvTranspose2d, a norm: none of them on the networks' path) is widened to C16 with zero
    channels — and a C16 tensor of <= 4 real channels in front of a thin-input layer narrowed — by plain torch ops."""
    if x.shape[-1] == C:
        return x
    if x.shape[-1] < C:
        return torch.nn.functional.pad(x, (0, C - x.shape[-1]))
    return x[..., :C].contiguous()


class ConvTranspose2d(nn.ConvTranspose2d, _Cached):
    """nn.ConvTranspose2d(k3,s2,p1,op1) executed as four sub-pixel phase convolutions."""

    def packed(self):
        # weight (Cin_T, Cout_T, k, k) == OIHW of the Conv2d(Cout_T -> Cin_T) it is the adjoint of
        return self._cached(lambda: ops.PackedConv(self.weight, self.bias, cpad(self.out_channels), cpad(self.in_channels)))

    def forward_nhwc(self, x, act=ACT_NONE, want_stats=None):
        x = restore_width(x, cpad(self.in_channels))
        return ops.ConvTranspose2dFn.apply(x, self.weight, self.bias, self.packed(), self.stride[0], self.padding[0],
                                           self.output_padding[0], act, want_stats)

    def forward(self, input, output_size=None):
        return ops.ToNCHW.apply(self.forward_nhwc(ops.ToNHWC.apply(input)), self.out_channels)


class _BatchNormMixin(_Cached):
    def _gb(self):
        C = cpad(self.num_features) if self._pad16 else self.num_features
        return self._cached(lambda: (_padded_vec(self.weight, C), _padded_vec(self.bias, C)))

    def forward_act(self, x, act=ACT_NONE, res=None):
        """res: a residual added before the activation (ResnetBlock with --norm batch: ReLU(x + BN(conv(...))),
        networks.py:23-31 -> modules.py:227-235) — folded into the apply pass like InstanceNorm's"""
        if self._pad16:
            x = restore_width(x, cpad(self.num_features))
        C = x.shape[-1]
        g, b = self._gb()
        if not self.training:  # eval mode: normalise with the running buffers (model.eval(), train.py:258)
            return ops.NormAct.apply(x, self.weight, self.bias, res, "bn_eval", act, self.eps, g, b,
                                     self.running_mean.contiguous(), self.running_var.contiguous(), 0.0)
        if sync_bn_active():  # statistics over every rank's shard
            y = ops.SyncBatchNormAct.apply(x, self.weight, self.bias, ACT_NONE if res is not None else act, self.eps, g, b,
                                           self.running_mean, self.running_var, self.momentum)
            with torch.no_grad():
                self.num_batches_tracked += 1
            return y if res is None else add_act(y, res, act)
        if C == self.num_features and self.running_mean.is_contiguous() and self.running_var.is_contiguous():
            # no channel padding (every BatchNorm of the reference's networks: widths are multiples of 16): the statistics
            # kernel updates the running buffers in place
            y = ops.NormAct.apply(x, self.weight, self.bias, res, "bn", act, self.eps, g, b, self.running_mean,
                                  self.running_var, self.momentum)
        else:
            rm = torch.zeros(C, device=x.device, dtype=torch.float32)
            rv = torch.ones(C, device=x.device, dtype=torch.float32)
            rm[:self.num_features].copy_(self.running_mean)
            rv[:self.num_features].copy_(self.running_var)
            y = ops.NormAct.apply(x, self.weight, self.bias, res, "bn", act, self.eps, g, b, rm, rv, self.momentum)
            with torch.no_grad():
                self.running_mean.copy_(rm[:self.num_features])
                self.running_var.copy_(rv[:self.num_features])
        with torch.no_grad():
            self.num_batches_tracked += 1
        return y


def add_act(y, res, act):
    """act(y + res) through the norm kernels with constant statistics (mean 0, rstd 1, gamma 1, beta 0: the 'bn_eval' form of
    ops.NormAct, whose backward is dx = gy, dres = gy) — the residual behind a SyncBatchNorm, which has no fused form"""
    C = y.shape[-1]
    one, zero = torch.ones(C, device=y.device), torch.zeros(C, device=y.device)
    return ops.NormAct.apply(y, one, zero, res, "bn_eval", act, 0.0, one, zero, zero, one, 0.0)


class BatchNorm2d(nn.BatchNorm2d, _BatchNormMixin):
    _pad16 = True

    def forward(self, input):
        return ops.ToNCHW.apply(self.forward_act(ops.ToNHWC.apply(input)), self.num_features)


class BatchNorm1d(nn.BatchNorm1d, _BatchNormMixin):
    _pad16 = False

    def forward(self, input):
        return self.forward_act(input)


class Linear(nn.Linear):
    def forward_act(self, x, act=ACT_NONE, out_cols=None):
        return ops.LinearFn.apply(x, self.weight, self.bias, act, out_cols or self.out_features)

    def forward(self, input):
        return self.forward_act(input)


######################################################################
# Superclass of all Modules that take two inputs  (modules.py:15-17)
######################################################################
class TwoInputModule(nn.Module):
    def forward(self, input1, input2):
        raise NotImplementedError


class MergeModule(TwoInputModule):
    """o = module2(module1(x), z)  (modules.py:25-37)"""

    def __init__(self, module1, module2):
        super(MergeModule, self).__init__()
        self.module1 = module1
        self.module2 = module2

    def forward(self, input1, input2):
        return self.module2.forward(self.module1.forward(input1), input2)


class TwoInputSequential(nn.Sequential, TwoInputModule):
    """nn.Sequential that threads `input2` to every TwoInputModule child (modules.py:44-56)."""

    def __init__(self, *args):
        super(TwoInputSequential, self).__init__(*args)

    def forward(self, input1, input2):
        x = ops.ToNHWC.apply(input1, _starts_with_conv(self))
        y, C = run_sequence(list(self._modules.values()), x, input1.shape[1], cond_bank(self, as_latent(input2)))
        return ops.ToNCHW.apply(y, C)


def _starts_with_conv(seq):
    """does the layer list take its input straight into a Conv2d (then an image input may be stored C4, ops.cimg)?"""
    for m in seq._modules.values():
        if isinstance(m, nn.ReflectionPad2d):
            continue
        return isinstance(m, Conv2d) or (isinstance(m, MergeModule) and isinstance(m.module1, Conv2d))
    return False


def as_latent(z):
    """(N, nl, 1, 1) or (N, nl) -> contiguous (N, nl) (networks.py:427-428 does the same reshape)."""
    if z is None:
        return None
    return z.reshape(z.shape[0], -1).contiguous()


######################################################################
# InstanceNorm  (modules.py:64-98): biased variance, learnable scale~N(0,0.02) / shift=0
######################################################################
class InstanceNorm(nn.Module, _Cached):
    def __init__(self, num_features, affine=True, eps=1e-5):
        super(InstanceNorm, self).__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        self.scale = Parameter(torch.Tensor(num_features))
        self.shift = Parameter(torch.Tensor(num_features))
        self.reset_parameters()

    def reset_parameters(self):
        if self.affine:
            self.scale.data.normal_(mean=0., std=0.02)
            self.shift.data.zero_()
        self._acg_cache = None

    def _gb(self):
        C = cpad(self.num_features)
        if self.affine:
            return self._cached(lambda: (_padded_vec(self.scale, C), _padded_vec(self.shift, C)))
        dev = self.scale.device
        return torch.ones(C, device=dev), torch.zeros(C, device=dev)

    def forward_act(self, x, act=ACT_NONE, res=None, lazy_dres=None, stats=None, s16_out=False, s16_dx=False, sums=None):
        x = restore_width(x, cpad(self.num_features))
        g, b = self._gb()
        y = ops.NormAct.apply(x, self.scale, self.shift, res, "in", act, self.eps, g, b, None, None, 0.0, lazy_dres, stats,
                              s16_out, s16_dx, res is not None and ops.is_s16(res), sums)
        return ops.tag_s16(y) if s16_out else y

    def forward(self, input):
        return ops.ToNCHW.apply(self.forward_act(ops.ToNHWC.apply(input)), self.num_features)


InstanceNorm2d = nn.InstanceNorm2d if USE_PYTORCH_IN else InstanceNorm


######################################################################
# CondInstanceNorm  (modules.py:104-132): scale/shift = ReLU(1x1 conv(z)); UNBIASED variance
######################################################################
class CondInstanceNorm(TwoInputModule):
    def __init__(self, x_dim, z_dim, eps=1e-5):
        super(CondInstanceNorm, self).__init__()
        self.eps = eps
        self.x_dim, self.z_dim = x_dim, z_dim
        # torch.nn.Conv2d holders keep the reference's keys (shift_conv.0.weight (C, nl, 1, 1), ...); on a
        # (N, nl, 1, 1) latent a 1x1 conv is a dense layer, run by acg_linear_fwd with the ReLU fused.
        self.shift_conv = nn.Sequential(Conv2d(z_dim, x_dim, kernel_size=1, padding=0, bias=True), nn.ReLU(True))
        self.scale_conv = nn.Sequential(Conv2d(z_dim, x_dim, kernel_size=1, padding=0, bias=True), nn.ReLU(True))

    def forward_act(self, x, z, act=ACT_NONE, stats=None, s16_out=False, s16_dx=False, sums=None):
        x = restore_width(x, cpad(self.x_dim))
        Cp = x.shape[-1]
        bank = getattr(z, "_acg_bank", None)   # every scale / shift of the generator computed in one launch (cond_bank)
        if bank is not None and id(self) in bank and bank[id(self)][0].shape == (x.shape[0], Cp):
            sc, sh = bank[id(self)]
        else:
            sh = ops.LinearFn.apply(z, self.shift_conv[0].weight, self.shift_conv[0].bias, ACT_RELU, Cp)
            sc = ops.LinearFn.apply(z, self.scale_conv[0].weight, self.scale_conv[0].bias, ACT_RELU, Cp)
        y = ops.NormAct.apply(x, sc, sh, None, "cin", act, self.eps, None, None, None, None, 0.0, None, stats, s16_out, s16_dx,
                              False, sums)
        return ops.tag_s16(y) if s16_out else y

    def forward(self, input, noise):
        y = self.forward_act(ops.ToNHWC.apply(input), as_latent(noise))
        return ops.ToNCHW.apply(y, self.x_dim)


def cond_bank(root, z):
    """Tag latent z with the (scale, shift) pairs of every CondInstanceNorm under `root`, computed by ONE dense layer
    (ops.CondBankFn) instead of two small ones per norm; norms of a width of their own (not the majority's) keep theirs."""
    if not ops.COND_BANK:
        return z
    norms = [m for m in root.modules() if isinstance(m, CondInstanceNorm)]
    if len(norms) < 2:
        return z
    widths = [m.x_dim for m in norms]
    C = max(set(widths), key=widths.count)
    norms = [m for m in norms if m.x_dim == C and C % 16 == 0 and m.z_dim == norms[0].z_dim]
    if len(norms) < 2 or 2 * len(norms) > ops._lib.MAX_SEGMENTS // 2:
        return z
    params = []
    for m in norms:
        params += [m.scale_conv[0].weight, m.scale_conv[0].bias, m.shift_conv[0].weight, m.shift_conv[0].bias]
    outs = ops.CondBankFn.apply(z, C, *params)
    z = z.view(z.shape)   # the tag goes on an alias: the caller's tensor may feed other networks
    z._acg_bank = {id(m): (outs[2 * k], outs[2 * k + 1]) for k, m in enumerate(norms)}
    return z


_ACTS = {nn.ReLU: ACT_RELU, nn.LeakyReLU: ACT_LRELU, nn.Tanh: ACT_TANH}


def _act_of(m):
    for cls, a in _ACTS.items():
        if isinstance(m, cls):
            if cls is nn.LeakyReLU and abs(m.negative_slope - 0.2) > 1e-12:
                raise NotImplementedError("LeakyReLU slope %g (kernels implement the reference's 0.2)" % m.negative_slope)
            return a
    return None


def run_sequence(mods, x, C, z=None, res=None, last_block=False):
    """Interpret a reference-style layer list on an NHWC C16 tensor with peephole fusion:
         [ReflectionPad2d] Conv2d|ConvTranspose2d|MergeModule(conv, CondIN) [norm] [activation]
    `res`: residual input folded into the LAST norm of the list together with a ReLU
    (ResnetBlock / CINResnetBlock: out = ReLU(x + conv_block(x)), modules.py:185-188, 232-235); `last_block`: no block
    follows this one, so a pre-split block writes its output fp32 (what the layer behind the trunk reads).
    Returns (tensor, real channel count)."""
    i, n = 0, len(mods)
    last_norm = max([k for k, m in enumerate(mods) if isinstance(m, (InstanceNorm, CondInstanceNorm, BatchNorm2d))
                     or (isinstance(m, MergeModule))] + [-1]) if res is not None else -1
    reflect = 0
    skip_routed = False
    relu_link = None   # set by a conv+ReLU whose output goes straight into the next convolution (ops.ReluLink)
    skip_grad = None   # ops.SkipGrad slot shared by the block's first convolution and its last norm
    # Pre-split ("S16") storage of the residual trunk (ops.S16Plan): the norm in front of the first block writes its
    # output pre-split when that block can take it, every tensor a 3x3 trunk convolution reads stays pre-split from there
    # (block outputs, conv + ReLU outputs, the gradients the norms and the fused data gradients write), and the first
    # last block writes its output fp32 for the layer behind it (a block that cannot know it is last — called on its own —
    # leaves a pre-split tensor, decoded here).  s16 = this list is the inside of such a block.
    s16 = res is not None and ops.is_s16(x)
    nconv = 0
    # ops.NormSums: a norm whose pre-split output goes to ONE trunk convolution (inside a block: the next convolution of the
    # list; a block's output or the norm in front of the first block: the next block's first convolution, whose data gradient
    # already includes the skip gradient) shares a slot with it — the slot travels on the tensor between blocks
    ns_prev = getattr(x, "_acg_ns", None) if s16 else None
    while i < n:
        m = mods[i]
        if isinstance(m, nn.ReflectionPad2d):
            reflect = m.padding[0]
            i += 1
            continue
        if isinstance(m, nn.Dropout):   # --use_dropout: behind the first ReLU of a residual block (modules.py:167-168, 214-215)
            if m.training and m.p > 0.0:
                if ops.is_s16(x):
                    raise NotImplementedError("dropout on a pre-split tensor")
                x = ops.DropoutFn.apply(x, C, m.p)
            i += 1
            continue
        if isinstance(m, (ResnetBlock, CINResnetBlock)):
            x = m.forward_nhwc(x, z, last=not (i + 1 < n and isinstance(mods[i + 1], (ResnetBlock, CINResnetBlock))))
            ns_prev = None   # (the slot of the norm in front of the trunk belongs to the first block's first convolution)
            i += 1
            continue
        if res is None and ops.is_s16(x):
            x = ops.S16Decode.apply(x)
        conv, norm, stats = None, None, None
        if isinstance(m, MergeModule):
            conv, norm = m.module1, m.module2
            i += 1
        elif isinstance(m, (Conv2d, ConvTranspose2d)):
            conv = m
            i += 1
            if i < n and isinstance(mods[i], (InstanceNorm, CondInstanceNorm, BatchNorm2d)):
                norm = mods[i]
                i += 1
        elif isinstance(m, nn.Sigmoid):
            raise NotImplementedError("use_sigmoid / --no_lsgan: the reference's BCE branch is broken (model.py:59-63); "
                                      "only LSGAN is implemented")
        else:
            raise NotImplementedError("run_sequence: unexpected layer %s" % type(m).__name__)
        norm_idx = i - 1
        act = ACT_NONE
        if i < n and _act_of(mods[i]) is not None:
            act = _act_of(mods[i])
            i += 1
        # convolution (activation fused only when no norm follows)
        cact = act if norm is None else ACT_NONE
        if isinstance(conv, ConvTranspose2d):
            if reflect:
                raise NotImplementedError("reflection pad before ConvTranspose2d")
            stats = ops.ConvStats() if isinstance(norm, (InstanceNorm, CondInstanceNorm)) else None
            x = conv.forward_nhwc(x, cact, stats)
        else:  # an (Cond)InstanceNorm right behind the conv can take its statistics from the conv epilogue
            skip_here = res is not None and not skip_routed and x is res  # the block's FIRST convolution
            link_in, relu_link = relu_link, None
            link_out = None
            if cact == ACT_RELU:  # pad-conv-ReLU-pad-conv: the next convolution's data gradient applies this ReLU's mask
                k = i + 1 if (i < n and isinstance(mods[i], nn.ReflectionPad2d)) else i
                if k < n and isinstance(mods[k], Conv2d):
                    link_out = relu_link = ops.ReluLink()
            stats = ops.ConvStats() if isinstance(norm, (InstanceNorm, CondInstanceNorm)) else None
            if skip_here:
                skip_grad = ops.SkipGrad()
            plan = None
            if s16:
                # first convolution: writes pre-split iff it is a conv + ReLU (then its dy arrives pre-split and masked from the
                # next convolution's data gradient), dx fp32 + skip gradient; second: dy pre-split from the block-output norm,
                # dx pre-split with the first one's ReLU mask where that link exists (else fp32 for the norm in between)
                plan = ops.S16Plan(x=True, y=(cact == ACT_RELU and link_out is not None), gy=norm is not None,
                                   dx=(nconv > 0 and link_in is not None))
                if nconv == 0 and not skip_here:
                    raise NotImplementedError("pre-split trunk: the block's first layer must be its first convolution")
            nconv += 1
            # (outside the trunk: a norm + ReLU whose output goes to this convolution alone left its slot on the tensor — taken
            # where the data gradient runs on the row pipeline, acg_conv2d_bwd_data_sums_supported)
            if plan is None and res is None and ns_prev is None:
                ns_prev = getattr(x, "_acg_ns", None)
            ns_conv, ns_prev = (ns_prev if ((plan is not None and (skip_here or nconv > 1)) or (plan is None and res is None)) else None), None
            x = conv.forward_nhwc(x, cact, reflect, stats, skip_here, link_out, link_in, skip_grad if skip_here else None, plan,
                                  ns_conv)
            if skip_here:  # the skip connection continues from the conv's identity output: its gradient is added
                x, res = x  # inside that conv's data-gradient epilogue
                skip_routed = True
        reflect = 0
        C = conv.out_channels
        if norm is not None:
            fuse_res = res is not None and norm_idx == last_norm
            if fuse_res and act != ACT_NONE:
                raise NotImplementedError("residual fusion expects the block to end with its norm")
            # pre-split output: inside a pre-split block always (its consumer is a trunk convolution or the next block);
            # outside, when the next module is a block that can take it
            emit = s16 and not (fuse_res and last_block)
            if not s16 and res is None and act == ACT_RELU and i < n and isinstance(mods[i], (ResnetBlock, CINResnetBlock)) \
                    and isinstance(norm, (InstanceNorm, CondInstanceNorm)):
                emit = mods[i].s16_ok(x)
            # outside the trunk: the next layer is a zero-padded convolution whose data gradient (the gradient w.r.t. this norm's
            # output) can leave the norm's backward sums (ops.NormSums on fp32 tensors)
            rows_ns = (not s16 and res is None and not emit and ops.NORM_SUMS and act in (ACT_NONE, ACT_RELU) and i < n
                       and isinstance(mods[i], Conv2d) and isinstance(norm, (InstanceNorm, CondInstanceNorm))
                       and mods[i].dgrad_sums_ok(x))
            if isinstance(norm, CondInstanceNorm):
                if fuse_res:
                    raise NotImplementedError("residual after CondInstanceNorm")
                if s16 and act != ACT_RELU:
                    raise NotImplementedError("pre-split trunk: CondInstanceNorm without ReLU")
                ns_prev = ops.NormSums() if (emit or rows_ns) else None
                x = norm.forward_act(x, z, act, stats.part if stats is not None else None, emit, s16, ns_prev)
                if ns_prev is not None:
                    x._acg_ns = ns_prev
            elif isinstance(norm, InstanceNorm):
                # skip_routed: `res` is the identity output of the block's first convolution, i.e. its gradient goes to that
                # convolution's data-gradient epilogue and nowhere else -> it may stay un-materialised (ops.NormAct lazy_dres)
                if s16 and not (fuse_res and skip_routed):
                    raise NotImplementedError("pre-split trunk: InstanceNorm that is not the block output")
                ns_prev = ops.NormSums() if (emit or rows_ns) else None
                x = norm.forward_act(x, ACT_RELU if fuse_res else act, res if fuse_res else None,
                                     skip_grad if (fuse_res and skip_routed) else None, stats.part if stats is not None else None,
                                     emit, s16, ns_prev)
                if ns_prev is not None:
                    x._acg_ns = ns_prev
            else:   # BatchNorm2d (--norm batch; E_B always): a block-output norm takes the residual + ReLU in its apply pass
                x = norm.forward_act(x, ACT_RELU if fuse_res else act, res if fuse_res else None)
    if res is None and ops.is_s16(x):   # the list ended with a block
        x = ops.S16Decode.apply(x)
    return x, C


def _block_s16_ok(block, x):
    """can this residual block run on pre-split tensors (ops.S16Plan) for an input of x's shape?  Reflection-padded
    3x3 stride-1 C -> C convolutions, no dropout, and a library that takes pre-split operands in all three passes."""
    key = (tuple(x.shape), ops.CONFIG_EPOCH, ops.S16_ENABLED)
    cache = block.__dict__.setdefault("_acg_s16_ok", {})
    if key not in cache:
        mods = list(block.conv_block._modules.values())
        convs = [m.module1 if isinstance(m, MergeModule) else m for m in mods if isinstance(m, (Conv2d, MergeModule))]
        pads = [m for m in mods if isinstance(m, nn.ReflectionPad2d)]
        N, H, W, C = x.shape
        ok = (len(convs) == 2 and len(pads) == 2 and all(p.padding[0] == 1 for p in pads)
              and not any(isinstance(m, nn.Dropout) for m in mods)
              and all(c.kernel_size[0] == 3 and c.stride[0] == 1 and c.padding[0] == 0 and c.in_channels == C
                      and c.out_channels == C for c in convs)
              and ops.conv_s16_supported(N, H, W, C, 3, 1, PAD_REFLECT))
        cache[key] = bool(ok)
    return cache[key]


# padding of the residual blocks' 3x3 convolutions: 'reflect' = a ReflectionPad2d(1) module in front of an unpadded conv,
# 'zero' = the conv's own padding=1
_PADDING = {"reflect": (lambda: [nn.ReflectionPad2d(1)], 0), "zero": (lambda: [], 1)}


def _padding(padding_type):
    if padding_type not in _PADDING:
        raise NotImplementedError('padding [%s] is not implemented' % padding_type)
    return _PADDING[padding_type]


def _pad_front(padding_type):
    return _padding(padding_type)[0]()


def _conv_pad(padding_type):
    return _padding(padding_type)[1]


def _dropout(use_dropout):
    return [nn.Dropout(0.5)] if use_dropout else []


######################################################################
# CINResnetBlock  (modules.py:139-188)
######################################################################
class CINResnetBlock(TwoInputModule):
    def __init__(self, x_dim, z_dim, padding_type, norm_layer, use_dropout, use_bias):
        super(CINResnetBlock, self).__init__()
        self.conv_block = self.build_conv_block(x_dim, z_dim, padding_type, norm_layer, use_dropout, use_bias)
        self.relu = nn.ReLU(True)
        # the reference re-registers each child under a numeric name (modules.py:145-146): the
        # state_dict therefore carries aliased keys `model.1x.<j>.…`; reproduce them so that
        # reference checkpoints load with strict=True.
        for idx, module in enumerate(self.conv_block):
            self.add_module(str(idx), module)

    def build_conv_block(self, x_dim, z_dim, padding_type, norm_layer, use_dropout, use_bias):
        # stage 1 normalises with the latent-conditioned norm (MergeModule threads z to it), stage 2 with a plain
        # InstanceNorm; module ORDER is the checkpoint schema (state_dict keys conv_block.1.module1..., conv_block.4/5)
        conv = lambda: Conv2d(x_dim, x_dim, kernel_size=3, padding=_conv_pad(padding_type), bias=use_bias)
        stages = [[MergeModule(conv(), norm_layer(x_dim, z_dim)), nn.ReLU(True)] + _dropout(use_dropout),
                  [conv(), InstanceNorm2d(x_dim, affine=True)]]
        return TwoInputSequential(*[m for st in stages for m in _pad_front(padding_type) + st])

    def s16_ok(self, x):
        return _block_s16_ok(self, x)

    def forward_nhwc(self, x, z, last=False):
        y, _ = run_sequence(list(self.conv_block._modules.values()), x, None, z, res=x, last_block=last)
        return y

    def forward(self, x, noise):
        C = x.shape[1]
        return ops.ToNCHW.apply(self.forward_nhwc(ops.ToNHWC.apply(x), as_latent(noise)), C)


######################################################################
# ResnetBlock  (modules.py:193-235): pad-conv-ReLU-pad-conv-IN ; ReLU(x + out)
######################################################################
class ResnetBlock(nn.Module):
    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super(ResnetBlock, self).__init__()
        self.conv_block = self.build_conv_block(dim, padding_type, norm_layer, use_dropout, use_bias)
        self.relu = nn.ReLU(True)

    def build_conv_block(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        # two 3x3 stages; only the SECOND is normalised (no norm after the first conv: modules.py:211-215)
        conv = lambda: Conv2d(dim, dim, kernel_size=3, padding=_conv_pad(padding_type), bias=use_bias)
        stages = [[conv(), nn.ReLU(True)] + _dropout(use_dropout),
                  [conv(), norm_layer(dim)]]
        return Sequential(*[m for st in stages for m in _pad_front(padding_type) + st])

    def s16_ok(self, x):
        return _block_s16_ok(self, x)

    def forward_nhwc(self, x, z=None, last=False):
        y, _ = run_sequence(list(self.conv_block._modules.values()), x, None, None, res=x, last_block=last)
        return y

    def forward(self, x):
        C = x.shape[1]
        return ops.ToNCHW.apply(self.forward_nhwc(ops.ToNHWC.apply(x)), C)


class Sequential(nn.Sequential):
    """nn.Sequential whose forward (NCHW in / NCHW out) runs the fused HIP pipeline."""

    def forward(self, input):
        if input.dim() == 2:
            return run_dense(list(self._modules.values()), input)
        x = ops.ToNHWC.apply(input, _starts_with_conv(self))
        y, C = run_sequence(list(self._modules.values()), x, input.shape[1])
        return ops.ToNCHW.apply(y, C)


def run_dense(mods, x):
    """Linear [BatchNorm1d] [LeakyReLU] chains on (N, C) activations (DiscriminatorLatent)."""
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        if not isinstance(m, Linear):
            if isinstance(m, nn.Sigmoid):
                raise NotImplementedError("use_sigmoid: only LSGAN is implemented")
            raise NotImplementedError("run_dense: unexpected layer %s" % type(m).__name__)
        i += 1
        bn = None
        if i < n and isinstance(mods[i], BatchNorm1d):
            bn = mods[i]
            i += 1
        act = ACT_NONE
        if i < n and _act_of(mods[i]) is not None:
            act = _act_of(mods[i])
            i += 1
        if bn is None:
            x = m.forward_act(x, act, out_cols=(m.out_features + 3) // 4 * 4)
        else:
            if m.out_features % 4:
                raise NotImplementedError("BatchNorm1d width must be a multiple of 4")
            x = bn.forward_act(m.forward_act(x, ACT_NONE), act)
    return x
