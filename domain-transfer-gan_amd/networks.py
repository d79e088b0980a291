"""Network factories and classes — the API surface of
/root/reference/augmented_cyclegan/networks.py (define_G, define_stochastic_G, define_D_A,
define_D_B, define_LAT_D, define_E, print_network, weights_init and the six network classes,
same constructor arguments, same state_dict keys), executed by HIP kernels.

Differences, all documented in SURVEY.md §0:
  * `n_blocks` is honoured (the reference hard-codes 3 and ignores the argument, networks.py:173,225);
    the factories default to 3 = reference-faithful and accept `n_blocks=` as an extension.
  * multi-GPU is one process per GPU with gradient all-reduce (dist.py), so `gpu_ids` only selects
    "on the GPU" (non-empty) as in the reference; nn.parallel.data_parallel is not used.
  * LatentEncoder accepts any S >= 64 by averaging the mu/logvar maps over space (identity at the
    reference's 1x1 map, S = 64).
  * `define_D` is provided as a dispatcher alias (north-star name; the reference has define_D_A/B).
"""
import functools

import torch
import torch.nn as nn

from . import modules, ops
from .modules import (ResnetBlock, CondInstanceNorm, TwoInputSequential, CINResnetBlock, InstanceNorm2d,  # noqa: F401
                      Conv2d, ConvTranspose2d, BatchNorm2d, BatchNorm1d, Linear, Sequential, run_sequence,
                      run_dense, as_latent, mark_dirty, cond_bank)


###############################################################################
# Functions
###############################################################################
def weights_init(m):
    """networks.py:13-21"""
    classname = m.__class__.__name__
    if getattr(m, "_acg_cache", None) is not None:
        m._acg_cache = None   # `.data` writes below do not bump tensor versions: drop the layer's packed copies by hand
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
        if hasattr(m.bias, 'data'):
            m.bias.data.fill_(0)
    elif classname.find('BatchNorm2d') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_norm_layer(norm_type='instance'):
    """networks.py:23-30"""
    if norm_type == 'batch':
        norm_layer = functools.partial(BatchNorm2d, affine=True)
    elif norm_type == 'instance':
        norm_layer = functools.partial(InstanceNorm2d, affine=True)
    else:
        raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
    return norm_layer


def _finish(net, gpu_ids):
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.cuda()
    net.apply(weights_init)
    return net


def define_G(input_nc, output_nc, ngf, norm='instance', which_model_netG='resnet', use_dropout=False, gpu_ids=[],
             n_blocks=3):
    """networks.py:33-48"""
    norm_layer = get_norm_layer(norm_type=norm)
    netG = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout,
                           n_blocks=n_blocks, gpu_ids=gpu_ids)
    return _finish(netG, gpu_ids)


def define_stochastic_G(nlatent, input_nc, output_nc, ngf, norm='instance', which_model_netG='resnet',
                        use_dropout=False, gpu_ids=[], n_blocks=3):
    """networks.py:51-68"""
    netG = CINResnetGenerator(nlatent, input_nc, output_nc, ngf, norm_layer=CondInstanceNorm,
                              use_dropout=use_dropout, n_blocks=n_blocks, gpu_ids=gpu_ids)
    return _finish(netG, gpu_ids)


def define_D_A(input_nc, ndf, which_model_netD, norm, use_sigmoid=False, gpu_ids=[]):
    """networks.py:71-84"""
    netD = Discriminator_edges(input_nc, ndf, norm_layer=get_norm_layer(norm_type=norm), use_sigmoid=use_sigmoid,
                               gpu_ids=gpu_ids)
    return _finish(netD, gpu_ids)


def define_D_B(input_nc, ndf, which_model_netD, norm, use_sigmoid=False, gpu_ids=[]):
    """networks.py:87-100"""
    netD = Discriminator(input_nc, ndf, norm_layer=get_norm_layer(norm_type=norm), use_sigmoid=use_sigmoid,
                         gpu_ids=gpu_ids)
    return _finish(netD, gpu_ids)


def define_D(input_nc, ndf, which_model_netD='basic', norm='instance', use_sigmoid=False, gpu_ids=[], domain='B'):
    """North-star alias: dispatches to define_D_A / define_D_B (the reference has no define_D)."""
    f = define_D_A if str(domain).upper() == 'A' else define_D_B
    return f(input_nc, ndf, which_model_netD, norm, use_sigmoid, gpu_ids)


def define_LAT_D(nlatent, ndf, use_sigmoid=False, gpu_ids=[]):
    """networks.py:102-114"""
    netD = DiscriminatorLatent(nlatent, ndf, use_sigmoid=use_sigmoid, gpu_ids=gpu_ids)
    return _finish(netD, gpu_ids)


def define_E(nlatent, input_nc, nef, norm='batch', gpu_ids=[]):
    """networks.py:116-127"""
    netE = LatentEncoder(nlatent, input_nc, nef, norm_layer=get_norm_layer(norm_type=norm), gpu_ids=gpu_ids)
    return _finish(netE, gpu_ids)


def print_network(net, out_f=None):
    """networks.py:130-138"""
    num_params = 0
    for param in net.parameters():
        num_params += param.numel()
    if out_f is not None:
        out_f.write(net.__repr__() + "\n")
        out_f.write('Total number of parameters: %d\n' % num_params)
        out_f.flush()


##############################################################################
# Network classes — built from layer tables (same module order, hence same state_dict keys, as the reference)
##############################################################################
def _act(kind):
    return {"relu": lambda: nn.ReLU(True), "lrelu": lambda: nn.LeakyReLU(0.2, True), "tanh": lambda: nn.Tanh()}[kind]()


def _generator_layers(norm, input_nc, output_nc, ngf, n_blocks, make_block):
    """Stem (7x7 reflect, 3x3, 3x3 stride 2) -> n_blocks residual blocks at 4*ngf -> tail (ConvTranspose, 3x3, 7x7 + tanh).
    `norm(c)` builds the normalisation layer for c channels (InstanceNorm or CondInstanceNorm)."""
    layers = [nn.ReflectionPad2d(3)]
    # (cin, cout, kernel, stride, padding) of the three stem convolutions; each is followed by norm + ReLU
    for cin, cout, k, stride, pad in ((input_nc, ngf, 7, 1, 0), (ngf, 2 * ngf, 3, 1, 1), (2 * ngf, 4 * ngf, 3, 2, 1)):
        layers += [Conv2d(cin, cout, kernel_size=k, padding=pad, stride=stride, bias=True), norm(cout), _act("relu")]
    layers += [make_block(4 * ngf) for _ in range(n_blocks)]
    layers += [ConvTranspose2d(4 * ngf, 2 * ngf, kernel_size=3, stride=2, padding=1, output_padding=1, bias=True),
               norm(2 * ngf), _act("relu"),
               Conv2d(2 * ngf, ngf, kernel_size=3, padding=1, stride=1, bias=True), norm(ngf), _act("relu"),
               Conv2d(ngf, output_nc, kernel_size=7, padding=3), _act("tanh")]       # zero padding here (networks.py:187, 242)
    return layers


class CINResnetGenerator(nn.Module):
    """A -> B generator conditioned on the latent code through CondInstanceNorm — networks.py:149-197"""

    def __init__(self, nlatent, input_nc, output_nc, ngf=64, norm_layer=CondInstanceNorm, use_dropout=False,
                 n_blocks=9, gpu_ids=[], padding_type='reflect'):
        assert (n_blocks >= 0)
        super(CINResnetGenerator, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent, self.input_nc, self.output_nc = nlatent, input_nc, output_nc
        block = lambda c: CINResnetBlock(x_dim=c, z_dim=nlatent, padding_type=padding_type, norm_layer=norm_layer,
                                         use_dropout=use_dropout, use_bias=True)
        self.model = TwoInputSequential(*_generator_layers(lambda c: norm_layer(c, nlatent), input_nc, output_nc, ngf,
                                                           n_blocks, block))

    def forward_nhwc(self, x, z):
        """x: NHWC C16, z: (N, >=nlatent) -> NHWC C16 (output_nc valid channels)"""
        return run_sequence(list(self.model._modules.values()), x, self.input_nc, cond_bank(self.model, z))[0]

    def forward(self, input, noise):
        return self.model(input, noise)


class ResnetGenerator(nn.Module):
    """B -> A deterministic generator — networks.py:203-252"""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=InstanceNorm2d, use_dropout=False, n_blocks=9,
                 gpu_ids=[], padding_type='reflect'):
        assert (n_blocks >= 0)
        super(ResnetGenerator, self).__init__()
        self.gpu_ids = gpu_ids
        self.input_nc, self.output_nc = input_nc, output_nc
        block = lambda c: ResnetBlock(c, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout,
                                      use_bias=True)
        self.model = Sequential(*_generator_layers(norm_layer, input_nc, output_nc, ngf, n_blocks, block))

    def forward_nhwc(self, x):
        return run_sequence(list(self.model._modules.values()), x, self.input_nc)[0]

    def forward(self, input):
        return self.model(input)


def _patch_discriminator(input_nc, ndf, norm_layer, use_sigmoid, specs, head):
    """specs: (cout multiplier, kernel, stride) of the conv -> [norm] -> LeakyReLU(0.2) stages (the first has no norm);
    head: (kernel, padding) of the final 1-channel convolution."""
    seq, cin = [], input_nc
    for i, (mult, k, stride) in enumerate(specs):
        seq.append(Conv2d(cin, mult * ndf, kernel_size=k, stride=stride, padding=1, bias=True))
        if i > 0:
            seq.append(norm_layer(mult * ndf))
        seq.append(_act("lrelu"))
        cin = mult * ndf
    seq.append(Conv2d(cin, 1, kernel_size=head[0], stride=1, padding=head[1], bias=True))
    if use_sigmoid:
        seq.append(nn.Sigmoid())
    return Sequential(*seq)


class _ImageD(nn.Module):
    def forward_nhwc(self, x):
        return run_sequence(list(self.model._modules.values()), x, None)[0]

    def forward(self, input):
        return self.model(input)


class Discriminator(_ImageD):
    """D_B — networks.py:308-349: 4x4 convs, strides 2,2,1,1 then a 4x4 pad-1 head"""

    def __init__(self, input_nc, ndf=64, norm_layer=BatchNorm2d, use_sigmoid=False, gpu_ids=[]):
        super(Discriminator, self).__init__()
        self.gpu_ids = gpu_ids
        self.model = _patch_discriminator(input_nc, ndf, norm_layer, use_sigmoid,
                                          specs=((1, 4, 2), (2, 4, 2), (4, 4, 1), (4, 4, 1)), head=(4, 1))


class Discriminator_edges(_ImageD):
    """D_A — networks.py:352-393: four 3x3 stride-2 convs then a 4x4 valid head"""

    def __init__(self, input_nc, ndf=64, norm_layer=BatchNorm2d, use_sigmoid=False, gpu_ids=[]):
        super(Discriminator_edges, self).__init__()
        self.gpu_ids = gpu_ids
        self.model = _patch_discriminator(input_nc, ndf, norm_layer, use_sigmoid,
                                          specs=((1, 3, 2), (2, 3, 2), (4, 3, 2), (4, 3, 2)), head=(4, 0))


class DiscriminatorLatent(nn.Module):
    """D_z_B — networks.py:396-433: three Linear/BatchNorm1d/LeakyReLU stages and a Linear head"""

    def __init__(self, nlatent, ndf, use_sigmoid=False, gpu_ids=[]):
        super(DiscriminatorLatent, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent = nlatent
        seq, width = [], nlatent
        for _ in range(3):
            seq += [Linear(width, ndf), BatchNorm1d(ndf), _act("lrelu")]
            width = ndf
        seq.append(Linear(ndf, 1))
        if use_sigmoid:
            seq.append(nn.Sigmoid())
        self.model = Sequential(*seq)

    def forward_dense(self, z):
        """z: (N, >=nlatent) -> (N, 4) with column 0 valid"""
        mods = list(self.model._modules.values())
        fused = self._fused_args(mods, z)
        if fused is not None:
            return fused
        return run_dense(mods, z)

    def _fused_args(self, mods, z):
        """the whole chain as one launch per direction (ops.LatentMLPFn) when it is the standard train-mode chain, the batch
        fits one workgroup's LDS and BatchNorm statistics are local to this rank; None otherwise (layer by layer)"""
        if len(mods) != 10 or not self.training or modules.sync_bn_active():
            return None
        lins, bns = mods[0:10:3], mods[1:9:3]
        if not (all(isinstance(m, Linear) for m in lins) and all(isinstance(m, BatchNorm1d) for m in bns)
                and all(modules._act_of(m) == modules.ACT_LRELU for m in mods[2:9:3])):
            return None
        H = lins[0].out_features
        if any(b.num_features != H or b.eps != bns[0].eps or b.momentum != bns[0].momentum or not b.affine
               or not b.track_running_stats or b.momentum is None for b in bns):
            return None
        if not ops.latent_mlp_supported(z.shape[0], lins[0].in_features, H):
            return None
        out = ops.LatentMLPFn.apply(z, bns[0].eps, bns[0].momentum,
                                    [b.running_mean for b in bns] + [b.running_var for b in bns],
                                    *([m.weight for m in lins] + [m.bias for m in lins] + [b.weight for b in bns]
                                      + [b.bias for b in bns]))
        with torch.no_grad():
            for b in bns:
                b.num_batches_tracked += 1
        return out

    def forward(self, input):
        if input.dim() == 4:
            input = input.view(input.size(0), self.nlatent)
        return self.forward_dense(input.contiguous())[:, :1]


class LatentEncoder(nn.Module):
    """E_B — networks.py:438-482: five conv stages (only the first has a bias and no norm), then two 1x1 heads"""

    def __init__(self, nlatent, input_nc, nef, norm_layer, gpu_ids=[]):
        super(LatentEncoder, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent = nlatent
        seq = [Conv2d(input_nc, nef, kernel_size=3, stride=2, padding=1, bias=True), _act("relu")]
        # (cin, cout, kernel, stride, padding)
        for cin, cout, k, stride, pad in ((nef, 2 * nef, 3, 2, 1), (2 * nef, 4 * nef, 3, 2, 1), (4 * nef, 8 * nef, 3, 2, 1),
                                          (8 * nef, 8 * nef, 4, 1, 0)):
            seq += [Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False), norm_layer(cout), _act("relu")]
        self.conv_modules = Sequential(*seq)
        self.enc_mu = Conv2d(8 * nef, nlatent, kernel_size=1, stride=1, padding=0, bias=True)
        self.enc_logvar = Conv2d(8 * nef, nlatent, kernel_size=1, stride=1, padding=0, bias=True)

    def forward_nhwc(self, x):
        """x: NHWC C16 -> (mu, logvar), each (N, cpad(nlatent)) with nlatent valid columns"""
        h = run_sequence(list(self.conv_modules._modules.values()), x, None)[0]
        return ops.SpatialMean.apply(self.enc_mu.forward_nhwc(h)), ops.SpatialMean.apply(self.enc_logvar.forward_nhwc(h))

    def forward(self, input):
        mu, logvar = self.forward_nhwc(ops.ToNHWC.apply(input, True))
        return mu[:, :self.nlatent], logvar[:, :self.nlatent]
