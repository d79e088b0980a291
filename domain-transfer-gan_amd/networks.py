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

from . import ops
from .modules import (ResnetBlock, CondInstanceNorm, TwoInputSequential, CINResnetBlock, InstanceNorm2d,  # noqa: F401
                      Conv2d, ConvTranspose2d, BatchNorm2d, BatchNorm1d, Linear, Sequential, run_sequence,
                      run_dense, as_latent)


###############################################################################
# Functions
###############################################################################
def weights_init(m):
    """networks.py:13-21"""
    classname = m.__class__.__name__
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
# Network Classes
##############################################################################
class CINResnetGenerator(nn.Module):
    """networks.py:149-197"""

    def __init__(self, nlatent, input_nc, output_nc, ngf=64, norm_layer=CondInstanceNorm, use_dropout=False,
                 n_blocks=9, gpu_ids=[], padding_type='reflect'):
        assert (n_blocks >= 0)
        super(CINResnetGenerator, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent, self.input_nc, self.output_nc = nlatent, input_nc, output_nc
        model = [
            nn.ReflectionPad2d(3),
            Conv2d(input_nc, ngf, kernel_size=7, padding=0, stride=1, bias=True),
            norm_layer(ngf, nlatent),
            nn.ReLU(True),
            Conv2d(ngf, 2 * ngf, kernel_size=3, padding=1, stride=1, bias=True),
            norm_layer(2 * ngf, nlatent),
            nn.ReLU(True),
            Conv2d(2 * ngf, 4 * ngf, kernel_size=3, padding=1, stride=2, bias=True),
            norm_layer(4 * ngf, nlatent),
            nn.ReLU(True)
        ]
        for i in range(n_blocks):
            model += [CINResnetBlock(x_dim=4 * ngf, z_dim=nlatent, padding_type=padding_type, norm_layer=norm_layer,
                                     use_dropout=use_dropout, use_bias=True)]
        model += [
            ConvTranspose2d(4 * ngf, 2 * ngf, kernel_size=3, stride=2, padding=1, output_padding=1, bias=True),
            norm_layer(2 * ngf, nlatent),
            nn.ReLU(True),
            Conv2d(2 * ngf, ngf, kernel_size=3, padding=1, stride=1, bias=True),
            norm_layer(ngf, nlatent),
            nn.ReLU(True),
            Conv2d(ngf, output_nc, kernel_size=7, padding=3),
            nn.Tanh()
        ]
        self.model = TwoInputSequential(*model)

    def forward_nhwc(self, x, z):
        """x: NHWC C16, z: (N, >=nlatent) -> NHWC C16 (output_nc valid channels)"""
        y, _ = run_sequence(list(self.model._modules.values()), x, self.input_nc, z)
        return y

    def forward(self, input, noise):
        return self.model(input, noise)


class ResnetGenerator(nn.Module):
    """networks.py:203-252"""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=InstanceNorm2d, use_dropout=False, n_blocks=9,
                 gpu_ids=[], padding_type='reflect'):
        assert (n_blocks >= 0)
        super(ResnetGenerator, self).__init__()
        self.gpu_ids = gpu_ids
        self.input_nc, self.output_nc = input_nc, output_nc
        model = [
            nn.ReflectionPad2d(3),
            Conv2d(input_nc, ngf, kernel_size=7, padding=0, stride=1, bias=True),
            norm_layer(ngf),
            nn.ReLU(True),
            Conv2d(ngf, 2 * ngf, kernel_size=3, padding=1, stride=1, bias=True),
            norm_layer(2 * ngf),
            nn.ReLU(True),
            Conv2d(2 * ngf, 4 * ngf, kernel_size=3, padding=1, stride=2, bias=True),
            norm_layer(4 * ngf),
            nn.ReLU(True),
        ]
        for i in range(n_blocks):
            model += [ResnetBlock(4 * ngf, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout,
                                  use_bias=True)]
        model += [
            ConvTranspose2d(4 * ngf, 2 * ngf, kernel_size=3, stride=2, padding=1, output_padding=1, bias=True),
            norm_layer(2 * ngf),
            nn.ReLU(True),
            Conv2d(2 * ngf, ngf, kernel_size=3, padding=1, bias=True),
            norm_layer(ngf),
            nn.ReLU(True),
            Conv2d(ngf, output_nc, kernel_size=7, padding=3),
            nn.Tanh()
        ]
        self.model = Sequential(*model)

    def forward_nhwc(self, x):
        y, _ = run_sequence(list(self.model._modules.values()), x, self.input_nc)
        return y

    def forward(self, input):
        return self.model(input)


class _ImageD(nn.Module):
    def forward_nhwc(self, x):
        y, _ = run_sequence(list(self.model._modules.values()), x, None)
        return y

    def forward(self, input):
        return self.model(input)


class Discriminator(_ImageD):
    """D_B — networks.py:308-349"""

    def __init__(self, input_nc, ndf=64, norm_layer=BatchNorm2d, use_sigmoid=False, gpu_ids=[]):
        super(Discriminator, self).__init__()
        self.gpu_ids = gpu_ids
        use_bias = True
        kw = 4
        sequence = [
            Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=1, bias=True),
            nn.LeakyReLU(0.2, True),
            Conv2d(ndf, 2 * ndf, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(2 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(2 * ndf, 4 * ndf, kernel_size=kw, stride=1, padding=1, bias=use_bias),
            norm_layer(4 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(4 * ndf, 4 * ndf, kernel_size=kw, stride=1, padding=1, bias=use_bias),
            norm_layer(4 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(4 * ndf, 1, kernel_size=kw, stride=1, padding=1)
        ]
        if use_sigmoid:
            sequence += [nn.Sigmoid()]
        self.model = Sequential(*sequence)


class Discriminator_edges(_ImageD):
    """D_A — networks.py:352-393"""

    def __init__(self, input_nc, ndf=64, norm_layer=BatchNorm2d, use_sigmoid=False, gpu_ids=[]):
        super(Discriminator_edges, self).__init__()
        self.gpu_ids = gpu_ids
        use_bias = True
        kw = 3
        sequence = [
            Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=1, bias=True),
            nn.LeakyReLU(0.2, True),
            Conv2d(ndf, 2 * ndf, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(2 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(2 * ndf, 4 * ndf, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(4 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(4 * ndf, 4 * ndf, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(4 * ndf),
            nn.LeakyReLU(0.2, True),
            Conv2d(4 * ndf, 1, kernel_size=4, stride=1, padding=0, bias=True)
        ]
        if use_sigmoid:
            sequence += [nn.Sigmoid()]
        self.model = Sequential(*sequence)


class DiscriminatorLatent(nn.Module):
    """D_z_B — networks.py:396-433"""

    def __init__(self, nlatent, ndf, use_sigmoid=False, gpu_ids=[]):
        super(DiscriminatorLatent, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent = nlatent
        sequence = [
            Linear(nlatent, ndf),
            BatchNorm1d(ndf),
            nn.LeakyReLU(0.2, True),
            Linear(ndf, ndf),
            BatchNorm1d(ndf),
            nn.LeakyReLU(0.2, True),
            Linear(ndf, ndf),
            BatchNorm1d(ndf),
            nn.LeakyReLU(0.2, True),
            Linear(ndf, 1)
        ]
        if use_sigmoid:
            sequence += [nn.Sigmoid()]
        self.model = Sequential(*sequence)

    def forward_dense(self, z):
        """z: (N, >=nlatent) -> (N, 4) with column 0 valid"""
        return run_dense(list(self.model._modules.values()), z)

    def forward(self, input):
        if input.dim() == 4:
            input = input.view(input.size(0), self.nlatent)
        return self.forward_dense(input.contiguous())[:, :1]


class LatentEncoder(nn.Module):
    """E_B — networks.py:438-482"""

    def __init__(self, nlatent, input_nc, nef, norm_layer, gpu_ids=[]):
        super(LatentEncoder, self).__init__()
        self.gpu_ids = gpu_ids
        self.nlatent = nlatent
        use_bias = False
        kw = 3
        sequence = [
            Conv2d(input_nc, nef, kernel_size=kw, stride=2, padding=1, bias=True),
            nn.ReLU(True),
            Conv2d(nef, 2 * nef, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(2 * nef),
            nn.ReLU(True),
            Conv2d(2 * nef, 4 * nef, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(4 * nef),
            nn.ReLU(True),
            Conv2d(4 * nef, 8 * nef, kernel_size=kw, stride=2, padding=1, bias=use_bias),
            norm_layer(8 * nef),
            nn.ReLU(True),
            Conv2d(8 * nef, 8 * nef, kernel_size=4, stride=1, padding=0, bias=use_bias),
            norm_layer(8 * nef),
            nn.ReLU(True),
        ]
        self.conv_modules = Sequential(*sequence)
        self.enc_mu = Conv2d(8 * nef, nlatent, kernel_size=1, stride=1, padding=0, bias=True)
        self.enc_logvar = Conv2d(8 * nef, nlatent, kernel_size=1, stride=1, padding=0, bias=True)

    def forward_nhwc(self, x):
        """x: NHWC C16 -> (mu, logvar), each (N, cpad(nlatent)) with nlatent valid columns"""
        h, _ = run_sequence(list(self.conv_modules._modules.values()), x, None)
        mu = ops.SpatialMean.apply(self.enc_mu.forward_nhwc(h))
        logvar = ops.SpatialMean.apply(self.enc_logvar.forward_nhwc(h))
        return mu, logvar

    def forward(self, input):
        mu, logvar = self.forward_nhwc(ops.ToNHWC.apply(input))
        return mu[:, :self.nlatent], logvar[:, :self.nlatent]
