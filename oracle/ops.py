"""ORACLE (test infrastructure): NumPy/C restatement of every op on the hot path,
forward AND adjoint.  NCHW layout, weight OIHW — the reference's own layout.

Each function cites the reference line (or the torch semantic) it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

from .tape import T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _lib(dtype):
    """Load (building on demand with gcc) the plain-C conv restatement for dtype."""
    key = np.dtype(dtype).name
    if key not in _LIBS:
        tag = {"float32": "f32", "float64": "f64"}[key]
        path = os.path.join(_HERE, "_build", "libconv_ref_%s.so" % tag)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        lib = ctypes.CDLL(path)
        ip = ctypes.c_int
        vp = ctypes.c_void_p
        lib.conv_fwd.argtypes = [vp, vp, vp, vp] + [ip] * 9
        lib.conv_dgrad.argtypes = [vp, vp, vp] + [ip] * 9
        lib.conv_wgrad.argtypes = [vp, vp, vp] + [ip] * 9
        assert lib.conv_ref_real_bytes() == np.dtype(dtype).itemsize
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(a)


# ----------------------------------------------------------------------------
# padding (torch ReflectionPad2d / zero `padding=` of Conv2d) and adjoints
# ----------------------------------------------------------------------------
def pad_fwd(x, p, mode):
    if p == 0:
        return x
    m = "reflect" if mode == "reflect" else "constant"
    return np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)), mode=m)


def pad_bwd(gp, p, mode):
    """Adjoint of pad_fwd.  Reflection: border gradients fold back onto the
    mirrored interior pixels (torch reflection_pad2d_backward)."""
    if p == 0:
        return gp
    if mode != "reflect":
        return _c(gp[:, :, p:-p, p:-p])
    g = gp.copy()
    H = g.shape[2]
    for i in range(p):  # rows: padded row i mirrors padded row 2p-i; row H-1-i mirrors H-1-2p+i
        g[:, :, 2 * p - i, :] += g[:, :, i, :]
        g[:, :, H - 1 - 2 * p + i, :] += g[:, :, H - 1 - i, :]
    g = g[:, :, p:-p, :]
    W = g.shape[3]
    for j in range(p):
        g[:, :, :, 2 * p - j] += g[:, :, :, j]
        g[:, :, :, W - 1 - 2 * p + j] += g[:, :, :, W - 1 - j]
    return _c(g[:, :, :, p:-p])


# ----------------------------------------------------------------------------
# raw conv primitives on pre-padded input (plain C, oracle/csrc/conv_ref.c)
# ----------------------------------------------------------------------------
def _conv_fwd_raw(xp, w, b, s):
    N, Ci, Hp, Wp = xp.shape
    Co, _, K, _ = w.shape
    Ho, Wo = (Hp - K) // s + 1, (Wp - K) // s + 1
    y = np.empty((N, Co, Ho, Wo), xp.dtype)
    xp, w = _c(xp), _c(w)
    bb = _c(b) if b is not None else None
    _lib(xp.dtype).conv_fwd(_p(xp), _p(w), _p(bb) if bb is not None else None, _p(y),
                            N, Ci, Hp, Wp, Co, K, s, Ho, Wo)
    return y


def _conv_dgrad_raw(dy, w, s, Hp, Wp):
    N, Co, Ho, Wo = dy.shape
    _, Ci, K, _ = w.shape
    dxp = np.empty((N, Ci, Hp, Wp), dy.dtype)
    dy, w = _c(dy), _c(w)
    _lib(dy.dtype).conv_dgrad(_p(dy), _p(w), _p(dxp), N, Ci, Hp, Wp, Co, K, s, Ho, Wo)
    return dxp


def _conv_wgrad_raw(xp, dy, K, s):
    N, Ci, Hp, Wp = xp.shape
    _, Co, Ho, Wo = dy.shape
    dw = np.empty((Co, Ci, K, K), xp.dtype)
    xp, dy = _c(xp), _c(dy)
    _lib(xp.dtype).conv_wgrad(_p(xp), _p(dy), _p(dw), N, Ci, Hp, Wp, Co, K, s, Ho, Wo)
    return dw


# ----------------------------------------------------------------------------
# nn.Conv2d (+ optional preceding nn.ReflectionPad2d) — networks.py:159-160 etc.
# ----------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, pad=0, pad_mode="zero"):
    K = w.v.shape[2]
    xp = pad_fwd(x.v, pad, pad_mode)
    y = _conv_fwd_raw(xp, w.v, None if b is None else b.v, stride)
    Hp, Wp = xp.shape[2], xp.shape[3]

    def bw(g):
        gx = gw = gb = None
        if x.req:
            gx = pad_bwd(_conv_dgrad_raw(g, w.v, stride, Hp, Wp), pad, pad_mode)
        if w.req:
            gw = _conv_wgrad_raw(xp, g, K, stride)
        if b is not None and b.req:
            gb = g.sum(axis=(0, 2, 3))
        return (gx, gw, gb) if b is not None else (gx, gw)

    return T(y, (x, w, b) if b is not None else (x, w), bw)


# ----------------------------------------------------------------------------
# nn.ConvTranspose2d(k3,s2,p1,op1) — networks.py:178-179, 231-234.
# torch semantic: the adjoint (data-gradient) of Conv2d(Cout->Cin... ) with the
# SAME weight tensor, weight layout (Cin_T, Cout_T, kh, kw).
# ----------------------------------------------------------------------------
def conv_transpose2d(x, w, b, stride=2, pad=1, out_pad=1):
    N, Ci, H, W = x.v.shape
    K = w.v.shape[2]
    Ho = (H - 1) * stride - 2 * pad + K + out_pad
    Wo = (W - 1) * stride - 2 * pad + K + out_pad
    Hp, Wp = Ho + 2 * pad, Wo + 2 * pad

    def crop(a):
        return _c(a[:, :, pad:pad + Ho, pad:pad + Wo])

    y = crop(_conv_dgrad_raw(x.v, w.v, stride, Hp, Wp))
    if b is not None:
        y = y + b.v[None, :, None, None]

    def bw(g):
        gp = pad_fwd(g, pad, "zero")
        gx = _conv_fwd_raw(gp, w.v, None, stride) if x.req else None
        gx = None if gx is None else _c(gx[:, :, :H, :W])
        gw = _conv_wgrad_raw(gp, x.v, K, stride) if w.req else None
        gb = g.sum(axis=(0, 2, 3)) if (b is not None and b.req) else None
        return (gx, gw, gb) if b is not None else (gx, gw)

    return T(y, (x, w, b) if b is not None else (x, w), bw)


# ----------------------------------------------------------------------------
# InstanceNorm — modules.py:64-97 (biased variance, eps inside rsqrt, affine)
# ----------------------------------------------------------------------------
def instance_norm(x, scale, shift, eps=1e-5):
    xv = x.v
    mean = xv.mean(axis=(2, 3), keepdims=True)
    cen = xv - mean
    rstd = 1.0 / np.sqrt((cen ** 2).mean(axis=(2, 3), keepdims=True) + eps)
    rstd = rstd.astype(xv.dtype)
    xhat = cen * rstd
    y = xhat * scale.v[None, :, None, None] + shift.v[None, :, None, None]

    def bw(g):
        gs = (g * xhat).sum(axis=(0, 2, 3))
        gb = g.sum(axis=(0, 2, 3))
        gh = g * scale.v[None, :, None, None]
        gx = rstd * (gh - gh.mean(axis=(2, 3), keepdims=True)
                     - xhat * (gh * xhat).mean(axis=(2, 3), keepdims=True))
        return gx, gs, gb

    return T(y, (x, scale, shift), bw)


# ----------------------------------------------------------------------------
# CondInstanceNorm normalisation — modules.py:121-131 (UNBIASED variance via
# Tensor.var default; per-sample per-channel affine from the latent code)
# scale, shift: (N, C, 1, 1) tape values (outputs of the 1x1 convs + ReLU)
# ----------------------------------------------------------------------------
def cond_instance_norm(x, scale, shift, eps=1e-5):
    xv = x.v
    hw = xv.shape[2] * xv.shape[3]
    mean = xv.mean(axis=(2, 3), keepdims=True)
    cen = xv - mean
    var = (cen ** 2).sum(axis=(2, 3), keepdims=True) / (hw - 1)
    rstd = (1.0 / np.sqrt(var + eps)).astype(xv.dtype)
    xhat = cen * rstd
    y = xhat * scale.v + shift.v

    def bw(g):
        gs = (g * xhat).sum(axis=(2, 3), keepdims=True)
        gb = g.sum(axis=(2, 3), keepdims=True)
        gh = g * scale.v
        gx = rstd * (gh - gh.mean(axis=(2, 3), keepdims=True)
                     - xhat * (gh * xhat).sum(axis=(2, 3), keepdims=True) / (hw - 1))
        return gx, gs, gb

    return T(y, (x, scale, shift), bw)


# ----------------------------------------------------------------------------
# BatchNorm2d / BatchNorm1d in TRAIN mode — networks.py:407-415, 450-466
# torch semantic: normalise with biased batch variance; running_var updated with
# the UNBIASED one; momentum 0.1; eps 1e-5.  `stats` = dict(running_mean,
# running_var, num_batches_tracked) updated in place when training.
# ----------------------------------------------------------------------------
def batch_norm(x, weight, bias, stats, training=True, eps=1e-5, momentum=0.1):
    xv = x.v
    axes = (0, 2, 3) if xv.ndim == 4 else (0,)
    bshape = (1, -1, 1, 1) if xv.ndim == 4 else (1, -1)
    cnt = xv.size // xv.shape[1]
    if training:
        mean = xv.mean(axis=axes)
        var = ((xv - mean.reshape(bshape)) ** 2).mean(axis=axes)
        if stats is not None:
            unb = var * (cnt / max(cnt - 1, 1))
            stats["running_mean"] = ((1 - momentum) * stats["running_mean"] + momentum * mean).astype(xv.dtype)
            stats["running_var"] = ((1 - momentum) * stats["running_var"] + momentum * unb).astype(xv.dtype)
            stats["num_batches_tracked"] = stats["num_batches_tracked"] + 1
    else:
        mean, var = stats["running_mean"], stats["running_var"]
    rstd = (1.0 / np.sqrt(var + eps)).astype(xv.dtype).reshape(bshape)
    xhat = (xv - mean.reshape(bshape)) * rstd
    y = xhat * weight.v.reshape(bshape) + bias.v.reshape(bshape)

    def bw(g):
        gw = (g * xhat).sum(axis=axes)
        gb = g.sum(axis=axes)
        gh = g * weight.v.reshape(bshape)
        if training:
            gx = rstd * (gh - gh.mean(axis=axes, keepdims=True)
                         - xhat * (gh * xhat).mean(axis=axes, keepdims=True))
        else:
            gx = rstd * gh
        return gx, gw, gb

    return T(y, (x, weight, bias), bw)


# ----------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------
def relu(x):
    m = x.v > 0
    return T(np.where(m, x.v, 0).astype(x.v.dtype), (x,), lambda g: (g * m,))


def leaky_relu(x, slope=0.2):
    m = x.v > 0
    return T(np.where(m, x.v, x.v * slope).astype(x.v.dtype), (x,),
             lambda g: (np.where(m, g, g * slope).astype(g.dtype),))


def tanh(x):
    y = np.tanh(x.v)
    return T(y, (x,), lambda g: (g * (1 - y * y),))


# ----------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------
def dropout(x, keep, p=0.5):
    """torch.nn.Dropout(p) in training mode (modules.py:167-168, 214-215: `nn.Dropout(0.5)` behind the first ReLU of both
    block types): y = x * keep / (1 - p) with `keep` the Bernoulli(1 - p) draw — GIVEN here (the fixtures inject the same
    draw into the reference, tools/make_goldens.py), so the op is a pure function of its inputs."""
    m = (np.asarray(keep) != 0).astype(x.v.dtype) * x.v.dtype.type(1.0 / (1.0 - p))
    return T(x.v * m, (x,), lambda g: (g * m,))


def dropout_keep(seed, k, shape, p=0.5):
    """the keep mask of the k-th Dropout call of a fixture (NCHW, the reference's layout): regenerated from the seed on
    every box, like the parameters (oracle/recipe.py)"""
    return np.random.RandomState((seed * 1000003 + k) & 0x7FFFFFFF).rand(*shape) >= p


def add(a, b):
    return T(a.v + b.v, (a, b), lambda g: (g, g))


def scale(a, s):
    s = a.v.dtype.type(s)
    return T(a.v * s, (a,), lambda g: (g * s,))


def cat_channels(a, b):
    ca = a.v.shape[1]
    return T(np.concatenate([a.v, b.v], axis=1), (a, b),
             lambda g: (_c(g[:, :ca]), _c(g[:, ca:])))


def reshape(a, shape):
    old = a.v.shape
    return T(a.v.reshape(shape), (a,), lambda g: (g.reshape(old),))


def spatial_mean(a):
    """(N,C,H,W) -> (N,C): the build's documented extension of LatentEncoder to
    S != 64.  Identity when H=W=1, which is the only case the
    reference itself defines (networks.py:482)."""
    hw = a.v.shape[2] * a.v.shape[3]
    shp = a.v.shape
    return T(a.v.mean(axis=(2, 3)), (a,),
             lambda g: (np.broadcast_to(g[:, :, None, None] / hw, shp).astype(g.dtype),))


def linear(x, w, b):
    """nn.Linear: y = x W^T + b — networks.py:406-418."""
    y = x.v @ w.v.T + b.v

    def bw(g):
        return g @ w.v, g.T @ x.v, g.sum(axis=0)

    return T(y, (x, w, b), bw)


# ----------------------------------------------------------------------------
# losses — model.py:56-72 (LSGAN branch), F.l1_loss (mean), model.py:45-53
# ----------------------------------------------------------------------------
def mse_to_const(pred, target):
    """F.mse_loss(pred, full_like(pred, target)) — model.py:65-70."""
    d = pred.v - pred.v.dtype.type(target)
    n = d.size
    return T(np.asarray((d * d).mean(), pred.v.dtype), (pred,),
             lambda g: (g * 2.0 * d / n,))


def l1_loss(a, b):
    """F.l1_loss(a, b), mean reduction — model.py:117,391,486."""
    d = a.v - b.v
    n = d.size
    sg = np.sign(d)
    return T(np.asarray(np.abs(d).mean(), a.v.dtype), (a, b),
             lambda g: (g * sg / n, -g * sg / n))


def kld_std_gauss(mu, logvar):
    """model.py:45-53: -0.5*sum(logvar + 1 - mu^2 - exp(logvar), dim=1)."""
    e = np.exp(logvar.v)
    v = -0.5 * np.sum(logvar.v + 1.0 - mu.v ** 2 - e, axis=1)
    return T(v.astype(mu.v.dtype), (mu, logvar),
             lambda g: (g[:, None] * mu.v, g[:, None] * (-0.5) * (1.0 - e)))


def mean0(a):
    n = a.v.shape[0]
    shp = a.v.shape
    return T(np.asarray(a.v.mean(axis=0), a.v.dtype), (a,),
             lambda g: (np.broadcast_to(g / n, shp).astype(a.v.dtype),))


def log_prob_gaussian(z, mu, logvar):
    """model.py:31-34."""
    e = np.exp(logvar.v)
    d = z.v - mu.v
    v = -0.5 * logvar.v - d * d / (2.0 * e) - 0.5 * np.log(2 * np.pi)

    def bw(g):
        gz = -g * d / e
        return gz, -gz, g * (-0.5 + d * d / (2.0 * e))

    return T(v.astype(mu.v.dtype), (z, mu, logvar), bw)


def mean_all(a):
    n = a.v.size
    shp = a.v.shape
    return T(np.asarray(a.v.mean(), a.v.dtype), (a,),
             lambda g: (np.broadcast_to(g / n, shp).astype(a.v.dtype),))


def gauss_reparametrize(mu, logvar, eps):
    """model.py:15-22 with the N(0,1) draw `eps` (N, n_sample, nl) supplied by the
    caller: z = clamp(eps*exp(0.5*logvar) + mu, -4, 4) -> (N*n_sample, nl, 1, 1)."""
    std = np.exp(0.5 * logvar.v)
    z = eps * std[:, None, :] + mu.v[:, None, :]
    inside = (z > -4.0) & (z < 4.0)
    zc = np.clip(z, -4.0, 4.0)
    shp = zc.shape

    def bw(g):
        g = g.reshape(shp) * inside
        return g.sum(axis=1), (g * eps * std[:, None, :] * 0.5).sum(axis=1)

    return T(zc.reshape(shp[0] * shp[1], shp[2], 1, 1).astype(mu.v.dtype), (mu, logvar), bw)
