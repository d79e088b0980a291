"""torch.autograd.Function wrappers over the C ABI (libacgan_hip.so).

PyTorch is plumbing here: it owns device memory (caching allocator), the stream and the
autograd tape; every forward/backward computation on activations is a HIP kernel reached
through ctypes.  Internal activation layout: fp32 NHWC with channels padded to 16 ("C16"),
carried as torch tensors of shape (N, H, W, Cp) — or (N, Cp) for latent / MLP activations.

There is no CPU path: tensors must live on a ROCm device and the library must load.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, ConvDesc  # noqa: F401

_VP = ctypes.c_void_p


def cpad(c):
    """stored channel count for `c` real channels (feature maps, latents: multiples of 16)"""
    return (int(c) + 15) // 16 * 16


def cimg(c):
    """stored channel count of an IMAGE tensor (network inputs / outputs, their gradients): 4 for up to 4 real channels
    ("C4": one 16-byte load per pixel instead of a 64-byte row that is three quarters padding), else as cpad.  Only the thin
    side of a thin convolution layer reads or writes such a tensor (conv_thin_sides); everything else sees multiples of 16."""
    return 4 if int(c) <= 4 else cpad(c)


def conv_thin_sides(Or, Ir, K):
    """-> (input may be stored C4, output is stored C4) for a Conv2d of real widths Ir -> Or: the library's thin-channel
    kernels (K flattened over (tap, 4 channels)) take the thin side of a layer with K > 1 whose other side is wider"""
    ti, to = 1 <= Ir <= 4, 1 <= Or <= 4
    return (ti and not to and K > 1), (to and not ti and K > 1)


def _ptr(t):
    return None if t is None else _VP(t.data_ptr())


def _stream():
    return _VP(torch.cuda.current_stream().cuda_stream)


def _check(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.AcgError("acgan_hip kernels need tensors on a ROCm device (got %s); there is no CPU path" % t.device)
        if t.dtype != torch.float32:
            raise _lib.AcgError("acgan_hip kernels are fp32 (got %s)" % t.dtype)
        if not t.is_contiguous():
            raise _lib.AcgError("non-contiguous tensor reached a kernel")


_WS = {}


def _debug_switch(name):
    """A/B switches (ACGAN_NO_*: run the un-fused / previous path, for interleaved timings on one GPU box and for the
    bit-identity tests of the fusions).  Development aids: honoured only when ACGAN_DEBUG_SWITCHES is set."""
    return os.environ.get("ACGAN_DEBUG_SWITCHES") is not None and os.environ.get(name) is not None


def workspace(nbytes, slot=0):
    """Stream-ordered scratch (one buffer per device/slot, grown on demand)."""
    dev = torch.cuda.current_device()
    buf = _WS.get((dev, slot))
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device="cuda")
        _WS[(dev, slot)] = buf
    return buf


CONFIG_EPOCH = 0  # bumped whenever the packed-weight format changes (impl / precision switch)
_PRECISION = "bf16x3"  # the library default (acg_set_conv_precision)


def set_conv_impl(name):
    """'mfma' (product path) or 'direct' (naive HIP kernels, cross-check only)."""
    global CONFIG_EPOCH
    _lib.call("acg_set_conv_impl", {"mfma": _lib.IMPL_MFMA, "direct": _lib.IMPL_DIRECT}[name])
    CONFIG_EPOCH += 1


def set_precision(name):
    """Arithmetic of the MFMA convolution kernels: 'bf16x3' (default: each fp32 operand split into bf16 hi + lo in
    LDS, lo*hi + hi*lo + hi*hi on the bf16 matrix pipe, fp32 accumulate — operands keep 16 mantissa bits, ~4e-6 rms
    per conv, well inside the 1e-3 parity bar), 'f32' (exact-fp32 matrix pipe, the strict mode the tight tests pin the
    kernels with) or 'bf16' (operands rounded to bf16, fp32 accumulate: a numerics PROBE for tests/test_hip_bf16.py, not offered by
    bench.py / options.py — no faster than bf16x3 with fp32 tensors, DESIGN_LOG.md A.4).  HBM
    tensors stay fp32 in every mode."""
    global CONFIG_EPOCH, _PRECISION
    _lib.call("acg_set_conv_precision", {"f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "bf16x3": _lib.PREC_BF16X3}[name])
    _PRECISION = name
    CONFIG_EPOCH += 1


def get_precision():
    return _PRECISION


# ----------------------------------------------------------------------------------------------
# layout
# ----------------------------------------------------------------------------------------------
class ToNHWC(torch.autograd.Function):
    """(N,C,H,W) -> (N,H,W,Cp)"""

    @staticmethod
    def forward(ctx, x, img=False):
        """img: the tensor is an image that goes into a convolution (cimg: C4 storage for <= 4 channels)"""
        x = x.contiguous()
        _check(x)
        N, C, H, W = x.shape
        ctx.C = C
        Cp = cimg(C) if img else cpad(C)
        y = torch.empty((N, H, W, Cp), device=x.device, dtype=torch.float32)
        _lib.call("acg_nchw_to_nhwc16", _ptr(x), _ptr(y), N, C, H, W, Cp, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, Cp = g.shape
        d = torch.empty((N, ctx.C, H, W), device=g.device, dtype=torch.float32)
        _lib.call("acg_nhwc16_to_nchw", _ptr(g), _ptr(d), N, ctx.C, H, W, Cp, _stream())
        return d, None


class ToNCHW(torch.autograd.Function):
    """(N,H,W,Cp) -> (N,C,H,W)"""

    @staticmethod
    def forward(ctx, x, C):
        x = x.contiguous()
        _check(x)
        N, H, W, Cp = x.shape
        ctx.Cp = Cp
        y = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
        _lib.call("acg_nhwc16_to_nchw", _ptr(x), _ptr(y), N, C, H, W, Cp, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, C, H, W = g.shape
        d = torch.empty((N, H, W, ctx.Cp), device=g.device, dtype=torch.float32)
        _lib.call("acg_nchw_to_nhwc16", _ptr(g), _ptr(d), N, C, H, W, ctx.Cp, _stream())
        return d, None


class Concat(torch.autograd.Function):
    """torch.cat((a, b), 1) on C16 tensors (model.py:410, 472)."""

    @staticmethod
    def forward(ctx, a, b, Ca, Cb):
        a, b = a.contiguous(), b.contiguous()
        _check(a, b)
        N, H, W, Cap = a.shape
        Cbp = b.shape[3]
        Cdp = cimg(Ca + Cb)   # (an image again: the encoder's first layer takes it)
        ctx.dims = (Ca, Cap, Cb, Cbp, Cdp)
        y = torch.empty((N, H, W, Cdp), device=a.device, dtype=torch.float32)
        _lib.call("acg_concat_channels", _ptr(a), Ca, Cap, _ptr(b), Cb, Cbp, _ptr(y), Cdp, N * H * W, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        Ca, Cap, Cb, Cbp, Cdp = ctx.dims
        N, H, W, _ = g.shape
        ga = torch.empty((N, H, W, Cap), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gb = torch.empty((N, H, W, Cbp), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if ga is not None or gb is not None:
            _lib.call("acg_split_channels", _ptr(g), Cdp, _ptr(ga), Ca, Cap, _ptr(gb), Cb, Cbp, N * H * W, _stream())
        return ga, gb, None, None


# ----------------------------------------------------------------------------------------------
# convolutions
# ----------------------------------------------------------------------------------------------
class PackedConv(object):
    """Device-side packed forms of one conv weight (+ padded bias); see acg_pack_conv_weight."""

    def __init__(self, weight, bias, Ci, Co):
        Or, Ir, K, _ = weight.shape
        self.Or, self.Ir, self.K, self.Ci, self.Co = Or, Ir, K, Ci, Co   # Ci / Co: the PACKED widths (multiples of 16)
        # stored widths of the layer's input / output tensors: C4 on the thin side of a thin layer (image tensors)
        tin, tout = conv_thin_sides(Or, Ir, K)
        self.Cis, self.Cos = (4 if tin else Ci), (4 if tout else Co)
        dev = weight.device
        self.wf = torch.empty(_lib.query("acg_packed_wf_elems", K, Ci, Co), device=dev, dtype=torch.float32)
        self.wb = torch.empty(_lib.query("acg_packed_wb_elems", K, Ci, Co), device=dev, dtype=torch.float32)
        self.bias = None
        self.refresh(weight, bias)

    def refresh(self, weight, bias):
        w = weight.detach().contiguous()
        _check(w)
        _lib.call("acg_pack_conv_weight", _ptr(w), self.Or, self.Ir, self.K, self.Ci, self.Co, _ptr(self.wf), _ptr(self.wb),
                  _stream())
        self.refresh_bias(bias)

    def refresh_bias(self, bias):
        w = self.wf
        if bias is not None:
            n = bias.numel()
            npad = cpad(n)
            if n == npad and bias.is_contiguous() and bias.dtype == torch.float32:
                self.bias = bias.detach()   # no padding needed: the kernels read the parameter itself
            else:
                if self.bias is None or self.bias.data_ptr() == bias.data_ptr():
                    self.bias = torch.empty(npad, device=w.device, dtype=torch.float32)
                _lib.call("acg_pad_vector", _ptr(bias.detach()), n, _ptr(self.bias), npad, _stream())


def repack_many(entries):
    """entries: [(PackedConv, weight, bias)] whose parameters the optimiser just changed: refresh the packed copies IN PLACE —
    the regular layers of the list in one launch (acg_pack_conv_weights_multi), thin layers and the exact-fp32 mode layer by
    layer.  (In place: nothing of a step reads a network's old weights after its Adam update.)"""
    multi = []
    for pk, w, b in entries:
        if _lib.query("acg_pack_conv_weights_multi_supported", pk.Or, pk.Ir, pk.K):
            multi.append((pk, w, b))
        else:
            pk.refresh(w, b)
    if not multi:
        return
    arr = (_lib.PackItem * len(multi))()
    keep = []
    for i, (pk, w, b) in enumerate(multi):
        wd = w.detach().contiguous()
        _check(wd)
        keep.append(wd)
        arr[i].w, arr[i].wf, arr[i].wb = wd.data_ptr(), pk.wf.data_ptr(), pk.wb.data_ptr()
        arr[i].Or, arr[i].Ir, arr[i].K, arr[i].Ci, arr[i].Co = pk.Or, pk.Ir, pk.K, pk.Ci, pk.Co
    _lib.call("acg_pack_conv_weights_multi", arr, len(multi), _stream())
    _fused("packed_weights_multi", len(multi))
    for pk, w, b in multi:
        pk.refresh_bias(b)


class ConvTimer(object):
    """bench.py hook: HIP events (on the launch stream) around the launches of ONE pass ("fwd", "dgrad", "wgrad", or
    "dgrad_sums": the data-gradient launches that also emit the backward sums of the norm in front, ops.NormSums) of ONE
    conv shape, so the roofline numerator/denominator come from the live timed region."""

    def __init__(self, match, kind="fwd"):
        self.match, self.kind, self.events, self.kernel = match, kind, [], None

    class _Span(object):
        def __init__(self, timers, mid=False):
            self.timers, self.em = timers, None
            if timers:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if mid:   # weight gradient: the library records this event between its main kernel and the split-K reduction
                    self.em = torch.cuda.Event(enable_timing=True)
                    self.em.record()   # (creates the hipEvent_t; the library's record replaces this timestamp)
                    _lib.call("acg_debug_mid_event", ctypes.c_void_p(self.em.cuda_event))
                self.e0.record()

        def done(self):
            if self.timers:
                self.e1.record()
                if self.em is not None:
                    _lib.call("acg_debug_mid_event", None)   # (an entry point without a reduction must not leave it armed)
                k = _lib.query("acg_last_kernel").decode()   # what the dispatcher actually launched
                for t in self.timers:
                    t.events.append((self.e0, self.e1, k, self.em))
                    t.kernel = k

    @staticmethod
    def span(kind, d):
        """-> object whose done() closes the bracket (a no-op unless a timer of this kind matches descriptor d)"""
        return ConvTimer._Span([t for t in CONV_TIMERS if t.kind == kind and t.match(d)], mid=(kind == "wgrad"))

    def ms(self):
        return [a.elapsed_time(b) for a, b, _, _ in self.events]

    def ms_main(self):
        """the main kernel alone where the pass has a second launch behind it (weight gradient: its split-K reduction)"""
        return [self._main(a, b, m) for a, b, _, m in self.events]

    @staticmethod
    def _main(a, b, m):
        v = a.elapsed_time(m) if m is not None else -1.0
        return v if v > 0.0 else a.elapsed_time(b)   # (not recorded by the library: the place-holder stamp lies before `a`)

    def by_kernel(self, main=False):
        """-> {kernel label: [ms per launch]} (one pass of one layer can be served by several template instances)"""
        out = {}
        for a, b, k, m in self.events:
            out.setdefault(k, []).append(self._main(a, b, m) if main else a.elapsed_time(b))
        return out


CONV_TIMERS = []   # bench.py appends ConvTimer objects; every matching forward launch is bracketed by HIP events


def conv_desc(N, Hi, Wi, Ci, Co, K, stride, pad, pad_mode, Cir=0, Cor=0):
    """Cir / Cor: real (unpadded) channel counts — let the library pick the thin-channel kernels (0 = unknown)."""
    Ho = (Hi + 2 * pad - K) // stride + 1
    Wo = (Wi + 2 * pad - K) // stride + 1
    return ConvDesc(N, Hi, Wi, Ci, Ho, Wo, Co, K, stride, pad, pad_mode, Cir, Cor)


STATS_ROWS = 128
CONV_STATS_ENABLED = not _debug_switch("ACGAN_NO_CONV_STATS")  # A/B switch


class ConvStats(object):
    """Slot the caller hands to a convolution AND to the (Cond)InstanceNorm behind it (modules.run_sequence): where the
    kernel supports it the convolution's epilogue emits the norm's per-tile statistics (acg_conv2d_fwd_stats) and leaves
    them here: `part` [N][Ho*Wo/STATS_ROWS][2][Co] (mean, M2 per 128-pixel tile), else None."""

    def __init__(self):
        self.part = None


class SkipGrad(object):
    """Slot shared by a residual block's first convolution (whose identity output feeds the skip connection) and the
    block's last norm: the norm's backward returns dy itself as the gradient of the skip input and leaves the sign bitmask
    of the block output here; the convolution's data-gradient epilogue adds dy where the bit is set — the skip gradient
    dy * (y > 0) is never written to memory."""

    def __init__(self):
        self.mask = self.dy = None

    def take(self, dskip):
        """-> (skip gradient, sign bitmask or None).  When the norm left its gradient un-materialised (mask set), what autograd
        delivers must be that very tensor: anything else (a hook, a clone, a sum with another gradient) cannot be told apart
        from dy * mask any more, so it is refused instead of being added unmasked."""
        mask, dy = self.mask, self.dy
        self.mask = self.dy = None
        if mask is None:
            return dskip, None
        if dy is None or dy.data_ptr() != dskip.data_ptr() or dy.shape != dskip.shape:
            raise _lib.AcgError("the un-materialised skip gradient of a residual block was altered on its way to the block's "
                                "first convolution (tensor hook / extra consumer of the skip tensor); set ACGAN_NO_LAZY_DRES=1")
        return dskip, mask


# Parameter gradients straight into .grad: a model.FlatNet marks its parameters `_acg_direct_grad`; their .grad tensors are
# preset views of the network's flat gradient buffer, and the weight-gradient / norm-parameter kernels ADD into them
# (accumulate=1) instead of returning a fresh tensor for autograd's AccumulateGrad to add with one more kernel per parameter
# (564 five-microsecond launches per training step).  The Function then returns None for that parameter, so the
# post-accumulate hooks of the data-parallel exchange (dist.hook_params) are fired by hand.
DIRECT_GRAD = not _debug_switch("ACGAN_NO_DIRECT_GRAD")   # A/B switch


def _direct_grad(*params):
    """-> list of .grad targets when EVERY given parameter (None entries skipped) takes direct accumulation, else None"""
    if not DIRECT_GRAD:
        return None
    out = []
    for p in params:
        if p is None:
            out.append(None)
            continue
        g = getattr(p, "grad", None)
        if not getattr(p, "_acg_direct_grad", False) or g is None or not g.is_contiguous() or g.dtype != torch.float32:
            return None
        out.append(g)
    return out


def _grads_done(*params):
    for p in params:
        if p is not None:
            h = getattr(p, "_acg_grad_hook", None)
            if h is not None:
                h(p)


LAZY_DRES = not _debug_switch("ACGAN_NO_LAZY_DRES")   # A/B switch (SkipGrad)


class NormSums(object):
    """Slot shared by a (Cond)InstanceNorm and THE convolution that consumes its output (modules.run_sequence): the
    convolution's data gradient is the gradient w.r.t. the norm's output, and where its kernel supports it
    (acg_conv2d_bwd_data_s16_sums) the epilogue leaves the first pass of the norm's backward — per-tile sums of gy and
    gy * xhat — in `part`, so the norm's backward skips its own pass over dy and x.  The norm fills what the kernel needs at
    forward time; `dx` remembers the tensor the sums belong to: a gradient that reaches the norm as anything else (another
    consumer of the norm's output, a hook) makes the norm recompute them."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.x = self.mean = self.rstd = self.gp = self.bp = self.mask = None
        self.gstride = self.act = 0
        self.part = self.dx = None

    def desc(self, part):
        d = _lib.NormSumsDesc()
        d.x, d.mean, d.rstd = _ptr(self.x), _ptr(self.mean), _ptr(self.rstd)
        d.gamma, d.beta, d.gstride = _ptr(self.gp), _ptr(self.bp), self.gstride
        d.sign_mask, d.act, d.part = _ptr(self.mask), self.act, _ptr(part)
        return d


NORM_SUMS = not _debug_switch("ACGAN_NO_NORM_SUMS")   # A/B switch
NORM_SUMS_USED = 0   # norm backward passes that took their sums from a data-gradient epilogue (tests read it)
# Which fused paths the step actually took (bench.py prints them per step as `fused_paths`, so a fusion that a torch-side
# change — a hook, a clone, a shape — silently turned off shows in the bench line): launches per key since the last clear()
FUSED = {}


def _fused(key, n=1):
    FUSED[key] = FUSED.get(key, 0) + n


class ReluLink(object):
    """Hand-off between the two convolutions of a pad-conv-ReLU-pad-conv chain (ResnetBlock, modules.py:211-227): the
    SECOND convolution's data-gradient epilogue can mask its result with the sign of its own input (= the first one's ReLU
    output), which makes it the gradient w.r.t. the first convolution's pre-activation; it then sets `done`, and the first
    convolution's backward skips its activation-backward pass (3 tensor streams).  Valid only where the ReLU output has no
    other consumer."""

    def __init__(self):
        self.done = False
        self.mask = None   # sign bitmask of the ReLU output where the producing convolution stored one (pre-split trunk)


RELU_LINK = not _debug_switch("ACGAN_NO_RELU_LINK")   # A/B switch
RELU_MASK = not _debug_switch("ACGAN_NO_RELU_MASK")   # A/B switch: the link carries a sign bitmask instead of the activation


# Pre-split ("S16") activation storage of the residual trunk (include/acgan_hip.h, acg_s16_encode): an S16 tensor is carried
# as an fp32-typed torch tensor of the same shape whose BYTES are (bf16 hi, bf16 lo) groups, tagged `_acg_s16`; only the
# Functions below read or write it, and every one of them knows from its forward-time plan which of its tensors are S16
# (autograd only ever hands such a gradient from the one Function that wrote it to the one that reads it).
S16_ENABLED = not _debug_switch("ACGAN_NO_S16")   # A/B switch


def is_s16(t):
    return getattr(t, "_acg_s16", False)


def tag_s16(t):
    t._acg_s16 = True
    return t


class S16Plan(object):
    """which tensors of one convolution are pre-split: x (input), y (output; then dy arrives pre-split and already masked
    by the consumer's data gradient), gy (dy arrives pre-split from the norm behind the convolution), dx (what the data
    gradient writes: pre-split with the ReLU mask of x, or fp32)"""

    def __init__(self, x=False, y=False, gy=False, dx=False):
        self.x, self.y, self.gy, self.dx = x, y, gy, dx


def conv_s16_supported(N, Hi, Wi, C, K, pad, pad_mode):
    """can the stride-1 C -> C convolution take pre-split operands in forward, data gradient and weight gradient?"""
    if not S16_ENABLED or _PRECISION != "bf16x3" or not (LAZY_DRES and RELU_LINK and NORM_SIGN_MASK and CONV_STATS_ENABLED):
        return False
    d = conv_desc(N, Hi, Wi, C, C, K, 1, pad, pad_mode, C, C)
    return bool(_lib.query("acg_conv2d_s16_supported", ctypes.byref(d)))


class S16Decode(torch.autograd.Function):
    """pre-split -> fp32 where the residual trunk hands over to a layer that reads fp32 (its gradient is fp32 already)"""

    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _lib.call("acg_s16_decode", _ptr(x), _ptr(y), x.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g


class S16Encode(torch.autograd.Function):
    """fp32 -> pre-split (a trunk entered from a tensor that was not written pre-split)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.call("acg_s16_encode", _ptr(x), _ptr(y), x.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (+ preceding ReflectionPad2d) + bias + fused activation.  want_stats (a ConvStats slot or None): the
    caller runs an (Cond)InstanceNorm on the output next — where the kernel supports it the epilogue emits that norm's
    per-tile partial statistics, saving the norm one read of the tensor.  want_identity: also return x itself as a second
    output; a ResnetBlock feeds that alias to its skip connection, so the skip gradient arrives HERE and is added
    inside the data-gradient epilogue (acg_conv2d_bwd_data_add) instead of by a separate autograd accumulation;
    skip_grad (a SkipGrad slot or None) lets that gradient arrive un-materialised.  link_out / link_in: ReluLink."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, stride, pad, pad_mode, act, want_stats=None, want_identity=False,
                link_out=None, link_in=None, skip_grad=None, s16=None, norm_sums=None):
        x = x.contiguous()
        _check(x)
        N, Hi, Wi, Ci = x.shape
        if Ci != packed.Cis:
            raise _lib.AcgError("conv: input has %d stored channels, the layer takes %d" % (Ci, packed.Cis))
        d = conv_desc(N, Hi, Wi, Ci, packed.Cos, packed.K, stride, pad, pad_mode, packed.Ir, packed.Or)
        if d.Ho <= 0 or d.Wo <= 0:
            raise _lib.AcgError("conv: input %dx%d too small for kernel %d" % (Hi, Wi, packed.K))
        y = torch.empty((N, d.Ho, d.Wo, packed.Cos), device=x.device, dtype=torch.float32)
        span = ConvTimer.span("fwd", d)
        if s16 is not None and s16.x:   # pre-split input (and, for a conv + ReLU inside the trunk, output)
            part = None
            if want_stats is not None and act == ACT_NONE and not s16.y:
                part = torch.empty((N, (d.Ho * d.Wo) // STATS_ROWS, 2, packed.Co), device=x.device, dtype=torch.float32)
                want_stats.part = part
            if s16.y and act == ACT_RELU and link_out is not None and RELU_MASK and packed.Co % 32 == 0 and \
                    _lib.query("acg_conv2d_bwd_data_s16_sums_supported", ctypes.byref(d)):
                # conv + ReLU feeding the next trunk convolution: its data gradient needs only the SIGN of y (same shape: d fits both)
                link_out.mask = torch.empty((y.numel() + 31) // 32, device=x.device, dtype=torch.int32)
                _fused("conv_fwd_s16_relu_bitmask")
                _lib.call("acg_conv2d_fwd_s16_mask", ctypes.byref(d), _ptr(x), _ptr(packed.wf), _ptr(packed.bias if bias is not None else None),
                          _ptr(y), _ptr(link_out.mask), _stream())
            else:
                _fused("conv_fwd_s16")
                _lib.call("acg_conv2d_fwd_s16", ctypes.byref(d), _ptr(x), _ptr(packed.wf), _ptr(packed.bias if bias is not None else None),
                          _ptr(y), act, _ptr(part), 1 if s16.y else 0, _stream())
        elif want_stats is not None and CONV_STATS_ENABLED and act == ACT_NONE and \
                _lib.query("acg_conv2d_fwd_stats_supported", ctypes.byref(d)):
            part = torch.empty((N, (d.Ho * d.Wo) // STATS_ROWS, 2, packed.Co), device=x.device, dtype=torch.float32)
            _fused("conv_fwd_tile_stats")
            _lib.call("acg_conv2d_fwd_stats", ctypes.byref(d), _ptr(x), _ptr(packed.wf),
                      _ptr(packed.bias if bias is not None else None), _ptr(y), _ptr(part), _stream())
            want_stats.part = part
        else:
            _lib.call("acg_conv2d_fwd", ctypes.byref(d), _ptr(x), _ptr(packed.wf), _ptr(packed.bias if bias is not None else None),
                      _ptr(y), act, _stream())
        span.done()
        ctx.d, ctx.packed, ctx.act, ctx.has_bias = d, packed, act, bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        ctx.link_out, ctx.link_in = (link_out if act == ACT_RELU else None), link_in
        ctx.skip_grad = skip_grad
        ctx.s16 = s16 if (s16 is not None and s16.x) else None
        ctx.norm_sums = norm_sums if NORM_SUMS else None
        ctx.save_for_backward(x, y if (act != ACT_NONE and ctx.s16 is None) else None)
        if want_identity:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, y = ctx.saved_tensors
        d, pk = ctx.d, ctx.packed
        dy = dy.contiguous()
        st = _stream()
        if ctx.s16 is not None:
            return Conv2dFn._backward_s16(ctx, x, dy, dskip)
        if ctx.link_out is not None and ctx.link_out.done:
            ctx.link_out.done = False   # the consumer's data-gradient epilogue already applied this ReLU's mask
            g = dy
        elif ctx.act != ACT_NONE:
            g = torch.empty_like(dy)
            _lib.call("acg_act_bwd", _ptr(dy), _ptr(y), _ptr(g), dy.numel(), ctx.act, st)
        else:
            g = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nb = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
            ws = workspace(nb) if nb else None
            smask = None
            span = ConvTimer.span("dgrad", d)
            if dskip is not None:
                dskip = dskip.contiguous()
                if ctx.skip_grad is not None:   # the skip gradient is dskip * sign-bitmask (NormAct lazy_dres)
                    dskip, smask = ctx.skip_grad.take(dskip)
            if dskip is not None and _lib.query("acg_conv2d_bwd_data_add_supported", ctypes.byref(d)) and \
                    (smask is None or (d.Hi * d.Wi * (d.Ci // 4)) % 8 == 0):
                _fused("dgrad_skip_addend")
                _lib.call("acg_conv2d_bwd_data_add", ctypes.byref(d), _ptr(g), _ptr(pk.wb), _ptr(dskip), _ptr(smask), _ptr(dx),
                          _ptr(ws), nb, st)
                smask = None
            elif (dskip is None and ctx.link_in is not None and RELU_LINK
                  and _lib.query("acg_conv2d_bwd_data_add_supported", ctypes.byref(d))):
                _fused("dgrad_relu_link")
                _lib.call("acg_conv2d_bwd_data_relu", ctypes.byref(d), _ptr(g), _ptr(pk.wb), _ptr(x), _ptr(dx), _ptr(ws), nb, st)
                ctx.link_in.done = True
            elif (dskip is None and ctx.norm_sums is not None and ctx.norm_sums.x is not None and ctx.norm_sums.mask is None and
                  tuple(ctx.norm_sums.x.shape) == tuple(dx.shape) and _lib.query("acg_conv2d_bwd_data_sums_supported", ctypes.byref(d))):
                # dx is the gradient w.r.t. the output of the norm in front: the data-gradient kernel (row pipeline, four-phase
                # stride-2 tile, generic tile, thin-row kernel of the head) leaves that norm's backward sums
                ns = ctx.norm_sums
                part = torch.empty((d.N, (d.Hi * d.Wi) // STATS_ROWS, 2, d.Ci), device=dx.device, dtype=torch.float32)
                desc = ns.desc(part)
                _fused("dgrad_f32_norm_sums")
                _lib.call("acg_conv2d_bwd_data_sums", ctypes.byref(d), _ptr(g), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb, ctypes.byref(desc), st)
                ns.part, ns.dx = part, dx
            else:
                _lib.call("acg_conv2d_bwd_data", ctypes.byref(d), _ptr(g), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb, st)
                if dskip is not None:
                    if smask is not None:   # not fusable here: materialise the masked skip gradient
                        m = torch.empty_like(dskip)
                        _lib.call("acg_mask_apply", _ptr(dskip), _ptr(smask), _ptr(m), dskip.numel(), st)
                        dskip = m
                    dx = dx + dskip
            span.done()
        if ctx.needs_input_grad[1]:
            direct = _direct_grad(ctx.wparam, ctx.bparam)
            if direct is not None:
                dw, db = direct
            else:
                dw = torch.empty((pk.Or, pk.Ir, pk.K, pk.K), device=x.device, dtype=torch.float32)
                db = torch.empty(pk.Or, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            nb = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
            ws = workspace(nb)
            span = ConvTimer.span("wgrad", d)
            _lib.call("acg_conv2d_bwd_weight", ctypes.byref(d), _ptr(x), _ptr(g), _ptr(dw), _ptr(db), pk.Or, pk.Ir, _ptr(ws),
                      nb, 1 if direct is not None else 0, st)
            span.done()
            if direct is not None:
                dw = db = None
                _grads_done(ctx.wparam, ctx.bparam)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None


def _conv_backward_s16(ctx, x, dy, dskip):
    """Conv2dFn.backward on pre-split operands: x and dy are S16 (dy comes from the norm behind this convolution, or — for
    a conv + ReLU — from the next convolution's data gradient, which already applied the ReLU mask)."""
    d, pk, p = ctx.d, ctx.packed, ctx.s16
    st = _stream()
    if p.y:
        if ctx.link_out is None or not ctx.link_out.done:
            raise _lib.AcgError("pre-split trunk: the gradient of a conv + ReLU output must come from the next convolution's "
                                "fused data gradient")
        ctx.link_out.done = False
    dx = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty_like(x)
        nb = _lib.query("acg_conv2d_bwd_data_workspace_bytes", ctypes.byref(d))
        ws = workspace(nb) if nb else None
        ns = ctx.norm_sums if not p.dx else None
        if ns is not None and not (ns.x is not None and tuple(ns.x.shape) == tuple(dx.shape) and
                                   _lib.query("acg_conv2d_bwd_data_s16_sums_supported", ctypes.byref(d))):
            ns = None
        span = ConvTimer.span("dgrad_sums" if ns is not None else "dgrad", d)
        if p.dx:     # the gradient w.r.t. the pre-activation of the conv + ReLU in front, pre-split for its own backward
            if dskip is not None or ctx.link_in is None:
                raise _lib.AcgError("pre-split trunk: unexpected skip gradient / missing ReLU link")
            if ctx.link_in.mask is not None and ctx.link_in.mask.numel() == (x.numel() + 31) // 32 and \
                    _lib.query("acg_conv2d_bwd_data_s16_sums_supported", ctypes.byref(d)):
                _fused("dgrad_s16_relu_bitmask")
                _lib.call("acg_conv2d_bwd_data_s16_mask", ctypes.byref(d), _ptr(dy), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb,
                          _ptr(ctx.link_in.mask), st)
            else:
                _fused("dgrad_s16_relu_src")
                _lib.call("acg_conv2d_bwd_data_s16", ctypes.byref(d), _ptr(dy), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb, None, None,
                          _ptr(x), 1, st)
            ctx.link_in.done = True
        else:
            smask = None
            if dskip is not None:
                dskip = dskip.contiguous()
                if ctx.skip_grad is not None:
                    dskip, smask = ctx.skip_grad.take(dskip)
            if ns is not None:
                # dx is the gradient w.r.t. the output of the norm in front: its backward sums leave with the tiles
                part = torch.empty((d.N, (d.Hi * d.Wi) // STATS_ROWS, 2, d.Ci), device=dx.device, dtype=torch.float32)
                desc = ns.desc(part)
                _fused("dgrad_s16_norm_sums")
                if smask is not None:
                    _fused("dgrad_s16_lazy_skip")
                _lib.call("acg_conv2d_bwd_data_s16_sums", ctypes.byref(d), _ptr(dy), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb, _ptr(dskip),
                          _ptr(smask), ctypes.byref(desc), st)
                ns.part, ns.dx = part, dx
            else:
                _fused("dgrad_s16_plain")
                if smask is not None:
                    _fused("dgrad_s16_lazy_skip")
                _lib.call("acg_conv2d_bwd_data_s16", ctypes.byref(d), _ptr(dy), _ptr(pk.wb), _ptr(dx), _ptr(ws), nb, _ptr(dskip),
                          _ptr(smask), None, 0, st)
        span.done()
    if ctx.needs_input_grad[1]:
        direct = _direct_grad(ctx.wparam, ctx.bparam)
        if direct is not None:
            dw, db = direct
        else:
            dw = torch.empty((pk.Or, pk.Ir, pk.K, pk.K), device=x.device, dtype=torch.float32)
            db = torch.empty(pk.Or, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        nb = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
        ws = workspace(nb)
        span = ConvTimer.span("wgrad", d)
        _fused("wgrad_s16")
        _lib.call("acg_conv2d_bwd_weight_s16", ctypes.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), pk.Or, pk.Ir, _ptr(ws),
                  nb, 1 if direct is not None else 0, st)
        span.done()
        if direct is not None:
            dw = db = None
            _grads_done(ctx.wparam, ctx.bparam)
    else:
        dw = db = None
    return (dx, dw, db) + (None,) * 12


Conv2dFn._backward_s16 = staticmethod(_conv_backward_s16)


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(k3, s2, p1, op1) + bias + fused activation.  `packed` holds the weight
    (Cin_T, Cout_T, k, k) packed as the OIHW weight of the Conv2d it is the adjoint of."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, stride, pad, out_pad, act, want_stats=None):
        x = x.contiguous()
        _check(x)
        N, H, W, Cs = x.shape
        K = packed.K
        Hl = (H - 1) * stride - 2 * pad + K + out_pad
        Wl = (W - 1) * stride - 2 * pad + K + out_pad
        # underlying conv: large side (Hl, Wl, packed.Ci) -> small side (H, W, packed.Co)
        if Cs != packed.Co:
            raise _lib.AcgError("conv_transpose: input has %d stored channels, expected %d" % (Cs, packed.Co))
        d = conv_desc(N, Hl, Wl, packed.Ci, packed.Co, K, stride, pad, PAD_ZERO)
        if (d.Ho, d.Wo) != (H, W):
            raise _lib.AcgError("conv_transpose: inconsistent geometry")
        y = torch.empty((N, Hl, Wl, packed.Ci), device=x.device, dtype=torch.float32)
        if want_stats is not None and CONV_STATS_ENABLED and act == ACT_NONE and \
                _lib.query("acg_conv_transpose2d_fwd_stats_supported", ctypes.byref(d)):
            part = torch.empty((N, (Hl * Wl) // STATS_ROWS, 2, packed.Ci), device=x.device, dtype=torch.float32)
            _lib.call("acg_conv_transpose2d_fwd_stats", ctypes.byref(d), _ptr(x), _ptr(packed.wb),
                      _ptr(packed.bias if bias is not None else None), _ptr(y), _ptr(part), _stream())
            want_stats.part = part
        else:
            _lib.call("acg_conv_transpose2d_fwd", ctypes.byref(d), _ptr(x), _ptr(packed.wb),
                      _ptr(packed.bias if bias is not None else None), _ptr(y), act, _stream())
        ctx.d, ctx.packed, ctx.act, ctx.has_bias = d, packed, act, bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(x, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        d, pk = ctx.d, ctx.packed
        dy = dy.contiguous()
        st = _stream()
        if ctx.act != ACT_NONE:
            g = torch.empty_like(dy)
            _lib.call("acg_act_bwd", _ptr(dy), _ptr(y), _ptr(g), dy.numel(), ctx.act, st)
        else:
            g = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("acg_conv_transpose2d_bwd_data", ctypes.byref(d), _ptr(g), _ptr(pk.wf), _ptr(dx), st)
        if ctx.needs_input_grad[1]:
            direct = _direct_grad(ctx.wparam, ctx.bparam)
            if direct is not None:
                dw, db = direct
            else:
                dw = torch.empty((pk.Or, pk.Ir, pk.K, pk.K), device=x.device, dtype=torch.float32)
                db = torch.empty(pk.Ir, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            nb = _lib.query("acg_conv2d_bwd_weight_workspace_bytes", ctypes.byref(d))
            ws = workspace(nb)
            _lib.call("acg_conv_transpose2d_bwd_weight", ctypes.byref(d), _ptr(x), _ptr(g), _ptr(dw), _ptr(db), pk.Or,
                      pk.Ir, _ptr(ws), nb, 1 if direct is not None else 0, st)
            if direct is not None:
                dw = db = None
                _grads_done(ctx.wparam, ctx.bparam)
        return dx, dw, db, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# normalisation (+ fused activation / residual)
# ----------------------------------------------------------------------------------------------
NORM_SIGN_MASK = not _debug_switch("ACGAN_NO_NORM_MASK")   # A/B switch


class NormAct(torch.autograd.Function):
    """y = act(norm(x) * gamma + beta [+ res]).

    kind 'in'  : InstanceNorm      (modules.py:64-97)   gamma/beta (C,), biased variance
    kind 'cin' : CondInstanceNorm  (modules.py:104-132) gamma/beta (N, Cp), UNBIASED variance
    kind 'bn'  : BatchNorm2d/1d in train mode           gamma/beta (C,), biased; running stats updated
    x: (N,H,W,Cp) or (N,Cp).  gamma_p / beta_p are the C16-padded device copies for 'in'/'bn'.
    """

    @staticmethod
    def forward(ctx, x, gamma, beta, res, kind, act, eps, gamma_p, beta_p, run_mean, run_var, momentum, lazy_dres=None,
                stats=None, s16_out=False, s16_dx=False, s16_res=False, sums=None):
        """sums: a NormSums slot shared with the consumer of y, or None.  lazy_dres: a SkipGrad slot (the caller guarantees that the gradient w.r.t. `res` goes only to the Conv2dFn holding
        the same slot) or None.  stats: per-tile (mean, M2) partials a convolution epilogue produced (ConvStats.part)."""
        x = x.contiguous()
        _check(x)
        C = x.shape[-1]
        N = x.shape[0]
        rows = x.numel() // C
        if kind in ("bn", "bn_eval"):
            G, P = 1, rows
        else:
            G, P = N, rows // N
        unbiased = 1 if kind == "cin" else (2 if kind == "bn_eval" else 0)
        if kind == "cin":
            # (N, Cp) rows, possibly column blocks of a wider matrix (CondBankFn): the kernels take the row stride
            if gamma.shape != (N, C) or beta.shape != (N, C):
                raise _lib.AcgError("cin: scale/shift must be (N, Cp)")
            ok = gamma.stride(1) == 1 and beta.stride(1) == 1 and gamma.stride(0) == beta.stride(0) and gamma.stride(0) >= C \
                and gamma.stride(0) % 4 == 0 and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0
            if ok:
                gp, bp, gstride = gamma, beta, gamma.stride(0)
            else:
                gp, bp, gstride = gamma.contiguous(), beta.contiguous(), C
        else:
            gp, bp, gstride = gamma_p, beta_p, 0
        _check(gp if gp.is_contiguous() else None, bp if bp.is_contiguous() else None)   # (strided scale / shift: checked above)
        st = _stream()
        mean = torch.empty(G * C, device=x.device, dtype=torch.float32)
        rstd = torch.empty(G * C, device=x.device, dtype=torch.float32)
        if kind == "bn_eval":  # running statistics (real length) -> padded mean / rstd
            _lib.call("acg_bn_eval_stats", _ptr(run_mean), _ptr(run_var), run_mean.numel(), C, eps, _ptr(mean), _ptr(rstd), st)
        else:
            part = stats if kind in ("in", "cin") else None
            if part is not None:  # the producing convolution already reduced 128-pixel tiles: merge only
                _fused("norm_stats_from_conv_epilogue")
                _lib.call("acg_norm_stats_from_partials", _ptr(part), G, P, C, STATS_ROWS, eps, unbiased, _ptr(mean),
                          _ptr(rstd), st)
            else:
                nb = _lib.query("acg_norm_workspace_bytes", G, P, C)
                ws = workspace(nb)
                _lib.call("acg_norm_stats", _ptr(x), G, P, C, eps, unbiased, _ptr(mean), _ptr(rstd), _ptr(run_mean),
                          _ptr(run_var), momentum, _ptr(ws), nb, st)
        y = torch.empty_like(x)
        if res is not None:
            res = res.contiguous()
        # without a residual the backward recomputes the activation mask from x (gamma/beta): y is neither saved nor read.
        # With one it needs sign(y): the apply pass stores it as one bit per element (1/32 of a tensor) where it can.
        need_y = act != ACT_NONE and res is not None
        mask = None
        if need_y and NORM_SIGN_MASK and act in (ACT_RELU, ACT_LRELU) and (P * (C // 4)) % 8 == 0 and any(ctx.needs_input_grad):
            mask = torch.empty((G * P * C + 31) // 32, device=x.device, dtype=torch.int32)
        # pre-split (S16) output for the convolution behind this norm; a residual that arrives tagged S16 is read that way
        fmt = (2 if s16_out else 0) | (1 if (res is not None and s16_res) else 0)
        _lib.call("acg_norm_apply", _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gp), _ptr(bp), gstride, _ptr(res), _ptr(y), _ptr(mask),
                  G, P, C, act, fmt, st)
        ctx.s16_dx = bool(s16_dx)
        ctx.sums = None
        if sums is not None and NORM_SUMS and kind in ("in", "cin") and x.dim() == 4 and act in (ACT_NONE, ACT_RELU) and \
                (act == ACT_NONE or mask is not None or not need_y) and any(ctx.needs_input_grad):
            sums.x, sums.mean, sums.rstd, sums.gp, sums.bp, sums.gstride, sums.mask, sums.act = x, mean, rstd, gp, bp, gstride, mask, act
            ctx.sums = sums
        ctx.cfg = (kind, act, G, P, C, unbiased, gstride, res is not None, gamma.shape)
        ctx.gparam, ctx.bparam = (gamma, beta) if kind != "cin" else (None, None)
        ctx.lazy_dres = lazy_dres if (lazy_dres is not None and LAZY_DRES and mask is not None) else None
        ctx.save_for_backward(x, y if (need_y and mask is None) else None, mean, rstd, gp, bp, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, gp, bp, mask = ctx.saved_tensors
        kind, act, G, P, C, unbiased, gstride, has_res, gshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.lazy_dres is None) else None
        if ctx.s16_dx and dres is not None:
            raise _lib.AcgError("norm: a pre-split dx needs the un-materialised skip gradient")
        direct = _direct_grad(ctx.gparam, ctx.bparam) if (kind != "cin" and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
        if direct is not None:      # shared affine parameters: add into their .grad (first gshape[0] = real channels)
            dgamma, dbeta = direct
        else:
            npar = G * C if gstride else gshape[0]
            dgamma = torch.empty(npar, device=x.device, dtype=torch.float32)
            dbeta = torch.empty(npar, device=x.device, dtype=torch.float32)
        nb = _lib.query("acg_norm_workspace_bytes", G, P, C)
        ws = workspace(nb)
        part = None
        if ctx.sums is not None:   # the consumer's data gradient already summed gy and gy * xhat per tile — of THIS dy?
            s = ctx.sums
            if s.part is not None and s.dx is not None and s.dx.data_ptr() == dy.data_ptr() and s.dx.shape == dy.shape:
                part = s.part
            s.clear()
        if part is not None:
            global NORM_SUMS_USED
            NORM_SUMS_USED += 1
            _fused("norm_bwd_sums_from_dgrad")
            _lib.call("acg_norm_bwd_partials", _ptr(dy), _ptr(y), _ptr(mask), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gp), _ptr(bp),
                      gstride, _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), 0 if gstride else gshape[0],
                      1 if direct is not None else 0, G, P, C, act, unbiased, 1 if ctx.s16_dx else 0, _ptr(part), part.shape[1],
                      _ptr(ws), nb, _stream())
        else:
            _lib.call("acg_norm_bwd", _ptr(dy), _ptr(y), _ptr(mask), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gp), _ptr(bp), gstride, _ptr(dx),
                      _ptr(dres), _ptr(dgamma), _ptr(dbeta), 0 if gstride else gshape[0], 1 if direct is not None else 0, G, P, C,
                      act, unbiased, 1 if ctx.s16_dx else 0, _ptr(ws), nb, _stream())
        if mask is not None:
            _fused("norm_bwd_sign_bitmask")
        if kind == "cin":
            dg, db = dgamma.view(G, C), dbeta.view(G, C)
        elif direct is not None:
            dg = db = None
            _grads_done(ctx.gparam, ctx.bparam)
        else:
            dg, db = dgamma, dbeta
        if has_res and act == ACT_NONE:
            dres = dy
        elif ctx.lazy_dres is not None:
            ctx.lazy_dres.mask, ctx.lazy_dres.dy = mask, dy
            dres = dy
        return dx, dg, db, dres, None, None, None, None, None, None, None, None, None, None, None, None, None, None


# nn.Dropout inside the residual blocks (--use_dropout, options.py:65 -> modules.py:167-168, 214-215).  DROPOUT_SOURCE: None =
# draw the keep bits on the device (torch's Philox generator: 32 Bernoulli(1/2) bits per random word); a callable
# (nhwc_shape, real_channels, p) -> int32 bit words injects the draw (parity tests feed the fixture's masks to both sides)
DROPOUT_SOURCE = None


def dropout_keep_bits(shape, C, p):
    if DROPOUT_SOURCE is not None:
        return DROPOUT_SOURCE(tuple(shape), C, p)
    if p != 0.5:
        raise NotImplementedError("Dropout(p=%g): the device draw implements the reference's p = 0.5 (one random bit per element)" % p)
    n = 1
    for d in shape:
        n *= d
    dev = torch.device("cuda", torch.cuda.current_device())
    return torch.randint(-2 ** 31, 2 ** 31, ((n + 31) // 32,), device=dev, dtype=torch.int64).to(torch.int32)


def pack_keep_bits(keep_nchw, Cp, device):
    """a boolean NCHW keep mask (the reference's layout) -> bit words over the NHWC C16 tensor (pad channels: dropped)"""
    k = torch.as_tensor(keep_nchw, device=device).to(torch.bool)
    N, C, H, W = k.shape
    full = torch.zeros((N, H, W, Cp), device=device, dtype=torch.int64)
    full[..., :C] = k.permute(0, 2, 3, 1).to(torch.int64)
    words = (full.reshape(-1, 32) << torch.arange(32, device=device, dtype=torch.int64)).sum(1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


class DropoutFn(torch.autograd.Function):
    """y = keep ? x / (1 - p) : 0 on an NHWC C16 tensor (acg_dropout_apply); backward: the same map on dy"""

    @staticmethod
    def forward(ctx, x, C, p):
        x = x.contiguous()
        _check(x)
        if x.numel() % 32:
            raise _lib.AcgError("dropout: tensor size must be a multiple of 32 elements")
        bits = dropout_keep_bits(x.shape, C, p)
        if bits.numel() * 32 != x.numel() or bits.dtype != torch.int32 or not bits.is_cuda:
            raise _lib.AcgError("dropout: keep bits must be %d int32 words on the device" % (x.numel() // 32))
        y = torch.empty_like(x)
        ctx.scale = 1.0 / (1.0 - p)
        _lib.call("acg_dropout_apply", _ptr(x), _ptr(bits), ctx.scale, _ptr(y), x.numel(), _stream())
        ctx.save_for_backward(bits)
        return y

    @staticmethod
    def backward(ctx, dy):
        (bits,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.call("acg_dropout_apply", _ptr(dy), _ptr(bits), ctx.scale, _ptr(dx), dy.numel(), _stream())
        return dx, None, None


class SyncBatchNormAct(torch.autograd.Function):
    """BatchNorm (train mode) with statistics over ALL ranks' batches: per-rank (count, mean, M2) from the stats
    kernel, one all-reduce of [C x 3] floats, Chan-style combination; backward all-reduces the two per-channel sums
    before the apply pass.  With equal shards this makes N ranks == 1 rank on the concatenated batch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, act, eps, gamma_p, beta_p, run_mean, run_var, momentum):
        import torch.distributed as td
        x = x.contiguous()
        _check(x, gamma_p, beta_p)
        C = x.shape[-1]
        P = x.numel() // C
        st = _stream()
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        rstd = torch.empty(C, device=x.device, dtype=torch.float32)
        nb = _lib.query("acg_norm_workspace_bytes", 1, P, C)
        ws = workspace(nb)
        _lib.call("acg_norm_stats", _ptr(x), 1, P, C, eps, 0, _ptr(mean), _ptr(rstd), None, None, 0.0, _ptr(ws), nb, st)
        # local biased variance back from rstd, then combine across ranks: E[x], E[x^2] weighted by the pixel counts
        var = rstd.pow(-2) - eps
        pack = torch.cat([mean * P, (var + mean * mean) * P, mean.new_full((1,), float(P))])
        if td.get_backend() == "gloo" and pack.is_cuda:
            h = pack.cpu(); td.all_reduce(h); pack = h.to(x.device)
        else:
            td.all_reduce(pack)
        # the global pixel count rides in the same all-reduce and stays on the device (no host sync): the statistics are
        # exact for unequal shards too.  The backward's 1/Ptot is a launch argument and uses the equal-shard contract of
        # the data-parallel step (only then is the mean of the rank gradients the global-batch gradient, dist.py).
        cnt = pack[2 * C:]
        Ptot = float(P) * td.get_world_size()
        gmean = pack[:C] / cnt
        gvar = (pack[C:2 * C] / cnt - gmean * gmean).clamp_(min=0.0)
        grstd = (gvar + eps).rsqrt()
        if run_mean is not None:
            nreal = run_mean.numel()
            with torch.no_grad():
                run_mean.mul_(1 - momentum).add_(gmean[:nreal], alpha=momentum)
                run_var.mul_(1 - momentum).add_(gvar[:nreal] * (cnt / (cnt - 1).clamp(min=1.0)) * momentum)
        gmean, grstd = gmean.contiguous(), grstd.contiguous()
        y = torch.empty_like(x)
        _lib.call("acg_norm_apply", _ptr(x), _ptr(gmean), _ptr(grstd), _ptr(gamma_p), _ptr(beta_p), 0, None, _ptr(y), None, 1, P, C,
                  act, 0, st)
        ctx.cfg = (act, P, int(Ptot), C, gamma.shape)
        ctx.save_for_backward(x, y if act != ACT_NONE else None, gmean, grstd, gamma_p)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as td
        x, y, mean, rstd, gp = ctx.saved_tensors
        act, P, Ptot, C, gshape = ctx.cfg
        dy = dy.contiguous()
        st = _stream()
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        nb = _lib.query("acg_norm_workspace_bytes", 1, P, C)
        ws = workspace(nb)
        _lib.call("acg_norm_bwd_sums", _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(sums), 1, P, C, act, _ptr(ws),
                  nb, st)
        local = sums.clone()  # parameter gradients stay per-rank (the gradient all-reduce averages them later)
        if td.get_backend() == "gloo" and sums.is_cuda:
            h = sums.cpu(); td.all_reduce(h); sums = h.to(x.device)
        else:
            td.all_reduce(sums)
        dx = torch.empty_like(x)
        _lib.call("acg_norm_bwd_apply", _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gp), 0, _ptr(sums), _ptr(dx),
                  None, 1, P, Ptot, C, act, 0, st)
        # the later gradient all-reduce takes the MEAN over ranks while the single-process gradient is the SUM over all
        # samples of the global-batch-mean loss; losses here are per-rank means, so local sums are already right.
        return dx, local[C:][:gshape[0]], local[:C][:gshape[0]], None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# small dense layers
# ----------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y[N, Op] = act(x[:, :I] @ w.T + b), zero in columns >= O."""

    @staticmethod
    def forward(ctx, x, w, b, act, Op):
        x = x.contiguous()
        w = w.contiguous()
        _check(x, w, b)
        N, ldx = x.shape
        O, I = w.shape[0], w.numel() // w.shape[0]
        if I > ldx:
            raise _lib.AcgError("linear: input has %d columns, weight expects %d" % (ldx, I))
        y = torch.empty((N, Op), device=x.device, dtype=torch.float32)
        _lib.call("acg_linear_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), N, I, ldx, O, Op, act, _stream())
        ctx.cfg = (N, I, ldx, O, Op, act, w.shape)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, I, ldx, O, Op, act, wshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty(wshape, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        db = torch.empty(O, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        _lib.call("acg_linear_bwd", _ptr(dy), _ptr(y), _ptr(x), _ptr(w), _ptr(dx), _ptr(dw), _ptr(db), N, I, ldx, O, Op, act,
                  _stream())
        return dx, dw, db, None, None


COND_BANK = not _debug_switch("ACGAN_NO_COND_BANK")   # A/B switch


class CondBankFn(torch.autograd.Function):
    """The scale / shift layers of ALL CondInstanceNorms of a generator (modules.py:104-132: two ReLU(1x1 conv(z)) per norm,
    19 norms in the 9-block generator) as ONE dense layer: y = ReLU(z @ Wcat.T + bcat), Wcat = the 2L weights stacked;
    output k of the tuple is the (N, C) column block k of y (row stride 2L*C — NormAct reads scale / shift through that
    stride).  38 forward and 76 backward launches plus the ~80 small additions with which autograd sums the 38 gradients
    w.r.t. z and accumulates the second pass into .grad become 3 + 4.  params = w0, b0, w1, b1, ... (w_k: (C, I, 1, 1))."""

    @staticmethod
    def forward(ctx, z, C, *params):
        z = z.contiguous()
        ws, bs = params[0::2], params[1::2]
        I = ws[0].numel() // C
        O = C * len(ws)
        N, ldx = z.shape
        if any(w.numel() != C * I or b.numel() != C for w, b in zip(ws, bs)) or I > ldx or len(ws) > _lib.MAX_SEGMENTS // 2:
            raise _lib.AcgError("cond bank: layers of different widths")
        wcat = torch.cat([w.detach().reshape(C, I) for w in ws], 0)
        bcat = torch.cat([b.detach() for b in bs], 0)
        y = torch.empty((N, O), device=z.device, dtype=torch.float32)
        _lib.call("acg_linear_fwd", _ptr(z), _ptr(wcat), _ptr(bcat), _ptr(y), N, I, ldx, O, O, ACT_RELU, _stream())
        ctx.cfg = (N, I, ldx, O, C)
        ctx.params = params
        ctx.save_for_backward(z, wcat, y)
        outs = tuple(y[:, k * C:(k + 1) * C] for k in range(len(ws)))
        return outs

    @staticmethod
    def backward(ctx, *grads):
        z, wcat, y = ctx.saved_tensors
        N, I, ldx, O, C = ctx.cfg
        params = ctx.params
        zero = None
        cols = []
        for g in grads:
            if g is None:
                if zero is None:
                    zero = torch.zeros((N, C), device=z.device, dtype=torch.float32)
                g = zero
            cols.append(g)
        dy = torch.cat(cols, 1)
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        if dz is not None and ldx > I:
            dz.zero_()
        need_p = any(ctx.needs_input_grad[2:])
        dw = torch.empty((O, I), device=z.device, dtype=torch.float32) if need_p else None
        db = torch.empty(O, device=z.device, dtype=torch.float32) if need_p else None
        _lib.call("acg_linear_bwd", _ptr(dy), _ptr(y), _ptr(z), _ptr(wcat), _ptr(dz), _ptr(dw), _ptr(db), N, I, ldx, O, O, ACT_RELU,
                  _stream())
        pg = [None] * len(params)
        if need_p:
            direct = _direct_grad(*params)
            if direct is not None:   # add the slices into the parameters' .grad, all in one launch per source buffer
                for src, tens, width in ((dw, direct[0::2], C * I), (db, direct[1::2], C)):
                    sg = _lib.Segments()
                    for k, t in enumerate(tens):
                        sg.dst[k], sg.off[k], sg.len[k] = t.data_ptr(), k * width, width
                    sg.n = len(tens)
                    _lib.call("acg_segments_accumulate", _ptr(src), ctypes.byref(sg), 1, _stream())
                _grads_done(*params)
            else:
                for k in range(len(params) // 2):
                    pg[2 * k] = dw[k * C:(k + 1) * C].view_as(params[2 * k])
                    pg[2 * k + 1] = db[k * C:(k + 1) * C]
        return (dz, None) + tuple(pg)


LATENT_MLP = not _debug_switch("ACGAN_NO_LATENT_MLP")   # A/B switch


def latent_mlp_supported(N, I, H):
    return LATENT_MLP and bool(_lib.query("acg_latent_mlp_supported", N, I, H))


class LatentMLPFn(torch.autograd.Function):
    """DiscriminatorLatent's whole dense chain (networks.py:396-433) as one launch per direction: three Linear /
    BatchNorm1d(train) / LeakyReLU(0.2) stages and the Linear head.  `params` = 4 weights, 4 biases, 3 BN weights, 3 BN
    biases (autograd inputs); `buffers` = 3 running means + 3 running variances, updated in place.  -> (N, 4), column 0
    valid (the layout LinearFn gives the head)."""

    NPARAM = 14

    @staticmethod
    def forward(ctx, z, eps, momentum, buffers, *params):
        z = z.contiguous()
        _check(z, *params)
        _check(*buffers)
        N, ldz = z.shape
        H, I = params[0].shape
        for l in range(4):
            want = ((H if l < 3 else 1), (I if l == 0 else H))
            if tuple(params[l].shape) != want or params[4 + l].numel() != want[0] or not params[l].is_contiguous():
                raise _lib.AcgError("latent mlp: layer %d has shape %s, expected %s" % (l, tuple(params[l].shape), want))
        if I > ldz:
            raise _lib.AcgError("latent mlp: input has %d columns, the first layer expects %d" % (ldz, I))
        pp = _lib.LatentMlpParams()
        for l in range(4):
            pp.w[l], pp.b[l] = params[l].data_ptr(), params[4 + l].data_ptr()
        for l in range(3):
            pp.gamma[l], pp.beta[l] = params[8 + l].data_ptr(), params[11 + l].data_ptr()
            pp.run_mean[l], pp.run_var[l] = buffers[l].data_ptr(), buffers[3 + l].data_ptr()
        a_save = torch.empty((3, N, H), device=z.device, dtype=torch.float32)
        stats = torch.empty((3, 2, H), device=z.device, dtype=torch.float32)
        out = torch.empty((N, 4), device=z.device, dtype=torch.float32)
        _lib.call("acg_latent_mlp_fwd", ctypes.byref(pp), _ptr(z), ldz, N, I, H, float(eps), float(momentum), _ptr(a_save),
                  _ptr(stats), _ptr(out), _stream())
        ctx.cfg = (N, ldz, I, H)
        ctx.params = params
        ctx.save_for_backward(z, a_save, stats, *[p.detach() for p in params])
        return out

    @staticmethod
    def backward(ctx, dout):
        z, a_save, stats = ctx.saved_tensors[:3]
        ws = ctx.saved_tensors[3:]
        N, ldz, I, H = ctx.cfg
        dout = dout.contiguous()
        pp, gg = _lib.LatentMlpParams(), _lib.LatentMlpGrads()
        for l in range(4):
            pp.w[l], pp.b[l] = ws[l].data_ptr(), ws[4 + l].data_ptr()
        for l in range(3):
            pp.gamma[l], pp.beta[l] = ws[8 + l].data_ptr(), ws[11 + l].data_ptr()
        want = [ctx.needs_input_grad[4 + i] for i in range(LatentMLPFn.NPARAM)]
        direct = _direct_grad(*ctx.params) if all(want) else None
        if direct is not None:
            grads = direct
        else:
            grads = [torch.empty_like(ws[i]) if want[i] else None for i in range(LatentMLPFn.NPARAM)]
        for l in range(4):
            gg.dw[l] = grads[l].data_ptr() if grads[l] is not None else None
            gg.db[l] = grads[4 + l].data_ptr() if grads[4 + l] is not None else None
        for l in range(3):
            gg.dgamma[l] = grads[8 + l].data_ptr() if grads[8 + l] is not None else None
            gg.dbeta[l] = grads[11 + l].data_ptr() if grads[11 + l] is not None else None
        dz = torch.zeros_like(z) if ctx.needs_input_grad[0] else None
        _lib.call("acg_latent_mlp_bwd", ctypes.byref(pp), ctypes.byref(gg), _ptr(z), ldz, N, I, H, _ptr(a_save), _ptr(stats),
                  _ptr(dout), _ptr(dz), 1 if direct is not None else 0, _stream())
        if direct is not None:
            _grads_done(*ctx.params)
            grads = [None] * LatentMLPFn.NPARAM
        return (dz, None, None, None) + tuple(grads)


class SpatialMean(torch.autograd.Function):
    """(N,H,W,Cp) -> (N,Cp): mean over H*W (identity at 1x1; LatentEncoder extension)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _check(x)
        N, H, W, Cp = x.shape
        ctx.shape = x.shape
        y = torch.empty((N, Cp), device=x.device, dtype=torch.float32)
        _lib.call("acg_spatial_mean_fwd", _ptr(x), _ptr(y), N, H * W, Cp, _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, Cp = ctx.shape
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _lib.call("acg_spatial_mean_bwd", _ptr(g), _ptr(dx), N, H * W, Cp, _stream())
        return dx


# ----------------------------------------------------------------------------------------------
# losses (device scalars; no host sync)
# ----------------------------------------------------------------------------------------------
def _red_ws():
    nb = _lib.query("acg_reduce_workspace_bytes", 0)
    return workspace(nb, slot=1), nb


class MseConst(torch.autograd.Function):
    """F.mse_loss(pred, full_like(pred, target)) over the C valid channels (model.py:65-70)."""

    @staticmethod
    def forward(ctx, p, C, target):
        p = p.contiguous()
        _check(p)
        Cp = p.shape[-1]
        npix = p.numel() // Cp
        out = torch.empty((), device=p.device, dtype=torch.float32)
        ws, nb = _red_ws()
        _lib.call("acg_mse_const_fwd", _ptr(p), npix, C, Cp, float(target), _ptr(out), _ptr(ws), nb, _stream())
        ctx.cfg = (npix, C, Cp, float(target))
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        npix, C, Cp, target = ctx.cfg
        g = g.contiguous()
        dp = torch.empty_like(p)
        _lib.call("acg_mse_const_bwd", _ptr(p), npix, C, Cp, target, _ptr(g), _ptr(dp), _stream())
        return dp, None, None


class L1(torch.autograd.Function):
    """F.l1_loss(a, b) (mean) over the C valid channels (model.py:391, 468, 486, 494)."""

    @staticmethod
    def forward(ctx, a, b, C):
        a, b = a.contiguous(), b.contiguous()
        _check(a, b)
        Cp = a.shape[-1]
        npix = a.numel() // Cp
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws, nb = _red_ws()
        _lib.call("acg_l1_fwd", _ptr(a), _ptr(b), npix, C, Cp, _ptr(out), _ptr(ws), nb, _stream())
        ctx.cfg = (npix, C, Cp)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        npix, C, Cp = ctx.cfg
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            _lib.call("acg_l1_bwd", _ptr(a), _ptr(b), npix, C, Cp, _ptr(g), _ptr(da), _ptr(db), _stream())
        return da, db, None


def mean_valid(x, C, out=None):
    """mean over the C valid channels of a C16 tensor -> device scalar (no grad)."""
    x = x.detach().contiguous()
    _check(x)
    Cp = x.shape[-1]
    if out is None:
        out = torch.empty((), device=x.device, dtype=torch.float32)
    ws, nb = _red_ws()
    _lib.call("acg_mean_fwd", _ptr(x), x.numel() // Cp, C, Cp, _ptr(out), _ptr(ws), nb, _stream())
    return out


def sumsq(flat, out):
    """out[0] = sum(flat^2) (first half of clip_grad_norm)."""
    _check(flat)
    ws, nb = _red_ws()
    _lib.call("acg_sumsq", _ptr(flat), flat.numel(), _ptr(out), _ptr(ws), nb, _stream())
    return out


def clip_adam_multi(groups, max_norm, lr, beta1, beta2, eps, step, step_dev=None):
    """clip_grad_norm + Adam for all networks of a phase in three launches; groups = [(p, g, m, v, sumsq), ...] flat fp32
    buffers of each network (sumsq: 1-element tensor that receives the gradient sum of squares).  step_dev: int32 device
    tensor holding the number of COMPLETED steps (graph capture: the kernel reads it instead of `step`)."""
    n = len(groups)
    arr = (_lib.AdamGroup * n)()
    for i, (p, g, m, v, ss) in enumerate(groups):
        _check(p, g, m, v, ss)
        if not (p.numel() == g.numel() == m.numel() == v.numel()):
            raise _lib.AcgError("clip_adam_multi: group %d buffers differ in size" % i)
        arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n, arr[i].sumsq = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                                           p.numel(), ss.data_ptr())
    nb = _lib.query("acg_clip_adam_multi_workspace_bytes", n)
    ws = workspace(nb, slot=1)
    _lib.call("acg_clip_adam_multi", arr, n, float(max_norm), float(lr), float(beta1), float(beta2), float(eps), int(step),
              _ptr(step_dev), _ptr(ws), nb, _stream())


def adam_step(p, g, m, v, sumsq_t, max_norm, lr, beta1, beta2, eps, step, scale_grads=True):
    """clip (coefficient from the device-side sum of squares) + Adam on one flat buffer."""
    _check(p, g, m, v)
    _lib.call("acg_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(sumsq_t), float(max_norm), float(lr),
              float(beta1), float(beta2), float(eps), int(step), 1 if scale_grads else 0, _stream())
