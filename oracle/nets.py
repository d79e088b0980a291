"""ORACLE (test infrastructure): the six networks of the reference, restated on
the NumPy tape.  Parameter names are the reference's state_dict keys (unique
tensors only — the aliased duplicates created by CINResnetBlock.__init__,
modules.py:145-146, are not separate tensors).

`n_blocks` is a real parameter here; the reference hard-codes 3
(networks.py:173, 225) — default 3 is reference-faithful.
"""
from collections import OrderedDict

import numpy as np

from . import ops
from .tape import leaf


class Net(object):
    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.params = OrderedDict()   # key -> T leaf (trainable)
        self.buffers = OrderedDict()  # key -> ndarray (BatchNorm running stats)
        self.shapes = OrderedDict()
        self.training = True

    def _add(self, key, shape):
        self.shapes[key] = tuple(shape)
        self.params[key] = leaf(np.zeros(shape, self.dtype), name=key)
        return key

    def _add_bn(self, prefix, c):
        self._add(prefix + ".weight", (c,))
        self._add(prefix + ".bias", (c,))
        self.buffers[prefix + ".running_mean"] = np.zeros(c, self.dtype)
        self.buffers[prefix + ".running_var"] = np.ones(c, self.dtype)
        self.buffers[prefix + ".num_batches_tracked"] = np.zeros((), np.int64)

    def load(self, values):
        """values: dict key -> ndarray (params and optionally buffers)."""
        for k, p in self.params.items():
            p.v = np.array(values[k], self.dtype).reshape(self.shapes[k])
        for k in self.buffers:
            if k in values:
                self.buffers[k] = np.array(values[k], self.buffers[k].dtype)

    def state(self):
        out = OrderedDict((k, p.v) for k, p in self.params.items())
        out.update(self.buffers)
        return out

    def parameters(self):
        return list(self.params.values())

    def P(self, key):
        return self.params[key]

    # `--use_dropout` (options.py:65): the Dropout(0.5) behind the first ReLU of every residual block.  The keep masks are
    # GIVEN (self.drop = callable(shape) -> NCHW keep mask, one call per Dropout forward, in call order); eval mode: identity
    drop = None

    def _drop(self, x):
        if not getattr(self, "use_dropout", False) or not self.training:
            return x
        if self.drop is None:
            raise RuntimeError("oracle: use_dropout needs an injected mask source (net.drop)")
        return ops.dropout(x, self.drop(x.v.shape))

    def _bn(self, x, prefix):
        stats = {k: self.buffers[prefix + "." + k] for k in ("running_mean", "running_var", "num_batches_tracked")}
        y = ops.batch_norm(x, self.P(prefix + ".weight"), self.P(prefix + ".bias"), stats, self.training)
        for k, v in stats.items():
            self.buffers[prefix + "." + k] = v
        return y


# ------------------------------------------------------------------ generators
class ResnetGenerator(Net):
    """networks.py:203-252 (+ modules.py:193-235 ResnetBlock)."""

    def __init__(self, input_nc, output_nc, ngf, n_blocks=3, dtype=np.float32, norm="instance", use_dropout=False):
        Net.__init__(self, dtype)
        self.nb = n_blocks
        self.norm, self.use_dropout = norm, use_dropout   # networks.py:23-31 get_norm_layer; modules.py:214-215
        self.j = 1 if use_dropout else 0
        self._conv("model.1", ngf, input_nc, 7); self._in("model.2", ngf)
        self._conv("model.4", 2 * ngf, ngf, 3); self._in("model.5", 2 * ngf)
        self._conv("model.7", 4 * ngf, 2 * ngf, 3); self._in("model.8", 4 * ngf)
        for i in range(n_blocks):
            b = "model.%d.conv_block" % (10 + i)
            # the Dropout module sits at index 3 and shifts the second stage's state_dict keys (modules.py:214-228)
            self._conv(b + ".1", 4 * ngf, 4 * ngf, 3)
            self._conv(b + ".%d" % (4 + self.j), 4 * ngf, 4 * ngf, 3)
            self._in(b + ".%d" % (5 + self.j), 4 * ngf)
        t = 10 + n_blocks
        self.t = t
        self._add("model.%d.weight" % t, (4 * ngf, 2 * ngf, 3, 3)); self._add("model.%d.bias" % t, (2 * ngf,))
        self._in("model.%d" % (t + 1), 2 * ngf)
        self._conv("model.%d" % (t + 3), ngf, 2 * ngf, 3); self._in("model.%d" % (t + 4), ngf)
        self._conv("model.%d" % (t + 6), output_nc, ngf, 7)

    def _conv(self, p, co, ci, k):
        self._add(p + ".weight", (co, ci, k, k)); self._add(p + ".bias", (co,))

    def _in(self, p, c):
        if self.norm == "batch":
            return self._add_bn(p, c)
        self._add(p + ".scale", (c,)); self._add(p + ".shift", (c,))

    def _c(self, x, p, **kw):
        return ops.conv2d(x, self.P(p + ".weight"), self.P(p + ".bias"), **kw)

    def _n(self, x, p):
        if self.norm == "batch":
            return self._bn(x, p)
        return ops.instance_norm(x, self.P(p + ".scale"), self.P(p + ".shift"))

    def forward(self, x):
        t = self.t
        h = ops.relu(self._n(self._c(x, "model.1", pad=3, pad_mode="reflect"), "model.2"))
        h = ops.relu(self._n(self._c(h, "model.4", pad=1), "model.5"))
        h = ops.relu(self._n(self._c(h, "model.7", pad=1, stride=2), "model.8"))
        for i in range(self.nb):
            b = "model.%d.conv_block" % (10 + i)
            o = ops.relu(self._c(h, b + ".1", pad=1, pad_mode="reflect"))       # modules.py:211-212: no norm
            o = self._drop(o)                                                    # modules.py:214-215
            o = self._n(self._c(o, b + ".%d" % (4 + self.j), pad=1, pad_mode="reflect"), b + ".%d" % (5 + self.j))
            h = ops.relu(ops.add(h, o))                                          # modules.py:232-235
        h = ops.conv_transpose2d(h, self.P("model.%d.weight" % t), self.P("model.%d.bias" % t))
        h = ops.relu(self._n(h, "model.%d" % (t + 1)))
        h = ops.relu(self._n(self._c(h, "model.%d" % (t + 3), pad=1), "model.%d" % (t + 4)))
        return ops.tanh(self._c(h, "model.%d" % (t + 6), pad=3))                 # zero pad 3: networks.py:242


class CINResnetGenerator(Net):
    """networks.py:149-197 (+ modules.py:104-188 CondInstanceNorm / CINResnetBlock)."""

    def __init__(self, nlatent, input_nc, output_nc, ngf, n_blocks=3, dtype=np.float32, use_dropout=False):
        Net.__init__(self, dtype)
        self.nb, self.nl = n_blocks, nlatent
        self.use_dropout = use_dropout
        self.j = 1 if use_dropout else 0   # Dropout at conv_block index 3 shifts the later keys (modules.py:167-181)
        self._conv("model.1", ngf, input_nc, 7); self._cin("model.2", ngf)
        self._conv("model.4", 2 * ngf, ngf, 3); self._cin("model.5", 2 * ngf)
        self._conv("model.7", 4 * ngf, 2 * ngf, 3); self._cin("model.8", 4 * ngf)
        for i in range(n_blocks):
            b = "model.%d.conv_block" % (10 + i)
            self._conv(b + ".1.module1", 4 * ngf, 4 * ngf, 3)
            self._cin(b + ".1.module2", 4 * ngf)
            self._conv(b + ".%d" % (4 + self.j), 4 * ngf, 4 * ngf, 3)
            self._add(b + ".%d.scale" % (5 + self.j), (4 * ngf,)); self._add(b + ".%d.shift" % (5 + self.j), (4 * ngf,))
        t = 10 + n_blocks
        self.t = t
        self._add("model.%d.weight" % t, (4 * ngf, 2 * ngf, 3, 3)); self._add("model.%d.bias" % t, (2 * ngf,))
        self._cin("model.%d" % (t + 1), 2 * ngf)
        self._conv("model.%d" % (t + 3), ngf, 2 * ngf, 3); self._cin("model.%d" % (t + 4), ngf)
        self._conv("model.%d" % (t + 6), output_nc, ngf, 7)

    def _conv(self, p, co, ci, k):
        self._add(p + ".weight", (co, ci, k, k)); self._add(p + ".bias", (co,))

    def _cin(self, p, c):
        for br in ("shift_conv", "scale_conv"):
            self._add("%s.%s.0.weight" % (p, br), (c, self.nl, 1, 1))
            self._add("%s.%s.0.bias" % (p, br), (c,))

    def _c(self, x, p, **kw):
        return ops.conv2d(x, self.P(p + ".weight"), self.P(p + ".bias"), **kw)

    def _n(self, x, z, p):
        # modules.py:123-124: shift/scale = ReLU(1x1 conv(noise)); noise is (N, nl, 1, 1)
        sh = ops.relu(ops.conv2d(z, self.P(p + ".shift_conv.0.weight"), self.P(p + ".shift_conv.0.bias")))
        sc = ops.relu(ops.conv2d(z, self.P(p + ".scale_conv.0.weight"), self.P(p + ".scale_conv.0.bias")))
        return ops.cond_instance_norm(x, sc, sh)

    def forward(self, x, z):
        t = self.t
        h = ops.relu(self._n(self._c(x, "model.1", pad=3, pad_mode="reflect"), z, "model.2"))
        h = ops.relu(self._n(self._c(h, "model.4", pad=1), z, "model.5"))
        h = ops.relu(self._n(self._c(h, "model.7", pad=1, stride=2), z, "model.8"))
        for i in range(self.nb):
            b = "model.%d.conv_block" % (10 + i)
            o = ops.relu(self._n(self._c(h, b + ".1.module1", pad=1, pad_mode="reflect"), z, b + ".1.module2"))
            o = self._drop(o)                                                    # modules.py:167-168
            o = self._c(o, b + ".%d" % (4 + self.j), pad=1, pad_mode="reflect")
            o = ops.instance_norm(o, self.P(b + ".%d.scale" % (5 + self.j)), self.P(b + ".%d.shift" % (5 + self.j)))   # modules.py:180-181
            h = ops.relu(ops.add(h, o))
        h = ops.conv_transpose2d(h, self.P("model.%d.weight" % t), self.P("model.%d.bias" % t))
        h = ops.relu(self._n(h, z, "model.%d" % (t + 1)))
        h = ops.relu(self._n(self._c(h, "model.%d" % (t + 3), pad=1), z, "model.%d" % (t + 4)))
        return ops.tanh(self._c(h, "model.%d" % (t + 6), pad=3))


# -------------------------------------------------------------- discriminators
class _ConvD(Net):
    norm = "instance"   # `--norm batch` (options.py:64, networks.py:23-31): BatchNorm2d in place of InstanceNorm

    def _conv(self, p, co, ci, k):
        self._add(p + ".weight", (co, ci, k, k)); self._add(p + ".bias", (co,))

    def _in(self, p, c):
        if self.norm == "batch":
            return self._add_bn(p, c)
        self._add(p + ".scale", (c,)); self._add(p + ".shift", (c,))

    def _c(self, x, p, **kw):
        return ops.conv2d(x, self.P(p + ".weight"), self.P(p + ".bias"), **kw)

    def _n(self, x, p):
        if self.norm == "batch":
            return self._bn(x, p)
        return ops.instance_norm(x, self.P(p + ".scale"), self.P(p + ".shift"))


class Discriminator(_ConvD):
    """D_B — networks.py:308-349: k4, strides 2,2,1,1,1, pad 1."""

    def __init__(self, input_nc, ndf, dtype=np.float32, norm="instance"):
        Net.__init__(self, dtype)
        self.norm = norm
        self._conv("model.0", ndf, input_nc, 4)
        self._conv("model.2", 2 * ndf, ndf, 4); self._in("model.3", 2 * ndf)
        self._conv("model.5", 4 * ndf, 2 * ndf, 4); self._in("model.6", 4 * ndf)
        self._conv("model.8", 4 * ndf, 4 * ndf, 4); self._in("model.9", 4 * ndf)
        self._conv("model.11", 1, 4 * ndf, 4)

    def forward(self, x):
        h = ops.leaky_relu(self._c(x, "model.0", stride=2, pad=1))
        h = ops.leaky_relu(self._n(self._c(h, "model.2", stride=2, pad=1), "model.3"))
        h = ops.leaky_relu(self._n(self._c(h, "model.5", stride=1, pad=1), "model.6"))
        h = ops.leaky_relu(self._n(self._c(h, "model.8", stride=1, pad=1), "model.9"))
        return self._c(h, "model.11", stride=1, pad=1)


class Discriminator_edges(_ConvD):
    """D_A — networks.py:352-393: four k3 s2 p1 convs then a k4 p0 head."""

    def __init__(self, input_nc, ndf, dtype=np.float32, norm="instance"):
        Net.__init__(self, dtype)
        self.norm = norm
        self._conv("model.0", ndf, input_nc, 3)
        self._conv("model.2", 2 * ndf, ndf, 3); self._in("model.3", 2 * ndf)
        self._conv("model.5", 4 * ndf, 2 * ndf, 3); self._in("model.6", 4 * ndf)
        self._conv("model.8", 4 * ndf, 4 * ndf, 3); self._in("model.9", 4 * ndf)
        self._conv("model.11", 1, 4 * ndf, 4)

    def forward(self, x):
        h = ops.leaky_relu(self._c(x, "model.0", stride=2, pad=1))
        h = ops.leaky_relu(self._n(self._c(h, "model.2", stride=2, pad=1), "model.3"))
        h = ops.leaky_relu(self._n(self._c(h, "model.5", stride=2, pad=1), "model.6"))
        h = ops.leaky_relu(self._n(self._c(h, "model.8", stride=2, pad=1), "model.9"))
        return self._c(h, "model.11", stride=1, pad=0)


class DiscriminatorLatent(Net):
    """D_z_B — networks.py:396-433: Linear/BatchNorm1d/LeakyReLU x3 + Linear."""

    def __init__(self, nlatent, ndf, dtype=np.float32):
        Net.__init__(self, dtype)
        self.nl = nlatent
        dims = [(0, nlatent, ndf), (3, ndf, ndf), (6, ndf, ndf), (9, ndf, 1)]
        for idx, i, o in dims:
            self._add("model.%d.weight" % idx, (o, i)); self._add("model.%d.bias" % idx, (o,))
            if idx != 9:
                self._add_bn("model.%d" % (idx + 1), o)

    def forward(self, z):
        h = ops.reshape(z, (z.v.shape[0], self.nl)) if z.v.ndim == 4 else z    # networks.py:427-428
        for idx in (0, 3, 6):
            h = ops.linear(h, self.P("model.%d.weight" % idx), self.P("model.%d.bias" % idx))
            h = ops.leaky_relu(self._bn(h, "model.%d" % (idx + 1)))
        return ops.linear(h, self.P("model.9.weight"), self.P("model.9.bias"))


# --------------------------------------------------------------------- encoder
class LatentEncoder(Net):
    """E_B — networks.py:438-482 (norm='batch', model.py:363-364)."""

    def __init__(self, nlatent, input_nc, nef, dtype=np.float32):
        Net.__init__(self, dtype)
        self._add("conv_modules.0.weight", (nef, input_nc, 3, 3)); self._add("conv_modules.0.bias", (nef,))
        chans = [(2, nef, 2 * nef, 3), (5, 2 * nef, 4 * nef, 3), (8, 4 * nef, 8 * nef, 3), (11, 8 * nef, 8 * nef, 4)]
        for idx, ci, co, k in chans:
            self._add("conv_modules.%d.weight" % idx, (co, ci, k, k))          # bias=False: networks.py:442
            self._add_bn("conv_modules.%d" % (idx + 1), co)
        for head in ("enc_mu", "enc_logvar"):
            self._add(head + ".weight", (nlatent, 8 * nef, 1, 1)); self._add(head + ".bias", (nlatent,))

    def forward(self, x):
        h = ops.relu(ops.conv2d(x, self.P("conv_modules.0.weight"), self.P("conv_modules.0.bias"), stride=2, pad=1))
        for idx in (2, 5, 8):
            h = ops.conv2d(h, self.P("conv_modules.%d.weight" % idx), None, stride=2, pad=1)
            h = ops.relu(self._bn(h, "conv_modules.%d" % (idx + 1)))
        h = ops.conv2d(h, self.P("conv_modules.11.weight"), None, stride=1, pad=0)
        h = ops.relu(self._bn(h, "conv_modules.12"))
        mu = ops.conv2d(h, self.P("enc_mu.weight"), self.P("enc_mu.bias"))
        lv = ops.conv2d(h, self.P("enc_logvar.weight"), self.P("enc_logvar.bias"))
        # networks.py:482 flattens; defined by the reference only for a 1x1 map (S=64).
        # For larger maps the build's extension is the spatial mean (identity at 1x1).
        return ops.spatial_mean(mu), ops.spatial_mean(lv)
