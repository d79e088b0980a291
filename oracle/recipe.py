"""ORACLE (test infrastructure): deterministic parameter / input recipes.

Parameters are regenerated from (seed, network name, state_dict key) on every
box, so fixtures never carry the 22 MB of weights.  Each key
gets its own RandomState seeded by crc32 — independent of iteration order and
of torch's RNG streams (and of the reference's double-init quirk, §3.3).

Two flavours:
  "init" — the reference's own distributions (networks.py:13-21 weights_init,
           modules.py:78-81 InstanceNorm.reset_parameters; Linear/BN1d get a
           small normal instead of torch's default uniform — distributions, not
           streams, are what matters for the step).
  "rich" — O(1)-scale values everywhere so that every term of every adjoint is
           numerically exercised (IN scale ~ N(1,0.3) instead of ~0, non-zero
           biases/shifts).
"""
import zlib

import numpy as np


def _rs(seed, net, key):
    return np.random.RandomState((zlib.crc32(("%s/%s" % (net, key)).encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def param(seed, net, key, shape, flavour="rich"):
    rs = _rs(seed, net, key)
    leafname = key.split(".")[-1]
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    if flavour == "init":
        if leafname == "weight" and len(shape) == 4:
            v = rs.normal(0.0, 0.02, shape)
        elif leafname == "weight" and len(shape) == 2:
            v = rs.normal(0.0, 1.0 / np.sqrt(fan_in), shape)
        elif leafname == "weight":  # BatchNorm affine weight
            v = rs.normal(1.0, 0.02, shape)
        elif leafname == "scale":
            v = rs.normal(0.0, 0.02, shape)
        else:  # bias / shift
            v = np.zeros(shape)
    else:
        if leafname == "weight" and len(shape) >= 2:
            v = rs.normal(0.0, 1.0 / np.sqrt(fan_in), shape)
        elif leafname in ("weight", "scale"):
            v = rs.normal(1.0, 0.3, shape)
        else:
            v = rs.normal(0.0, 0.2, shape)
    return v.astype(np.float32)


def fill(net, net_name, seed=0, flavour="rich"):
    """net: oracle.nets.Net — returns {key: ndarray} and loads it."""
    vals = {k: param(seed, net_name, k, shp, flavour) for k, shp in net.shapes.items()}
    net.load(vals)
    return vals


def values_for(shapes, net_name, seed=0, flavour="rich"):
    return {k: param(seed, net_name, k, tuple(shp), flavour) for k, shp in shapes.items()}


def inputs(seed, N, nc_a, nc_b, S, nlatent):
    """Synthetic batch as in SURVEY.md §8d: U(-1,1) images, N(0,1) latent."""
    rs = np.random.RandomState(1234 + seed)
    A = rs.uniform(-1, 1, (N, nc_a, S, S)).astype(np.float32)
    B = rs.uniform(-1, 1, (N, nc_b, S, S)).astype(np.float32)
    z = rs.normal(0, 1, (N, nlatent, 1, 1)).astype(np.float32)
    return A, B, z
