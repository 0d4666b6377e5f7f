"""Deterministic, platform-independent tensor fill.

Weights never travel as fixtures: the golden generator (oracle/gen_golden.py)
applies this rule to the imported reference model, and the tests/bench apply
the very same rule to this repo's model on the GPU box.

The rule is a counter-based hash (splitmix64 finaliser) of (crc32(name), flat
index) mapped to a uniform float32 in [lo, hi).  numpy only, no RNG state.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def tag_of(name):
    """Stable 32-bit tag of a tensor name (independent of dict order)."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def uniform(shape, tag, lo=0.0, hi=1.0):
    """float32 array of `shape`; element i = lo + (hi-lo) * u(tag, i), u in [0,1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(tag) * np.uint64(0x100000001B3) + idx)
        key = _splitmix64(key ^ np.uint64(tag))
    u = (key >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fill_rule(name, shape):
    """The parameter fill rule, by state_dict key suffix.

    conv / linear weights: U(-a, a), a = g * sqrt(6 / fan_in); g = 0.5 keeps the
    19-block residual trunk's activations O(10) without normalisation (eval-BN),
    heads use g = 0.1 so that exp() in the box decode stays finite.
    """
    leaf = name.split(".")[-1]
    tag = tag_of(name)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return uniform(shape, tag, -0.1, 0.1)
    if leaf == "running_var":
        return uniform(shape, tag, 0.8, 1.2)
    if len(shape) >= 2:  # conv [O,I,kh,kw] or linear [O,I]
        fan_in = int(np.prod(shape[1:]))
        g = 0.1 if ("classconv" in name or "bbox3dconv" in name) else 0.5
        if "fusion" in name and name.endswith("fc2.weight"):
            g = 0.25
        a = g * float(np.sqrt(6.0 / fan_in))
        return uniform(shape, tag, -a, a)
    if leaf == "weight":  # BN gamma
        return uniform(shape, tag, 0.8, 1.2)
    if leaf == "bias":
        return uniform(shape, tag, -0.1, 0.1)
    return uniform(shape, tag, -0.1, 0.1)


def fill_state_dict(module):
    """In-place deterministic fill of every tensor in module.state_dict()."""
    import torch

    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            key = k[7:] if k.startswith("module.") else k
            arr = fill_rule(key, tuple(v.shape))
            v.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(v.dtype).reshape(v.shape))
    return module


def synthetic_points(n, lim, seed):
    """Synthetic LiDAR points [n,3] float32 (SURVEY.md §8(d) generator, hash based).

    x~U(xmin-5, xmax+5), y~U(ymin-5, ymax+5), z~U(zmin-0.4, zmax+0.4); no exact zeros.
    lim = (xmin, xmax, ymin, ymax, zmin, zmax).
    """
    xmin, xmax, ymin, ymax, zmin, zmax = lim
    t = 0x51D0 + int(seed) * 7919
    p = np.empty((n, 3), dtype=np.float32)
    p[:, 0] = uniform((n,), t + 1, xmin - 5.0, xmax + 5.0)
    p[:, 1] = uniform((n,), t + 2, ymin - 5.0, ymax + 5.0)
    p[:, 2] = uniform((n,), t + 3, zmin - 0.4, zmax + 0.4)
    p[p == 0.0] = np.float32(1e-3)
    return p


def synthetic_image(h, w, seed):
    """uint8 [3,h,w] image, uniform over 0..255."""
    u = uniform((3, h, w), 0xA11CE + int(seed) * 104729, 0.0, 256.0)
    return np.minimum(u, 255.0).astype(np.uint8)
