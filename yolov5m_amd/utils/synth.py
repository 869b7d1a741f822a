"""Counter-based synthetic weights / inputs, reproducible WITHOUT torch's RNG.

value(key, i) = u01(splitmix64(fnv1a64(key) + i * GOLDEN)) -- identical on any host, so the GPU box
and the build container construct bit-identical weights from nothing but the key names.
"""
import numpy as np
import torch

from ..arch import state_dict_spec
from ..config import ANCHORS, STRIDES

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def u01(key: str, n: int, offset: int = 0) -> np.ndarray:
    """n float32 uniforms in [0,1) for stream `key`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        z = np.uint64(fnv1a64(key)) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def uniform(key, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    v = u01(key, n) * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def synth_state_dict(first_out=48, nc=80, seed="y5m", gain=6.0):
    """Reference-layout state_dict with He-uniform conv weights (bound sqrt(gain/fan_in)) and
    non-trivial BN affine / running statistics so eval-mode folding is exercised."""
    sd = {}
    for key, shape, kind in state_dict_spec(first_out, nc):
        k = f"{seed}/{key}"
        if kind in ("conv", "head_w"):
            fan_in = shape[1] * shape[2] * shape[3]
            b = float(np.sqrt(gain / fan_in))
            sd[key] = uniform(k, shape, -b, b)
        elif kind == "head_b":
            sd[key] = uniform(k, shape, -0.1, 0.1)
        elif kind == "bn_w":
            sd[key] = uniform(k, shape, 0.5, 1.5)
        elif kind in ("bn_b", "bn_rm"):
            sd[key] = uniform(k, shape, -0.1, 0.1)
        elif kind == "bn_rv":
            sd[key] = uniform(k, shape, 0.5, 1.5)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
        elif kind == "anchors":
            a = torch.tensor(ANCHORS).float().view(3, -1, 2)
            sd[key] = a / torch.tensor(STRIDES).float().view(3, 1, 1)
    return sd


def synth_images(B, H, W, seed="img"):
    return uniform(f"{seed}/{B}x{H}x{W}", (B, 3, H, W), 0.0, 1.0)


def synth_labels(B, boxes_per_image=8, nc=80, seed="lab"):
    """SURVEY 8(d) config 2 recipe: cls~U{0..nc-1}, xy~U(.05,.95), wh~U(.02,.5). (nt,6) rows."""
    nt = B * boxes_per_image
    u = u01(f"{seed}/{B}x{boxes_per_image}", nt * 5).reshape(nt, 5)
    img = np.repeat(np.arange(B, dtype=np.float32), boxes_per_image)
    cls = np.floor(u[:, 0] * nc).astype(np.float32)
    xy = 0.05 + 0.9 * u[:, 1:3]
    wh = 0.02 + 0.48 * u[:, 3:5]
    return torch.from_numpy(np.concatenate([img[:, None], cls[:, None], xy, wh], 1).astype(np.float32))
