"""Torch-CPU fp32 restatement of the reference model forward -- TEST INFRASTRUCTURE ONLY.

Functional (no nn.Module): consumes a reference-layout ``state_dict`` (481 keys) and
replays reference model.py:210-239 with ATen fp32 ops. Floating point, so a torch fp32 reference is
the permitted oracle form; pinned against the imported reference in tests/golden (G5).
Also used as the ``cpu_baseline`` ("port") in bench.py.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # reference model.py:17
BN_MOM = 0.03        # reference model.py:17

# Quantisation-aware variant (forward(..., quant=True)): the SAME network with values rounded to bf16 at the points where
# the MI355X throughput path stores bf16 -- the input image, the packed weights, every raw conv output y (its BatchNorm
# statistics come from the f32 accumulators, before the rounding) and every activation z = silu(bn(y)) (+ residual); the
# head output stays f32. Rounding is a straight-through estimator for autograd, so `.backward()` gives the gradients of
# this quantised network. It is the yardstick for the bf16 path: the distance between the HIP bf16 results and THIS is
# kernel error, the distance between this and the f32 oracle is what bf16 storage costs by construction.
_QUANT = False


def _q(t):
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach()) if _QUANT else t


def _cbl(sd, prefix, x, k, s, p, training, new_stats=None, res=None):
    """reference model.py:12-28 (Conv2d(bias=False) -> BatchNorm2d(eps 1e-3, mom 0.03) -> SiLU)."""
    w = _q(sd[prefix + ".cbl.0.weight"])
    y = F.conv2d(x, w, None, stride=s, padding=p)
    g, b = sd[prefix + ".cbl.1.weight"], sd[prefix + ".cbl.1.bias"]
    rm, rv = sd[prefix + ".cbl.1.running_mean"], sd[prefix + ".cbl.1.running_var"]
    if _QUANT and training:
        # statistics of the f32 accumulators, normalisation of the bf16-stored y (what bn_act reads back)
        mean = y.mean((0, 2, 3))
        var = y.var((0, 2, 3), unbiased=False)
        sc = g / torch.sqrt(var + BN_EPS)
        yq = _q(y)
        z = F.silu(yq * sc.view(1, -1, 1, 1) + (b - mean * sc).view(1, -1, 1, 1))
        return _q(z if res is None else z + res)           # (the residual is added in f32, one rounding: bn_act)
    if training:
        if new_stats is not None:
            rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(y, rm if new_stats is not None else None, rv if new_stats is not None else None,
                         g, b, True, BN_MOM, BN_EPS)
        if new_stats is not None:
            new_stats[prefix + ".cbl.1.running_mean"] = rm
            new_stats[prefix + ".cbl.1.running_var"] = rv
    else:
        y = F.batch_norm(y, rm, rv, g, b, False, BN_MOM, BN_EPS)
    return _q(F.silu(y) if res is None else F.silu(y) + res)


def _c3(sd, prefix, x, depth, backbone, training, ns):
    """reference model.py:54-92."""
    y = _cbl(sd, prefix + ".c1", x, 1, 1, 0, training, ns)
    for d in range(depth):
        if backbone:   # Bottleneck with residual, model.py:32-50
            t = _cbl(sd, f"{prefix}.seq.{d}.c1", y, 1, 1, 0, training, ns)
            y = _cbl(sd, f"{prefix}.seq.{d}.c2", t, 3, 1, 1, training, ns, res=y)
        else:          # plain [1x1, 3x3] pair, model.py:82-87
            t = _cbl(sd, f"{prefix}.seq.{d}.0", y, 1, 1, 0, training, ns)
            y = _cbl(sd, f"{prefix}.seq.{d}.1", t, 3, 1, 1, training, ns)
    sk = _cbl(sd, prefix + ".c_skipped", x, 1, 1, 0, training, ns)
    return _cbl(sd, prefix + ".c_out", torch.cat([y, sk], 1), 1, 1, 0, training, ns)


def _sppf(sd, prefix, x, training, ns):
    """reference model.py:96-112."""
    x = _cbl(sd, prefix + ".c1", x, 1, 1, 0, training, ns)
    p1 = F.max_pool2d(x, 5, 1, 2)
    p2 = F.max_pool2d(p1, 5, 1, 2)
    p3 = F.max_pool2d(p2, 5, 1, 2)
    return _cbl(sd, prefix + ".c_out", torch.cat([x, p1, p2, p3], 1), 1, 1, 0, training, ns)


BACKBONE_DEPTH = {2: 2, 4: 4, 6: 6, 8: 2}    # reference model.py:187-193


def forward(sd, x, training=True, nc=80, naxs=3, new_stats=None, quant=False):
    """reference model.py:210-239 + HEADS.forward :165-175. Returns list of 3 (B,3,ny,nx,5+nc).
    quant: the bf16-storage variant described at _QUANT."""
    global _QUANT
    _QUANT = bool(quant)
    try:
        return _forward(sd, _q(x), training, nc, naxs, new_stats)
    finally:
        _QUANT = False


def _forward(sd, x, training, nc, naxs, new_stats):
    assert x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0
    ns = new_stats
    bb = []
    x = _cbl(sd, "backbone.0", x, 6, 2, 2, training, ns)
    for idx in range(1, 10):
        if idx in (1, 3, 5, 7):
            x = _cbl(sd, f"backbone.{idx}", x, 3, 2, 1, training, ns)
        elif idx == 9:
            x = _sppf(sd, "backbone.9", x, training, ns)
        else:
            x = _c3(sd, f"backbone.{idx}", x, BACKBONE_DEPTH[idx], True, training, ns)
        if idx in (4, 6):
            bb.append(x)
    nk, outs = [], []
    for idx in range(8):
        if idx in (0, 2):
            x = _cbl(sd, f"neck.{idx}", x, 1, 1, 0, training, ns)
            nk.append(x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")     # model.py:225
            x = torch.cat([x, bb.pop(-1)], 1)
        elif idx in (4, 6):
            x = _cbl(sd, f"neck.{idx}", x, 3, 2, 1, training, ns)
            x = torch.cat([x, nk.pop(-1)], 1)
        else:
            x = _c3(sd, f"neck.{idx}", x, 2, False, training, ns)
            if idx > 2:
                outs.append(x)
    res = []
    for i, o in enumerate(outs):
        y = F.conv2d(o, _q(sd[f"head.out_convs.{i}.weight"]), sd[f"head.out_convs.{i}.bias"])
        bs, _, gy, gx = y.shape
        res.append(y.view(bs, naxs, 5 + nc, gy, gx).permute(0, 1, 3, 4, 2).contiguous())
    return res
