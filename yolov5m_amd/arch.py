"""Static description of the YOLOv5m topology (reference model.py:178-239).

One table drives three consumers: the nn.Module mirror (parameter names / shapes = the 481-key
state_dict contract), the native execution plan (yolov5m_amd/engine.py) and the
synthetic-weight generator. Nothing here is executable model code.
"""
from collections import namedtuple

# a Conv-BN-SiLU unit (reference model.py:12-28)
CBL = namedtuple("CBL", "name cin cout k s p")

BN_EPS = 1e-3      # model.py:17
BN_MOMENTUM = 0.03  # model.py:17


def _c3(prefix, cin, cout, width, depth, backbone):
    c_ = int(width * cin)                                   # model.py:73
    out = [CBL(f"{prefix}.c1", cin, c_, 1, 1, 0), CBL(f"{prefix}.c_skipped", cin, c_, 1, 1, 0)]
    for d in range(depth):
        if backbone:                                        # Bottleneck, model.py:44-47
            out += [CBL(f"{prefix}.seq.{d}.c1", c_, c_, 1, 1, 0), CBL(f"{prefix}.seq.{d}.c2", c_, c_, 3, 1, 1)]
        else:                                               # model.py:82-87
            out += [CBL(f"{prefix}.seq.{d}.0", c_, c_, 1, 1, 0), CBL(f"{prefix}.seq.{d}.1", c_, c_, 3, 1, 1)]
    out.append(CBL(f"{prefix}.c_out", 2 * c_, cout, 1, 1, 0))
    return out


# (kind, args) per backbone / neck index -- reference model.py:184-206
def blocks(first_out=48):
    f = first_out
    backbone = [
        ("cbl", dict(cin=3, cout=f, k=6, s=2, p=2)),
        ("cbl", dict(cin=f, cout=2 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=2 * f, cout=2 * f, width=0.5, depth=2, backbone=True)),
        ("cbl", dict(cin=2 * f, cout=4 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=4 * f, cout=4 * f, width=0.5, depth=4, backbone=True)),
        ("cbl", dict(cin=4 * f, cout=8 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=8 * f, cout=8 * f, width=0.5, depth=6, backbone=True)),
        ("cbl", dict(cin=8 * f, cout=16 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=16 * f, cout=16 * f, width=0.5, depth=2, backbone=True)),
        ("sppf", dict(cin=16 * f, cout=16 * f)),
    ]
    neck = [
        ("cbl", dict(cin=16 * f, cout=8 * f, k=1, s=1, p=0)),
        ("c3", dict(cin=16 * f, cout=8 * f, width=0.25, depth=2, backbone=False)),
        ("cbl", dict(cin=8 * f, cout=4 * f, k=1, s=1, p=0)),
        ("c3", dict(cin=8 * f, cout=4 * f, width=0.25, depth=2, backbone=False)),
        ("cbl", dict(cin=4 * f, cout=4 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=8 * f, cout=8 * f, width=0.5, depth=2, backbone=False)),
        ("cbl", dict(cin=8 * f, cout=8 * f, k=3, s=2, p=1)),
        ("c3", dict(cin=16 * f, cout=16 * f, width=0.5, depth=2, backbone=False)),
    ]
    return backbone, neck


def cbl_list(first_out=48):
    """All 79 CBL units in state_dict (module registration) order."""
    backbone, neck = blocks(first_out)
    out = []
    for part, blks in (("backbone", backbone), ("neck", neck)):
        for idx, (kind, a) in enumerate(blks):
            pre = f"{part}.{idx}"
            if kind == "cbl":
                out.append(CBL(pre, a["cin"], a["cout"], a["k"], a["s"], a["p"]))
            elif kind == "c3":
                out += _c3(pre, a["cin"], a["cout"], a["width"], a["depth"], a["backbone"])
            else:  # sppf, model.py:96-112
                c_ = a["cin"] // 2
                out += [CBL(f"{pre}.c1", a["cin"], c_, 1, 1, 0), CBL(f"{pre}.c_out", 4 * c_, a["cout"], 1, 1, 0)]
    return out


def state_dict_spec(first_out=48, nc=80, naxs=3, ch=None):
    """Ordered [(key, shape, kind)] of the reference state_dict (481 entries for the default config).
    kind in {conv, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, anchors, head_w, head_b}."""
    ch = ch or (first_out * 4, first_out * 8, first_out * 16)
    spec = []
    for c in cbl_list(first_out):
        spec.append((f"{c.name}.cbl.0.weight", (c.cout, c.cin, c.k, c.k), "conv"))
        spec.append((f"{c.name}.cbl.1.weight", (c.cout,), "bn_w"))
        spec.append((f"{c.name}.cbl.1.bias", (c.cout,), "bn_b"))
        spec.append((f"{c.name}.cbl.1.running_mean", (c.cout,), "bn_rm"))
        spec.append((f"{c.name}.cbl.1.running_var", (c.cout,), "bn_rv"))
        spec.append((f"{c.name}.cbl.1.num_batches_tracked", (), "bn_nbt"))
    spec.append(("head.anchors", (3, naxs, 2), "anchors"))
    for i, c in enumerate(ch):
        spec.append((f"head.out_convs.{i}.weight", ((5 + nc) * naxs, c, 1, 1), "head_w"))
        spec.append((f"head.out_convs.{i}.bias", ((5 + nc) * naxs,), "head_b"))
    return spec
