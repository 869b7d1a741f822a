#!/usr/bin/env python
"""Per-launch timing of the eval-mode (inference) forward: one HIP-event pair per entry of the engine's launch list.
usage: eval_layers.py [B] [S]   (default 128 1280: BASELINE.json configs[4])"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import config, _lib
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.utils.synth import synth_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
torch.manual_seed(0)
m = YOLOV5m(48, 80, config.ANCHORS, (192, 384, 768)).to("cuda"); m.compute_dtype = "bf16"; m.eval()
x = synth_images(8, S, S).to("cuda").repeat(B // 8, 1, 1, 1)
with torch.no_grad():
    for _ in range(2):
        m(x)
    eng = next(reversed(m._engines.values()))
    L = _lib.lib()
    torch.cuda.synchronize()
    rows = []
    for fn, args in list(eng.pack) + list(eng.fwd):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(*args); e1.record()
        rows.append((fn, e0, e1))
    torch.cuda.synchronize()
tot = 0.0
agg = {}
for fn, e0, e1 in rows:
    ms = e0.elapsed_time(e1)
    tot += ms
    a = getattr(fn, "__defaults__", None)
    a = a[0] if a and isinstance(a[0], _lib.ConvArgs) else None
    if a is not None:
        buf = ctypes.create_string_buffer(192)
        L.y5m_conv_kernel_name(ctypes.byref(a), eng.dtype, buf, 192)
        fl = 2.0 * a.M * a.N * a.K
        key = f"conv M={a.M:8d} N={a.N:4d} K={a.K:5d} taps={a.th}x{a.tw} s={a.sy} {buf.value.decode()}"
        r = agg.setdefault(key, [0, 0.0, 0.0, (a.M * (a.K // (a.th * a.tw)) + a.M * a.N) * 2.0]); r[0] += 1; r[1] += ms; r[2] += fl
    else:
        r = agg.setdefault(getattr(fn, "__name__", "op"), [0, 0.0, 0.0, 0.0]); r[0] += 1; r[1] += ms
for k, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    extra = f"  {fl / ms / 1e9:7.1f} TF/s  {by * n / ms / 1e9:6.2f} TB/s(alg)" if fl else ""
    print(f"{ms:8.3f} ms x{n:3d}  {k}{extra}")
print(f"total {tot:.2f} ms  (B={B} @ {S}x{S})")
