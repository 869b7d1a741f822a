#!/usr/bin/env python
"""Phase timing of the halo-patch conv kernel (experiment build with -DHL_TIMING: Y5M_LIB=build/exp/lib_TIMING.so).
The kernel writes, for (block 0, waves 0 and 4), the cycles spent in each phase (work, barrier wait) behind the zero page.
usage: halo_timing.py B C H W N"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolov5m_amd import _lib
from yolov5m_amd._lib import ConvArgs, EPI_RAW_STATS, EPI_DGRAD, BF16

B, Cin, H, W, Cout = [int(v) for v in sys.argv[1:6]]
epi = EPI_DGRAD if len(sys.argv) > 6 and sys.argv[6] == "dgrad" else EPI_RAW_STATS
L = _lib.lib()
dev = "cuda"
M = B * H * W
K = 9 * Cin
Kp = (K + 63) // 64 * 64
Np = Cout
x = torch.randn(M * Cin, device=dev).bfloat16()
y = torch.zeros(M * Cout, device=dev).bfloat16()
w = (torch.randn(Np * Kp, device=dev) * 0.05).bfloat16()
z = torch.zeros(4096, dtype=torch.uint8, device=dev)
a = ConvArgs()
a.zeros = z.data_ptr()
a.inp, a.w, a.out = x.data_ptr(), w.data_ptr(), y.data_ptr()
a.B, a.Hin, a.Win, a.ldin, a.Hg, a.Wg, a.sy, a.sx = B, H, W, Cin, H, W, 1, 1
a.th, a.tw, a.dh0, a.dhs, a.dw0, a.dws = 3, 3, -1, 1, -1, 1
a.Cin, a.K, a.Kp, a.N, a.M = Cin, K, Kp, Cout, M
a.Hout, a.Wout, a.ldout, a.osy, a.osx = H, W, Cout, 1, 1
a.Np, a.epi = Np, epi
if epi == EPI_RAW_STATS:
    stats = torch.zeros(L.y5m_conv_stats_rows(ctypes.byref(a), BF16) * 2 * Np, device=dev)
    a.stats = stats.data_ptr()
assert L.y5m_conv_is_halo(ctypes.byref(a), BF16)
for _ in range(3):
    _lib.check(L.y5m_conv(ctypes.byref(a), BF16, _lib.stream_ptr()), "conv")
torch.cuda.synchronize()
t = z.view(torch.int64)[32:48].cpu().tolist()
names = ["R0", "M0", "R1", "M1"]
for wv, off in ((0, 0), (4, 8)):
    print(f"wave {wv}: " + "  ".join(f"{names[i]} work={t[off + 2 * i]} wait={t[off + 2 * i + 1]}" for i in range(4)),
          " total", sum(t[off:off + 8]))
