"""Driver of tests/emu/probes.hip (TEST INFRASTRUCTURE): the same probe kernels on the CPU executor and on the GPU.
inputs() -> {probe id: uint32 array}; run(call) -> {name: uint32 array} where call(which, in_array) -> out_array (1024 words).
expected_mfma(): the two matrix products in numpy, laid out as the C/D register layout of the guides says
(col = lane & 15, row = 4 (lane >> 4) + reg) -- a truth that does not depend on the executor."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "probes.hip")
GPU_LIB = os.path.join(HERE, "libprobes_gfx950.so")
NAMES = ["mfma_bf16", "mfma_f32", "tr16", "dpp", "lanes", "buffer", "lds_dma"]
OUT_WORDS = 1024


def _bf16_bits(x):
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def matrices():
    """small-integer, ASYMMETRIC operands (a transposed or row/column-swapped layout cannot pass): exact in bf16 and in f32 sums"""
    i, k = np.arange(16)[:, None], np.arange(32)[None, :]
    A = ((i * 3 + k * 5) % 7 - 3).astype(np.float32)
    B = ((i * 5 + k * 2 + 1) % 9 - 4).astype(np.float32)
    A4 = ((i * 7 + np.arange(4)[None, :] * 3) % 11 - 5).astype(np.float32) * 0.5
    B4 = ((i * 2 + np.arange(4)[None, :] * 5 + 3) % 13 - 6).astype(np.float32) * 0.25
    return A, B, A4, B4


def inputs():
    A, B, A4, B4 = matrices()
    ab = np.concatenate([_bf16_bits(A).reshape(-1), _bf16_bits(B).reshape(-1)])                 # 1024 x u16 = 512 words
    out = {0: ab.view(np.uint32).copy(), 1: np.concatenate([A4.reshape(-1), B4.reshape(-1)]).view(np.uint32).copy()}
    for w in (2, 3, 4):
        out[w] = np.zeros(4, dtype=np.uint32)
    out[5] = (np.arange(64, dtype=np.uint32) * 2654435761 % 1000003).astype(np.uint32)           # 256 bytes = num_records
    out[6] = (np.arange(256, dtype=np.uint32) + 0x1000000).astype(np.uint32)
    return out


def expected_mfma():
    A, B, A4, B4 = matrices()
    res = {}
    for name, a, b in (("mfma_bf16", A, B), ("mfma_f32", A4, B4)):
        D = a.astype(np.float64) @ b.astype(np.float64).T                                        # D[i][n] = sum_k A[i][k] B[n][k]
        out = np.zeros((64, 4), dtype=np.float32)
        for lane in range(64):
            for r in range(4):
                out[lane, r] = D[4 * (lane >> 4) + r, lane & 15]
        res[name] = out.reshape(-1).view(np.uint32)
    return res


def run(call):
    ins = inputs()
    res = {}
    for which, name in enumerate(NAMES):
        o = np.asarray(call(which, ins[which]), dtype=np.uint32)
        n = {"mfma_bf16": 256, "mfma_f32": 256, "tr16": 256, "dpp": 512, "lanes": 640, "buffer": 256, "lds_dma": 512}[name]
        res[name] = o[:n].copy()
    return res


def build_gpu():
    """hipcc --offload-arch=gfx950 -> tests/emu/libprobes_gfx950.so (cross-compiles without a GPU; called by __graft_entry__.build())"""
    if os.path.exists(GPU_LIB) and os.path.getmtime(GPU_LIB) >= os.path.getmtime(SRC):
        return GPU_LIB
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", GPU_LIB, SRC])
    return GPU_LIB


def build_emu():
    import build as B
    from translate import translate
    out = os.path.join(B.OUT_DIR, "libprobes_emu.so")
    rt = os.path.join(HERE, "emu_rt.cpp")
    deps = [SRC, rt, os.path.join(HERE, "include", "emu_rt.h"), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "translate.py")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(B.OUT_DIR, exist_ok=True)
    cpp = os.path.join(B.OUT_DIR, "probes.cpp")
    with open(SRC) as f:
        text = translate(f.read(), SRC)
    with open(cpp, "w") as f:
        f.write(f'#line 1 "{SRC}"\n' + text)
    subprocess.check_call([B.CXX] + B.FLAGS + ["-shared", cpp, rt, "-o", out])
    return out


def run_emu():
    L = ctypes.CDLL(build_emu())
    L.emu_probe.restype = ctypes.c_int
    L.emu_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def call(which, a):
        a = np.ascontiguousarray(a)
        o = np.zeros(OUT_WORDS, dtype=np.uint32)
        rc = L.emu_probe(which, a.ctypes.data, o.ctypes.data, None)
        assert rc == 0, (which, rc)
        return o
    return run(call)


def run_gpu():
    import torch
    L = ctypes.CDLL(GPU_LIB)
    L.emu_probe.restype = ctypes.c_int
    L.emu_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    def call(which, a):
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()
        o = torch.zeros(OUT_WORDS, dtype=torch.int32, device="cuda")
        rc = L.emu_probe(which, t.data_ptr(), o.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0, (which, rc)
        return o.cpu().numpy().view(np.uint32)
    return run(call)
