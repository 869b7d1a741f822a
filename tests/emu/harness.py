"""Test harness of the CPU lane-level executor (tests/emu/include/emu_rt.h). TEST INFRASTRUCTURE.

`with emulated():` swaps the handle yolov5m_amd._lib holds for build/emu/liby5m_emu.so -- the SAME kernel sources compiled
for x86-64, every GPU thread a fiber -- and the package's one device seam (`_lib._device_guard`, a test-only hook) for one that
lets host tensors through, both as scoped mock.patch objects, so that the product's own
Python (ops.py, engine.py, NativeTrainStep, parallel.py) drives the product's own kernel code on this CPU-only container.
Nothing in yolov5m_amd/ knows about it: outside this context manager the package has no CPU path and refuses host tensors
(tests/test_abi.py). Streams and events do not exist here (launches are synchronous, plans run with Y5M_OVERLAP=0), so stream
ordering is NOT what these tests cover -- that is the -m gpu suite's job.

Captured graphs DO exist, as a recording: inside `with torch.cuda.graph(g):` every launching C-ABI call (last argument = stream)
is appended to g instead of executed, and so is every in-place torch op (fills, the step counter) through a TorchDispatchMode;
`g.replay()` runs the list. An op that would ALLOCATE inside a capture raises -- the rule a real capture imposes. That puts the
graph bookkeeping of NativeTrainStep (one graph per plan, segments, eviction, _check_hyper, accumulation) under test on a CPU."""
import contextlib
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


_REC = []                 # stack of op lists being recorded (innermost capture last)
_QUERY = ("workspace_bytes", "eligible", "kernel_name", "owner_ptrs", "_is_", "tile_n", "stats_rows", "stages_stats", "acc_slots",
          "fuse_enabled", "persistent_cu", "device_ok", "last_error", "version", "geometry")


class _LibProxy:
    """the ctypes handle, with every LAUNCHING entry point (signature ends with the stream) recordable"""

    def __init__(self, L, sigs):
        self._L, self._w = L, {}
        self._launching = {n for n, (res, args) in sigs.items() if args and args[-1] is ctypes.c_void_p and not any(q in n for q in _QUERY)}

    def __getattr__(self, name):
        w = self._w.get(name)
        if w is None:
            fn = getattr(self._L, name)
            if name in self._launching:
                def w(*a, _fn=fn):
                    if _REC:
                        _REC[-1].append((_fn, a, None))
                        return 0
                    return _fn(*a)
            else:
                w = fn
            self._w[name] = w
        return w


class FakeGraph:
    """stand-in of torch.cuda.CUDAGraph: the recorded launches and in-place torch ops of one capture"""

    def __init__(self):
        self.ops = []

    def replay(self):
        for fn, a, kw in self.ops:
            fn(*a) if kw is None else fn(*a, **kw)

    def reset(self):
        self.ops = []


@contextlib.contextmanager
def fake_capture(g, *a, **k):
    from torch.utils._python_dispatch import TorchDispatchMode

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            name = func._schema.name.split("::")[1]
            if name.endswith("_"):
                g.ops.append((func, args, kwargs))
                return args[0]
            if name in ("select", "slice", "view", "reshape", "as_strided", "detach", "alias", "expand", "unsqueeze", "squeeze",
                        "_unsafe_view", "permute", "transpose", "t", "narrow", "unbind", "split", "_local_scalar_dense", "item"):
                return func(*args, **kwargs)            # views / host reads of existing tensors: no allocation
            raise RuntimeError(f"capture: aten::{name} inside a captured region (it would allocate, or is not replayable)")
    _REC.append(g.ops)
    try:
        with Rec():
            yield g
    finally:
        _REC.pop()


def emu_lib_path():
    """build/emu/liby5m_emu.so (built on demand); Y5M_EMU_LIB names another build of it (the broken variants of test_emu_checks.py)"""
    if os.environ.get("Y5M_EMU_LIB"):
        return os.environ["Y5M_EMU_LIB"]
    import build
    return build.build()


class _HostGuard:
    """stand-in of yolov5m_amd._lib._DeviceGuard while the executor's library is loaded: host tensors, no streams"""
    device = "cpu"

    def stream(self):
        return ctypes.c_void_p(0)

    def tensors(self, *tensors):
        pass

    def module_device(self, dev):
        pass


@contextlib.contextmanager
def emulated(cus=None):
    """Scoped: every replacement below is a mock.patch that is undone when the block is left, however it is left."""
    from unittest import mock
    from yolov5m_amd import _lib
    import yolov5m_amd.model  # noqa: F401  (registers its entry points before the handle is swapped)
    path = emu_lib_path()
    L = ctypes.CDLL(path)
    for name, (res, args) in list(_lib._SIGS.items()):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L = _LibProxy(L, _lib._SIGS)

    class _Props:
        total_memory = 64 << 30
    env = {"Y5M_OVERLAP": "0"}
    if cus is not None:
        env["Y5M_EMU_CUS"] = str(cus)
    with contextlib.ExitStack() as es:
        es.enter_context(mock.patch.dict(os.environ, env))
        es.enter_context(mock.patch.object(_lib, "_lib", L))                       # the library handle
        es.enter_context(mock.patch.object(_lib, "_device_guard", _HostGuard()))  # the package's one device seam (test-only hook)
        es.enter_context(mock.patch.object(torch.cuda, "synchronize", lambda *a, **k: None))
        es.enter_context(mock.patch.object(torch.cuda, "memory_allocated", lambda *a, **k: 0))
        es.enter_context(mock.patch.object(torch.cuda, "CUDAGraph", FakeGraph))
        es.enter_context(mock.patch.object(torch.cuda, "graph", fake_capture))
        es.enter_context(mock.patch.object(torch.cuda, "get_device_properties", lambda *a, **k: _Props()))
        yield L
