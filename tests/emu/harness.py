"""Test harness of the CPU lane-level executor (tests/emu/include/emu_rt.h). TEST INFRASTRUCTURE.

`with emulated():` swaps the handle yolov5m_amd._lib holds for build/emu/liby5m_emu.so -- the SAME kernel sources compiled
for x86-64, every GPU thread a fiber -- and lets host tensors through the wrappers' device checks, so that the product's own
Python (ops.py, engine.py, NativeTrainStep, parallel.py) drives the product's own kernel code on this CPU-only container.
Nothing in yolov5m_amd/ knows about it: outside this context manager the package has no CPU path and refuses host tensors
(tests/test_abi.py). Streams, events and graphs do not exist here (launches are synchronous): plans run with Y5M_OVERLAP=0
and use_graph=False, so stream ordering is NOT what these tests cover -- that is the -m gpu suite's job."""
import contextlib
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def emu_lib_path():
    import build
    return build.build()


@contextlib.contextmanager
def emulated(cus=None):
    from yolov5m_amd import _lib
    import yolov5m_amd.model  # noqa: F401  (registers its entry points before the handle is swapped)
    path = emu_lib_path()
    saved_env = {k: os.environ.get(k) for k in ("Y5M_OVERLAP", "Y5M_EMU_CUS")}
    os.environ["Y5M_OVERLAP"] = "0"
    if cus is not None:
        os.environ["Y5M_EMU_CUS"] = str(cus)
    L = ctypes.CDLL(path)
    for name, (res, args) in list(_lib._SIGS.items()):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    saved = dict(lib=_lib._lib, stream_ptr=_lib.stream_ptr, require_cuda=_lib.require_cuda,
                 require_cuda_device=getattr(_lib, "require_cuda_device", None),
                 sync=torch.cuda.synchronize, mem=torch.cuda.memory_allocated, props=torch.cuda.get_device_properties,
                 zero=dict(_lib._zero_pages))
    _lib._lib = L
    saved["device"], _lib.DEVICE = _lib.DEVICE, "cpu"
    _lib.stream_ptr = lambda: ctypes.c_void_p(0)
    _lib.require_cuda = lambda *t: None
    _lib.require_cuda_device = lambda dev: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.memory_allocated = lambda *a, **k: 0

    class _Props:
        total_memory = 64 << 30
    torch.cuda.get_device_properties = lambda *a, **k: _Props()
    try:
        yield L
    finally:
        _lib._lib = saved["lib"]
        _lib.DEVICE = saved["device"]
        _lib.stream_ptr, _lib.require_cuda = saved["stream_ptr"], saved["require_cuda"]
        if saved["require_cuda_device"] is not None:
            _lib.require_cuda_device = saved["require_cuda_device"]
        torch.cuda.synchronize, torch.cuda.memory_allocated = saved["sync"], saved["mem"]
        torch.cuda.get_device_properties = saved["props"]
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
