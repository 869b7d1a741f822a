"""ctypes binding of liby5m.so (the C ABI declared in include/y5m.h).

There is NO CPU fallback: importing this module without the built library raises, and every entry
point refuses non-CUDA tensors. PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liby5m.so")

c_void_p, c_int, c_int64, c_float, c_double, c_size_t = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_size_t)


class Y5MError(RuntimeError):
    pass


class Targets(ctypes.Structure):
    """mirror of y5m_targets (include/y5m.h)"""
    _fields_ = [("count", c_void_p), ("bagg", c_void_p), ("tbox", c_void_p), ("anch", c_void_p),
                ("tcls", c_void_p)]


class ConvDesc(ctypes.Structure):
    """mirror of y5m_conv (include/y5m.h)"""
    _fields_ = [("B", c_int), ("Hi", c_int), ("Wi", c_int), ("Cin", c_int), ("ldin", c_int),
                ("Ho", c_int), ("Wo", c_int), ("Cout", c_int), ("ldout", c_int),
                ("k", c_int), ("s", c_int), ("p", c_int), ("dtype", c_int)]


_SIGS = {
    "y5m_version": (ctypes.c_char_p, []),
    "y5m_last_error": (ctypes.c_char_p, []),
    "y5m_device_ok": (c_int, []),
    "y5m_decode_scale": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p,
                                 c_int64, c_int64, c_void_p]),
    "y5m_decode_targets_scale": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64,
                                         c_int64, c_void_p]),
    "y5m_nms_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "y5m_nms": (c_int, [c_void_p, c_int, c_int64, c_float, c_double, c_int, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_size_t, c_void_p]),
    "y5m_iou": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "y5m_iou_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p,
                            c_void_p]),
    "y5m_build_targets_workspace_bytes": (c_size_t, [c_int, c_int]),
    "y5m_build_targets": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                  c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "y5m_compute_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_void_p, c_void_p, c_int]),
    "y5m_compute_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                 c_int, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
}

_lib = None


def lib():
    """Load liby5m.so or fail loudly (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Y5MError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(_SIGS.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def register(name, restype, argtypes):
    """Used by sibling modules to declare further entry points next to their wrappers."""
    _SIGS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes


def exported_symbols():
    return sorted(_SIGS.keys())


def check(rc, what):
    if rc != 0:
        raise Y5MError(f"{what} failed (rc={rc}): {lib().y5m_last_error().decode()}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Y5MError("yolov5m_amd ops run on the MI355X only: got a CPU tensor (no CPU fallback).")


def int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


def float_array(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])


def ptr_array(tensors):
    return (c_void_p * len(tensors))(*[t.data_ptr() if t is not None else 0 for t in tensors])
