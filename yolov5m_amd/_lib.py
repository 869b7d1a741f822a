"""ctypes binding of liby5m.so (the C ABI declared in include/y5m.h).

There is NO CPU fallback: importing this module without the built library raises, and every entry
point refuses non-CUDA tensors. PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Y5M_LIB") or os.path.join(_HERE, "liby5m.so")     # Y5M_LIB: experiment builds (tools/)

c_void_p, c_int, c_int64, c_float, c_double, c_size_t = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_size_t)


class Y5MError(RuntimeError):
    pass


class Targets(ctypes.Structure):
    """mirror of y5m_targets (include/y5m.h)"""
    _fields_ = [("count", c_void_p), ("bagg", c_void_p), ("tbox", c_void_p), ("anch", c_void_p),
                ("tcls", c_void_p)]


class ConvArgs(ctypes.Structure):
    """mirror of y5m_conv_args (include/y5m.h) -- field order is the ABI"""
    _fields_ = ([("inp", c_void_p), ("w", c_void_p), ("out", c_void_p)] +
                [(n, c_int) for n in ("B", "Hin", "Win", "ldin", "Hg", "Wg", "sy", "sx", "th", "tw", "dh0", "dhs",
                                      "dw0", "dws", "Cin", "K", "Kp", "N", "M", "Hout", "Wout", "ldout", "osy",
                                      "osx", "ooy", "oox", "epi", "act", "accumulate")] +
                [("scale", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("ldres", c_int),
                 ("stats", c_void_p), ("Np", c_int), ("naxs", c_int), ("nch", c_int), ("tiles_m", c_int),
                 ("tiles_n", c_int), ("zeros", c_void_p),
                 ("bn_acc", c_void_p)])


class WgradArgs(ctypes.Structure):
    """mirror of y5m_wgrad_args (include/y5m.h)"""
    _fields_ = ([("dy", c_void_p), ("x", c_void_p), ("dwgt", c_void_p)] +
                [(n, c_int) for n in ("B", "Hin", "Win", "ldx", "Hg", "Wg", "sy", "sx", "th", "tw", "dh0", "dhs",
                                      "dw0", "dws", "C", "N", "M", "lddy", "lddw", "ksplit", "tiles_n", "tiles_c")] +
                [("zeros", c_void_p), ("reserved0", c_int), ("pad_", c_int)])


class BwdPwSeg(ctypes.Structure):
    """mirror of y5m_bwd_pw_seg (include/y5m.h)"""
    _fields_ = [("c0", c_int), ("cn", c_int), ("lddz", c_int), ("pad_", c_int), ("dz", c_void_p), ("acc", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("mean", c_void_p),
                ("invstd", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("dw", c_void_p)]


class BwdPwArgs(ctypes.Structure):
    """mirror of y5m_bwd_pw_args (include/y5m.h)"""
    _fields_ = ([("y", c_void_p), ("x", c_void_p), ("wd", c_void_p), ("dx", c_void_p), ("res", c_void_p), ("M", c_int64)] +
                [(n, c_int) for n in ("ldy", "ldx", "Kp", "lddx", "ldres", "lddw", "N", "C", "accumulate", "act", "nseg")] +
                [("seg", BwdPwSeg * 2)])


class BwdStemArgs(ctypes.Structure):
    """mirror of y5m_bwd_stem_args (include/y5m.h)"""
    _fields_ = ([("dz", c_void_p), ("y", c_void_p), ("x", c_void_p), ("dwgt", c_void_p)] +
                [(n, c_int) for n in ("B", "H", "W", "lddz", "ldy", "ldx", "lddw", "N", "C", "act", "pad_", "pad2_")] +
                [(n, c_void_p) for n in ("acc", "scale", "shift", "mean", "invstd", "dgamma", "dbeta")])


_zero_pages = {}


def zero_page(device):
    """64 zero bytes on `device` (the `zeros` field of y5m_conv_args / y5m_wgrad_args)."""
    key = str(device)
    t = _zero_pages.get(key)
    if t is None:
        t = _zero_pages[key] = torch.zeros(64, dtype=torch.uint8, device=device)
    return t


class PackJob(ctypes.Structure):
    """mirror of y5m_pack_job (include/y5m.h)"""
    _fields_ = ([("src", c_void_p), ("dst", c_void_p)] +
                [(n, c_int) for n in ("Cout", "Cin", "KH", "KW", "mode", "kh0", "khs", "th", "kw0", "kws", "tw",
                                      "rows_p", "Kp", "cstride", "ldd", "pad_")] + [("start", c_int64)])


class FoldJob(ctypes.Structure):
    """mirror of y5m_fold_job (include/y5m.h)"""
    _fields_ = ([(n, c_void_p) for n in ("gamma", "beta", "running_mean", "running_var", "scale", "shift")] +
                [("C", c_int), ("start", c_int)])


EPI_RAW_STATS, EPI_AFFINE_ACT, EPI_HEAD, EPI_DGRAD = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1
F32, BF16 = 0, 1


_SIGS = {
    "y5m_version": (ctypes.c_char_p, []),
    "y5m_last_error": (ctypes.c_char_p, []),
    "y5m_device_ok": (c_int, []),
    "y5m_persistent_cu_count": (c_int, []),
    "y5m_r4_kernel_forms": (c_int, []),
    "y5m_decode_scale": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p,
                                 c_int64, c_int64, c_void_p]),
    "y5m_class_obj_accuracy": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "y5m_decode_targets_scale": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64,
                                         c_int64, c_void_p]),
    "y5m_nms_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "y5m_nms_aladdin": (c_int, [c_void_p, c_int, c_int64, c_double, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
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
    "y5m_compute_loss_sparse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                 c_int, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "y5m_yolo_build_targets": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                       c_void_p, c_void_p]),
    "y5m_compute_loss_dense_workspace_bytes": (c_size_t, [c_int, c_int, c_void_p, c_void_p, c_int]),
    "y5m_compute_loss_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "y5m_compute_loss_dense_sparse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                              c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p,
                                              c_size_t, c_void_p]),
    "y5m_compute_loss_dense_owner_ptrs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p]),
    "y5m_conv_tile_n": (c_int, [c_int]),
    "y5m_conv_is_pointwise": (c_int, [c_void_p, c_int]),
    "y5m_conv_stats_rows": (c_int, [c_void_p, c_int]),
    "y5m_conv_kernel_name": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "y5m_conv_multi_kernel_name": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int]),
    "y5m_wgrad_kernel_name": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "y5m_wgrad_geometry": (c_int, [c_void_p, c_int, c_void_p]),
    "y5m_conv_is_halo": (c_int, [c_void_p, c_int]),
    "y5m_conv_stages_stats": (c_int, [c_void_p, c_int]),
    "y5m_conv": (c_int, [c_void_p, c_int, c_void_p]),
    "y5m_conv_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "y5m_wgrad": (c_int, [c_void_p, c_int, c_void_p]),
    "y5m_pack_weights": (c_int, [c_void_p] + [c_int] * 11 + [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "y5m_pack_weights_batched": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p]),
    "y5m_bwd_pw": (c_int, [c_void_p, c_int, c_void_p]),
    "y5m_bwd_pw_eligible": (c_int, [c_void_p, c_int]),
    "y5m_bwd_stem": (c_int, [c_void_p, c_void_p]),
    "y5m_bwd_stem_eligible": (c_int, [c_void_p]),
    "y5m_unpack_wgrad": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "y5m_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "y5m_s2d_input": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "y5m_bn_finalize_workspace_bytes": (c_size_t, [c_int]),
    "y5m_bn_acc_slots": (c_int, []),
    "y5m_bn_fuse_enabled": (c_int, []),
    "y5m_bn_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                c_void_p]),
    "y5m_bn_fold_batched": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    "y5m_bn_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "y5m_bn_act": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int,
                           c_int, c_int, c_void_p]),
    "y5m_bn_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "y5m_bn_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                           c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_int,
                           c_void_p]),
    "y5m_bn_act_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "y5m_bn_bwd_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "y5m_bn_bwd_fused_phase": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]),
    "y5m_add": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "y5m_upsample2x": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "y5m_upsample2x_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                   c_void_p]),
    "y5m_sppf_pool_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "y5m_sppf_pool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_size_t, c_int, c_void_p]),
    "y5m_sppf_pool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "y5m_sppf_pool_tiled": (c_int, [c_int, c_int, c_int, c_int]),
    "y5m_maxpool5_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "y5m_maxpool5_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                 c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "y5m_head_grad_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                   c_void_p]),
    "y5m_head_grad_pack_sparse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "y5m_compute_loss_owner_ptrs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "y5m_adam_workspace_bytes": (c_size_t, []),
    "y5m_grad_norm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "y5m_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_double,
                              c_double, c_double, c_double, c_double, c_void_p, c_void_p]),
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
        if os.environ.get("Y5M_LIB"):
            # another build of the library (timing / instrumentation builds under build/exp/): say so, loudly -- a stray
            # Y5M_LIB in the environment must not go unnoticed in a training run
            import sys
            sys.stderr.write(f"[yolov5m_amd] WARNING: Y5M_LIB is set: using {LIB_PATH} instead of the in-tree liby5m.so "
                             "(experiment build; not for training)\n")
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


class _DeviceGuard:
    """The ONE seam between this package and the device: which stream launches go to, where the wrappers create their own tensors,
    and what they refuse. The product has exactly this implementation -- "cuda", host tensors refused, no CPU fallback
    (tests/test_abi.py::test_no_cpu_fallback). TEST-ONLY hook: tests/emu/harness.py replaces `_device_guard` (together with the
    library handle `_lib`) inside a scoped mock.patch to drive the kernel sources compiled for its CPU executor; nothing else may."""
    device = "cuda"

    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def tensors(self, *tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise Y5MError("yolov5m_amd ops run on the MI355X only: got a CPU tensor (no CPU fallback).")

    def module_device(self, dev):
        if torch.device(dev).type != "cuda":
            raise Y5MError("YOLOV5m runs on the MI355X only: call .to('cuda') first (no CPU fallback)")


_device_guard = _DeviceGuard()


def stream_ptr():
    return _device_guard.stream()


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def require_cuda(*tensors):
    _device_guard.tensors(*tensors)


def device():
    """where the wrappers create their own tensors (there is no other choice: no CPU fallback)"""
    return _device_guard.device


def require_cuda_device(dev):
    _device_guard.module_device(dev)


def int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


def float_array(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])


def ptr_array(tensors):
    return (c_void_p * len(tensors))(*[t.data_ptr() if t is not None else 0 for t in tensors])
