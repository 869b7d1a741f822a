"""ctypes bindings of oracle/liby5m_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liby5m_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "csrc", "y5m_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        p = ctypes.c_void_p
        i64 = ctypes.c_int64
        L.orc_nms_tv012.restype = i64
        L.orc_nms_tv012.argtypes = [p, p, i64, ctypes.c_double, p]
        L.orc_non_max_suppression.restype = i64
        L.orc_non_max_suppression.argtypes = [p, i64, ctypes.c_float, ctypes.c_double, i64, p, p]
        L.orc_nms_aladdin.restype = i64
        L.orc_nms_aladdin.argtypes = [p, i64, ctypes.c_double, ctypes.c_float, ctypes.c_int, i64, p]
        L.orc_build_targets_ultra.restype = i64
        L.orc_build_targets_ultra.argtypes = [p, i64, p, i64, i64, i64, ctypes.c_float,
                                              p, p, p, p, p, p, p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def nms_tv012(boxes, scores, iou_threshold):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib().orc_nms_tv012(_ptr(boxes), _ptr(scores), n, float(iou_threshold), _ptr(keep))
    return keep[:k].copy()


def non_max_suppression_image(boxes, iou_threshold, threshold, max_detections=300):
    """One image of reference utils/bboxes_utils.py:175-209. boxes (N,6) -> (rows (k,6), idx (k,))."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    N = boxes.shape[0]
    out = np.empty((max(max_detections, 1), 6), dtype=np.float32)
    idx = np.empty(max(max_detections, 1), dtype=np.int64)
    k = lib().orc_non_max_suppression(_ptr(boxes), N, float(threshold), float(iou_threshold),
                                      int(max_detections), _ptr(out), _ptr(idx))
    return out[:k].copy(), idx[:k].copy()


def nms_aladdin(boxes, iou_threshold, threshold, box_format="corners", max_detections=300):
    """reference utils/bboxes_utils.py:129-173 for one list: (N,6) rows -> kept indices (keep order)"""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    idx = np.empty(max(boxes.shape[0], 1), dtype=np.int64)
    k = lib().orc_nms_aladdin(_ptr(boxes), boxes.shape[0], float(threshold), float(iou_threshold),
                              1 if box_format == "midpoint" else 0, int(max_detections), _ptr(idx))
    return idx[:k].copy()


def build_targets_ultra_scale(targets, anchors, ny, nx, anchor_t=4.0):
    """One scale of ComputeLoss.build_targets. Returns dict(b,a,gj,gi,tbox,anch,tcls)."""
    targets = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 6)
    anchors = np.ascontiguousarray(anchors, dtype=np.float32).reshape(-1, 2)
    nt, na = targets.shape[0], anchors.shape[0]
    cap = max(5 * na * nt, 1)
    b = np.empty(cap, np.int64); a = np.empty(cap, np.int64)
    gj = np.empty(cap, np.int64); gi = np.empty(cap, np.int64)
    tbox = np.empty((cap, 4), np.float32); anch = np.empty((cap, 2), np.float32)
    tcls = np.empty(cap, np.int64)
    n = lib().orc_build_targets_ultra(_ptr(targets), nt, _ptr(anchors), na, int(ny), int(nx),
                                      float(anchor_t), _ptr(b), _ptr(a), _ptr(gj), _ptr(gi),
                                      _ptr(tbox), _ptr(anch), _ptr(tcls))
    return dict(b=b[:n].copy(), a=a[:n].copy(), gj=gj[:n].copy(), gi=gi[:n].copy(),
                tbox=tbox[:n].copy(), anch=anch[:n].copy(), tcls=tcls[:n].copy())
