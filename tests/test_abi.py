"""CPU: the C-ABI shared library loads and exports every symbol include/y5m.h declares, and the
product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "y5m.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(y5m_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "y5m_nms" in syms and "y5m_compute_loss" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    from yolov5m_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build liby5m.so first (__graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(L, s)]
    assert not missing, f"declared in include/y5m.h but not exported: {missing}"
    assert L.y5m_version is not None


def test_python_binding_covers_header():
    from yolov5m_amd import _lib
    import yolov5m_amd  # noqa: F401  (registers every wrapper's signatures)
    bound = set(_lib.exported_symbols())
    missing = [s for s in _declared_symbols() if s not in bound]
    assert not missing, f"no ctypes signature registered for: {missing}"


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    from yolov5m_amd import _lib
    from yolov5m_amd.utils.bboxes_utils import non_max_suppression, intersection_over_union
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes
    with pytest.raises(_lib.Y5MError):
        non_max_suppression(torch.zeros(1, 4, 6), 0.5, 0.1)
    with pytest.raises(_lib.Y5MError):
        intersection_over_union(torch.zeros(3, 4), torch.zeros(3, 4))
    with pytest.raises(_lib.Y5MError):
        cells_to_bboxes([torch.zeros(1, 3, 2, 2, 85)] * 3, torch.ones(3, 3, 2), [8, 16, 32], is_pred=True)


def test_native_train_step_refuses_a_loss_it_has_no_launch_list_for():
    """NativeTrainStep enqueues native build-targets + loss launches; it knows the reference's two losses
    (ComputeLoss, YOLO_LOSS -- train.py:102-106) and refuses anything else by TYPE at construction, before touching a device,
    instead of failing on a missing attribute inside the first step"""
    from yolov5m_amd import _lib
    from yolov5m_amd.utils.training_utils import NativeTrainStep
    with pytest.raises(_lib.Y5MError, match="ComputeLoss or YOLO_LOSS"):
        NativeTrainStep(object(), lambda *a, **k: 0.0)
