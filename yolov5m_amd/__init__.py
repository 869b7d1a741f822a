"""yolov5m_amd -- MI355X-native (gfx950) YOLOv5m hot path behind the reference's Python call signatures.

    from yolov5m_amd.model import YOLOV5m                       # reference model.py
    from yolov5m_amd.ultralytics_loss import ComputeLoss        # reference ultralytics_loss.py
    from yolov5m_amd.utils.plot_utils import cells_to_bboxes    # reference utils/plot_utils.py
    from yolov5m_amd.utils.bboxes_utils import non_max_suppression, intersection_over_union

All arithmetic runs in hand-written HIP kernels inside liby5m.so (C ABI: include/y5m.h). There is no
CPU fallback: ops raise Y5MError on CPU tensors or when the library is missing.
"""
from . import _lib, config  # noqa: F401
from .model import YOLOV5m  # noqa: F401
from .ultralytics_loss import ComputeLoss  # noqa: F401
from .loss import YOLO_LOSS  # noqa: F401
from .utils.plot_utils import cells_to_bboxes, make_grids  # noqa: F401
from .utils.bboxes_utils import non_max_suppression, intersection_over_union, iou_width_height  # noqa: F401

__all__ = ["YOLOV5m", "ComputeLoss", "YOLO_LOSS", "cells_to_bboxes", "make_grids", "non_max_suppression",
           "intersection_over_union", "iou_width_height", "config"]
