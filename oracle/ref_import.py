"""Import the real reference (/root/reference) in THIS container to generate golden vectors.

TEST INFRASTRUCTURE ONLY. Never runs on the GPU box (/root/reference does not exist there) and is
never imported by the product. Recipe = SURVEY.md Appendix D: third-party modules that are absent
from the image (torchvision, albumentations, cv2, imagesize, torchmetrics) are stubbed in
``sys.modules`` before the reference is put on ``sys.path``. The only stub with arithmetic in it is
``torchvision.ops.nms``, which forwards to the C restatement in oracle/csrc/y5m_oracle.c
(orc_nms_tv012) -- i.e. NMS goldens pin the reference's *wrapper* (filter, corner conversion, class
offset, truncation) around our restatement of torchvision, not torchvision itself.
"""
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"


def available():
    import os
    return os.path.isdir(REF)


def _install_stubs():
    import torch
    import torch.nn.functional as F
    from . import cnative

    sys.dont_write_bytecode = True
    for name in ("albumentations", "cv2", "imagesize"):
        sys.modules.setdefault(name, MagicMock())
    tm = types.ModuleType("torchmetrics")
    tmd = types.ModuleType("torchmetrics.detection")
    tmm = types.ModuleType("torchmetrics.detection.mean_ap")
    tmm.MeanAveragePrecision = MagicMock()
    tm.detection = tmd
    tmd.mean_ap = tmm
    sys.modules.setdefault("torchmetrics", tm)
    sys.modules.setdefault("torchmetrics.detection", tmd)
    sys.modules.setdefault("torchmetrics.detection.mean_ap", tmm)

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvo = types.ModuleType("torchvision.ops")

    class InterpolationMode:
        NEAREST = "nearest"

    class Resize:
        def __init__(self, size, interpolation=None):
            self.size = size

        def __call__(self, x):
            return F.interpolate(x, size=list(self.size), mode="nearest")

    def nms(boxes, scores, iou_threshold):
        keep = cnative.nms_tv012(boxes.detach().contiguous().float().numpy(),
                                 scores.detach().contiguous().float().numpy(), float(iou_threshold))
        return torch.from_numpy(keep)

    tvt.InterpolationMode = InterpolationMode
    tvt.Resize = Resize
    tvo.nms = nms
    tv.transforms = tvt
    tv.ops = tvo
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.modules.setdefault("torchvision.ops", tvo)


_loaded = None


def load():
    """Returns a namespace with the reference's hot-path symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (only available in the build container)")
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import config as ref_config
    from model import YOLOV5m
    from ultralytics_loss import ComputeLoss
    from loss import YOLO_LOSS
    from utils.plot_utils import cells_to_bboxes, make_grids
    from utils.bboxes_utils import (non_max_suppression, intersection_over_union, iou_width_height,
                                    non_max_suppression_aladdin)
    ns = types.SimpleNamespace(
        config=ref_config, YOLOV5m=YOLOV5m, ComputeLoss=ComputeLoss, YOLO_LOSS=YOLO_LOSS,
        cells_to_bboxes=cells_to_bboxes, make_grids=make_grids,
        non_max_suppression=non_max_suppression, intersection_over_union=intersection_over_union,
        iou_width_height=iou_width_height, non_max_suppression_aladdin=non_max_suppression_aladdin)
    _loaded = ns
    return ns
