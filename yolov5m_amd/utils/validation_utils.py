"""Eval path of the reference (utils/validation_utils.py:11-144), same class and method names.

The heavy parts are the native hot path: eval-mode model forward (folded-BN conv epilogues), decode of predictions
AND of dense targets (`cells_to_bboxes`, is_pred True / False), per-image NMS at CONF_THRESHOLD, and the masked counting of check_class_accuracy
(y5m_class_obj_accuracy). What stays in torch is the dict packing. mAP itself is torchmetrics'
MeanAveragePrecision -- not part of the hot path: used when importable, otherwise `map_pr_rec` returns the
(preds, targets) lists it would have been fed."""
import csv
import os

import torch

from .bboxes_utils import non_max_suppression
from .plot_utils import cells_to_bboxes


class DenseTargets:
    """The dense target builder of the reference's datasets (Validation_Dataset.__getitem__, dataset.py:337-414; the training
    dataset's :124-196 is the same code): per image, every box claims the cell of its best free anchor on each scale
    ([x_cell, y_cell, w_cell, h_cell, 1, class]) and marks further well-matching anchors as ignored (-1) -- for a WHOLE BATCH
    in one native launch (y5m_yolo_build_targets, the kernel behind YOLO_LOSS.build_targets: the algorithm is the same,
    loss.py:101-192). State like the dataset object's: `anchors` (3,3,2) in stride units (dataset.py:240), which the
    reference's iou_width_height divides by 640 IN PLACE once per box (utils/bboxes_utils.py:18) -- reproduced, bit for bit,
    because which anchors match depends on it. Returns [(B,3,ny,nx,6)] * 3 on the device: what YOLO_EVAL.check_class_accuracy
    and cells_to_bboxes(is_pred=False) consume."""

    def __init__(self, anchors, S=(8, 16, 32), ignore_iou_thresh=0.5, device=None, strided=False):
        """anchors: config.ANCHORS (pixels, what the dataset constructor receives, :216) -- or, with strided=True, a tensor
        already in stride units such as model.head.anchors"""
        import numpy as np
        from .. import _lib
        self._np, self._lib = np, _lib
        self.S = [int(v) for v in S]
        dev = device or _lib.device()
        a = torch.as_tensor(anchors, dtype=torch.float32)
        if strided:
            a = a.reshape(3, 3, 2).clone()
        else:                                           # :240
            a = a.view(3, -1, 2) / torch.tensor(self.S).repeat(6, 1).T.reshape(3, 3, 2)
        self._anc = [a.to(dev).contiguous(), torch.zeros((3, 3, 2), dtype=torch.float32, device=dev)]
        _lib.require_cuda(self._anc[0])
        self.ignore_iou_thresh = float(ignore_iou_thresh)

    @property
    def anchors(self):
        """the dataset object's `self.anchors` after every in-place decay so far (CPU copy)"""
        return self._anc[0].detach().to("cpu")

    def __call__(self, labels, img_hw):
        """labels: per image an (n_i, 5) array [cls, x, y, w, h] (normalised, dataset.py:320-323); img_hw: (H, W) of the batch"""
        np, _lib = self._np, self._lib
        L = _lib.lib()
        dev = self._anc[0].device
        B = len(labels)
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum([len(b) for b in labels])
        flat = (np.concatenate([np.asarray(b, np.float64).reshape(-1, 5) for b in labels], 0) if off[-1] else
                np.zeros((1, 5), np.float64))
        d_boxes = torch.from_numpy(np.ascontiguousarray(flat)).to(dev, non_blocking=True)
        d_off = torch.from_numpy(off).to(dev, non_blocking=True)
        shapes = [(int(img_hw[0] / s), int(img_hw[1] / s)) for s in self.S]                      # :337-339
        dense = [torch.empty((B, 3, ny, nx, 6), dtype=torch.float32, device=dev) for ny, nx in shapes]
        _lib.check(L.y5m_yolo_build_targets(_lib.ptr(d_boxes), _lib.ptr(d_off), B, _lib.int_array([s[0] for s in shapes]),
                                            _lib.int_array([s[1] for s in shapes]), _lib.int_array(self.S),
                                            _lib.ptr(self._anc[0]), _lib.ptr(self._anc[1]), self.ignore_iou_thresh,
                                            _lib.ptr_array(dense), _lib.stream_ptr()), "y5m_yolo_build_targets")
        if off[-1]:
            self._anc.reverse()
        self._keep = (d_boxes, d_off)                   # until the launch has run
        return dense


class YOLO_EVAL:
    def __init__(self, save_logs, conf_threshold, nms_iou_thresh, map_iou_thresh, device, filename, resume):
        """reference :12-42"""
        self.save_logs = save_logs
        self.conf_threshold = conf_threshold
        self.nms_iou_thresh = nms_iou_thresh
        self.map_iou_threshold = map_iou_thresh
        self.device = device
        self.filename = filename
        if self.save_logs and not resume:
            folder = os.path.join("train_eval_metrics", filename)
            os.makedirs(folder, exist_ok=True)
            with open(os.path.join(folder, "eval.csv"), "w") as f:
                csv.writer(f).writerow(["epoch", "class_accuracy", "obj_accuracy", "map50", "map75"])
        self.class_accuracy = None
        self.noobj_accuracy = None
        self.obj_accuracy = None

    def check_class_accuracy(self, model, loader):
        """reference :44-83. Returns (class_accuracy, obj_accuracy) as 0-dim tensors. The per-scale masked counting (:58-68) is
        one native launch per scale (y5m_class_obj_accuracy) into three device counters -- no boolean-mask gathers, no host
        sync inside the loop. NOTE the reference reads the objectness from channel 0 (`out[i][..., 0]`, :66), not channel 4:
        parity is defined on that."""
        from .. import _lib
        L = _lib.lib()
        model.eval()
        counts = None
        for images, y in loader:
            images = images.to(self.device).float() / 255                                     # :52-53
            with torch.no_grad():
                out = model(images)
            if counts is None:
                counts = torch.zeros(3, dtype=torch.int64, device=out[0].device)             # objects, class hits, objectness hits
            for i in range(3):
                o = out[i] if (out[i].dtype == torch.float32 and out[i].is_contiguous()) else out[i].float().contiguous()
                yi = y[i].to(o.device, non_blocking=True).float().contiguous()
                _lib.require_cuda(o, yi)
                if o.shape[:-1] != yi.shape[:-1] or yi.shape[-1] != 6:
                    raise _lib.Y5MError(f"check_class_accuracy: scale {i}: logits {tuple(o.shape)} vs targets {tuple(yi.shape)}")
                _lib.check(L.y5m_class_obj_accuracy(_lib.ptr(o), _lib.ptr(yi), o.numel() // o.shape[-1], o.shape[-1],
                                                    float(self.conf_threshold), _lib.ptr(counts), _lib.stream_ptr()),
                           "y5m_class_obj_accuracy")
        if counts is None:
            counts = torch.zeros(3, dtype=torch.int64, device=self.device)
        class_accuracy = counts[1] / (counts[0] + 1e-16)                                      # :72-73
        obj_accuracy = counts[2] / (counts[0] + 1e-16)
        if self.save_logs:
            self.class_accuracy = round(float(class_accuracy), 3)
            self.obj_accuracy = round(float(obj_accuracy), 3)
        model.train()
        return class_accuracy, obj_accuracy

    def eval_boxes(self, model, loader, anchors):
        """reference :93-128: the (preds, targets) lists handed to MeanAveragePrecision.update"""
        model.eval()
        preds, targets = [], []
        for images, labels in loader:
            images = images.to(self.device).float() / 255
            with torch.no_grad():
                predictions = model(images)
            pred_boxes = cells_to_bboxes(predictions, anchors, strides=model.head.stride, is_pred=True, to_list=False)
            true_boxes = cells_to_bboxes([l.to(self.device) for l in labels], anchors, strides=model.head.stride,
                                         is_pred=False, to_list=False)
            pred_boxes = non_max_suppression(pred_boxes, iou_threshold=self.nms_iou_thresh, threshold=self.conf_threshold,
                                             tolist=False, max_detections=300)
            true_boxes = non_max_suppression(true_boxes, iou_threshold=self.nms_iou_thresh, threshold=self.conf_threshold,
                                             tolist=False, max_detections=300)
            preds.append(dict(boxes=pred_boxes[..., 2:], scores=pred_boxes[..., 1], labels=pred_boxes[..., 0]))
            targets.append(dict(boxes=true_boxes[..., 2:], labels=true_boxes[..., 0]))
        model.train()
        return preds, targets

    def map_pr_rec(self, model, loader, anchors, epoch):
        """reference :85-144"""
        preds, targets = self.eval_boxes(model, loader, anchors)
        try:
            from torchmetrics.detection.mean_ap import MeanAveragePrecision
        except Exception:
            return preds, targets                       # torchmetrics is not part of this build
        metric = MeanAveragePrecision()
        metric.update(preds, targets)
        metrics = metric.compute()
        map50, map75 = metrics["map_50"], metrics["map_75"]
        if self.save_logs:
            with open(os.path.join("train_eval_metrics", self.filename, "eval.csv"), "a") as f:
                csv.writer(f).writerow([epoch, self.class_accuracy, self.obj_accuracy, map50.item(), map75.item()])
        return map50, map75
