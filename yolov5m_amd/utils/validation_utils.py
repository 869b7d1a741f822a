"""Eval path of the reference (utils/validation_utils.py:11-144), same class and method names.

The heavy parts are the native hot path: eval-mode model forward (folded-BN conv epilogues), decode of predictions
AND of dense targets (`cells_to_bboxes`, is_pred True / False), per-image NMS at CONF_THRESHOLD. What stays in torch
is the bookkeeping around them (boolean-mask counting, dict packing). mAP itself is torchmetrics'
MeanAveragePrecision -- not part of the hot path: used when importable, otherwise `map_pr_rec` returns the
(preds, targets) lists it would have been fed."""
import csv
import os

import torch

from .bboxes_utils import non_max_suppression
from .plot_utils import cells_to_bboxes


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
        """reference :44-83. Returns (class_accuracy, obj_accuracy) as 0-dim tensors. NOTE the reference reads the
        objectness from channel 0 (`out[i][..., 0]`, :66), not channel 4: parity is defined on that."""
        model.eval()
        tot_class_preds, correct_class = 0, 0
        tot_obj, correct_obj = 0, 0
        for images, y in loader:
            images = images.to(self.device).float() / 255                                     # :52-53
            with torch.no_grad():
                out = model(images)
            for i in range(3):
                yi = y[i].to(self.device)
                obj = yi[..., 4] == 1                                                         # :60
                correct_class += torch.sum(torch.argmax(out[i][..., 5:][obj], dim=-1) == yi[..., 5][obj])
                tot_class_preds += torch.sum(obj)
                obj_preds = torch.sigmoid(out[i][..., 0]) > self.conf_threshold               # :66
                correct_obj += torch.sum(obj_preds[obj] == yi[..., 4][obj])
                tot_obj += torch.sum(obj)
        class_accuracy = correct_class / (tot_class_preds + 1e-16)
        obj_accuracy = correct_obj / (tot_obj + 1e-16)
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
