"""The detect flow -- the INTENDED path of the reference's detect.py:31-54 (the file itself raises two TypeErrors before it reaches
the model: :35 calls load_model_checkpoint with keywords its signature does not have, :54 passes `to_list=` to a function whose
parameter is `tolist`): checkpoint -> image(s) -> /255 -> eval forward -> cells_to_bboxes(is_pred=True) ->
non_max_suppression(iou 0.45, threshold 0.25). Everything between the uint8 image and the kept rows runs on the MI355X: the input stage
(y5m_preprocess_u8), the model (engine.py), the decode (y5m_decode_scale) and the NMS (y5m_nms); the one host sync is the read of the
per-image row counts. Plotting (plot_image, detect.py:55) is visualisation and out of scope.

    from yolov5m_amd.detect import detect
    rows = detect(model, images)          # list (per image) of [class, score, x1, y1, x2, y2] rows

    python -m yolov5m_amd.detect --model_name model_1 --checkpoint 8 --img some.jpg      # the reference's CLI flags
"""
import argparse
import os

import torch

from . import _lib, config
from .utils.bboxes_utils import non_max_suppression
from .utils.plot_utils import cells_to_bboxes
from .utils.training_utils import preprocess_u8


def _as_batch(images, device):
    """detect.py:46-50: HWC uint8 array -> CHW -> [None] -> float / 255, for one image or a batch, on the device.
    Accepts a (H, W, 3) / (B, H, W, 3) uint8 array (PIL / numpy layout), a (B, 3, H, W) uint8 tensor, or float tensors already in
    0..255 (the reference divides whatever it gets by 255)."""
    t = torch.as_tensor(images)
    if t.dim() == 3:
        t = t[None]
    if t.dim() != 4:
        raise _lib.Y5MError("detect: expected one (H, W, 3) image or a batch (B, H, W, 3) / (B, 3, H, W)")
    if t.shape[1] != 3 and t.shape[-1] == 3:
        t = t.permute(0, 3, 1, 2)                           # :46 img.transpose((2, 0, 1))
    if t.shape[1] != 3:
        raise _lib.Y5MError("detect: images must have 3 channels")
    if t.shape[2] % 32 or t.shape[3] % 32:
        raise _lib.Y5MError(f"detect: image size {tuple(t.shape[2:])} is not a multiple of 32 (model.py:211 asserts the same)")
    t = t.to(device, non_blocking=True)
    if t.dtype == torch.uint8:
        return preprocess_u8(t.contiguous())                # :49 `.float() / 255` as one native launch on the uint8 batch
    return t.float() / 255


def detect(model, images, iou_threshold=0.45, threshold=0.25, max_detections=300, tolist=True):
    """detect.py:46-54 for one image or a batch. Returns what non_max_suppression returns: a list (per image) of
    [class, score, x1, y1, x2, y2] rows (tolist=True), or one concatenated tensor (tolist=False, bboxes_utils.py:209)."""
    dev = next(model.parameters()).device
    _lib.require_cuda_device(dev)
    x = _as_batch(images, dev)
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():                               # :51-52
            out = model(x)
            boxes = cells_to_bboxes(out, model.head.anchors, model.head.stride, is_pred=True, to_list=False)   # :53
            return non_max_suppression(boxes, iou_threshold=iou_threshold, threshold=threshold,
                                       max_detections=max_detections, tolist=tolist)                           # :54
    finally:
        model.train(was_training)


def main(argv=None):
    """the reference's command line (detect.py:22-27); --checkpoint takes the epoch number or the file name"""
    import numpy as np
    from PIL import Image
    from .model import YOLOV5m
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", type=str, default="model_1", help="folder inside SAVED_CHECKPOINT")
    ap.add_argument("--checkpoint", type=str, default="checkpoint_epoch_8.pth.tar", help="checkpoint inside SAVED_CHECKPOINT/model_name")
    ap.add_argument("--img", type=str, required=True, help="path of the image to predict")
    ap.add_argument("--nc", type=int, default=len(getattr(config, "FLIR", [])) or 80)
    args = ap.parse_args(argv)
    model = YOLOV5m(first_out=config.FIRST_OUT, nc=args.nc, anchors=config.ANCHORS,
                    ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(config.DEVICE)
    path = os.path.join("SAVED_CHECKPOINT", args.model_name, args.checkpoint if args.checkpoint.endswith(".tar")
                        else f"checkpoint_epoch_{args.checkpoint}.pth.tar")
    model.load_state_dict(torch.load(path, map_location=config.DEVICE, weights_only=True)["state_dict"])
    img = np.array(Image.open(args.img).convert("RGB"))
    h, w = img.shape[0] // 32 * 32, img.shape[1] // 32 * 32
    rows = detect(model, img[:h, :w])[0]                    # (cropped to a multiple of 32: the reference asserts instead)
    for r in rows:
        print(f"class {int(r[0])} score {r[1]:.3f} box {r[2]:.1f} {r[3]:.1f} {r[4]:.1f} {r[5]:.1f}")
    print(f"{len(rows)} detections")


if __name__ == "__main__":
    main()
