"""The training flow of the reference's train.py:56-140 on the native path, with the data loaders handed in (datasets, augmentation and
plotting are out of scope): model -> optimizer -> resume from SAVED_CHECKPOINT -> loss (YOLO_LOSS unless
ultralytics_loss, train.py:102-106) -> per epoch: train_loop, YOLO_EVAL on the validation loader, save_checkpoint. The optimizer is the
fused NativeTrainStep by default (fused=False: torch.optim.Adam exactly as train.py:61, autograd through the same native kernels);
both write / read the reference's `{"state_dict", "optimizer"}` checkpoints (utils/utils.py:56-82).

    from yolov5m_amd.train import train
    train(train_loader, val_loader=None, epochs=..., ultralytics_loss=False, filename=None, resume=False)

    python -m yolov5m_amd.train --synthetic 4 --bs 8 --size 320 --epochs 2 [--ultralytics_loss] [--resume --filename model_1]
runs the same flow on synthetic uint8 batches (no dataset in this repository): what bench.py's step does, through the reference's
entry points, with checkpoints. Under `python -m torch.distributed.run --nproc-per-node N -m yolov5m_amd.train ...` it is data parallel
(one process per GPU, gradients summed over RCCL, rank 0 writes the checkpoints).
"""
import argparse
import os

import numpy as np
import torch

from . import _lib, config, parallel
from .loss import YOLO_LOSS
from .model import YOLOV5m
from .ultralytics_loss import ComputeLoss
from .utils.training_utils import NativeTrainStep, train_loop
from .utils.utils import load_model_checkpoint, load_optim_checkpoint, make_checkpoint, save_checkpoint
from .utils.validation_utils import YOLO_EVAL


def _run_name(resume, filename, root):
    """train.py:63-92: model_<n + 1> for a new run, the given name for a resumed one (+ its last saved epoch)"""
    os.makedirs(root, exist_ok=True)
    if resume:
        if not filename:
            raise _lib.Y5MError("train: resume needs the run's filename (train.py:79-85)")
        epochs = [int(f.split(".")[0].split("_")[-1]) for f in os.listdir(os.path.join(root, filename)) if f.startswith("checkpoint_epoch_")]
        if not epochs:
            raise _lib.Y5MError(f"train: no checkpoint under {os.path.join(root, filename)}")
        return filename, max(epochs)
    saved = [int(n.split("_")[1]) for n in os.listdir(root) if n.startswith("model_") and n.split("_")[1].isdigit()]
    return filename or f"model_{max(saved) + 1 if saved else 1}", None


def train(train_loader, val_loader=None, epochs=1, ultralytics_loss=False, rect=False, filename=None, resume=False, save_model=True,
          save_logs=False, only_eval=False, fused=True, use_graph=True, nc=config.nc, dtype="bf16", nt_max=1024, grad_hook=None,
          checkpoint_root="SAVED_CHECKPOINT", model=None):
    """train.py:56-140. Returns (model, optim, per-epoch mean losses). `train_loader` yields (images uint8 or float 0..255 (B,3,H,W),
    labels): labels = the (nt, 6) tensor of collate_fn_ultra with ultralytics_loss, the tuple of per-image (n_i, 5) arrays of collate_fn
    otherwise (dataset.py:199-209). `val_loader` (optional) yields what YOLO_EVAL expects (images, dense targets of the 3 scales)."""
    if model is None:
        model = YOLOV5m(first_out=config.FIRST_OUT, nc=nc, anchors=config.ANCHORS,
                        ch=(config.FIRST_OUT * 4, config.FIRST_OUT * 8, config.FIRST_OUT * 16)).to(config.DEVICE)        # :58-59
    model.compute_dtype = dtype
    # data parallel (no counterpart in the reference; parallel.py): when the process group is up (parallel.init_from_env under
    # torch.distributed.run) every rank runs this flow on ITS loader, the gradients are summed over the ranks before the optimizer
    # (bucketed and overlapped with the backward pass when no accumulation is in the way), rank 0 writes the checkpoints
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if world > 1:
        if not fused:
            raise _lib.Y5MError("train: data parallelism goes through the fused step (the gradient exchange is its grad_hook)")
        grad_hook = grad_hook or parallel.GradAllReduce(world)
    filename, last_epoch = _run_name(resume, filename, checkpoint_root)
    if world > 1:                                           # (one run name for all ranks: rank 0's)
        names = [filename]
        dist.broadcast_object_list(names, src=0)
        filename = names[0]
    loss_fn = (ComputeLoss(model, save_logs=save_logs, filename=filename, resume=resume) if ultralytics_loss else
               YOLO_LOSS(model, rect_training=rect, save_logs=save_logs, filename=filename, resume=resume))               # :102-106
    if fused:
        optim = NativeTrainStep(model, loss_fn, lr=config.LEARNING_RATE, weight_decay=config.WEIGHT_DECAY, nt_max=nt_max,
                                use_graph=use_graph, grad_hook=grad_hook)
    else:
        optim = torch.optim.Adam(model.parameters(), lr=config.LEARNING_RATE, weight_decay=config.WEIGHT_DECAY)        # :61
    starting_epoch = 1
    if resume:                                                                                                            # :79-86
        load_model_checkpoint(filename, model, last_epoch, root=checkpoint_root)
        load_optim_checkpoint(filename, optim, last_epoch, root=checkpoint_root)
        starting_epoch = last_epoch + 1
    parallel.broadcast_parameters(model)                    # (no-op with one rank: every replica starts from rank 0's weights)
    evaluate = YOLO_EVAL(save_logs=save_logs, conf_threshold=config.CONF_THRESHOLD, nms_iou_thresh=config.NMS_IOU_THRESH,
                         map_iou_thresh=config.MAP_IOU_THRESH, device=config.DEVICE, filename=filename, resume=resume)   # :108-111
    losses = []
    for epoch in range(starting_epoch, epochs + starting_epoch):                                                          # :115
        model.train()
        if not only_eval:
            losses.append(train_loop(model=model, loader=train_loader, loss_fn=loss_fn, optim=optim, scaler=None, epoch=epoch,
                                     num_epochs=epochs + starting_epoch, multi_scale_training=not rect))                 # :120-123
        model.eval()
        if val_loader is not None:
            evaluate.check_class_accuracy(model, val_loader)                                                              # :127
            evaluate.map_pr_rec(model, val_loader, anchors=model.head.anchors, epoch=epoch)                               # :129
        if save_model and rank == 0:
            save_checkpoint(make_checkpoint(model, optim), folder_path=checkpoint_root, filename=filename, epoch=epoch)   # :136-140
    return model, optim, losses


class SyntheticLoader:
    """n batches of uint8 images + labels in the format the chosen loss's collate function produces (dataset.py:199-209); the same
    batches every epoch (seeded), 8 boxes per image as in BASELINE.json configs[2]"""

    def __init__(self, n, batch, size, ultralytics_loss, boxes_per_image=8, rank=0):
        from .utils.synth import synth_labels
        g = torch.Generator().manual_seed(rank)
        self.batches = []
        for i in range(n):
            img = torch.randint(0, 256, (batch, 3, size, size), generator=g, dtype=torch.uint8)
            lab = synth_labels(batch, boxes_per_image, seed=f"train/lab{i}/rank{rank}")
            if not ultralytics_loss:
                t = lab.numpy().astype(np.float64)
                lab = tuple(t[t[:, 0] == b][:, 1:] for b in range(batch))
            self.batches.append((img, lab))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def main(argv=None):
    ap = argparse.ArgumentParser(description="reference train.py:18-31 flags that concern the hot path, on synthetic batches")
    ap.add_argument("--synthetic", type=int, default=4, help="batches per epoch (this repository ships no dataset)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--ultralytics_loss", action="store_true")
    ap.add_argument("--rect", action="store_true", help="no multi_scale (train.py:122)")
    ap.add_argument("--nosavemodel", action="store_true")
    ap.add_argument("--nosavelogs", action="store_true")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--filename", type=str)
    ap.add_argument("--no-fused", action="store_true", help="torch.optim.Adam + autograd instead of the fused NativeTrainStep")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    a = ap.parse_args(argv)
    rank, _, world = parallel.init_from_env()               # torch.distributed.run: one process per GPU, RCCL (--bs is per GPU)
    loader = SyntheticLoader(a.synthetic, a.bs, a.size, a.ultralytics_loss, rank=rank)
    _, _, losses = train(loader, epochs=a.epochs, ultralytics_loss=a.ultralytics_loss, rect=a.rect, filename=a.filename, resume=a.resume,
                         save_model=not a.nosavemodel, save_logs=not a.nosavelogs, fused=not a.no_fused, dtype=a.dtype, nt_max=a.bs * 8)
    for i, l in enumerate(losses):
        print(f"[rank {rank} of {world}] epoch {i + 1}: training_loss {l:.4f}")


if __name__ == "__main__":
    main()
