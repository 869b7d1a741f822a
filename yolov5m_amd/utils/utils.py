"""Checkpoint wire format of the reference (utils/utils.py:56-82): one `.pth.tar` per epoch holding
{"state_dict": model.state_dict(), "optimizer": optimizer.state_dict()} under
<folder_path>/<filename>/checkpoint_epoch_<epoch>.pth.tar, loaded back from SAVED_CHECKPOINT/<model_name>/.

Same signatures and file layout, so checkpoints written by the reference load here and vice versa: the model's
481 state_dict keys are the reference's (model.py) and the native optimizer exports / imports
torch.optim.Adam's state_dict (NativeTrainStep.optimizer_state_dict / load_optimizer_state_dict)."""
import os

import torch

from .. import config


def _optim_state(optim):
    return optim.optimizer_state_dict() if hasattr(optim, "optimizer_state_dict") else optim.state_dict()


def make_checkpoint(model, optim):
    """the dict train.py hands to save_checkpoint (train.py: {"state_dict": ..., "optimizer": ...})"""
    return {"state_dict": model.state_dict(), "optimizer": _optim_state(optim)}


def save_checkpoint(state, folder_path, filename, epoch):
    """reference utils/utils.py:56-63"""
    path = os.path.join(folder_path, filename)
    os.makedirs(path, exist_ok=True)
    torch.save(state, os.path.join(path, f"checkpoint_epoch_{str(epoch)}.pth.tar"))


def _ckpt_path(model_name, last_epoch, root="SAVED_CHECKPOINT"):
    return os.path.join(root, model_name, f"checkpoint_epoch_{last_epoch}.pth.tar")


def load_model_checkpoint(model_name, model, last_epoch, root="SAVED_CHECKPOINT"):
    """reference utils/utils.py:66-73"""
    checkpoint = torch.load(_ckpt_path(model_name, last_epoch, root), map_location=config.DEVICE, weights_only=True)
    model.load_state_dict(checkpoint["state_dict"])


def load_optim_checkpoint(model_name, optim, last_epoch, root="SAVED_CHECKPOINT"):
    """reference utils/utils.py:76-82; `optim` is a torch optimizer or a NativeTrainStep"""
    checkpoint = torch.load(_ckpt_path(model_name, last_epoch, root), map_location=config.DEVICE, weights_only=True)
    if hasattr(optim, "load_optimizer_state_dict"):
        optim.load_optimizer_state_dict(checkpoint["optimizer"])
    else:
        optim.load_state_dict(checkpoint["optimizer"])
