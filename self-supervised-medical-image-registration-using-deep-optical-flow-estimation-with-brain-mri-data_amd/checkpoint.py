"""Checkpoint / weight-format interop with the reference: the state_dict keys of the mireg
modules equal the reference's, so its files load as they are; these helpers write and read the same containers.

  training_state.pt  {'epoch', 'model_state_dict', 'best_loss', 'optimizer_state_dict'}      reference train.py:150-156,183-188
  best_weight.pt     {'model_state_dict', 'loss_val', 'photo_loss_val', ..., 'smooth_loss'}  reference train.py:190-201
  RGB -> one-channel folding of FlyingChairs-pretrained first layers                         reference models.py:247,305-309
  Middlebury .flo files                                                                      reference flownet2/utils/flow_utils.py:20-57
Host-side only (torch.save / numpy); nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

FLO_TAG = b"PIEH"          # 202021.25 as float32, the Middlebury magic number


def save_training_state(path: str, model: torch.nn.Module, optimizer_state: dict, epoch: int, best_loss: float) -> None:
    """reference train.py:183-188.  optimizer_state: torch.optim.Adam.state_dict() or RegistrationTrainer.optimizer_state_dict()."""
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "best_loss": best_loss,
                "optimizer_state_dict": optimizer_state}, path)


def load_training_state(path: str, model: torch.nn.Module, trainer=None, optimizer=None, map_location="cpu") -> Tuple[int, float]:
    """reference train.py:150-156: restores model (+ optimizer / trainer moments) and returns (starting_epoch, best_loss)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model_state_dict"])
    if trainer is not None:
        trainer.load_optimizer_state_dict(ck["optimizer_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return int(ck["epoch"]) + 1, float(ck["best_loss"])


def save_best_weight(path: str, model: torch.nn.Module, val: Dict[str, float], train: Dict[str, float]) -> None:
    """reference train.py:195-201; val / train = {'loss', 'photo_loss', 'corr_loss', 'smooth_loss'} of the two splits."""
    out = {"model_state_dict": model.state_dict()}
    for k in ("loss", "photo_loss", "corr_loss", "smooth_loss"):
        out[f"{k}_val"] = val[k]
        out[k] = train[k]
    torch.save(out, path)


def fold_rgb_pretrained(state_dict: Dict[str, torch.Tensor], key: str = "conv1.0.weight", images: int = 2) -> Dict[str, torch.Tensor]:
    """FlyingChairs checkpoints take RGB frames; the MRI slices have one channel per image.  The reference sums each image's
    three input channels (models.py:305-309: two images, [Co,6,k,k] -> [Co,2,k,k]; models.py:247: PWC 'conv1a.0.weight',
    one image, [Co,3,k,k] -> [Co,1,k,k]).  Returns a shallow copy with `key` folded."""
    w = state_dict[key]
    if w.shape[1] != 3 * images:
        raise ValueError(f"{key}: expected {3 * images} input channels, got {tuple(w.shape)}")
    out = dict(state_dict)
    out[key] = torch.cat([w[:, 3 * i:3 * i + 3].sum(dim=1, keepdim=True) for i in range(images)], dim=1)
    return out


def write_flo(path: str, flow) -> None:
    """flow: (H, W, 2) array or (2, H, W) tensor, u then v -> Middlebury .flo (flow_utils.py:28-57)."""
    if isinstance(flow, torch.Tensor):
        flow = flow.detach().cpu().numpy()
    flow = np.asarray(flow, dtype=np.float32)
    if flow.ndim == 3 and flow.shape[0] == 2 and flow.shape[2] != 2:
        flow = np.transpose(flow, (1, 2, 0))
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError(f"write_flo expects (H, W, 2) or (2, H, W), got {flow.shape}")
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        f.write(FLO_TAG)
        np.array(w, dtype=np.int32).tofile(f)
        np.array(h, dtype=np.int32).tofile(f)
        np.ascontiguousarray(flow).tofile(f)             # rows of interleaved (u, v)


def read_flo(path: str) -> np.ndarray:
    """Middlebury .flo -> (H, W, 2) float32 (flow_utils.py:8-26)."""
    with open(path, "rb") as f:
        if f.read(4) != FLO_TAG:
            raise ValueError(f"{path}: not a .flo file (magic number mismatch)")
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError(f"{path}: truncated .flo file")
    return data.reshape(h, w, 2)
