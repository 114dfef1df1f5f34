"""opticalFlowReg -- the registration wrapper (reference models.py:208-289) on the HIP engine.

Same constructor contract (`conv_predictor` string), same attributes (`.predictor`, `.stn`) and the
same 4-tuple from forward.  API fix for SURVEY Q1: `forward(x, segs=None)` -- the segmentation /
deformation-grid branch only runs when `segs` is given, and the grid image is broadcast over the
batch (the reference only works at batch 1 there).  The label rounding stays on device (K14).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .flownets import FlowNetS


def grid_generator(device=None) -> torch.Tensor:
    """reference utils.py:15-23: 256x256 image with lines at rows / cols 7, 23, ..., 247."""
    g = torch.zeros(256, 256, device=device)
    idx = torch.arange(7, 255, 16, device=device)
    g[idx, :] = 1.0
    g[:, idx] = 1.0
    return g


def generate_grid(B: int, H: int, W: int, device) -> torch.Tensor:
    """reference models.py:195-204 (kept for API compatibility; the warp kernel computes pixel
    coordinates in-register and never materialises this tensor)."""
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack((xs, ys), dim=-1).unsqueeze(0).repeat(B, 1, 1, 1)


class opticalFlowReg(nn.Module):
    def __init__(self, conv_predictor: str = "flownets", precision: str = "bf16", pretrained: Optional[str] = None):
        super().__init__()
        name = conv_predictor.lower()
        if "raft" in name:
            raise NotImplementedError(
                f"predictor '{conv_predictor}' is outside the accelerated hot path (SURVEY section 8f); "
                "available: 'flownets' (default), 'flownetc', 'pwc', 'flownet2'")
        if "flownet2" in name:                              # reference models.py:212-225: FlowNet2(args, batchNorm=True)
            from .flownet2 import FlowNet2
            self.predictor = FlowNet2(None, batchNorm=True, precision=precision)
        elif "pwc" in name:
            from .pwcnet import PWCDCNet
            self.predictor = PWCDCNet(md=4, precision=precision)
        elif "flownetc" in name:
            from .flownetc import FlowNetC
            self.predictor = FlowNetC(None, batchNorm=True, precision=precision)
        else:
            self.predictor = FlowNetS(batchNorm=True, precision=precision)
        if pretrained is not None:  # never a hard-coded path
            self.load_pretrained(pretrained)
        self._grid = None

    def load_pretrained(self, path: str) -> None:
        """Weights from a checkpoint file: a registration checkpoint of the reference (`predictor.`-prefixed keys, train.py:183-201)
        or a bare predictor checkpoint as the optical-flow repositories publish them (un-prefixed keys, RGB first layers folded to
        one channel per image as reference models.py:247,305-309 does).  Raises when nothing in the file matches the model."""
        from .checkpoint import fold_rgb_pretrained
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd.get("model_state_dict", sd))
        target = self if any(k.startswith("predictor.") for k in sd) else self.predictor
        own = target.state_dict()
        sd = dict(sd)
        for k, v in list(sd.items()):                        # first layers trained on RGB: 3 (one image) or 6 (two images) input channels
            if k in own and v.dim() == 4 and v.shape != own[k].shape and v.shape[1] == 3 * own[k].shape[1] and v.shape[0] == own[k].shape[0]:
                sd = fold_rgb_pretrained(sd, k, images=own[k].shape[1])
        matched = [k for k in sd if k in own and sd[k].shape == own[k].shape]
        if not matched:
            raise RuntimeError(f"{path}: no tensor of the checkpoint matches this model's parameters "
                               f"(checkpoint keys like {list(sd)[:3]}, model keys like {list(own)[:3]})")
        target.load_state_dict({k: sd[k] for k in matched}, strict=False)

    def stn(self, flow: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
        return ops.stn(flow, frame)

    def forward(self, x: torch.Tensor, segs: Optional[torch.Tensor] = None):
        flow_predictions = self.predictor(x)
        moving = x[:, 1:2].float().contiguous()
        warped_images = [self.stn(flow, moving) for flow in flow_predictions]
        if segs is None:
            return flow_predictions, warped_images, 0, 0
        B = x.shape[0]
        m_seg = segs[:, 1:2].float().contiguous()
        warped_segs = self.stn(flow_predictions[0].detach(), m_seg)
        if self._grid is None or self._grid.device != x.device:
            self._grid = grid_generator(x.device).view(1, 1, 256, 256)
        warped_grid = self.stn(flow_predictions[0].detach(), self._grid.expand(B, 1, 256, 256).contiguous())
        return flow_predictions, warped_images, ops.seg_round(warped_segs), warped_grid
