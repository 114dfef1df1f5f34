"""Synthetic OASIS-style registration pairs: the reference's data pipeline
(dataset.py, MONAI + Analyze files) is CPU I/O outside the hot path, so benches and tests use this
generator.  fixed = smooth "brain" (random anisotropic Gaussians inside an elliptical mask, min-max
scaled like ScaleIntensityd, dataset.py:83); seg = 3 nested intensity thresholds (labels 0..3);
moving = elastic warp of fixed with a random 16-px control grid (Rand2DElasticd stand-in,
dataset.py:78).  Runs once at set-up on the CPU with plain torch; not part of any timed region.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def make_pairs(n: int, size: int = 256, seed: int = 6, magnitude=(0.0, 0.5), spacing: int = 16):
    """-> images (n,2,size,size) [fixed, moving] in [0,1], segs (n,2,size,size) labels {0,1,2,3} (float32)."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    fixed = torch.zeros(n, 1, size, size)
    for i in range(n):
        img = torch.zeros(size, size)
        for _ in range(6):
            cx, cy = (torch.rand(2, generator=g) - 0.5) * 0.9
            sx, sy = 0.12 + 0.35 * torch.rand(2, generator=g)
            amp = 0.4 + 0.6 * torch.rand(1, generator=g)
            img = img + amp * torch.exp(-((xs - cx) ** 2 / (2 * sx ** 2) + (ys - cy) ** 2 / (2 * sy ** 2)))
        a, b = 0.75 + 0.15 * torch.rand(2, generator=g)
        img = img * (((xs / a) ** 2 + (ys / b) ** 2) < 1).float()
        img = (img - img.min()) / (img.max() - img.min() + 1e-12)
        fixed[i, 0] = img
    seg_f = torch.bucketize(fixed, torch.tensor([0.25, 0.5, 0.75])).float()
    cg = size // spacing + 1
    mag = magnitude[0] + (magnitude[1] - magnitude[0]) * torch.rand(n, 1, 1, 1, generator=g)
    ctrl = (torch.rand(n, 2, cg, cg, generator=g) * 2 - 1) * mag * spacing          # pixels
    disp = F.interpolate(ctrl, size=(size, size), mode="bicubic", align_corners=True)
    gx = (xs.unsqueeze(0) + disp[:, 0] * 2 / size)
    gy = (ys.unsqueeze(0) + disp[:, 1] * 2 / size)
    grid = torch.stack((gx, gy), -1)
    moving = F.grid_sample(fixed, grid, mode="bicubic", padding_mode="zeros", align_corners=True).clamp(0, 1)
    seg_m = F.grid_sample(seg_f, grid, mode="nearest", padding_mode="zeros", align_corners=True)
    return torch.cat((fixed, moving), 1).contiguous(), torch.cat((seg_f, seg_m), 1).contiguous()


def elastic_deform(images: torch.Tensor, segs, ctrl: torch.Tensor):
    """On-device elastic deformation (the per-batch Rand2DElasticd step, reference dataset.py:78,150-152,205): images (B,C,H,W)
    in [0,1], segs (B,Cs,H,W) label maps or None, ctrl (B,2,gh,gw) control-grid displacements in pixels -> (warped images,
    warped segs).  Same arithmetic as make_pairs above (bicubic field, bicubic image / nearest label resampling, zero padding,
    align_corners=True) on the HIP kernels of csrc/augment.hip; no CPU fallback."""
    from . import _lib
    from .ops import _need_gpu, _stream
    _need_gpu(images, ctrl)
    B, C, H, W = images.shape
    if ctrl.shape[0] != B or ctrl.shape[1] != 2:
        raise RuntimeError(f"elastic_deform: ctrl {tuple(ctrl.shape)} does not match a batch of {B}")
    images, ctrl = images.contiguous(), ctrl.contiguous()
    st = _stream()
    disp = torch.empty(B, 2, H, W, device=images.device, dtype=torch.float32)
    _lib.call("mireg_resize_bicubic_fwd", ctrl.data_ptr(), disp.data_ptr(), B * 2, ctrl.shape[2], ctrl.shape[3], H, W, st)
    out = torch.empty_like(images)
    out_seg = None
    if segs is not None:
        _need_gpu(segs)
        segs = segs.contiguous()
        out_seg = torch.empty_like(segs)
    _lib.call("mireg_elastic_sample", images.data_ptr(), segs.data_ptr() if segs is not None else None, disp.data_ptr(),
              out.data_ptr(), out_seg.data_ptr() if out_seg is not None else None, B, C, segs.shape[1] if segs is not None else 0,
              H, W, st)
    return out, out_seg



def affine_deform(images: torch.Tensor, segs, theta: torch.Tensor):
    """On-device affine resampling (the per-batch RandAffined step, reference dataset.py:79,151): images (B,C,H,W), segs
    (B,Cs,H,W) or None, theta (B,2,3) -> F.grid_sample(x, F.affine_grid(theta, x.size())) with torch's defaults, bilinear for
    the images and nearest for the label maps; no CPU fallback."""
    from . import _lib
    from .ops import _need_gpu, _stream
    _need_gpu(images, theta)
    B, C, H, W = images.shape
    if tuple(theta.shape) != (B, 2, 3):
        raise RuntimeError(f"affine_deform: theta {tuple(theta.shape)} does not match a batch of {B}")
    images, theta = images.contiguous(), theta.contiguous()
    out = torch.empty_like(images)
    out_seg = None
    if segs is not None:
        _need_gpu(segs)
        segs = segs.contiguous()
        out_seg = torch.empty_like(segs)
    _lib.call("mireg_affine_sample2d", images.data_ptr(), segs.data_ptr() if segs is not None else None, theta.data_ptr(),
              out.data_ptr(), out_seg.data_ptr() if out_seg is not None else None, B, C, segs.shape[1] if segs is not None else 0,
              H, W, _stream())
    return out, out_seg
