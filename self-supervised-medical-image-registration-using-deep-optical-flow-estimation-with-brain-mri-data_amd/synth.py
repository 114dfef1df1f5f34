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


def _strides3(t: torch.Tensor):
    """Element strides (item, d, h, w) of an (N, D, H, W) view (any permutation / crop of a contiguous volume)."""
    return t.stride(0), t.stride(1), t.stride(2), t.stride(3)


def resample_volume(vol: torch.Tensor, size, mode: str = "bilinear", rot_k: int = 0) -> torch.Tensor:
    """Resized + Rotate90d of the reference's MONAI pipelines on device.  vol: (N, D, H, W) fp32 VIEW of device memory (Transposed and
    SpatialCropd of dataset.py:55-56,143-145 are `permute` / slicing on the caller's side: only strides change); size = (d, h, w)
    after the resize (d == D for the per-slice 2-D pipeline); mode "bilinear" (torch's linear interpolation with
    align_corners=False over every resized axis) or "nearest"; rot_k = numpy.rot90 k on the (h, w) axes.  Returns a contiguous
    (N, d, h', w')."""
    from . import _lib
    from .ops import _need_gpu, _stream
    _need_gpu(vol)
    if vol.dim() != 4 or vol.dtype != torch.float32:
        raise RuntimeError(f"resample_volume expects an (N, D, H, W) float32 view, got {tuple(vol.shape)} {vol.dtype}")
    N, D, H, W = vol.shape
    d, h, w = (int(v) for v in size)
    oh, ow = (w, h) if rot_k % 2 else (h, w)
    out = torch.empty(N, d, oh, ow, device=vol.device, dtype=torch.float32)
    _lib.call("mireg_resample_volume", vol.data_ptr(), *_strides3(vol), N, D, H, W, out.data_ptr(), *_strides3(out), d, h, w,
              {"bilinear": 0, "trilinear": 0, "nearest": 1}[mode], rot_k % 4, _stream())
    return out


def scale_intensity(x: torch.Tensor, minv: float = 0.0, maxv: float = 1.0) -> torch.Tensor:
    """ScaleIntensityd(minv, maxv) per item of a contiguous (N, ...) fp32 batch, in place (dataset.py:83,153,209)."""
    from . import _lib
    from .ops import _need_gpu, _stream
    _need_gpu(x)
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise RuntimeError("scale_intensity expects a contiguous float32 batch")
    n = x[0].numel()
    ws = torch.empty(x.shape[0] * 64 * 2, device=x.device, dtype=torch.float32)
    _lib.call("mireg_scale_intensity", x.data_ptr(), x.shape[0], n, float(minv), float(maxv), ws.data_ptr(), _stream())
    return x


def slices_from_volume(image: torch.Tensor, seg=None, z_range=(60, 140), yx_size=(176, 208), size: int = 256, rot_k: int = 1):
    """The volume -> slice front of volume2slices_ds / eval_random_ds (dataset.py:52-57,73-77) on device: image / seg are the
    Transposed volumes (Z, Y, X) fp32 on the GPU; SpatialCropd(roi_start=(z0, 0, 0), roi_end=(z1, Y, X)), one patch per slice,
    Resized to (size, size) (image bilinear, label map nearest), Rotate90d(k).  Returns (slices (n, 1, size, size), seg slices or
    None); the per-pair steps that follow (elastic_deform, concat, scale_intensity) are the other functions of this module."""
    z0, z1 = z_range
    crop = image[z0:z1, :yx_size[0], :yx_size[1]]
    out = resample_volume(crop.unsqueeze(0), (crop.shape[0], size, size), "bilinear", rot_k)[0].unsqueeze(1)
    out_seg = None
    if seg is not None:
        cs = seg[z0:z1, :yx_size[0], :yx_size[1]]
        out_seg = resample_volume(cs.unsqueeze(0), (cs.shape[0], size, size), "nearest", rot_k)[0].unsqueeze(1)
    return out, out_seg


def prepare_volume(image: torch.Tensor, size=(256, 256, 176), rot_k: int = 2) -> torch.Tensor:
    """volume_ds (dataset.py:141-148) on device: image = the Transposed volume (A0, A1, A2) fp32 on the GPU; Resized(size, trilinear,
    align_corners=False) and Rotate90d(k, spatial_axes=(0, 1)).  Returns (1, s0', s1', s2) contiguous in MONAI's axis order."""
    # the kernel rotates its last two logical axes: present the volume as (d, h, w) = (A2, A0, A1) through a permuted view
    v = image.permute(2, 0, 1).unsqueeze(0)
    out = resample_volume(v, (size[2], size[0], size[1]), "trilinear", rot_k)          # (1, s2, s0', s1')
    return out.permute(0, 2, 3, 1).contiguous()


def affine_deform3d(vol: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """The RandAffined step of volume_ds (dataset.py:150-152) as torch states it: F.grid_sample(vol, F.affine_grid(theta, vol.size()))
    with trilinear interpolation, zero padding, align_corners=False; vol (B, C, D, H, W) fp32, theta (B, 3, 4)."""
    from . import _lib
    from .ops import _need_gpu, _stream
    _need_gpu(vol, theta)
    B, C, D, H, W = vol.shape
    if tuple(theta.shape) != (B, 3, 4):
        raise RuntimeError(f"affine_deform3d: theta {tuple(theta.shape)} does not match a batch of {B}")
    vol, theta = vol.contiguous().float(), theta.contiguous().float()
    out = torch.empty_like(vol)
    _lib.call("mireg_affine_sample3d", vol.data_ptr(), theta.data_ptr(), out.data_ptr(), B, C, D, H, W, _stream())
    return out
