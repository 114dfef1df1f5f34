"""Per-sample evaluation metrics of the reference's inference loop on device (inference.py:66-75, utils.py:41-59): the
reference computes each of them sample by sample and pulls every scalar to the host; here one call covers the batch and the
results stay on the device.  Same names and argument order as reference utils.py (fixed, warped); there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib
from .ops import _need_gpu, _stream

MI_SCALE, MI_BINS = 1500.0, 1501          # utils.py:53-54: round(x * 1500) on [0, 1] images
_MI_WS: dict = {}


def _as_batch(fixed: torch.Tensor, warped: torch.Tensor):
    _need_gpu(fixed, warped)
    if fixed.shape != warped.shape:
        raise RuntimeError(f"metrics: fixed {tuple(fixed.shape)} and warped {tuple(warped.shape)} differ")
    return fixed.contiguous(), warped.contiguous()


def pair_metrics(fixed: torch.Tensor, warped: torch.Tensor, mutual_info: bool = True) -> Dict[str, torch.Tensor]:
    """fixed / warped: (B, ...) batches -> {"mse", "psnr", "corr"[, "ssim"][, "mi"]}: float64 tensors of shape (B,), one value per sample
    (what inference.py:66-75 feeds its running averages with)."""
    f, w = _as_batch(fixed, warped)
    B = f.shape[0]
    n = f.numel() // B
    dev, st = f.device, _stream()
    sums = torch.empty(B, 8, device=dev, dtype=torch.float64)
    out = torch.empty(B, 3, device=dev, dtype=torch.float64)
    _lib.call("mireg_pair_metrics", f.data_ptr(), w.data_ptr(), sums.data_ptr(), out.data_ptr(), B, n, st)
    res = {"mse": out[:, 0], "psnr": out[:, 1], "corr": out[:, 2]}
    if f.dim() >= 3 and f.numel() == B * f.shape[-2] * f.shape[-1] and min(f.shape[-2:]) >= 7:
        res["ssim"] = ssim_batch(f, w)
    if mutual_info:
        key = (dev, B)
        if key not in _MI_WS:                                  # zeroed once; the kernels leave the tables clean
            _MI_WS.clear()
            _MI_WS[key] = (torch.zeros(B * MI_BINS * MI_BINS, device=dev, dtype=torch.int32),
                           torch.zeros(B * 2 * MI_BINS, device=dev, dtype=torch.int32))
        joint, marg = _MI_WS[key]
        mi = torch.empty(B, device=dev, dtype=torch.float64)
        _lib.call("mireg_mutual_info", f.data_ptr(), w.data_ptr(), joint.data_ptr(), marg.data_ptr(), mi.data_ptr(), B, n, MI_BINS,
                  MI_SCALE, st)
        res["mi"] = mi
    return res


def ssim_batch(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0, win_size: int = 7) -> torch.Tensor:
    """a / b: (B, H, W) or (B, 1, H, W) -> (B,) float64: skimage.metrics.structural_similarity(a[i], b[i], data_range=...) with
    its defaults, as inference.py:70-71 calls it for the image pair and for the segmentation pair."""
    a, b = _as_batch(a, b)
    B, H, W = a.shape[0], a.shape[-2], a.shape[-1]
    if a.numel() != B * H * W:
        raise RuntimeError(f"ssim_batch expects one channel per sample, got {tuple(a.shape)}")
    out = torch.empty(B, device=a.device, dtype=torch.float64)
    _lib.call("mireg_ssim", a.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, int(win_size), float(data_range), _stream())
    return out


def structural_similarity(im1: torch.Tensor, im2: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Drop-in for the call at reference inference.py:70-71 on one (H, W) pair."""
    return ssim_batch(im1.unsqueeze(0), im2.unsqueeze(0), data_range)[0]


def _single(fixed: torch.Tensor, warped: torch.Tensor, key: str) -> torch.Tensor:
    return pair_metrics(fixed.unsqueeze(0), warped.unsqueeze(0), mutual_info=(key == "mi"))[key][0]


def MSE(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.MSE (utils.py:41-42) on one sample."""
    return _single(fixed, warped, "mse")


def PSNR(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.PSNR (utils.py:45-49)."""
    return _single(fixed, warped, "psnr")


def CORR(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.CORR (utils.py:58-59)."""
    return _single(fixed, warped, "corr")


def MI(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.MI (utils.py:52-55)."""
    return _single(fixed, warped, "mi")


def modified_hausdorff(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.modified_hausdorff (utils.py:187-199) on two (n, 2) point sets (device tensors): float64
    scalar.  The contour extraction in front of it (utils.py:154-170, skimage.measure.find_contours) is not part of mireg."""
    _need_gpu(A.float(), B.float())
    A, B = A.float().contiguous(), B.float().contiguous()
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != 2 or B.shape[1] != 2 or A.shape[0] == 0 or B.shape[0] == 0:
        raise RuntimeError(f"modified_hausdorff expects two non-empty (n, 2) point sets, got {tuple(A.shape)} and {tuple(B.shape)}")
    work = torch.empty(A.shape[0] + B.shape[0], device=A.device, dtype=torch.float32)
    out = torch.empty(1, device=A.device, dtype=torch.float64)
    _lib.call("mireg_modified_hausdorff", A.data_ptr(), A.shape[0], B.data_ptr(), B.shape[0], work.data_ptr(), out.data_ptr(), _stream())
    return out[0]


def extract_boundary_points(mask: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.extract_boundary_points (utils.py:155-170) on a binary (H, W) device mask: the (row, col) points of
    skimage.measure.find_contours(mask, 0.5) truncated to int, one per 4-neighbour pixel pair with differing values (raster order;
    skimage's repeated first vertex of closed contours is not reproduced).  Returns an (n, 2) float tensor (this call reads n back)."""
    pts, cnt = _boundary_sets(mask.float().unsqueeze(0), torch.ones(1, device=mask.device))
    return pts[0, :int(cnt[0])]


def _boundary_sets(segs: torch.Tensor, labels: torch.Tensor):
    """segs (m, H, W) fp32 label maps, labels (m,) -> (points (m, 2HW, 2), counts (m,) int32), all on device."""
    _need_gpu(segs, labels)
    segs, labels = segs.contiguous().float(), labels.contiguous().float()
    m, H, W = segs.shape
    dev = segs.device
    rowcnt = torch.empty(m * H, device=dev, dtype=torch.int32)
    rowoff = torch.empty(m * H, device=dev, dtype=torch.int32)
    counts = torch.empty(m, device=dev, dtype=torch.int32)
    pts = torch.empty(m, 2 * H * W, 2, device=dev, dtype=torch.float32)
    _lib.call("mireg_boundary_points", segs.data_ptr(), H * W, m, labels.data_ptr(), H, W, rowcnt.data_ptr(), rowoff.data_ptr(),
              counts.data_ptr(), pts.data_ptr(), 4 * H * W, _stream())
    return pts, counts


def dist_hausdorff(seg1: torch.Tensor, seg2: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference utils.dist_hausdorff (utils.py:201-211): mean over the labels 1..3 of the modified Hausdorff distance between
    the contour points of seg1 == label and seg2 == label; (H, W) device label maps in, float64 device scalar out.  Contour
    extraction, nearest-point search and the means all run on device without a host round trip (inference.py:66-75 calls this per
    sample; the reference goes through .cpu().numpy() and skimage three times per call)."""
    _need_gpu(seg1, seg2)
    H, W = seg1.shape[-2:]
    segs = torch.stack([seg1.reshape(H, W).float(), seg2.reshape(H, W).float()] * 3)          # masks 2p, 2p+1 = pair p
    labels = torch.tensor([1, 1, 2, 2, 3, 3], device=seg1.device, dtype=torch.float32)
    pts, counts = _boundary_sets(segs, labels)
    cap = 2 * H * W
    work = torch.empty(3 * 2 * cap, device=seg1.device, dtype=torch.float32)
    out = torch.empty(4, device=seg1.device, dtype=torch.float64)
    _lib.call("mireg_hausdorff_pairs", pts.data_ptr(), 4 * H * W, counts.data_ptr(), 3, cap, work.data_ptr(), out.data_ptr(), _stream())
    return out[0]
