"""Op-level CPU restatement (torch fp32) of the registration hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it restates (paths relative to /root/reference).  The
restatements are written in closed form with explicit index arithmetic (no
F.grid_sample / F.interpolate) so that they double as the specification of
what each HIP kernel computes per output element.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

F32 = torch.float32


# ----------------------------------------------------------------------------
# bilinear resampling helpers
# ----------------------------------------------------------------------------
def _src_index(out_size: int, in_size: int, align_corners: bool) -> torch.Tensor:
    """Source coordinate of every output index for F.interpolate(bilinear).

    align_corners=True : src = dst * (in-1)/(out-1)   (0 when out == 1)
    align_corners=False: src = max((dst+0.5)*in/out - 0.5, 0)
    """
    dst = torch.arange(out_size, dtype=F32)
    if align_corners:
        scale = (in_size - 1) / (out_size - 1) if out_size > 1 else 0.0
        return dst * F32_scalar(scale)
    scale = in_size / out_size
    return torch.clamp((dst + 0.5) * F32_scalar(scale) - 0.5, min=0.0)


def F32_scalar(v: float) -> float:
    """Round a python double to fp32 (the value torch would use in-kernel)."""
    return float(np.float32(v))


def resize_bilinear(img: torch.Tensor, size: Tuple[int, int], align_corners: bool) -> torch.Tensor:
    """F.interpolate(img, size, mode='bilinear', align_corners=...) restated.

    Used by stn (models.py:258, align_corners=True), the losses
    (loss.py:11,54, align_corners=False) and FlowNetS (FlowNetS/FlowNetS.py:82,
    default align_corners=False; flow values are NOT rescaled).
    """
    B, C, H, W = img.shape
    h, w = size
    sy = _src_index(h, H, align_corners)
    sx = _src_index(w, W, align_corners)
    y0 = sy.floor().long().clamp(max=H - 1)
    x0 = sx.floor().long().clamp(max=W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    x1 = (x0 + 1).clamp(max=W - 1)
    ly = (sy - y0.to(F32)).view(1, 1, h, 1)
    lx = (sx - x0.to(F32)).view(1, 1, 1, w)
    top = img[:, :, y0][:, :, :, x0] * (1 - lx) + img[:, :, y0][:, :, :, x1] * lx
    bot = img[:, :, y1][:, :, :, x0] * (1 - lx) + img[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def _bilinear_zeros(img: torch.Tensor, px: torch.Tensor, py: torch.Tensor) -> torch.Tensor:
    """Sample img (B,C,H,W) at pixel coords px,py (B,h,w), zero outside.

    This is F.grid_sample(mode='bilinear', padding_mode='zeros') after the
    un-normalisation step, restated tap by tap.
    """
    B, C, H, W = img.shape
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    wx1 = px - x0
    wy1 = py - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    x0 = x0.long()
    y0 = y0.long()
    out = torch.zeros(B, C, *px.shape[1:], dtype=img.dtype)
    flat = img.reshape(B, C, H * W)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, C, -1)
            tap = torch.gather(flat, 2, idx).reshape(B, C, *px.shape[1:])
            out = out + tap * (wy * wx * ok.to(img.dtype)).unsqueeze(1)
    return out


# ----------------------------------------------------------------------------
# a6 / a7: pixel grid + spatial transformer
# ----------------------------------------------------------------------------
def generate_grid(B: int, H: int, W: int) -> torch.Tensor:
    """models.py:195-204 -> (B,H,W,2), [...,0] = x index, [...,1] = y index."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=F32), torch.arange(W, dtype=F32), indexing="ij")
    return torch.stack((xs, ys), dim=-1).unsqueeze(0).repeat(B, 1, 1, 1)


def stn_coords(flow: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sampling coordinates of opticalFlowReg.stn (models.py:256-268).

    grid = (flow + pixel) * (2/w, 2/h) - 1, then grid_sample(align_corners=True)
    un-normalises with ((g + 1)/2) * (size - 1): the composite map is
    (x + u) * (w - 1) / w  -- NOT the identity at zero flow.
    The fp32 operation order of the reference is kept.
    """
    B, _, h, w = flow.shape
    g = generate_grid(B, h, w)
    gx = (flow[:, 0] + g[..., 0]) * F32_scalar(2 / w) - 1
    gy = (flow[:, 1] + g[..., 1]) * F32_scalar(2 / h) - 1
    px = ((gx + 1) / 2) * (w - 1)
    py = ((gy + 1) / 2) * (h - 1)
    return px, py


def stn(flow: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
    """opticalFlowReg.stn(flow, frame) (models.py:256-268).

    flow (B,2,h,w) ch0 = x-displacement, ch1 = y; frame (B,C,H,W) is first
    resized to (h,w) with align_corners=True, then bilinearly sampled with
    zero padding at stn_coords(flow).
    """
    h, w = flow.shape[2:]
    frame_r = resize_bilinear(frame, (h, w), align_corners=True)
    px, py = stn_coords(flow)
    return _bilinear_zeros(frame_r, px, py)


# ----------------------------------------------------------------------------
# a9-a12: losses
# ----------------------------------------------------------------------------
def charbonnier(x: torch.Tensor, alpha: float = 0.25, epsilon: float = 1.0e-9) -> torch.Tensor:
    """loss.py:33-35: (x^2 + eps^2)^alpha."""
    return torch.pow(x * x + epsilon ** 2, alpha)


def photometric_loss(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """loss.py:9-14: fixed resized (align_corners=False) to warped's size;
    sum(charbonnier(fixed - warped)) / B."""
    fixed_r = resize_bilinear(fixed, warped.shape[2:], align_corners=False)
    return charbonnier(fixed_r - warped).sum() / fixed.shape[0]


def correlation_loss(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """loss.py:52-64: global NCC over the whole batch tensor with an extra 1/B
    (SURVEY Q6); 1 - corr; corr := 1 when either centred tensor is all-zero."""
    b = warped.shape[0]
    fixed_r = resize_bilinear(fixed, warped.shape[2:], align_corners=False)
    vx = warped - warped.mean()
    vy = fixed_r - fixed_r.mean()
    if not bool(vx.any()) or not bool(vy.any()):
        return torch.tensor(0.0)
    corr = (1.0 / b) * (vx * vy).sum() / (torch.sqrt((vx * vx).sum()) * torch.sqrt((vy * vy).sum()))
    return 1.0 - corr


def smoothness_loss(flow: torch.Tensor) -> torch.Tensor:
    """loss.py:23-30: forward differences against a ZERO-extended copy (the
    last row / column terms are charbonnier(flow - 0)); channel sum / 2; / B."""
    b = flow.shape[0]
    down = torch.zeros_like(flow)
    down[:, :, :-1, :] = flow[:, :, 1:, :]
    right = torch.zeros_like(flow)
    right[:, :, :, :-1] = flow[:, :, :, 1:]
    s = charbonnier(flow - down) + charbonnier(flow - right)
    return (s.sum(dim=1) / 2).sum() / b


def ofe_loss(flows: Sequence[torch.Tensor], warped: Sequence[torch.Tensor], fixed: torch.Tensor,
             lamb_da: float = 0.5, gamma: float = 100.0, zeta: float = 100.0):
    """loss.py:66-84 OFEloss.  weights = 0.05*(i+1) as float64 (so the four
    returned scalars are float64); returns (p, c, s, p+s+c)."""
    n = len(flows)
    wts = torch.from_numpy(0.05 * np.arange(1, n + 1))
    p = c = s = 0
    for i in range(n):
        p = p + wts[i] * photometric_loss(fixed, warped[i])
        c = c + wts[i] * correlation_loss(fixed, warped[i])
        s = s + wts[i] * smoothness_loss(flows[i])
    p = 1 / n * gamma * p
    c = 1 / n * zeta * c
    s = 1 / n * lamb_da * s
    return p, c, s, p + s + c


# 3-D losses (loss.py:16-19, 38-50, 87-94)
def photometric_loss_3d(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    return charbonnier(fixed - warped).sum() / fixed.shape[0]


def correlation_loss_3d(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    b = warped.shape[0]
    vx = warped - warped.mean()
    vy = fixed - fixed.mean()
    if not bool(vx.any()) or not bool(vy.any()):
        return torch.tensor(0.0)
    corr = (1.0 / b) * (vx * vy).sum() / (torch.sqrt((vx * vx).sum()) * torch.sqrt((vy * vy).sum()))
    return 1.0 - corr


def aff_loss(warped: torch.Tensor, fixed: torch.Tensor, lamb_da: float = 1.0, gamma: float = 1.0):
    p = gamma * photometric_loss_3d(fixed, warped)
    c = lamb_da * correlation_loss_3d(fixed, warped)
    return p, c, p + c


# ----------------------------------------------------------------------------
# a14 / BASELINE config "3D FlowNetS on 128^3": dense 3-D flow.  The reference has no such path:
# these are the reference's 2-D formulas written per axis, pinned at op level against torch's own 5-D ops
# (F.interpolate trilinear, F.grid_sample) -- which is why, unlike the 2-D restatements above, they call them.
# ----------------------------------------------------------------------------
def resize_trilinear(x: torch.Tensor, size, align_corners: bool) -> torch.Tensor:
    import torch.nn.functional as F
    return F.interpolate(x, size=tuple(size), mode="trilinear", align_corners=align_corners)


def stn3d(flow: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
    """models.py:256-268 per axis: frame resized (align_corners=True) to the flow's size; grid = (flow + voxel index) *
    (2/w, 2/h, 2/d) - 1; grid_sample(align_corners=True, zeros).  flow channels 0/1/2 = x/y/z displacement."""
    import torch.nn.functional as F
    B, _, d, h, w = flow.shape
    frame_r = resize_trilinear(frame, (d, h, w), True) if tuple(frame.shape[2:]) != (d, h, w) else frame
    zz, yy, xx = torch.meshgrid(torch.arange(d, dtype=F32), torch.arange(h, dtype=F32), torch.arange(w, dtype=F32), indexing="ij")
    idx = torch.stack((xx, yy, zz), dim=-1).unsqueeze(0)                       # (1,d,h,w,3) in (x,y,z) order
    grid = (flow.permute(0, 2, 3, 4, 1) + idx) * torch.tensor([2 / w, 2 / h, 2 / d], dtype=F32) - 1
    return F.grid_sample(frame_r, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def smoothness_loss_3d(flow: torch.Tensor) -> torch.Tensor:
    """loss.py:21-29 with three axes and three channels: forward differences against a zero-extended copy, channel mean, / B."""
    b = flow.shape[0]
    s = 0
    for ax in (2, 3, 4):
        sh = torch.zeros_like(flow)
        src = [slice(None)] * 5
        dst = [slice(None)] * 5
        src[ax], dst[ax] = slice(1, None), slice(0, -1)
        sh[tuple(dst)] = flow[tuple(src)]
        s = s + charbonnier(flow - sh)
    return (s.sum(dim=1) / 3).sum() / b


def ofe_loss_3d(flows: Sequence[torch.Tensor], warped: Sequence[torch.Tensor], fixed: torch.Tensor,
                lamb_da: float = 0.5, gamma: float = 100.0, zeta: float = 100.0):
    """loss.py:66-84 over volumes (fixed resized with trilinear align_corners=False, as loss.py:11,54 do in 2-D)."""
    n = len(flows)
    wts = torch.from_numpy(0.05 * np.arange(1, n + 1))
    p = c = s = 0
    for i in range(n):
        fr = fixed if tuple(fixed.shape[2:]) == tuple(warped[i].shape[2:]) else resize_trilinear(fixed, warped[i].shape[2:], False)
        p = p + wts[i] * photometric_loss_3d(fr, warped[i])
        c = c + wts[i] * correlation_loss_3d(fr, warped[i])
        s = s + wts[i] * smoothness_loss_3d(flows[i])
    p = 1 / n * gamma * p
    c = 1 / n * zeta * c
    s = 1 / n * lamb_da * s
    return p, c, s, p + s + c


# ----------------------------------------------------------------------------
# a13: segmentation warp + Dice
# ----------------------------------------------------------------------------
def seg_round(warped_seg: torch.Tensor) -> torch.Tensor:
    """models.py:286: clip(rint(x), 0, 3) (numpy rint = round-half-even)."""
    return torch.clamp(torch.round(warped_seg), 0, 3)


def dice_average(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
    """utils.py:72-91: mean over labels 1..3 of 2|A&B| / (|A|+|B|)."""
    vals = []
    for lab in (1, 2, 3):
        a = (y_true == lab).to(F32).flatten()
        p = (y_pred == lab).to(F32).flatten()
        vals.append(((2.0 * (a * p).sum()) / (a.sum() + p.sum())).item())
    return float(np.mean(np.asarray(vals, dtype=np.float32)))


def grid_generator() -> torch.Tensor:
    """utils.py:15-23: 256x256 image with lines at rows/cols 7, 23, ..., 247."""
    g = torch.zeros(256, 256)
    idx = torch.arange(7, 255, 16)
    g[idx, :] = 1.0
    g[:, idx] = 1.0
    return g


# ----------------------------------------------------------------------------
# a3: cost volume (third-party op; PARITY UNPINNED, published definition)
# ----------------------------------------------------------------------------
def correlation(f1: torch.Tensor, f2: torch.Tensor, pad_size: int, kernel_size: int,
                max_displacement: int, stride1: int, stride2: int, corr_multiply: int = 1) -> torch.Tensor:
    """NVIDIA flownet2 Correlation as used at flownet2/networks/FlowNetC.py:31,88
    and PWC/models/PWCNet.py:69,200-259 (kernel_size=1, stride1=1, pad == md).

    out[b, (dy+R)*D + (dx+R), y, x] = (1/C) * sum_c f1[b,c,y,x] * f2[b,c,y+s2*dy,x+s2*dx]
    with R = md // s2, D = 2R+1, zero outside f2.  Same values as
    spatial_correlation_sample(k=1, patch=D, dilation_patch=s2)/C restated at
    FlowNetS/util.py:58-72.
    """
    assert kernel_size == 1 and stride1 == 1 and pad_size == max_displacement and corr_multiply == 1
    B, C, H, W = f1.shape
    R = max_displacement // stride2
    D = 2 * R + 1
    P = R * stride2
    f2p = torch.zeros(B, C, H + 2 * P, W + 2 * P, dtype=f1.dtype)
    f2p[:, :, P:P + H, P:P + W] = f2
    out = torch.empty(B, D * D, H, W, dtype=f1.dtype)
    for iy in range(D):
        for ix in range(D):
            oy = P + (iy - R) * stride2
            ox = P + (ix - R) * stride2
            out[:, iy * D + ix] = (f1 * f2p[:, :, oy:oy + H, ox:ox + W]).sum(dim=1) / C
    return out


# ----------------------------------------------------------------------------
# a5: PWC feature warp
# ----------------------------------------------------------------------------
def pwc_warp(x: torch.Tensor, flo: torch.Tensor) -> torch.Tensor:
    """PWCDCNet.warp (PWC/models/PWCNet.py:143-179).

    vgrid normalised with (W-1) but grid_sample runs with its default
    align_corners=False, so the sample coordinate is
    ((2*(x+u)/(W-1) - 1 + 1) * W - 1) / 2; the validity mask is the same
    sampling applied to ones, thresholded at 0.9999.
    """
    B, C, H, W = x.shape
    g = generate_grid(B, H, W)
    gx = 2.0 * (g[..., 0] + flo[:, 0]) / max(W - 1, 1) - 1.0
    gy = 2.0 * (g[..., 1] + flo[:, 1]) / max(H - 1, 1) - 1.0
    px = ((gx + 1) * W - 1) / 2
    py = ((gy + 1) * H - 1) / 2
    out = _bilinear_zeros(x, px, py)
    mask = _bilinear_zeros(torch.ones(B, 1, H, W, dtype=x.dtype), px, py)
    mask = (mask >= 0.9999).to(x.dtype)
    return out * mask


# ----------------------------------------------------------------------------
# the FlowNet2 stack's external custom layers (flownet2/models.py:40-88,136-180).
# NVIDIA/flownet2-pytorch resample2d_package / channelnorm_package are absent from the reference tree and unpinned:
# restated from their published definitions -- PARITY UNPINNED (no reference-run fixture exists for these two).
# ----------------------------------------------------------------------------
def resample2d(src: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """out[b,c,y,x] = bilinear(src[b,c], x + flow_x, y + flow_y); the four tap indices are clamped to the border."""
    B, C, H, W = src.shape
    yy, xx = torch.meshgrid(torch.arange(H, dtype=F32), torch.arange(W, dtype=F32), indexing="ij")
    xf, yf = xx + flow[:, 0], yy + flow[:, 1]
    x0, y0 = torch.floor(xf), torch.floor(yf)
    a, b = (xf - x0).unsqueeze(1), (yf - y0).unsqueeze(1)
    xl, xr = x0.long().clamp(0, W - 1), (x0.long() + 1).clamp(0, W - 1)
    yt, yb = y0.long().clamp(0, H - 1), (y0.long() + 1).clamp(0, H - 1)
    flat = src.reshape(B, C, -1)

    def at(yi, xi):
        return torch.gather(flat, 2, (yi * W + xi).reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, H, W)

    return (1 - a) * (1 - b) * at(yt, xl) + a * (1 - b) * at(yt, xr) + (1 - a) * b * at(yb, xl) + a * b * at(yb, xr)


def channelnorm(x: torch.Tensor) -> torch.Tensor:
    """sqrt(sum_c x^2) keeping a singleton channel; the layer's backward divides by (norm + 1e-9)."""
    return torch.sqrt((x * x).sum(dim=1, keepdim=True))


# ----------------------------------------------------------------------------
# per-sample evaluation metrics (utils.py:41-59)
# ----------------------------------------------------------------------------
def mse(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """utils.py:41-42."""
    return torch.mean(torch.pow(warped - fixed, 2))


def psnr(fixed: torch.Tensor, warped: torch.Tensor):
    """utils.py:45-49."""
    m = mse(fixed, warped)
    if m < 1.0e-10:
        return torch.tensor(100.0)
    return 10 * torch.log10(1.0 ** 2 / m)


def pearson(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """utils.py:58-59 CORR = torchmetrics.functional.pearson_corrcoef (dependency absent here, version unpinned): the
    published definition cov / (sigma_x sigma_y), in float64."""
    x, y = fixed.reshape(-1).double(), warped.reshape(-1).double()
    vx, vy = x - x.mean(), y - y.mean()
    return (vx * vy).sum() / torch.sqrt((vx * vx).sum() * (vy * vy).sum())


def ssim(im1: torch.Tensor, im2: torch.Tensor, data_range: float = 1.0, win_size: int = 7) -> float:
    """inference.py:70-71: skimage.metrics.structural_similarity(im1, im2, data_range=1.0) with its defaults (skimage absent here
    and unpinned -- PARITY UNPINNED): the published algorithm on the same scipy.ndimage.uniform_filter skimage calls -- 7x7 uniform
    window, sample covariance, K1 = 0.01, K2 = 0.03, float32 maps, float64 mean of the map cropped by 3 pixels per side."""
    from scipy.ndimage import uniform_filter
    x, y = im1.numpy().astype(np.float32), im2.numpy().astype(np.float32)
    npix = win_size ** 2
    cov_norm = npix / (npix - 1)
    ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
    uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean(dtype=np.float64))


def mutual_info(fixed: torch.Tensor, warped: torch.Tensor, scale: float = 1500.0) -> float:
    """utils.py:52-55: sklearn.metrics.mutual_info_score(round(fixed*1500), round(warped*1500)); restated from its published
    definition sum_ij n_ij/N * log(N n_ij / (a_i b_j)) (natural log); tests also call sklearn's own function when importable."""
    a = torch.round(fixed * scale).int().reshape(-1).numpy()
    b = torch.round(warped * scale).int().reshape(-1).numpy()
    n = a.size
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    cont = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
    np.add.at(cont, (ai, bi), 1)
    ra, cb = cont.sum(1), cont.sum(0)
    i, j = np.nonzero(cont)
    nij = cont[i, j].astype(np.float64)
    return float(np.sum(nij / n * (np.log(nij) + np.log(n) - np.log(ra[i]) - np.log(cb[j]))))


# ----------------------------------------------------------------------------
# a14: 3-D affine grid + trilinear sampling (models.py:187-188)
# ----------------------------------------------------------------------------
def affine_grid_sample_3d(vol: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """F.grid_sample(vol, F.affine_grid(theta, vol.size())) with both defaults
    (align_corners=False, trilinear, zeros) for vol (B,C,D,H,W), theta (B,3,4)."""
    B, C, D, H, W = vol.shape

    def lin(n):
        return (torch.arange(n, dtype=F32) * 2 + 1) / n - 1

    zz, yy, xx = torch.meshgrid(lin(D), lin(H), lin(W), indexing="ij")
    base = torch.stack((xx, yy, zz, torch.ones_like(xx)), dim=-1).reshape(1, -1, 4)
    g = torch.bmm(base.expand(B, -1, -1), theta.transpose(1, 2)).reshape(B, D, H, W, 3)
    px = ((g[..., 0] + 1) * W - 1) / 2
    py = ((g[..., 1] + 1) * H - 1) / 2
    pz = ((g[..., 2] + 1) * D - 1) / 2
    x0, y0, z0 = torch.floor(px), torch.floor(py), torch.floor(pz)
    out = torch.zeros(B, C, D, H, W, dtype=vol.dtype)
    flat = vol.reshape(B, C, -1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                wgt = ((px - x0) if dx else (1 - (px - x0))) * ((py - y0) if dy else (1 - (py - y0))) \
                    * ((pz - z0) if dz else (1 - (pz - z0)))
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                idx = ((zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1))
                tap = torch.gather(flat, 2, idx.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, D, H, W)
                out = out + tap * (wgt * ok.to(vol.dtype)).unsqueeze(1)
    return out


# ----------------------------------------------------------------------------
# a15: Adam step exactly as train.py:129 configures it
# ----------------------------------------------------------------------------
def adam_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg: List[torch.Tensor],
              exp_avg_sq: List[torch.Tensor], step: int, lr: float = 1e-4,
              betas=(0.9, 0.999), eps: float = 1e-4) -> None:
    """torch.optim.Adam(lr=1e-4, betas=(.9,.999), eps=lrMin=1e-4) (train.py:129,
    SURVEY Q7), no weight decay, no amsgrad; in-place on params / moments."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
