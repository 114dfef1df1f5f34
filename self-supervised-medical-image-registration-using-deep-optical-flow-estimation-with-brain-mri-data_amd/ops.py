"""Python op layer over the C ABI: thin torch.autograd.Function wrappers.

PyTorch is plumbing here (device memory, streams, autograd graph); every device
computation is a hand-written gfx950 kernel reached through include/mireg.h.
All ops raise if their inputs are not on a HIP device: there is no CPU path.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import _lib

F32 = torch.float32
SLOTS = 32          # MIREG_SUM_SLOTS: replicas of every moment row (include/mireg.h)


_COUNTS = {}


def pixel_counts(values, dev: torch.device) -> torch.Tensor:
    """int64 device vector of per-scale element counts, uploaded once per (device, values): a pageable host-to-device copy
    inside a hipGraph capture records the host pointer, which is gone by the time the graph replays."""
    key = (dev.type, dev.index, tuple(int(v) for v in values))
    t = _COUNTS.get(key)
    if t is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mireg: a new pixel-count table would be uploaded inside a hipGraph capture; run the step once eagerly first")
        t = _COUNTS[key] = torch.tensor(key[2], dtype=torch.int64, device=dev)
    return t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError(f"mireg ops run on the MI355X only (tensor is on {t.device}); there is no CPU fallback")
        if t.dtype != F32:
            raise RuntimeError(f"mireg op expects float32 tensors, got {t.dtype}")


def _flow_view(flow: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
    """Return (tensor, sb, sc, sp): element (b,c,y,x) lives at b*sb + c*sc + (y*w+x)*sp.
    NCHW-contiguous flows and the conv engine's NHWC (channels_last) flows both qualify as-is."""
    B, C, h, w = flow.shape
    sb, sc, sy, sx = flow.stride()
    sp = sx if w > 1 else (sy if h > 1 else 1)
    ok = sp > 0 and (h == 1 or w == 1 or sy == w * sx)
    if not ok:
        flow = flow.contiguous()
        sb, sc, sp = C * h * w, h * w, 1
    return flow, sb, sc, sp


def resize_bilinear(x: torch.Tensor, size: Tuple[int, int], align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size, mode='bilinear', align_corners=...) on the HIP kernel (autograd-aware)."""
    return _ResizeFn.apply(x, int(size[0]), int(size[1]), bool(align_corners))


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, w, align):
        _need_gpu(x)
        x = x.contiguous()
        N, C, H, W = x.shape
        out = torch.empty(N, C, h, w, device=x.device, dtype=F32)
        _lib.call("mireg_resize_bilinear_fwd", x.data_ptr(), out.data_ptr(), N, C, H, W, h, w,
                  C * H * W, H * W, 1, C * h * w, h * w, 1, int(align), _stream())
        ctx.shape = (N, C, H, W, h, w, align)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W, h, w, align = ctx.shape
        g = g.contiguous()
        gin = torch.empty(N, C, H, W, device=g.device, dtype=F32)
        _lib.call("mireg_resize_bilinear_bwd", g.data_ptr(), gin.data_ptr(), N, C, H, W, h, w,
                  C * H * W, H * W, 1, C * h * w, h * w, 1, int(align), 0.0, _stream())
        return gin, None, None, None


class _StnFn(torch.autograd.Function):
    """opticalFlowReg.stn (reference models.py:256-268): gradient flows to `flow` only."""

    @staticmethod
    def forward(ctx, flow, frame_r):
        _need_gpu(flow, frame_r)
        B, two, h, w = flow.shape
        if two != 2 or frame_r.shape[0] != B or tuple(frame_r.shape[2:]) != (h, w):
            raise RuntimeError(f"stn: flow {tuple(flow.shape)} / frame {tuple(frame_r.shape)} mismatch")
        flow, sb, sc, sp = _flow_view(flow)
        frame_r = frame_r.contiguous()
        C = frame_r.shape[1]
        out = torch.empty(B, C, h, w, device=flow.device, dtype=F32)
        _lib.call("mireg_stn_warp_fwd", flow.data_ptr(), sb, sc, sp, frame_r.data_ptr(), None, out.data_ptr(), None,
                  B, C, h, w, _stream())
        ctx.save_for_backward(flow, frame_r)
        return out

    @staticmethod
    def backward(ctx, g):
        flow, frame_r = ctx.saved_tensors
        B, _, h, w = flow.shape
        _, sb, sc, sp = _flow_view(flow)
        g = g.contiguous()
        gflow = torch.empty(B, 2, h, w, device=g.device, dtype=F32)
        _lib.call("mireg_stn_warp_bwd", flow.data_ptr(), sb, sc, sp, frame_r.data_ptr(), g.data_ptr(),
                  gflow.data_ptr(), 2 * h * w, h * w, 1, 0.0, B, frame_r.shape[1], h, w, _stream())
        return gflow, None


def stn(flow: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
    """Warp `frame` (B,C,H,W) with `flow` (B,2,h,w): resize to (h,w) (align_corners=True) then sample at
    (x+u)(w-1)/w with zero padding -- the exact coordinate convention of the reference."""
    h, w = flow.shape[2:]
    frame = frame.detach()
    if tuple(frame.shape[2:]) != (h, w):
        frame = resize_bilinear(frame, (h, w), True)
    return _StnFn.apply(flow, frame)


class _OFELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fixed, lamb_da, gamma, zeta, n, *tensors):
        flows, warped = tensors[:n], tensors[n:]
        _need_gpu(fixed, *flows, *warped)
        dev = fixed.device
        B = fixed.shape[0]
        sums = torch.zeros(n, SLOTS, 8, device=dev, dtype=torch.float64)
        npix = pixel_counts([w.numel() for w in warped], dev)
        fixed_rs, wcs, fviews = [], [], []
        st = _stream()
        for i in range(n):
            wi = warped[i].contiguous()
            h, w = wi.shape[2:]
            fr = fixed.detach() if tuple(fixed.shape[2:]) == (h, w) else resize_bilinear(fixed.detach(), (h, w), False)
            fr = fr.contiguous()
            _lib.call("mireg_loss_partials", wi.data_ptr(), fr.data_ptr(), sums[i].data_ptr(), wi.numel(), st)
            fl, sb, sc, sp = _flow_view(flows[i])
            _lib.call("mireg_smoothness_fwd", fl.data_ptr(), sb, sc, sp, sums[i, 0, 6:].data_ptr(), B, fl.shape[2],
                      fl.shape[3], st)
            fixed_rs.append(fr)
            wcs.append(wi)
            fviews.append(fl)
        out = torch.empty(4, device=dev, dtype=torch.float64)
        _lib.call("mireg_ofe_finalize", sums.data_ptr(), npix.data_ptr(), n, B, float(lamb_da), float(gamma),
                  float(zeta), out.data_ptr(), st)
        ctx.n, ctx.B, ctx.hyper = n, B, (float(lamb_da), float(gamma), float(zeta))
        ctx.save_for_backward(sums, npix, *fviews, *wcs, *fixed_rs)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, gp, gc, gs, gt):
        n, B = ctx.n, ctx.B
        saved = ctx.saved_tensors
        sums, npix = saved[0], saved[1]
        flows, warped, fixed_rs = saved[2:2 + n], saved[2 + n:2 + 2 * n], saved[2 + 2 * n:2 + 3 * n]
        dev = sums.device
        zero = torch.zeros((), device=dev, dtype=torch.float64)
        g4 = torch.stack([zero if g is None else g.to(torch.float64) for g in (gp, gc, gs, gt)]).contiguous()
        coef = torch.empty(n, 8, device=dev, dtype=F32)
        st = _stream()
        lamb_da, gamma, zeta = ctx.hyper
        _lib.call("mireg_ofe_bwd_coef", sums.data_ptr(), npix.data_ptr(), n, B, lamb_da, gamma, zeta, g4.data_ptr(),
                  coef.data_ptr(), st)
        gflows, gwarped = [], []
        for i in range(n):
            gw = torch.empty_like(warped[i])
            _lib.call("mireg_loss_bwd", warped[i].data_ptr(), fixed_rs[i].data_ptr(), coef[i].data_ptr(),
                      gw.data_ptr(), gw.numel(), st)
            fl, sb, sc, sp = _flow_view(flows[i])
            _, _, h, w = fl.shape
            gf = torch.empty(B, 2, h, w, device=dev, dtype=F32)
            _lib.call("mireg_smoothness_bwd", fl.data_ptr(), sb, sc, sp, coef[i].data_ptr(), gf.data_ptr(),
                      2 * h * w, h * w, 1, 0.0, B, h, w, st)
            gflows.append(gf)
            gwarped.append(gw)
        return (None, None, None, None, None, *gflows, *gwarped)


def OFEloss(flow: Sequence[torch.Tensor], warped: Sequence[torch.Tensor], fixed: torch.Tensor,
            lamb_da: float = 0.5, gamma: float = 100.0, zeta: float = 100.0):
    """Drop-in for reference loss.OFEloss (loss.py:66-84): returns (p, c, s, total) float64 scalars.
    No host synchronisation happens inside (the reference has 2 per scale)."""
    n = len(flow)
    if n != len(warped) or n < 1:
        raise RuntimeError("OFEloss: flow and warped must be equally long, non-empty sequences")
    return _OFELossFn.apply(fixed, lamb_da, gamma, zeta, n, *flow, *warped)


def _single_scale_terms(fixed: torch.Tensor, warped: torch.Tensor, flow: torch.Tensor):
    """(photometric, 1 - ncc, smoothness) of ONE scale through the OFEloss kernels: weight 0.05 * (1/n = 1) * 20 = 1."""
    return _OFELossFn.apply(fixed, 20.0, 20.0, 20.0, 1, flow, warped)[:3]


def photometric_loss(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference loss.photometric_loss (loss.py:9-14); differentiable wrt `warped`."""
    B, _, h, w = warped.shape
    return _single_scale_terms(fixed, warped, torch.zeros(B, 2, h, w, device=warped.device, dtype=F32))[0]


def correlation_loss(fixed: torch.Tensor, warped: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference loss.correlation_loss (loss.py:52-64), incl. the degenerate-input guard and the 1/B factor."""
    B, _, h, w = warped.shape
    return _single_scale_terms(fixed, warped, torch.zeros(B, 2, h, w, device=warped.device, dtype=F32))[1]


def smoothness_loss(flow: torch.Tensor) -> torch.Tensor:
    """Drop-in for reference loss.smoothness_loss (loss.py:23-30); differentiable wrt `flow`."""
    B, _, h, w = flow.shape
    z = torch.zeros(B, 1, h, w, device=flow.device, dtype=F32)
    return _single_scale_terms(z, z, flow)[2]


def seg_round(x: torch.Tensor) -> torch.Tensor:
    """clip(rint(x), 0, 3) on device (reference does a CPU numpy round trip, models.py:286)."""
    _need_gpu(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("mireg_seg_round", x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    return out


def dice_batch(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """Per-sample mean Dice over labels 1..3 for (B, ...) label maps -> (B,) tensor, one launch group."""
    _need_gpu(y_true, y_pred)
    if y_true.shape[0] != y_pred.shape[0] or y_true.numel() != y_pred.numel():
        raise RuntimeError(f"dice_batch: label maps differ in size: {tuple(y_true.shape)} vs {tuple(y_pred.shape)}")
    B = y_true.shape[0]
    yt = y_true.contiguous().view(B, -1)
    yp = y_pred.contiguous().view(B, -1)
    counts = torch.empty(B, 9, device=yt.device, dtype=F32)
    dice = torch.empty(B, device=yt.device, dtype=F32)
    _lib.call("mireg_dice", yt.data_ptr(), yp.data_ptr(), counts.data_ptr(), dice.data_ptr(), B, yt.shape[1],
              _stream())
    return dice


def dice_average(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
    """Drop-in for reference utils.dice_average (utils.py:87-91) on ONE sample (any shape)."""
    return float(dice_batch(y_true.reshape(1, -1), y_pred.reshape(1, -1)).item())
