"""Volume (3-D) registration tail: trilinear resize, dense 3-D warp, 3-D OFEloss -- SURVEY section 8 row a14, BASELINE config
"3D FlowNetS on 128^3 volumes".  The reference has no dense 3-D flow path; these ops generalise its 2-D conventions axis by
axis (models.py:256-268 `stn`, loss.py:9-84) and are pinned at op level against torch on the CPU (tests/test_volume_gpu.py).
Flow channels 0/1/2 displace along x/y/z (W/H/D).  Everything runs on the HIP kernels of csrc/volume_ops.hip; no fallback.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import _lib
from .ops import SLOTS, _need_gpu, _stream, pixel_counts

F32 = torch.float32


def _flow_strides(flow: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
    """(tensor, sb, sc, sp): element (b,c,z,y,x) at b*sb + c*sc + ((z*h + y)*w + x)*sp -- planar and channel-last both qualify."""
    B, C, d, h, w = flow.shape
    sb, sc, sz, sy, sx = flow.stride()
    if sx > 0 and sy == w * sx and sz == h * w * sx:
        return flow, sb, sc, sx
    flow = flow.contiguous()
    return flow, C * d * h * w, d * h * w, 1


class _Resize3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align):
        _need_gpu(x)
        x, sn, sc, sp = _flow_strides(x)
        N, C, D, H, W = x.shape
        d, h, w = size
        out = torch.empty(N, C, d, h, w, device=x.device, dtype=F32)
        _lib.call("mireg_resize_trilinear_fwd", x.data_ptr(), sn, sc, sp, out.data_ptr(), N, C, D, H, W, d, h, w, int(align), _stream())
        ctx.shape = (N, C, D, H, W, d, h, w, align)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, D, H, W, d, h, w, align = ctx.shape
        g = g.contiguous()
        gin = torch.empty(N, C, D, H, W, device=g.device, dtype=F32)
        ws = torch.empty(N * C * d * (h * W + H * W), device=g.device, dtype=F32)       # the two intermediate planes of the separable adjoint
        _lib.call("mireg_resize_trilinear_bwd_sep", g.data_ptr(), gin.data_ptr(), C * D * H * W, D * H * W, 1, N, C, D, H, W, d, h, w,
                  int(align), 0.0, ws.data_ptr(), ws.numel(), _stream())
        return gin, None, None


def resize_trilinear(x: torch.Tensor, size: Sequence[int], align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size, mode='trilinear', align_corners=...) on the HIP kernel (autograd-aware)."""
    return _Resize3dFn.apply(x, tuple(int(s) for s in size), bool(align_corners))


class _Stn3dFn(torch.autograd.Function):
    """Dense 3-D warp; gradient flows to `flow` only (as in opticalFlowReg.stn, reference models.py:256-268)."""

    @staticmethod
    def forward(ctx, flow, frame_r):
        _need_gpu(flow, frame_r)
        B, three, d, h, w = flow.shape
        if three != 3 or frame_r.shape[0] != B or tuple(frame_r.shape[2:]) != (d, h, w):
            raise RuntimeError(f"stn3d: flow {tuple(flow.shape)} / frame {tuple(frame_r.shape)} mismatch")
        flow, sb, sc, sp = _flow_strides(flow)
        frame_r = frame_r.contiguous()
        C = frame_r.shape[1]
        out = torch.empty(B, C, d, h, w, device=flow.device, dtype=F32)
        _lib.call("mireg_stn3d_fwd", flow.data_ptr(), sb, sc, sp, frame_r.data_ptr(), out.data_ptr(), B, C, d, h, w, _stream())
        ctx.save_for_backward(flow, frame_r)
        return out

    @staticmethod
    def backward(ctx, g):
        flow, frame_r = ctx.saved_tensors
        B, _, d, h, w = flow.shape
        _, sb, sc, sp = _flow_strides(flow)
        g = g.contiguous()
        gflow = torch.empty(B, 3, d, h, w, device=g.device, dtype=F32)
        _lib.call("mireg_stn3d_bwd", flow.data_ptr(), sb, sc, sp, frame_r.data_ptr(), g.data_ptr(), gflow.data_ptr(), 0.0,
                  B, frame_r.shape[1], d, h, w, _stream())
        return gflow, None


def stn3d(flow: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
    """Warp `frame` (B,C,D,H,W) with `flow` (B,3,d,h,w): trilinear resize to (d,h,w) with align_corners=True, then sample at
    (i + flow_i)(n_i - 1)/n_i per axis with zero padding -- the reference's 2-D coordinate convention per axis."""
    size = tuple(flow.shape[2:])
    frame = frame.detach()
    if tuple(frame.shape[2:]) != size:
        frame = resize_trilinear(frame, size, True)
    return _Stn3dFn.apply(flow, frame)


class _OFELoss3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fixed, lamb_da, gamma, zeta, n, *tensors):
        flows, warped = tensors[:n], tensors[n:]
        _need_gpu(fixed, *flows, *warped)
        dev, B, st = fixed.device, fixed.shape[0], _stream()
        sums = torch.zeros(n, SLOTS, 8, device=dev, dtype=torch.float64)
        npix = pixel_counts([w.numel() for w in warped], dev)
        fixed_rs, wcs, fviews = [], [], []
        for i in range(n):
            wi = warped[i].contiguous()
            size = tuple(wi.shape[2:])
            fr = fixed.detach() if tuple(fixed.shape[2:]) == size else resize_trilinear(fixed.detach(), size, False)
            fr = fr.contiguous()
            _lib.call("mireg_loss_partials", wi.data_ptr(), fr.data_ptr(), sums[i].data_ptr(), wi.numel(), st)
            fl, sb, sc, sp = _flow_strides(flows[i])
            _lib.call("mireg_smoothness3d_fwd", fl.data_ptr(), sb, sc, sp, sums[i, 0, 6:].data_ptr(), B, *fl.shape[2:], st)
            fixed_rs.append(fr)
            wcs.append(wi)
            fviews.append(fl)
        out = torch.empty(4, device=dev, dtype=torch.float64)
        _lib.call("mireg_ofe_finalize", sums.data_ptr(), npix.data_ptr(), n, B, float(lamb_da), float(gamma), float(zeta),
                  out.data_ptr(), st)
        ctx.n, ctx.B, ctx.hyper = n, B, (float(lamb_da), float(gamma), float(zeta))
        ctx.save_for_backward(sums, npix, *fviews, *wcs, *fixed_rs)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, gp, gc, gs, gt):
        n, B = ctx.n, ctx.B
        saved = ctx.saved_tensors
        sums, npix = saved[0], saved[1]
        flows, warped, fixed_rs = saved[2:2 + n], saved[2 + n:2 + 2 * n], saved[2 + 2 * n:2 + 3 * n]
        dev, st = sums.device, _stream()
        zero = torch.zeros((), device=dev, dtype=torch.float64)
        g4 = torch.stack([zero if g is None else g.to(torch.float64) for g in (gp, gc, gs, gt)]).contiguous()
        coef = torch.empty(n, 8, device=dev, dtype=F32)
        lamb_da, gamma, zeta = ctx.hyper
        _lib.call("mireg_ofe_bwd_coef", sums.data_ptr(), npix.data_ptr(), n, B, lamb_da, gamma, zeta, g4.data_ptr(), coef.data_ptr(), st)
        gflows, gwarped = [], []
        for i in range(n):
            gw = torch.empty_like(warped[i])
            _lib.call("mireg_loss_bwd", warped[i].data_ptr(), fixed_rs[i].data_ptr(), coef[i].data_ptr(), gw.data_ptr(), gw.numel(), st)
            fl, sb, sc, sp = _flow_strides(flows[i])
            gf = torch.empty(B, 3, *fl.shape[2:], device=dev, dtype=F32)
            _lib.call("mireg_smoothness3d_bwd", fl.data_ptr(), sb, sc, sp, coef[i].data_ptr(), gf.data_ptr(), 0.0, B, *fl.shape[2:], st)
            gflows.append(gf)
            gwarped.append(gw)
        return (None, None, None, None, None, *gflows, *gwarped)


def OFEloss3d(flow: Sequence[torch.Tensor], warped: Sequence[torch.Tensor], fixed: torch.Tensor,
              lamb_da: float = 0.5, gamma: float = 100.0, zeta: float = 100.0):
    """loss.OFEloss (loss.py:66-84) over volumes: photometric_loss_3d / correlation_loss_3d per scale (fixed resized with
    trilinear align_corners=False, as loss.py:11,54 do in 2-D) + the three-axis smoothness; returns (p, c, s, total) float64."""
    n = len(flow)
    if n != len(warped) or n < 1:
        raise RuntimeError("OFEloss3d: flow and warped must be equally long, non-empty sequences")
    return _OFELoss3dFn.apply(fixed, lamb_da, gamma, zeta, n, *flow, *warped)


def smoothness_loss_3d(flow: torch.Tensor) -> torch.Tensor:
    """sum_c sum_axes charbonnier(flow - shifted) / 3 / B (loss.py:21-29 with three flow channels and three axes)."""
    _need_gpu(flow)
    fl, sb, sc, sp = _flow_strides(flow)
    B = fl.shape[0]
    s = torch.zeros(SLOTS, 8, device=fl.device, dtype=torch.float64)
    _lib.call("mireg_smoothness3d_fwd", fl.data_ptr(), sb, sc, sp, s.data_ptr(), B, *fl.shape[2:], _stream())
    return s[:, 0].sum() / 2.0 / B
