"""The FlowNet2 stack's glue layers as differentiable modules with the reference's class names (flownet2/models.py:10-11,
40-88,136-180): `Resample2d`, `ChannelNorm`, and nearest / bilinear x4 `Upsample`.  The two custom layers are EXTERNAL to the
reference tree (NVIDIA/flownet2-pytorch, unpinned): published definitions, parity unpinned.  First piece of
SURVEY section 8(f) rank 1; FlowNetSD / FlowNetFusion and the FlowNet2 chaining are not built yet.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .ops import _need_gpu, _stream, resize_bilinear


class _Resample2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, flow):
        _need_gpu(src, flow)
        B, C, H, W = src.shape
        if tuple(flow.shape) != (B, 2, H, W):
            raise RuntimeError(f"Resample2d: input {tuple(src.shape)} / flow {tuple(flow.shape)} mismatch")
        src, flow = src.contiguous(), flow.contiguous()
        out = torch.empty_like(src)
        _lib.call("mireg_resample2d_fwd", src.data_ptr(), flow.data_ptr(), out.data_ptr(), B, C, H, W, _stream())
        ctx.save_for_backward(src, flow)
        return out

    @staticmethod
    def backward(ctx, g):
        src, flow = ctx.saved_tensors
        B, C, H, W = src.shape
        g = g.contiguous()
        gsrc = torch.zeros_like(src) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if gsrc is None and gflow is None:
            return None, None
        _lib.call("mireg_resample2d_bwd", src.data_ptr(), flow.data_ptr(), g.data_ptr(),
                  gsrc.data_ptr() if gsrc is not None else None, gflow.data_ptr() if gflow is not None else None, B, C, H, W, _stream())
        return gsrc, gflow


class Resample2d(nn.Module):
    """Drop-in for flownet2 `Resample2d()` (kernel_size 1, bilinear): forward(input1, input2=flow)."""

    def forward(self, input1: torch.Tensor, input2: torch.Tensor) -> torch.Tensor:
        return _Resample2dFn.apply(input1, input2)


class _ChannelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous()
        B, C = x.shape[:2]
        npix = x.numel() // (B * C)
        out = torch.empty(B, 1, *x.shape[2:], device=x.device, dtype=torch.float32)
        _lib.call("mireg_channelnorm_fwd", x.data_ptr(), out.data_ptr(), B, C, npix, _stream())
        ctx.save_for_backward(x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, out = ctx.saved_tensors
        B, C = x.shape[:2]
        gin = torch.empty_like(x)
        _lib.call("mireg_channelnorm_bwd", x.data_ptr(), out.data_ptr(), g.contiguous().data_ptr(), gin.data_ptr(), B, C,
                  x.numel() // (B * C), _stream())
        return gin


class ChannelNorm(nn.Module):
    """Drop-in for flownet2 `ChannelNorm()` (norm_deg 2): (B, C, H, W) -> (B, 1, H, W)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _ChannelNormFn.apply(x)


class _NearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _need_gpu(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, C, H * k, W * k, device=x.device, dtype=torch.float32)
        _lib.call("mireg_upsample_nearest", x.data_ptr(), out.data_ptr(), B * C, H, W, k, 0, _stream())
        ctx.dims = (B, C, H, W, k)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, k = ctx.dims
        gin = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32)
        _lib.call("mireg_upsample_nearest", g.contiguous().data_ptr(), gin.data_ptr(), B * C, H, W, k, 1, _stream())
        return gin, None


class Upsample(nn.Module):
    """nn.Upsample(scale_factor=k, mode='bilinear' | 'nearest') as FlowNet2 uses it (flownet2/models.py:44,56,71,72)."""

    def __init__(self, scale_factor: int = 4, mode: str = "bilinear"):
        super().__init__()
        if mode not in ("bilinear", "nearest"):
            raise ValueError(f"Upsample: unsupported mode {mode!r}")
        self.k, self.mode = int(scale_factor), mode

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.mode == "nearest":
            return _NearestFn.apply(x, self.k)
        return resize_bilinear(x, (x.shape[2] * self.k, x.shape[3] * self.k), False)
