"""Cost-volume and PWC warp ops on the HIP kernels (csrc/correlation.hip).

`Correlation` is the drop-in for the external `correlation_package.Correlation` the reference imports
(flownet2/networks/FlowNetC.py:8,31; PWC/models/PWCNet.py:13,69): same constructor arguments, NCHW fp32
in / out.  Only the parameterisation the reference uses is supported (kernel_size=1, stride1=1,
pad_size == max_displacement, corr_multiply=1); anything else raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .engine import DT_BF16, DT_F32, PROFILER, View, _stream, rup


def _esz(code: int) -> int:
    return 2 if code == DT_BF16 else 4


def correlation_views(f1: View, f2: View, out: View, c_norm: int, md: int, s2: int, slope: float, code: int) -> None:
    """NHWC engine entry: out[..., D*D] = lrelu(corr(f1, f2)); f1/f2 channel extent padded with zeros to 8.
    Algorithmic HBM bytes: both feature maps read once, the cost volume written once."""
    cp = rup(f1.C, 8)
    assert f1.c0 + cp <= f1.ld and f2.c0 + cp <= f2.ld and (f1.B, f1.H, f1.W) == (f2.B, f2.H, f2.W) == (out.B, out.H, out.W)
    D = 2 * (md // s2) + 1
    PROFILER.call("correlation_fwd", float(f1.rows) * (2 * f1.C + D * D) * _esz(code), f"corr-fwd C={f1.C} {f1.H}x{f1.W}",
                  "mireg_correlation_fwd", f1.ptr, f1.ld, f2.ptr, f2.ld, out.ptr, out.ld, f1.B, f1.H, f1.W, cp, c_norm, md, s2,
                  slope, code, _stream(), unit="B")


def correlation_bwd_views(g: View, f1: View, f2: View, d1: View, d2: View, cp: int, c_norm: int, md: int, s2: int,
                          acc1: int, acc2: int, code: int) -> None:
    """d corr / d f1, d f2.  Algorithmic bytes: cost-volume gradient and both maps read once, both gradients written once."""
    D = 2 * (md // s2) + 1
    PROFILER.call("correlation_bwd", float(g.rows) * (D * D + 4 * c_norm) * _esz(code), f"corr-bwd C={c_norm} {f1.H}x{f1.W}",
                  "mireg_correlation_bwd", g.ptr, g.ld, f1.ptr, f1.ld, f2.ptr, f2.ld, d1.ptr, d1.ld, d2.ptr, d2.ld,
                  f1.B, f1.H, f1.W, cp, c_norm, md, s2, acc1, acc2, code, _stream(), unit="B")


def pwc_warp_views(x: View, flow32: View, scale: float, out: View, code: int) -> None:
    cp = rup(x.C, 8)
    assert x.c0 + cp <= x.ld and out.c0 + cp <= out.ld
    PROFILER.call("pwc_warp_fwd", float(x.rows) * (2 * x.C * _esz(code) + 8), f"warp-fwd C={x.C} {x.H}x{x.W}",
                  "mireg_pwc_warp_fwd", x.ptr, x.ld, flow32.ptr, flow32.ld, float(scale), out.ptr, out.ld, x.B, x.H, x.W, cp,
                  code, _stream(), unit="B")


class WarpBwdWorkspace:
    """Caller-owned scratch of mireg_pwc_warp_bwd_det for P = B*H*W pixels (counters start at zero and the kernel leaves them zero)."""

    def __init__(self, P: int, device):
        self.cnt = torch.zeros(P, device=device, dtype=torch.int32)
        self.off = torch.zeros(P + 1 + (P + 2047) // 2048, device=device, dtype=torch.int32)
        self.entries = torch.zeros(4 * P, 2, device=device, dtype=torch.int32)


def pwc_warp_bwd_views(x: View, flow32: View, scale: float, dw: View, dx32: View, dflow32: View, C: int, code: int,
                       wsb: WarpBwdWorkspace) -> None:
    """Deterministic backward of the PWC warp (no fp32 scatter atomics).  Algorithmic bytes: features and their warped gradient
    read once, fp32 feature gradient + flow gradient written once."""
    PROFILER.call("pwc_warp_bwd", float(x.rows) * (2 * C * _esz(code) + 4 * C + 16), f"warp-bwd C={C} {x.H}x{x.W}",
                  "mireg_pwc_warp_bwd_det", x.ptr, x.ld, flow32.ptr, flow32.ld, float(scale), dw.ptr, dw.ld, dx32.ptr, dx32.ld,
                  dflow32.ptr, dflow32.ld, wsb.cnt.data_ptr(), wsb.off.data_ptr(), wsb.entries.data_ptr(), x.B, x.H, x.W, C, code,
                  _stream(), unit="B")


class Correlation(nn.Module):
    def __init__(self, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        if kernel_size != 1 or stride1 != 1 or pad_size != max_displacement or corr_multiply != 1 or max_displacement % stride2:
            raise NotImplementedError("mireg.Correlation supports kernel_size=1, stride1=1, pad_size==max_displacement, "
                                      "corr_multiply=1 (the only parameterisations the reference uses)")
        self.md, self.s2 = max_displacement, stride2

    def forward(self, in1: torch.Tensor, in2: torch.Tensor) -> torch.Tensor:
        if not in1.is_cuda:
            raise RuntimeError("mireg.Correlation runs on the MI355X only; there is no CPU fallback")
        return _CorrelationFn.apply(in1, in2, self.md, self.s2)


class _CorrelationFn(torch.autograd.Function):
    """NCHW fp32 front end of mireg_correlation_fwd / _bwd (exact-fp32 MFMA kernels)."""

    @staticmethod
    def forward(ctx, in1, in2, md, s2):
        B, C, H, W = in1.shape
        cp = rup(C, 4)
        a = torch.zeros(B, H, W, cp, device=in1.device, dtype=torch.float32)
        b = torch.zeros(B, H, W, cp, device=in1.device, dtype=torch.float32)
        a[..., :C] = in1.detach().permute(0, 2, 3, 1)
        b[..., :C] = in2.detach().permute(0, 2, 3, 1)
        D = 2 * (md // s2) + 1
        out = torch.empty(B, H, W, D * D, device=in1.device, dtype=torch.float32)
        _lib.call("mireg_correlation_fwd", a.data_ptr(), cp, b.data_ptr(), cp, out.data_ptr(), D * D, B, H, W, cp, C,
                  md, s2, 1.0, DT_F32, _stream())
        ctx.save_for_backward(a, b)
        ctx.cfg = (B, C, H, W, cp, D, md, s2)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, C, H, W, cp, D, md, s2 = ctx.cfg
        gn = g.permute(0, 2, 3, 1).contiguous().float()                       # (B, H, W, D*D)
        da, db = torch.empty_like(a), torch.empty_like(b)
        _lib.call("mireg_correlation_bwd", gn.data_ptr(), D * D, a.data_ptr(), cp, b.data_ptr(), cp, da.data_ptr(), cp,
                  db.data_ptr(), cp, B, H, W, cp, C, md, s2, 0, 0, DT_F32, _stream())
        return da[..., :C].permute(0, 3, 1, 2), db[..., :C].permute(0, 3, 1, 2), None, None
