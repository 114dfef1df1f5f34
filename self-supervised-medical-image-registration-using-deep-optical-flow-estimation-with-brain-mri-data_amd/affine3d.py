"""3-D affine registration path (reference models.py:39-43,156-191 `conv_3d` / `affmodel`; loss.py:16-19,38-50,87-94).

`affmodel` keeps the reference's module names (conv1..conv6, fc) and forward signature `(x) -> (para, warped)`;
Conv3d + ReLU run on the same LDS-DMA implicit-GEMM kernel as the 2-D predictors (depth axis in the descriptor,
NDHWC volumes), the Linear layer is the same kernel with the whole remaining volume as one tap window, the
affine grid + trilinear sampling is one fused kernel, and `Affloss` reuses the moment / finalize kernels of
OFEloss.  Forward / evaluation only in this round (no 3-D backward kernels yet).
`fc_in` defaults to the reference's hard-wired 176*512 (256x256x176 volumes); pass the flattened size of
conv6's output for other volume sizes.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import _lib
from .engine import DT_BF16, DT_F32, ConvDesc, PackJob, View, Workspace, _stream, assign_tiles, rup, upload_table
from .ops import SLOTS

SPEC = [(2, 16, 7, (2, 2, 1)), (16, 32, 5, (2, 2, 1)), (32, 64, 3, (2, 2, 2)), (64, 128, 3, (2, 2, 2)),
        (128, 256, 3, (2, 2, 2)), (256, 512, 3, (2, 2, 2))]


class Conv3dLayer:
    """Conv3d(k, stride, padding=(k-1)//2) forward on mireg_conv_gemm with the depth axis enabled."""

    def __init__(self, weight: torch.Tensor, bias, stride: Tuple[int, int, int], pad: Tuple[int, int, int], ws: Workspace):
        self.weight, self.bias, self.stride, self.pad, self.ws = weight, bias, stride, pad, ws
        self.Co, self.Ci, self.kd, self.kh, self.kw = weight.shape
        self.Cip = rup(self.Ci, 8)
        self.Kf = self.kd * self.kh * self.kw * self.Cip
        self.packF = torch.zeros(self.Co, self.Kf, device=ws.device, dtype=ws.dtype)

    def pack_job(self) -> PackJob:
        j = PackJob()
        j.src, j.dst = self.weight.data_ptr(), self.packF.data_ptr()
        j.Co, j.Ci, j.kh, j.kw = self.Co, self.Ci, self.kd * self.kh, self.kw      # taps = kd*kh*kw, order (z, y, x)
        j.Cpad, j.Cop, j.ld, j.stride, j.nclass = self.Cip, rup(self.Co, 8), self.Kf, 1, 0
        return j

    def out_dims(self, D: int, H: int, W: int) -> Tuple[int, int, int]:
        f = lambda n, k, s, p: (n + 2 * p - k) // s + 1
        return (f(D, self.kd, self.stride[0], self.pad[0]), f(H, self.kh, self.stride[1], self.pad[1]),
                f(W, self.kw, self.stride[2], self.pad[2]))

    def run(self, x: torch.Tensor, dims: Tuple[int, int, int], y: torch.Tensor, slope: float) -> Tuple[int, int, int]:
        """x: (B, D, H, W, ld_x) NDHWC buffer, y: (B, Do, Ho, Wo, ld_y)."""
        B = x.shape[0]
        D, H, W = dims
        Do, Ho, Wo = self.out_dims(D, H, W)
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C, d.x_D = x.data_ptr(), x.shape[-1], H, W, self.Cip, D
        d.taps_z, d.taps_y, d.taps_x = self.kd, self.kh, self.kw
        d.mul_z, d.mul_y, d.mul_x = self.stride
        d.off_z, d.off_y, d.off_x = -self.pad[0], -self.pad[1], -self.pad[2]
        d.step_z = d.step_y = d.step_x = 1
        d.g_D, d.g_H, d.g_W, d.n_img = Do, Ho, Wo, B
        d.w, d.w_ld, d.N = self.packF.data_ptr(), self.Kf, self.Co
        d.x_bytes, d.w_bytes = x.numel() * x.element_size(), self.packF.numel() * self.packF.element_size()
        d.y, d.y_ld, d.y_D, d.y_H, d.y_W = y.data_ptr(), y.shape[-1], Do, Ho, Wo
        d.y_mul_z = d.y_mul_y = d.y_mul_x = 1
        d.bias = self.bias.data_ptr() if self.bias is not None else None
        d.slope, d.dtype, d.split_k = slope, self.ws.code, 1
        M, bn = B * Do * Ho * Wo, (128 if self.Co > 64 else (64 if self.Co > 32 else 32))
        tiles = ((M + 127) // 128) * ((self.Co + bn - 1) // bn)
        nk = (self.Kf + 31) // 32
        if tiles < 256 and nk >= 16:                        # deep / tiny layers (incl. the Linear): split K
            split = max(1, min((512 + tiles - 1) // tiles, nk // 4, 64))
            if split > 1:
                d.split_k, d.slab_cls_stride = split, split * M * self.Co
                self.ws.need_scratch(split * M * self.Co)
                d.slab = self.ws.get_scratch().data_ptr()
        _lib.call("mireg_conv_gemm", ctypes.byref(d), _stream())
        return Do, Ho, Wo


class affmodel(nn.Module):
    """Drop-in for reference models.affmodel (forward only)."""

    def __init__(self, fc_in: int = 176 * 512, precision: str = "bf16"):
        super().__init__()
        self.precision = precision
        for i, (cin, cout, k, s) in enumerate(SPEC, start=1):
            stride = (s[0], s[1], s[2])
            setattr(self, f"conv{i}", nn.Sequential(nn.Conv3d(cin, cout, k, stride, (k - 1) // 2), nn.ReLU(True)))
        self.flat = nn.Flatten()
        self.fc = nn.Linear(fc_in, 12)
        self._eng: Dict[tuple, dict] = {}

    def _engine(self, x: torch.Tensor) -> dict:
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (tuple(x.shape), x.device, dtype, self.conv1[0].weight.data_ptr())
        if key in self._eng:
            return self._eng[key]
        self._eng.clear()
        B, C, D, H, W = x.shape
        ws = Workspace(x.device, dtype)
        layers, bufs, dims = [], [], (D, H, W)
        for i in range(1, 7):
            conv = getattr(self, f"conv{i}")[0]
            lay = Conv3dLayer(conv.weight, conv.bias, tuple(conv.stride), tuple(conv.padding), ws)
            layers.append(lay)
            dims = lay.out_dims(*dims)
            bufs.append(torch.zeros(B, *dims, rup(lay.Co, 8), device=x.device, dtype=dtype))
        if 512 * dims[0] * dims[1] * dims[2] != self.fc.in_features:
            raise RuntimeError(f"affmodel: fc expects {self.fc.in_features} features, conv6 yields 512x{dims} = "
                               f"{512 * dims[0] * dims[1] * dims[2]} (the reference is hard-wired to 256x256x176 volumes)")
        # Linear == Conv3d whose single tap window is the whole conv6 volume; torch flattens (C, D, H, W)
        fcw = self.fc.weight.view(12, 512, *dims)
        fc = Conv3dLayer(fcw, self.fc.bias, (1, 1, 1), (0, 0, 0), ws)
        e = dict(ws=ws, layers=layers, bufs=bufs, fc=fc, fc_dims=dims,
                 x0=torch.zeros(B, D, H, W, 8, device=x.device, dtype=dtype),
                 para=torch.zeros(B, 1, 1, 1, 16, device=x.device, dtype=torch.float32),
                 paraT=torch.zeros(B, 1, 1, 1, 16, device=x.device, dtype=dtype))
        self._eng[key] = e
        return e

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("mireg.affmodel runs on the MI355X only; there is no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("affmodel training (3-D backward kernels) is not implemented yet; use torch.no_grad()")
        e = self._engine(x)
        ws, st = e["ws"], _stream()
        B, C, D, H, W = x.shape
        x = x.float().contiguous()
        jobs = [l.pack_job() for l in e["layers"]] + [e["fc"].pack_job()]
        units, dunits = assign_tiles(jobs, False)
        tab = upload_table(jobs, x.device)
        _lib.call("mireg_pack_weights", tab.data_ptr(), len(jobs), units, dunits, ws.code, st)
        _lib.call("mireg_nchw_to_nhwc", x.data_ptr(), e["x0"].data_ptr(), B, 2, 0, 2, D * H * W, 8, ws.code, st)
        src, dims = e["x0"], (D, H, W)
        for lay, buf in zip(e["layers"], e["bufs"]):
            dims = lay.run(src, dims, buf, 0.0)               # LeakyReLU with slope 0 == ReLU
            src = buf
        e["fc"].run(src, dims, e["paraT"], 1.0)
        _lib.call("mireg_cast_to_f32", e["para"].data_ptr(), 16, e["paraT"].data_ptr(), 16, B, 12, 1.0, 0.0, ws.code, st)
        para = e["para"].view(B, 16)[:, :12].reshape(B, 3, 4).contiguous()
        moving = x[:, 1:].contiguous()
        warped = torch.empty_like(moving)
        _lib.call("mireg_affine_sample3d", moving.data_ptr(), para.data_ptr(), warped.data_ptr(), B, 1, D, H, W, st)
        return para, warped


def Affloss(warped: torch.Tensor, fixed: torch.Tensor, lamb_da: float = 1.0, gamma: float = 1.0):
    """Drop-in for reference loss.Affloss (loss.py:87-94): (gamma * photometric_3d, lamb_da * ncc_3d, sum)."""
    if not warped.is_cuda:
        raise RuntimeError("mireg.Affloss runs on the MI355X only; there is no CPU fallback")
    w, f = warped.float().contiguous(), fixed.float().contiguous()
    B, n = w.shape[0], w.numel()
    sums = torch.zeros(1, SLOTS, 8, device=w.device, dtype=torch.float64)
    npix = torch.tensor([n], dtype=torch.int64, device=w.device)
    out = torch.empty(4, device=w.device, dtype=torch.float64)
    st = _stream()
    _lib.call("mireg_loss_partials", w.data_ptr(), f.data_ptr(), sums.data_ptr(), n, st)
    # OFE finalisation with one scale: weight 0.05 -> fold 1/0.05 into gamma / zeta
    _lib.call("mireg_ofe_finalize", sums.data_ptr(), npix.data_ptr(), 1, B, 0.0, gamma / 0.05, lamb_da / 0.05, out.data_ptr(), st)
    return out[0], out[1], out[0] + out[1]


def photometric_loss_3d(fixed: torch.Tensor, warped: torch.Tensor):
    """Drop-in for reference loss.photometric_loss_3d (loss.py:16-19) -- value only."""
    return Affloss(warped, fixed, 1.0, 1.0)[0]


def correlation_loss_3d(fixed: torch.Tensor, warped: torch.Tensor):
    """Drop-in for reference loss.correlation_loss_3d (loss.py:38-50) -- value only."""
    return Affloss(warped, fixed, 1.0, 1.0)[1]
