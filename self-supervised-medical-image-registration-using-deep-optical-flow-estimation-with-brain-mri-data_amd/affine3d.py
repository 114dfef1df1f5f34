"""3-D affine registration path (reference models.py:39-43,156-191 `conv_3d` / `affmodel`; loss.py:16-19,38-50,87-94).

`affmodel` keeps the reference's module names (conv1..conv6, fc) and forward signature `(x) -> (para, warped)`;
Conv3d + ReLU run on the same LDS-DMA implicit-GEMM kernel as the 2-D predictors (depth axis in the descriptor,
NDHWC volumes), the Linear layer is the same kernel with the whole remaining volume as one tap window, the
affine grid + trilinear sampling is one fused kernel, and `Affloss` reuses the moment / finalize kernels of
OFEloss.  Training: both are torch.autograd functions whose backward runs on the same kernels -- backward-data as one
GEMM launch per output-voxel parity class, backward-weights as one launch per depth tap into a shared slab, the ReLU
masks, the bias column sums, the sampler's d/d theta kernel and the Charbonnier / NCC gradient of OFEloss.
`fc_in` defaults to the reference's hard-wired 176*512 (256x256x176 volumes); pass the flattened size of
conv6's output for other volume sizes.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import engine as engine_mod
from .engine import DT_BF16, PROFILER, ConvDesc, Pack3dJob, PackJob, View, WoptJob, Workspace, _stream, assign_tiles, rup, upload_table
from .ops import SLOTS, pixel_counts

SPEC = [(2, 16, 7, (2, 2, 1)), (16, 32, 5, (2, 2, 1)), (32, 64, 3, (2, 2, 2)), (64, 128, 3, (2, 2, 2)),
        (128, 256, 3, (2, 2, 2)), (256, 512, 3, (2, 2, 2))]


@dataclass
class Vol:
    """Channel slice [c0, c0+C) of a channel-last volume buffer (B, D, H, W, ld)."""
    buf: torch.Tensor
    dims: Tuple[int, int, int]
    C: int
    c0: int = 0

    @property
    def B(self) -> int:
        return self.buf.shape[0]

    @property
    def ld(self) -> int:
        return self.buf.shape[-1]

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + self.c0 * self.buf.element_size()

    @property
    def rows(self) -> int:
        return self.buf.numel() // self.buf.shape[-1]

    @property
    def bytes_left(self) -> int:
        return (self.buf.numel() - self.c0) * self.buf.element_size()

    def slice(self, c0: int, C: int) -> "Vol":
        assert c0 >= 0 and self.c0 + c0 + C <= self.ld
        return Vol(self.buf, self.dims, C, self.c0 + c0)

    def view2d(self) -> View:
        """The same rows as an NHWC view (depth folded into H) for the dimension-agnostic row kernels (BatchNorm, masks)."""
        D, H, W = self.dims
        return View(self.buf.view(self.B, D * H, W, self.ld), self.B, D * H, W, self.C, self.c0)


def _vol(x, dims) -> Vol:
    return x if isinstance(x, Vol) else Vol(x, tuple(dims), x.shape[-1], 0)


WIDE_CLASSES = os.environ.get("MIREG_3D_DGRAD_WIDE_CLASSES", "0") == "1"        # A/B switch: large levels as per-class launches of the 8-wave tile
MERGE_CLASSES = os.environ.get("MIREG_3D_DGRAD_PER_CLASS", "0") != "1"     # A/B switch: one backward-data launch per parity class
WGRAD_PER_TAP = os.environ.get("MIREG_3D_WGRAD_PER_TAP", "0") == "1"    # A/B switch: one backward-weights launch per depth tap
FORCE_WIDE = None      # tests only: None = heuristic, (128,) / (256,) = always the 256-pixel tile at that width, () = never


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

    def run(self, x, dims: Tuple[int, int, int], y, slope: float, *, y32=None, accumulate: bool = False,
            bias: bool = True) -> Tuple[int, int, int]:
        """x: (B, D, H, W, ld_x) channel-last buffer or Vol, y: (B, Do, Ho, Wo, ld_y) buffer or Vol (None with y32 only)."""
        x = _vol(x, dims)
        B = x.B
        D, H, W = dims
        Do, Ho, Wo = self.out_dims(D, H, W)
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C, d.x_D = x.ptr, x.ld, H, W, self.Cip, D
        d.taps_z, d.taps_y, d.taps_x = self.kd, self.kh, self.kw
        d.mul_z, d.mul_y, d.mul_x = self.stride
        d.off_z, d.off_y, d.off_x = -self.pad[0], -self.pad[1], -self.pad[2]
        d.step_z = d.step_y = d.step_x = 1
        d.g_D, d.g_H, d.g_W, d.n_img = Do, Ho, Wo, B
        d.w, d.w_ld, d.N = self.packF.data_ptr(), self.Kf, self.Co
        d.x_bytes, d.w_bytes = x.bytes_left, self.packF.numel() * self.packF.element_size()
        if y is not None:
            y = _vol(y, (Do, Ho, Wo))
            d.y, d.y_ld = y.ptr, y.ld
        if y32 is not None:
            y32 = _vol(y32, (Do, Ho, Wo))
            d.y32, d.y32_ld = y32.ptr, y32.ld
        d.y_D, d.y_H, d.y_W = Do, Ho, Wo
        d.y_mul_z = d.y_mul_y = d.y_mul_x = 1
        d.bias = self.bias.data_ptr() if (bias and self.bias is not None) else None
        d.slope, d.dtype, d.split_k, d.accumulate = slope, self.ws.code, 1, int(accumulate)
        self._launch(d, B * Do * Ho * Wo, self.Co, self.Kf)
        return Do, Ho, Wo

    def _launch(self, d: ConvDesc, M: int, N: int, K: int) -> None:
        bn = 128 if N > 64 else (64 if N > 32 else 32)
        tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
        nk = (K + 31) // 32
        d.split_k = 1
        # volumes have the rows the 256-pixel 8-wave tile (conv_wide.hip) wants: take it wherever its grid still covers the chip
        # (FORCE_WIDE: tests; (tile_n,) forces it, () forbids it)
        d.algo, d.tile_n = 0, 0
        if self.ws.code == DT_BF16 and engine_mod.USE_WIDE and FORCE_WIDE != () and _lib.lib().mireg_conv_wide_eligible(ctypes.byref(d), None):
            t256, t128 = ((M + 255) // 256) * ((N + 255) // 256), ((M + 255) // 256) * ((N + 127) // 128)
            pick = FORCE_WIDE[0] if FORCE_WIDE else (256 if (N > 128 and t256 >= 224) else (128 if t128 >= 224 else 0))
            if pick:
                d.algo, d.tile_n = 3, pick
                PROFILER.launch("mireg_conv_gemm", d, "conv3d_gemm", 2.0 * M * N * K, f"{getattr(self, 'name', 'conv3d')} M={M} N={N} K={K} wide{pick}")
                return
        if tiles < 256 and nk >= 16:                        # deep / tiny layers (incl. the Linear): split K
            split = max(1, min((512 + tiles - 1) // tiles, nk // 4, 64))
            if split > 1:
                d.split_k, d.slab_cls_stride = split, split * M * N
                self.ws.need_scratch(split * M * N)
                d.slab = self.ws.get_scratch().data_ptr()
        PROFILER.launch("mireg_conv_gemm", d, "conv3d_gemm", 2.0 * M * N * K, f"{getattr(self, 'name', 'conv3d')} M={M} N={N} K={K}")

    # ---- backward (autograd of nn.Conv3d, reference models.py:39-43 trained through loss.backward()) ----
    def dgrad_classes(self) -> list:
        """Per output-voxel parity class (pz, py, px) of the backward-data form: the taps t = r + s*j that reach it, packed
        [Ci][(jz, jy, jx)][Cop] by mireg_pack_dgrad3d (persistent buffers; class order = the kernel's (cz*sy + cy)*sx + cx)."""
        if getattr(self, "_cls", None) is None:
            Cop = rup(self.Co, 8)
            self._cls = []
            for pz in range(self.stride[0]):
                for py in range(self.stride[1]):
                    for px in range(self.stride[2]):
                        par = (pz, py, px)
                        r = [(par[a] + self.pad[a]) % self.stride[a] for a in range(3)]
                        c = [(par[a] + self.pad[a]) // self.stride[a] for a in range(3)]
                        k = (self.kd, self.kh, self.kw)
                        nt = tuple(max(0, (k[a] - r[a] + self.stride[a] - 1) // self.stride[a]) for a in range(3))
                        if min(nt) == 0:
                            raise ValueError("Conv3d kernel smaller than its stride is not supported")
                        pk = torch.zeros(self.Ci, nt[0] * nt[1] * nt[2] * Cop, device=self.ws.device, dtype=self.ws.dtype)
                        self._cls.append(dict(par=par, c=c, nt=nt, pack=pk))
        return self._cls

    def pack3d_job(self) -> Pack3dJob:
        cls = self.dgrad_classes()
        j = Pack3dJob()
        j.src = self.weight.data_ptr()
        for i, k in enumerate(cls):
            j.dst[i] = k["pack"].data_ptr()
        j.Co, j.Ci, j.Cop = self.Co, self.Ci, rup(self.Co, 8)
        j.kd, j.kh, j.kw = self.kd, self.kh, self.kw
        j.sz, j.sy, j.sx = self.stride
        j.pz, j.py, j.px = self.pad
        return j

    @staticmethod
    def pack_dgrad_table(layers, ws: Workspace, from_fwd: bool = False) -> torch.Tensor:
        """One launch for the backward-data packs of `layers`; marks them fresh (dgrad() repacks a stale layer on its own).
        from_fwd: the layers' forward packs are current (just packed / rewritten by the packed-domain optimizer): transpose
        those (half the bytes, whole-line accesses) instead of gathering from the fp32 weights."""
        jobs, u = [], 0
        for l in layers:
            j = l.pack3d_job()
            j.unit0 = u
            if from_fwd:
                j.src = l.packF.data_ptr()
                u += l.kd * l.kh * l.kw * ((rup(l.Co, 8) + 63) // 64) * ((l.Ci + 63) // 64)
            else:
                u += l.Ci * ((rup(l.Co, 8) + 63) // 64)
            jobs.append(j)
        tab = upload_table(jobs, ws.device)
        _lib.call("mireg_pack_dgrad3d_fwd" if from_fwd else "mireg_pack_dgrad3d", tab.data_ptr(), len(jobs), u, ws.code, _stream())
        for l in layers:
            l.dfresh = True
        return tab

    def dgrad(self, gy, odims: Tuple[int, int, int], gx, idims: Tuple[int, int, int], *, slope: float = 1.0,
              accumulate: bool = False) -> None:
        """gx[(z,y,x), ci] = act(sum_{taps, co} gy[(z + p - t)/s ..., co] W[co][ci][t]); one GEMM launch per parity class.
        Also the forward of ConvTranspose3d (whose weight [Cin][Cout][k^3] is this layer's [Co][Ci][k^3])."""
        gy, gx = _vol(gy, odims), _vol(gx, idims)
        B = gy.B
        Cop = rup(self.Co, 8)
        if self.weight.dtype != torch.float32 or not self.weight.is_contiguous():
            raise RuntimeError("Conv3dLayer expects contiguous float32 weights")
        if not getattr(self, "dfresh", False):
            self._tab3 = self.pack_dgrad_table([self], self.ws)
        self.dfresh = False                                   # the packs serve one backward-data pass; weights may move after it
        live = []
        for k in self.dgrad_classes():
            g = [(idims[a] - k["par"][a] + self.stride[a] - 1) // self.stride[a] for a in range(3)]
            if min(g) > 0:
                live.append((k, g))
        if len(live) > 1 and MERGE_CLASSES and self._merged_dgrad(live, gy, odims, gx, idims, slope, accumulate):
            return
        for k, g in live:
            d = ConvDesc()
            d.x, d.x_ld, d.x_D, d.x_H, d.x_W, d.x_C = gy.ptr, gy.ld, odims[0], odims[1], odims[2], Cop
            d.taps_z, d.taps_y, d.taps_x = k["nt"]
            d.mul_z = d.mul_y = d.mul_x = 1
            d.off_z, d.off_y, d.off_x = k["c"]
            d.step_z = d.step_y = d.step_x = -1
            d.g_D, d.g_H, d.g_W, d.n_img = g[0], g[1], g[2], B
            d.w, d.w_ld, d.N = k["pack"].data_ptr(), k["pack"].shape[1], self.Ci
            d.x_bytes, d.w_bytes = gy.bytes_left, k["pack"].numel() * k["pack"].element_size()
            d.y, d.y_ld, d.y_D, d.y_H, d.y_W = gx.ptr, gx.ld, idims[0], idims[1], idims[2]
            d.y_mul_z, d.y_mul_y, d.y_mul_x = self.stride
            d.y_off_z, d.y_off_y, d.y_off_x = k["par"]
            d.slope, d.dtype, d.accumulate = slope, self.ws.code, int(accumulate)
            self._launch(d, B * g[0] * g[1] * g[2], self.Ci, k["pack"].shape[1])

    def _merged_dgrad(self, live, gy: Vol, odims, gx: Vol, idims, slope: float, accumulate: bool) -> bool:
        """All parity classes of a stride-2 backward-data pass in ONE ring-kernel launch (blockIdx.y = class, mireg_conv_cls with its
        depth fields) plus one split-K reduce, instead of up to eight of each: the coarse levels' launches are launch-latency sized and
        the large levels' eight grids (27 / 18 / 12 / 8 taps) leave CUs idle at each launch's tail.  False = take the per-class path."""
        B, Cop, N = gy.B, rup(self.Co, 8), self.Ci
        Ms = [B * g[0] * g[1] * g[2] for _, g in live]
        Ks = [k["pack"].shape[1] for k, _ in live]
        M, Kmin = max(Ms), min(Ks)
        # (measured, FlowNetS-3D: the merged launch also beats eight launches of the 8-wave tile on the large levels, 12.62 -> 12.25 ms per
        # step; WIDE_CLASSES restores the per-class form there, FORCE_WIDE (tests) always takes it)
        if FORCE_WIDE or (WIDE_CLASSES and self.ws.code == DT_BF16 and engine_mod.USE_WIDE and FORCE_WIDE != () and Cop >= 64 and N >= 16
                          and min(Ks) >= 64 and (((M + 255) // 256) * ((N + 127) // 128) >= 224)):
            return False
        k0, g0 = live[0]
        d = ConvDesc()
        d.x, d.x_ld, d.x_D, d.x_H, d.x_W, d.x_C = gy.ptr, gy.ld, odims[0], odims[1], odims[2], Cop
        d.mul_z = d.mul_y = d.mul_x = 1
        d.step_z = d.step_y = d.step_x = -1
        d.n_img, d.N = B, N
        d.x_bytes = gy.bytes_left
        d.y, d.y_ld, d.y_D, d.y_H, d.y_W = gx.ptr, gx.ld, idims[0], idims[1], idims[2]
        d.y_mul_z, d.y_mul_y, d.y_mul_x = self.stride
        d.slope, d.dtype, d.accumulate = slope, self.ws.code, int(accumulate)
        d.n_cls = len(live)
        for i, (k, g) in enumerate(live):
            c = d.cls[i]
            c.taps_z, c.taps_y, c.taps_x = k["nt"]
            c.off_z, c.off_y, c.off_x = k["c"]
            c.g_D, c.g_H, c.g_W = g
            c.y_off_z, c.y_off_y, c.y_off_x = k["par"]
            c.w, c.w_ld, c.w_bytes = k["pack"].data_ptr(), k["pack"].shape[1], k["pack"].numel() * k["pack"].element_size()
        # the launch's own class fields = class 0 (descriptor validation reads them)
        d.taps_z, d.taps_y, d.taps_x = k0["nt"]
        d.off_z, d.off_y, d.off_x = k0["c"]
        d.g_D, d.g_H, d.g_W = g0
        d.y_off_z, d.y_off_y, d.y_off_x = k0["par"]
        d.w, d.w_ld, d.w_bytes = d.cls[0].w, d.cls[0].w_ld, d.cls[0].w_bytes
        bn = 128 if N > 64 else (64 if N > 32 else 32)
        tiles = ((M + 127) // 128) * ((N + bn - 1) // bn) * len(live)
        nk = (Kmin + 31) // 32
        d.algo, d.tile_n, d.split_k = 1, 0, 1
        if tiles < 256 and nk >= 16:
            split = max(1, min((512 + tiles - 1) // tiles, nk // 4, 64))
            if split > 1:
                d.split_k, d.slab_cls_stride = split, split * M * N
                self.ws.need_scratch(len(live) * split * M * N)
                d.slab = self.ws.get_scratch().data_ptr()
        PROFILER.launch("mireg_conv_gemm", d, "conv3d_gemm", 2.0 * sum(m * N * kk for m, kk in zip(Ms, Ks)),
                        f"{getattr(self, 'name', 'conv3d')} dgrad {len(live)} classes M<={M} N={N}")
        return True

    def wgrad(self, x, idims: Tuple[int, int, int], gy, odims: Tuple[int, int, int]) -> None:
        """slab[z][co][(tz,ty,tx)*Cip + ci] = sum_voxels gy[v][co] x[v @ tap][ci]: one backward-weights launch per depth tap,
        each filling its column block of the shared slab (mireg_conv_desc.slab_ld)."""
        x, gy = _vol(x, idims), _vol(gy, odims)
        B = x.B
        P = B * odims[0] * odims[1] * odims[2]
        K2 = self.kh * self.kw * self.Cip
        tiles = ((self.Co + 127) // 128) * ((K2 + 127) // 128) * (1 if WGRAD_PER_TAP else self.kd)   # workgroups per pixel split
        nk = (P + 31) // 32
        split = 1 if tiles >= 256 else max(1, min(768 // tiles, max(nk // 8, 1)))      # up to three workgroups per CU resident
        if getattr(self, "slab", None) is None or self.slab.shape[0] != split:
            self.slab = torch.zeros(split, self.Co, self.Kf, device=x.buf.device, dtype=torch.float32)
        # one launch for all depth taps (grid.y = tap): the tap-variants of a pixel chunk run back to back and find its dy / x tiles in
        # the cache hierarchy; one launch per tap (WGRAD_PER_TAP, the round-2 form) swept both operands from HBM kd times
        for tz in (range(self.kd) if WGRAD_PER_TAP else (0,)):
            d = ConvDesc()
            d.x, d.x_ld, d.x_D, d.x_H, d.x_W, d.x_C = x.ptr, x.ld, idims[0], idims[1], idims[2], self.Cip
            d.taps_y, d.taps_x = self.kh, self.kw
            d.taps_z = 0 if WGRAD_PER_TAP else self.kd
            d.mul_z, d.mul_y, d.mul_x = self.stride
            d.off_z, d.off_y, d.off_x = tz - self.pad[0], -self.pad[1], -self.pad[2]
            d.step_y = d.step_x = 1
            d.g_D, d.g_H, d.g_W, d.n_img = odims[0], odims[1], odims[2], B
            d.y, d.y_ld, d.N = gy.ptr, gy.ld, self.Co
            d.split_k, d.dtype, d.stages = split, self.ws.code, 3
            d.slab, d.slab_ld = self.slab.data_ptr() + 4 * tz * K2, self.Kf
            d.x_bytes, d.w_bytes = x.bytes_left, gy.bytes_left
            d.algo = 1                                                            # the LDS-DMA ring kernel (the only one with a depth axis)
            PROFILER.launch("mireg_conv_wgrad", d, "conv3d_wgrad", 2.0 * P * self.Co * K2 * (1 if WGRAD_PER_TAP else self.kd),
                            f"conv3d-wgrad {'tz=%d' % tz if WGRAD_PER_TAP else 'all taps'}")

    @staticmethod
    def unpack_grads(pairs, ws: Workspace):
        """pairs = [(layer, torch-layout fp32 gradient)]: sum the split-K slabs in place (fully parallel, fixed order), then one
        transposing pass slab[co][(tap, ci)] -> grad[co][ci][tap].  Returns the device tables (keep them until the stream ran)."""
        st, red, r = _stream(), [], 0
        for lay, _ in pairs:
            if lay.slab.shape[0] > 1:
                j = WoptJob()
                j.slab = j.g = lay.slab.data_ptr()
                j.slab_stride, j.nsplit = lay.Co * lay.Kf, lay.slab.shape[0]
                j.Co, j.Ci, j.taps, j.Cpad, j.ld = lay.Co, lay.Ci, lay.kd * lay.kh * lay.kw, lay.Cip, lay.Kf
                j.runit0 = r
                r += (lay.Co * lay.Kf + 255) // 256
                red.append(j)
        tabs = []
        if red:
            tabs.append(upload_table(red, ws.device))
            _lib.call("mireg_wgrad_reduce", tabs[-1].data_ptr(), len(red), r, st)
        jobs = [lay.unpack_job(g, 1) for lay, g in pairs]
        units, _ = assign_tiles(jobs, True)
        tabs.append(upload_table(jobs, ws.device))
        _lib.call("mireg_unpack_wgrad", tabs[-1].data_ptr(), len(jobs), units, st)
        return tabs

    def unpack_job(self, grad: torch.Tensor, nsplit: int = 0) -> PackJob:
        j = PackJob()
        j.src, j.dst = self.slab.data_ptr(), grad.data_ptr()
        j.Co, j.Ci, j.kh, j.kw = self.Co, self.Ci, self.kd * self.kh, self.kw
        j.Cpad, j.Cop, j.ld, j.stride, j.nclass = self.Cip, rup(self.Co, 8), self.Kf, 1, 0
        j.nsplit, j.accumulate = (nsplit or self.slab.shape[0]), 0
        return j

    def bias_grad(self, gy: torch.Tensor, out: torch.Tensor) -> None:
        ws = self.ws
        if ws.colsum_ws is None or ws.colsum_ws.numel() < 64 * max(self.Co, 1024):
            ws.colsum_ws = torch.empty(64 * max(self.Co, 1024), device=ws.device, dtype=torch.float32)
        gy = _vol(gy, (1, 1, 1))
        _lib.call("mireg_colsum", gy.ptr, gy.ld, gy.rows, self.Co, out.data_ptr(), 0, ws.colsum_ws.data_ptr(),
                  ws.code, _stream())


class _AffmodelFn(torch.autograd.Function):
    """(x, *parameters) -> (para, warped); backward hands autograd the parameter gradients of the HIP backward pass."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod = mod
        para, warped = mod._forward_impl(x, keep=True)
        return para, warped

    @staticmethod
    def backward(ctx, g_para, g_warped):
        return (None, None, *ctx.mod._backward_impl(g_para, g_warped))


class affmodel(nn.Module):
    """Drop-in for reference models.affmodel: forward `(x) -> (para, warped)`; trains through loss.backward() with the
    parameter gradients produced by the HIP backward pass (one forward in flight per module: the engine owns the saved
    activations)."""

    def __init__(self, fc_in: int = 176 * 512, precision: str = "bf16"):
        super().__init__()
        self.precision = precision
        for i, (cin, cout, k, s) in enumerate(SPEC, start=1):
            stride = (s[0], s[1], s[2])
            setattr(self, f"conv{i}", nn.Sequential(nn.Conv3d(cin, cout, k, stride, (k - 1) // 2), nn.ReLU(True)))
        self.flat = nn.Flatten()
        self.fc = nn.Linear(fc_in, 12)
        self._eng: Dict[tuple, dict] = {}
        self._last = None

    def _params(self):
        out = []
        for i in range(1, 7):
            conv = getattr(self, f"conv{i}")[0]
            out += [conv.weight, conv.bias]
        return out + [self.fc.weight, self.fc.bias]

    def _engine(self, x: torch.Tensor) -> dict:
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (tuple(x.shape), x.device, dtype, self.conv1[0].weight.data_ptr())
        if key in self._eng:
            return self._eng[key]
        self._eng.clear()
        B, C, D, H, W = x.shape
        ws = Workspace(x.device, dtype)
        layers, bufs, dims, dlist = [], [], (D, H, W), [(D, H, W)]
        for i in range(1, 7):
            conv = getattr(self, f"conv{i}")[0]
            lay = Conv3dLayer(conv.weight, conv.bias, tuple(conv.stride), tuple(conv.padding), ws)
            layers.append(lay)
            dims = lay.out_dims(*dims)
            dlist.append(dims)
            bufs.append(torch.zeros(B, *dims, rup(lay.Co, 8), device=x.device, dtype=dtype))
        if 512 * dims[0] * dims[1] * dims[2] != self.fc.in_features:
            raise RuntimeError(f"affmodel: fc expects {self.fc.in_features} features, conv6 yields 512x{dims} = "
                               f"{512 * dims[0] * dims[1] * dims[2]} (the reference is hard-wired to 256x256x176 volumes)")
        # Linear == Conv3d whose single tap window is the whole conv6 volume; torch flattens (C, D, H, W)
        fcw = self.fc.weight.view(12, 512, *dims)
        fc = Conv3dLayer(fcw, self.fc.bias, (1, 1, 1), (0, 0, 0), ws)
        e = dict(ws=ws, layers=layers, bufs=bufs, fc=fc, fc_dims=dims, dims=dlist,
                 x0=torch.zeros(B, D, H, W, 8, device=x.device, dtype=dtype),
                 para=torch.zeros(B, 1, 1, 1, 16, device=x.device, dtype=torch.float32),
                 paraT=torch.zeros(B, 1, 1, 1, 16, device=x.device, dtype=dtype))
        self._eng[key] = e
        return e

    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("mireg.affmodel runs on the MI355X only; there is no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _AffmodelFn.apply(self, x, *self._params())
        return self._forward_impl(x, keep=False)

    def _forward_impl(self, x: torch.Tensor, keep: bool):
        e = self._engine(x)
        ws, st = e["ws"], _stream()
        B, C, D, H, W = x.shape
        x = x.detach().float().contiguous()
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
        self._last = dict(e=e, moving=moving, para=para, shape=(B, D, H, W)) if keep else None
        return para, warped

    def _backward_impl(self, g_para, g_warped):
        """d loss / d parameters in the order of `_params()` (autograd of reference models.py:182-191)."""
        if self._last is None:
            raise RuntimeError("affmodel backward without a saved forward (one forward in flight per module)")
        L, self._last = self._last, None
        e, (B, D, H, W) = L["e"], L["shape"]
        ws, st, dev = e["ws"], _stream(), L["moving"].device
        layers, bufs, dl, fc = e["layers"], e["bufs"], e["dims"], e["fc"]
        if "g" not in e:
            e["g"] = [torch.zeros_like(b) for b in bufs]
            e["gparaT"] = torch.zeros(B, 1, 1, 1, 16, device=dev, dtype=ws.dtype)
            e["aff_ws"] = torch.empty(B * 512 * 12, device=dev, dtype=torch.float32)
        gtheta = (g_para.reshape(B, 12).float().clone() if g_para is not None
                  else torch.zeros(B, 12, device=dev, dtype=torch.float32)).contiguous()
        if g_warped is not None:
            gw = g_warped.float().contiguous()
            _lib.call("mireg_affine_sample3d_bwd", L["moving"].data_ptr(), L["para"].data_ptr(), gw.data_ptr(),
                      gtheta.data_ptr(), e["aff_ws"].data_ptr(), 1, B, 1, D, H, W, st)
        gp = e["gparaT"]
        _lib.call("mireg_cast_from_f32", gp.data_ptr(), 16, gtheta.data_ptr(), 12, B, 12, 1.0, 0.0, ws.code, st)
        params = self._params()
        grads = [torch.zeros_like(p, dtype=torch.float32) for p in params]
        one = (1, 1, 1)
        e["_tab3"] = Conv3dLayer.pack_dgrad_table([fc] + layers[1:], ws)     # backward-data packs of this step's weights
        fc.wgrad(bufs[5], dl[6], gp, one)
        fc.bias_grad(gp, grads[13])
        fc.dgrad(gp, one, e["g"][5], dl[6])
        for i in range(5, -1, -1):
            g, lay = e["g"][i], layers[i]
            _lib.call("mireg_lrelu_bwd", g.data_ptr(), g.shape[-1], bufs[i].data_ptr(), bufs[i].shape[-1],
                      g.numel() // g.shape[-1], lay.Co, 0.0, ws.code, st)                  # ReLU mask of this layer's output
            xin = bufs[i - 1] if i > 0 else e["x0"]
            lay.wgrad(xin, dl[i], g, dl[i + 1])
            lay.bias_grad(g, grads[2 * i + 1])
            if i > 0:
                lay.dgrad(g, dl[i + 1], e["g"][i - 1], dl[i])
        e["_tab"] = Conv3dLayer.unpack_grads([(layers[i], grads[2 * i]) for i in range(6)] + [(fc, grads[12])], ws)
        return [g.to(p.dtype) for g, p in zip(grads, params)]


class _AfflossFn(torch.autograd.Function):
    """out4 = (gamma * photometric_3d, lamb_da * ncc_3d, 0, sum) as float64; backward = d/d warped (loss.py:16-19,38-50)."""

    @staticmethod
    def forward(ctx, warped, fixed, lamb_da, gamma):
        w, f = warped.detach().float().contiguous(), fixed.detach().float().contiguous()
        B, n = w.shape[0], w.numel()
        sums = torch.zeros(1, SLOTS, 8, device=w.device, dtype=torch.float64)
        npix = pixel_counts([n], w.device)
        out = torch.empty(4, device=w.device, dtype=torch.float64)
        st = _stream()
        _lib.call("mireg_loss_partials", w.data_ptr(), f.data_ptr(), sums.data_ptr(), n, st)
        # OFE finalisation with one scale: weight 0.05 -> fold 1/0.05 into gamma / zeta
        _lib.call("mireg_ofe_finalize", sums.data_ptr(), npix.data_ptr(), 1, B, 0.0, gamma / 0.05, lamb_da / 0.05, out.data_ptr(), st)
        ctx.saved = (w, f, sums, npix, B, n, lamb_da, gamma, warped.dtype)
        return out

    @staticmethod
    def backward(ctx, g4):
        w, f, sums, npix, B, n, lamb_da, gamma, dt = ctx.saved
        st = _stream()
        g4 = g4.to(torch.float64).contiguous()
        coef = torch.empty(8, device=w.device, dtype=torch.float32)
        gw = torch.empty_like(w)
        _lib.call("mireg_ofe_bwd_coef", sums.data_ptr(), npix.data_ptr(), 1, B, 0.0, gamma / 0.05, lamb_da / 0.05,
                  g4.data_ptr(), coef.data_ptr(), st)
        _lib.call("mireg_loss_bwd", w.data_ptr(), f.data_ptr(), coef.data_ptr(), gw.data_ptr(), n, st)
        return gw.to(dt), None, None, None


def Affloss(warped: torch.Tensor, fixed: torch.Tensor, lamb_da: float = 1.0, gamma: float = 1.0):
    """Drop-in for reference loss.Affloss (loss.py:87-94): (gamma * photometric_3d, lamb_da * ncc_3d, sum); differentiable
    with respect to `warped`."""
    if not warped.is_cuda:
        raise RuntimeError("mireg.Affloss runs on the MI355X only; there is no CPU fallback")
    out = _AfflossFn.apply(warped, fixed, float(lamb_da), float(gamma))
    return out[0], out[1], out[0] + out[1]


def photometric_loss_3d(fixed: torch.Tensor, warped: torch.Tensor):
    """Drop-in for reference loss.photometric_loss_3d (loss.py:16-19)."""
    return Affloss(warped, fixed, 1.0, 1.0)[0]


def correlation_loss_3d(fixed: torch.Tensor, warped: torch.Tensor):
    """Drop-in for reference loss.correlation_loss_3d (loss.py:38-50)."""
    return Affloss(warped, fixed, 1.0, 1.0)[1]
