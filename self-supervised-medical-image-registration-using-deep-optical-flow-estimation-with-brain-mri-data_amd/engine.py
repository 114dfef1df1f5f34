"""Host-side engine primitives: NHWC views, conv descriptors and launch helpers over the C ABI.

Everything here is plumbing (pointer arithmetic, descriptor filling, launch order); every byte of
arithmetic happens in csrc/*.hip.  Layout rules:
  * activations / gradients are NHWC with a pixel stride `ld` (elements), channel counts padded to 8
    with zeros, so a producer can write into a channel slice of a concat buffer (no torch.cat);
  * Conv2d weights [Co][Ci][kh][kw] are packed once per optimizer step into
      FWD   pack  [Co][(ky,kx,ci_pad)]                 (forward of a conv, backward-data of a deconv)
      DGRAD packs [Ci][(ty,tx,co_pad)] per parity class  (backward-data of a conv, forward of a deconv)
    ConvTranspose2d weights [Cin][Cout][kh][kw] are the Conv2d weights of the adjoint conv, so a
    deconvolution is just the same layer object used the other way round.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

F32 = torch.float32
DT_F32, DT_BF16 = 0, 1


def rup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class ConvCls(ctypes.Structure):
    _fields_ = [("taps_y", ctypes.c_int), ("taps_x", ctypes.c_int), ("off_y", ctypes.c_int), ("off_x", ctypes.c_int),
                ("g_H", ctypes.c_int), ("g_W", ctypes.c_int), ("y_off_y", ctypes.c_int), ("y_off_x", ctypes.c_int),
                ("taps_z", ctypes.c_int), ("off_z", ctypes.c_int), ("g_D", ctypes.c_int), ("y_off_z", ctypes.c_int),
                ("w", ctypes.c_void_p), ("w_ld", ctypes.c_long), ("w_bytes", ctypes.c_long)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("x_ld", ctypes.c_long), ("x_H", ctypes.c_int), ("x_W", ctypes.c_int),
                ("x_C", ctypes.c_int), ("taps_y", ctypes.c_int), ("taps_x", ctypes.c_int),
                ("mul_y", ctypes.c_int), ("mul_x", ctypes.c_int), ("off_y", ctypes.c_int), ("off_x", ctypes.c_int),
                ("step_y", ctypes.c_int), ("step_x", ctypes.c_int),
                ("g_H", ctypes.c_int), ("g_W", ctypes.c_int), ("n_img", ctypes.c_int),
                ("w", ctypes.c_void_p), ("w_ld", ctypes.c_long), ("N", ctypes.c_int),
                ("y", ctypes.c_void_p), ("y_ld", ctypes.c_long), ("y_H", ctypes.c_int), ("y_W", ctypes.c_int),
                ("y_mul_y", ctypes.c_int), ("y_mul_x", ctypes.c_int), ("y_off_y", ctypes.c_int), ("y_off_x", ctypes.c_int),
                ("y32", ctypes.c_void_p), ("y32_ld", ctypes.c_long),
                ("bias", ctypes.c_void_p), ("slope", ctypes.c_float), ("accumulate", ctypes.c_int), ("dtype", ctypes.c_int),
                ("split_k", ctypes.c_int), ("slab", ctypes.c_void_p), ("x_bytes", ctypes.c_long), ("w_bytes", ctypes.c_long),
                ("n_cls", ctypes.c_int), ("cls", ConvCls * 8), ("slab_cls_stride", ctypes.c_long),
                ("x_D", ctypes.c_int), ("taps_z", ctypes.c_int), ("mul_z", ctypes.c_int), ("off_z", ctypes.c_int),
                ("step_z", ctypes.c_int), ("g_D", ctypes.c_int), ("y_D", ctypes.c_int), ("y_mul_z", ctypes.c_int),
                ("y_off_z", ctypes.c_int), ("tile_n", ctypes.c_int), ("stages", ctypes.c_int), ("slab_ld", ctypes.c_long),
                ("algo", ctypes.c_int), ("tile_m", ctypes.c_int)]


class PackClass(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("ld", ctypes.c_long), ("ky0", ctypes.c_int), ("kx0", ctypes.c_int),
                ("nty", ctypes.c_int), ("ntx", ctypes.c_int)]


class PackJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int),
                ("kh", ctypes.c_int), ("kw", ctypes.c_int), ("Cpad", ctypes.c_int), ("Cop", ctypes.c_int),
                ("ld", ctypes.c_long), ("stride", ctypes.c_int), ("nclass", ctypes.c_int), ("cls", PackClass * 4),
                ("nsplit", ctypes.c_int), ("accumulate", ctypes.c_int), ("unit0", ctypes.c_int), ("dunit0", ctypes.c_int)]


class WoptJob(ctypes.Structure):
    _fields_ = [("slab", ctypes.c_void_p), ("slab_stride", ctypes.c_long), ("nsplit", ctypes.c_int),
                ("g", ctypes.c_void_p), ("p", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("F", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int), ("taps", ctypes.c_int),
                ("Cpad", ctypes.c_int), ("ld", ctypes.c_long), ("runit0", ctypes.c_int), ("unit0", ctypes.c_int)]


class TailJob(ctypes.Structure):
    _fields_ = [("flow", ctypes.c_void_p), ("fsb", ctypes.c_long), ("fsc", ctypes.c_long), ("fsp", ctypes.c_long),
                ("moving_r", ctypes.c_void_p), ("fixed_r", ctypes.c_void_p), ("warped", ctypes.c_void_p),
                ("gflow", ctypes.c_void_p), ("gsb", ctypes.c_long), ("gsc", ctypes.c_long), ("gsp", ctypes.c_long),
                ("sums", ctypes.c_void_p), ("coef", ctypes.c_void_p),
                ("h", ctypes.c_int), ("w", ctypes.c_int), ("blk0", ctypes.c_int), ("pad_", ctypes.c_int)]


class ZeroJob(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("bytes", ctypes.c_long), ("unit0", ctypes.c_int), ("pad_", ctypes.c_int)]


def zero_many_table(bufs: Sequence[torch.Tensor], device):
    """(device table, n, units) for mireg_zero_many over whole tensors."""
    jobs, u = [], 0
    for b in bufs:
        nb = b.numel() * b.element_size()
        jobs.append(ZeroJob(b.data_ptr(), nb, u, 0))
        u += (nb + 16383) // 16384
    return upload_table(jobs, device), len(jobs), u


def zero_tensors(bufs: Sequence[torch.Tensor]) -> None:
    """Zero whole tensors with ONE mireg_zero_many launch (the tables are cached per set of buffers): the step's only fills."""
    key = (bufs[0].device.index,) + tuple((b.data_ptr(), b.numel() * b.element_size()) for b in bufs)   # device, (address, bytes)
    tab = _ZERO_TABLES.get(key)
    if tab is None:
        tab = _ZERO_TABLES[key] = zero_many_table(bufs, bufs[0].device)
    _lib.call("mireg_zero_many", tab[0].data_ptr(), tab[1], tab[2], _stream())


_ZERO_TABLES: Dict[tuple, tuple] = {}


class Pack3dJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p * 8), ("Co", ctypes.c_int), ("Ci", ctypes.c_int),
                ("Cop", ctypes.c_int), ("kd", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int),
                ("sz", ctypes.c_int), ("sy", ctypes.c_int), ("sx", ctypes.c_int), ("pz", ctypes.c_int), ("py", ctypes.c_int),
                ("px", ctypes.c_int), ("unit0", ctypes.c_int)]


class AdamJob(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_long)]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def assign_tiles(jobs: Sequence["PackJob"], unpack: bool) -> Tuple[int, int]:
    """Fill unit0 / dunit0 of every job; returns (total_units, total_dgrad_units) for the launch."""
    u = d = 0
    for j in jobs:
        c = j.Ci if unpack else j.Cpad
        j.unit0, j.dunit0 = u, d
        u += j.Co * ((c + 63) // 64)
        if not unpack:
            taps = sum(j.cls[i].nty * j.cls[i].ntx for i in range(j.nclass))
            d += taps * ((j.Cop + 63) // 64) * ((j.Ci + 63) // 64)
    return u, d


def run_pack(jobs: Sequence["PackJob"], code: int, device) -> None:
    """One-off pack of a few layers (tests / micro-benches)."""
    units, dunits = assign_tiles(jobs, False)
    tab = upload_table(jobs, device)
    _lib.call("mireg_pack_weights", tab.data_ptr(), len(jobs), units, dunits, code, _stream())


def run_unpack(jobs: Sequence["PackJob"], device) -> None:
    units, _ = assign_tiles(jobs, True)
    tab = upload_table(jobs, device)
    _lib.call("mireg_unpack_wgrad", tab.data_ptr(), len(jobs), units, _stream())


_TABLE_CACHE: Dict[tuple, torch.Tensor] = {}


def upload_table(jobs: Sequence[ctypes.Structure], device) -> torch.Tensor:
    """An array of POD job structs as a device byte tensor.  Tables are cached by content (they are read-only for the kernels and a
    step rebuilds the same ones over persistent buffers): the steady state issues no host-to-device copy, which is also what lets a
    whole step be captured into a hipGraph -- a pageable upload is not capturable, and that, not a fault in a kernel, is what stopped
    the 3-D step's capture in round 2 (scratch/capture3d.py)."""
    arr = (type(jobs[0]) * len(jobs))(*jobs)
    raw = bytes(memoryview(arr).cast("B"))
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else -1, type(jobs[0]).__name__, raw)
    tab = _TABLE_CACHE.get(key)
    if tab is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mireg: a new job table would have to be uploaded inside a hipGraph capture; run the step eagerly once "
                               "before capturing so that its tables exist (buffers must be persistent)")
        if len(_TABLE_CACHE) >= 8192:
            _TABLE_CACHE.clear()
        tab = _TABLE_CACHE[key] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    return tab


@dataclass
class View:
    """A channel slice [c0, c0+C) of an NHWC buffer of shape (B, H, W, ld)."""
    buf: torch.Tensor
    B: int
    H: int
    W: int
    C: int
    c0: int = 0

    @property
    def ld(self) -> int:
        return self.buf.shape[-1]

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + self.c0 * self.buf.element_size()

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W

    @property
    def bytes_left(self) -> int:
        """Readable bytes from this view's first element to the end of its buffer."""
        return (self.buf.numel() - self.c0) * self.buf.element_size()

    def slice(self, c0: int, C: int) -> "View":
        assert c0 >= 0 and self.c0 + c0 + C <= self.ld
        return View(self.buf, self.B, self.H, self.W, C, self.c0 + c0)

    def nchw(self) -> torch.Tensor:
        """Logical (B, C, H, W) tensor aliasing this view (channels_last-style strides)."""
        return self.buf.view(self.B, self.H, self.W, self.ld)[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)


class Workspace:
    """Owns NHWC buffers (zero-initialised so pad channels stay zero forever) and scratch slabs."""

    def __init__(self, device, dtype: torch.dtype):
        self.device, self.dtype = device, dtype
        self.code = DT_BF16 if dtype == torch.bfloat16 else DT_F32
        self.scratch: Optional[torch.Tensor] = None
        self.scratch_elems = 0
        # launch-shape autotuning (tile width x split-K per contraction site): `tuning` is on during the trainer's
        # discarded tuning pass, `tuned` maps a launch site to its measured-best (tile_n, split_k)
        self.wgrad_stages = 3                              # mireg_conv_desc.stages of the backward-weights GEMM (3 or 4)
        self.colsum_ws: Optional[torch.Tensor] = None      # row-segment partials of the bias-gradient column sums
        self._tuning = False
        self.tuned: Dict[tuple, Tuple[int, int]] = {}
        self.tuned_wgrad: Dict[str, list] = {}             # layer name -> measured [split-K, algo] of its backward-weights GEMM

    def new(self, B: int, H: int, W: int, C: int, dtype: Optional[torch.dtype] = None, pad: int = 8) -> View:
        buf = torch.zeros(B, H, W, rup(C, pad), device=self.device, dtype=dtype or self.dtype)
        return View(buf, B, H, W, C, 0)

    # `tuning`: on during a discarded tuning pass -- the trainer's own (per workspace) or mireg.autotune's (TUNE_ALL, every workspace)
    TUNE_ALL = False

    @property
    def tuning(self) -> bool:
        return self._tuning or Workspace.TUNE_ALL

    @tuning.setter
    def tuning(self, on: bool) -> None:
        self._tuning = bool(on)

    def save_tuning(self, path: str) -> None:
        import json
        with open(path, "w") as f:
            json.dump({"sites": [[list(k), list(v)] for k, v in self.tuned.items()], "wgrad": self.tuned_wgrad}, f)

    def load_tuning(self, path: str) -> None:
        import json
        d = json.load(open(path))
        self.tuned = {tuple(k): (tuple(v) if len(v) == 4 else (v[0], v[1], 1, 0)) for k, v in d["sites"]}
        self.tuned_wgrad = dict(d["wgrad"])

    def need_scratch(self, elems: int) -> None:
        self.scratch_elems = max(self.scratch_elems, elems)

    def get_scratch(self) -> torch.Tensor:
        if self.scratch is None or self.scratch.numel() < self.scratch_elems:
            self.scratch = torch.empty(max(self.scratch_elems, 1), device=self.device, dtype=F32)
        return self.scratch


NUM_CU = 256


FORCE_TILE_N = int(os.environ.get('MIREG_TILE_N', '0'))   # experiments only
USE_STEM = True
WGRAD_ALGO = int(os.environ.get('MIREG_WGRAD_ALGO', '0'))         # tests / A-B runs: 0 auto, 1 ring kernel, 2 halo kernel required
FORCE_ALGO = None      # tests only: (algo, tile_m[, tile_n]) for every mireg_conv_gemm launch
THIN_GEMM_ROWS = int(os.environ.get('MIREG_THIN_GEMM_ROWS', '1024'))   # heads with at least this many pixels run as 1x1 GEMMs
USE_TINY = os.environ.get('MIREG_NO_TINY', '0') != '1'   # experiments / A-B runs only
# which forms of the 2->2 upsamplers take the pixel-parallel kernels: 1 forward, 2 backward-data, 4 backward-weights.  Round 2 kept
# backward-data off (mask 5) because two identically seeded trainers diverged with it; round 3 traced that to packed-fp32 VALU
# code in the kernel (csrc/Makefile): fixed at the compiler flag, all three forms are on.
TINY_MASK = int(os.environ.get('MIREG_TINY_MASK', '7'))
USE_HALO = os.environ.get('MIREG_NO_HALO', '0') != '1'   # experiments / A-B runs only
USE_WIDE = os.environ.get('MIREG_NO_WIDE', '0') != '1'   # experiments / A-B runs only: the 256-pixel 8-wave tile as a tuner candidate
USE_THIN = True     # module switch (tests compare the thin kernels with the GEMM path)


class LaunchProfiler:
    """Optional per-launch timing of the MFMA contractions (bench.py roofline leg): brackets every
    mireg_conv_gemm / mireg_conv_wgrad launch with events on the launch stream and books its algorithmic FLOPs."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (kernel family, flops, start event, stop event, tag)
        self.byte_records = []     # same, for the HBM-bound kernels: (family, algorithmic bytes, start, stop, tag)

    def launch(self, fn: str, desc, family: str, flops: float, tag: str = "") -> None:
        if not self.enabled:
            _lib.call(fn, ctypes.byref(desc), _stream())
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(fn, ctypes.byref(desc), _stream())
        b.record()
        self.records.append((family, flops, a, b, tag))

    def call(self, family: str, amount: float, tag: str, fn: str, *args, unit: str = "FLOP") -> None:
        """Same bookkeeping for a plain-argument entry point; unit "B" books algorithmic HBM bytes instead of FLOPs."""
        if not self.enabled:
            _lib.call(fn, *args)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(fn, *args)
        b.record()
        (self.records if unit == "FLOP" else self.byte_records).append((family, amount, a, b, tag))

    def summary(self, bytes_: bool = False) -> Dict[str, dict]:
        torch.cuda.synchronize()
        out: Dict[str, dict] = {}
        for fam, fl, a, b, _ in (self.byte_records if bytes_ else self.records):
            d = out.setdefault(fam, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += fl
            d["ms"] += a.elapsed_time(b)
        return out


PROFILER = LaunchProfiler()


def _split_for(tiles: int, nk: int, cap: int = 32) -> int:
    if tiles >= NUM_CU or nk < 8:
        return 1
    return max(1, min((2 * NUM_CU + tiles - 1) // tiles, nk // 4, cap))


class ConvLayer:
    """One Conv2d-shaped weight W[Co][Ci][kh][kw] (stride s, padding p, dilation d) and its packs.

    used as a convolution      : fwd = FWD form,  bwd-data = DGRAD form, bwd-weights = WGRAD(x, dy)
    used as a deconvolution    : fwd = DGRAD form, bwd-data = FWD form,  bwd-weights = WGRAD(dy, x)
    (ConvTranspose2d(Cin, Cout) weights are [Cin][Cout][kh][kw] == this class with Co=Cin, Ci=Cout.)
    """

    def __init__(self, name: str, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int, dil: int,
                 ws: Workspace):
        self.name, self.weight, self.bias = name, weight, bias
        self.Co, self.Ci, self.kh, self.kw = weight.shape
        self.s, self.p, self.d = stride, pad, dil
        self.ws = ws
        self.Cip, self.Cop = rup(self.Ci, 8), rup(self.Co, 8)
        dev, dt = ws.device, ws.dtype
        self.Kf = self.kh * self.kw * self.Cip
        self.packF = torch.zeros(self.Co, self.Kf, device=dev, dtype=dt)
        # DGRAD packs, one per parity class of the (larger) image
        self.classes = []
        for py in range(self.s):
            for px in range(self.s):
                ky0, kx0 = (py + self.p) % self.s, (px + self.p) % self.s
                nty = (self.kh - ky0 + self.s - 1) // self.s if self.kh > ky0 else 0
                ntx = (self.kw - kx0 + self.s - 1) // self.s if self.kw > kx0 else 0
                cy, cx = (py + self.p - ky0) // self.s, (px + self.p - kx0) // self.s
                K = nty * ntx * self.Cop
                pack = torch.zeros(self.Ci, max(K, 8), device=dev, dtype=dt)
                self.classes.append(dict(py=py, px=px, ky0=ky0, kx0=kx0, nty=nty, ntx=ntx, cy=cy, cx=cx, K=K, pack=pack))
        self.wgrad_slab: Optional[torch.Tensor] = None
        self.wgrad_split = 1
        self.wgrad_algo = 0                                 # mireg_conv_desc.algo of the backward-weights launch (0 auto, 1 ring, 2 halo)
        self.n_slots = 1
        self.grad_w: Optional[torch.Tensor] = None
        self.grad_b: Optional[torch.Tensor] = None
        # two-channel 3x3 heads (predict_flow): vector-ALU streaming kernels instead of a 2-column GEMM
        self.thin = USE_THIN and (self.Co, self.kh, self.kw, self.s, self.p, self.d) == (2, 3, 3, 1, 1, 1)
        self.thin_gemm = False                             # decided in plan_wgrad (forward decides per call)
        # 2 -> 2 channel ConvTranspose2d(4, 2, 1) upsamplers: pixel-parallel kernels on the master weight (csrc/thin_conv.hip)
        self.tiny = USE_THIN and USE_TINY and (self.Co, self.Ci, self.kh, self.kw, self.s, self.p, self.d) == (2, 2, 4, 4, 2, 1, 1)
        # 1-2 channel 7x7/s2 input convolutions: patch-staged kernels with K = (ky, kx, ci) (stem_conv.hip)
        self.stem = (USE_STEM and ws.code == DT_BF16 and (self.kh, self.kw, self.s, self.p, self.d) == (7, 7, 2, 3, 1)
                     and self.Ci <= 2 and self.Co == 64)
        self.gpack: Optional[torch.Tensor] = None       # packed-domain gradient [Co][Kf] (view of the trainer's flat buffer)

    # ---- pack jobs ----------------------------------------------------------------------------
    def pack_jobs(self) -> List[PackJob]:
        j = PackJob()
        j.src, j.dst = self.weight.data_ptr(), self.packF.data_ptr()
        j.Co, j.Ci, j.kh, j.kw = self.Co, self.Ci, self.kh, self.kw
        j.Cpad, j.Cop, j.ld, j.stride = self.Cip, self.Cop, self.Kf, self.s
        live = [c for c in self.classes if c["K"] > 0]
        j.nclass = len(live)
        for i, c in enumerate(live):
            j.cls[i] = PackClass(c["pack"].data_ptr(), c["pack"].shape[1], c["ky0"], c["kx0"], c["nty"], c["ntx"])
        return [j]

    def bind_gpack(self, gpack: torch.Tensor) -> None:
        """Packed-domain gradient target.  A layer whose wgrad needs no split writes it directly (no reduce)."""
        self.gpack = gpack
        if self.wgrad_slab is not None and self.wgrad_split * self.n_slots == 1:
            self.wgrad_slab = gpack.view(1, self.Co, self.Kf)

    def wopt_job(self, m: int = 0, v: int = 0) -> WoptJob:
        """Reduce (+ optimizer when m, v are given) job of this layer in the packed domain."""
        assert self.wgrad_slab is not None and self.gpack is not None, self.name
        j = WoptJob()
        direct = self.wgrad_slab.data_ptr() == self.gpack.data_ptr()
        j.slab, j.slab_stride, j.nsplit = self.wgrad_slab.data_ptr(), self.Co * self.Kf, 0 if direct else self.wgrad_split * self.n_slots
        j.g, j.p, j.m, j.v, j.F = self.gpack.data_ptr(), self.weight.data_ptr(), m, v, self.packF.data_ptr()
        j.Co, j.Ci, j.taps, j.Cpad, j.ld = self.Co, self.Ci, self.kh * self.kw, self.Cip, self.Kf
        return j

    def unpack_job(self, accumulate: bool = False) -> PackJob:
        if self.grad_w is None:
            self.grad_w = torch.zeros_like(self.weight, dtype=F32)
        assert self.wgrad_slab is not None
        j = PackJob()
        j.src, j.dst = self.wgrad_slab.data_ptr(), self.grad_w.data_ptr()
        j.Co, j.Ci, j.kh, j.kw = self.Co, self.Ci, self.kh, self.kw
        j.Cpad, j.Cop, j.ld, j.stride, j.nclass = self.Cip, self.Cop, self.Kf, self.s, 0
        j.nsplit, j.accumulate = self.wgrad_split * self.n_slots, int(accumulate)
        return j

    # ---- launches -----------------------------------------------------------------------------
    def _finish(self, d: ConvDesc, M: int, N: int, K: int, allow_split: bool, ncls: int = 1, site: str = "") -> None:
        """Launch shape of one contraction site: kernel (halo-staged / ring), tile and split-K.
        M = rows of the largest class, K = smallest class K (bounds the split)."""
        bk = 32 if self.ws.code == DT_BF16 else 16
        nk = (K + bk - 1) // bk
        d.dtype = self.ws.code

        def apply(bn: int, split: int, algo: int = 1, tm: int = 0) -> None:
            d.tile_n = bn if (N > 64 or algo == 3) else 0
            d.split_k, d.algo, d.tile_m = split, algo, tm
            d.slab = None
            if split > 1:
                d.slab_cls_stride = split * M * N
                self.ws.need_scratch(ncls * split * M * N)
                d.slab = self.ws.get_scratch().data_ptr()

        def tiles_for(bn: int, bm: int = 128) -> int:
            return ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * ncls
        bn = 128 if N > 64 else (64 if N > 32 else 32)
        if FORCE_TILE_N and N > 64:
            bn = FORCE_TILE_N
        key = (self.name, site, M, N, K, ncls)
        if FORCE_ALGO is not None:                          # tests / A-B runs: (algo, tile_m[, tile_n]); raises if not applicable
            if FORCE_ALGO[0] == 3:                          # (3, 256, tile_n[, split]): the 256-pixel 8-wave tile of conv_wide.hip
                apply(FORCE_ALGO[2] if len(FORCE_ALGO) > 2 else (256 if N > 128 else 128),
                      FORCE_ALGO[3] if (len(FORCE_ALGO) > 3 and allow_split) else 1, 3, 256)
                return
            apply(FORCE_ALGO[2] if len(FORCE_ALGO) > 2 else bn, 1 if FORCE_ALGO[0] == 2 else (_split_for(tiles_for(bn), nk) if allow_split else 1),
                  FORCE_ALGO[0], FORCE_ALGO[1])
            return
        if key in self.ws.tuned:
            apply(*self.ws.tuned[key])
            return
        # halo-staged kernel (conv_halo.hip): unit-stride gathers on 16/32/64-wide grids
        d.split_k, d.algo, d.tile_m = 1, 0, 0
        htiles = (ctypes.c_long * 2)()
        halo = USE_HALO and bool(_lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), htiles))
        if halo and N >= 64:                               # below 64 columns only as a measured candidate (half-empty 64-column tiles)
            tn = (N + bn - 1) // bn
            tm = 256 if (htiles[1] and htiles[1] * tn * ncls >= 224) else (128 if htiles[0] else 256)
            heur = (bn, 1, 2, tm)
        else:
            heur = (bn, _split_for(tiles_for(bn), nk) if allow_split else 1, 1, 0)
        if not (self.ws.tuning and allow_split and site):
            apply(*heur)
            return
        # measured choice: every (kernel, tile, split) whose grid is neither starved nor absurdly oversubscribed
        cands = {heur}
        for b in ((128, 64) if N > 64 else (bn,)):
            t = tiles_for(b)
            for sp in (1, 2, 3, 4, 6, 8, 11, 16, 22, 32):
                if sp <= max(nk // 4, 1) and 96 <= t * sp <= 2304 and ncls * sp * M * N <= (1 << 26):
                    cands.add((b, sp, 1, 0))
            if halo:
                for i, tm in enumerate((128, 256)):
                    if htiles[i]:
                        cands.add((max(b, 64) if N <= 64 else b, 1, 2, tm))
        # the 256-pixel 8-wave tile (conv_wide.hip): 256 or 128 columns, split-K where the grid would leave CUs idle
        if USE_WIDE and self.ws.code == DT_BF16 and bool(_lib.lib().mireg_conv_wide_eligible(ctypes.byref(d), None)):
            nk64 = (K + 63) // 64
            for b in ((256, 128) if N > 128 else (128,)):
                t = tiles_for(b, 256)
                for sp in (1, 2, 3, 4, 6, 8):
                    if sp <= max(nk64 // 4, 1) and 64 <= t * sp <= 1536 and ncls * sp * M * N <= (1 << 26):
                        cands.add((b, sp, 3, 256))
        best, best_t = heur, float("inf")
        st = _stream()
        for c in sorted(cands):
            apply(*c)
            _lib.call("mireg_conv_gemm", ctypes.byref(d), st)
            t = float("inf")
            for _ in range(2):                              # best of two 3-launch timings: candidates are often within 2-3 %
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    _lib.call("mireg_conv_gemm", ctypes.byref(d), st)
                b_.record()
                b_.synchronize()
                t = min(t, a.elapsed_time(b_))
            if t < best_t:
                best, best_t = c, t
        self.ws.tuned[key] = best
        apply(*best)

    def tiny_bwd_data_ok(self, x: View, y: View) -> bool:
        """True when run_fwd_form(x, y) takes the pixel-parallel 2->2 upsampler kernel (which can fuse a planar addend)."""
        return bool(self.tiny and (TINY_MASK & 2) and (x.H, x.W) == (2 * y.H, 2 * y.W))

    def run_fwd_form(self, x: View, y: Optional[View], *, y32: Optional[View] = None, slope: float = 1.0,
                     bias: bool = True, accumulate: bool = False, add_nchw: Optional[torch.Tensor] = None) -> None:
        """y[(oy,ox)] = act(sum_{ky,kx,ci} x[oy*s-p+ky*d, ox*s-p+kx*d, ci] W[co][ci][ky][kx] + b)."""
        assert x.C <= self.Cip and x.c0 + self.Cip <= x.ld, (self.name, x.C, self.Ci, x.ld)
        Ho = (x.H + 2 * self.p - self.d * (self.kh - 1) - 1) // self.s + 1
        Wo = (x.W + 2 * self.p - self.d * (self.kw - 1) - 1) // self.s + 1
        out = y if y is not None else y32
        assert (out.H, out.W) == (Ho, Wo), (self.name, (out.H, out.W), (Ho, Wo))
        if self.tiny and (TINY_MASK & 2) and y is not None and y32 is None and slope == 1.0 and not bias and (x.H, x.W) == (2 * Ho, 2 * Wo):
            _lib.call("mireg_tiny_deconv_bwd_data", x.ptr, x.ld, self.weight.data_ptr(), y.ptr, y.ld, int(accumulate),
                      add_nchw.data_ptr() if add_nchw is not None else None, x.B, Ho, Wo,
                      self.ws.code, _stream())
            return
        assert add_nchw is None, "add_nchw: only the 2->2 upsampler kernels fuse a planar addend"
        if self.stem and y is not None and y32 is None and not accumulate:
            PROFILER.call("stem_conv_fwd", 2.0 * x.B * Ho * Wo * self.Co * 49 * self.Ci, f"{self.name}:stem-fwd",
                          "mireg_stem_conv_fwd", x.ptr, x.ld, self.packF.data_ptr(), self.Kf, self.Cip,
                          self.bias.data_ptr() if (bias and self.bias is not None) else None, slope, y.ptr, y.ld,
                          x.B, x.H, x.W, self.Ci, self.Co, _stream())
            return
        if self.thin and slope == 1.0 and not accumulate and x.rows >= THIN_GEMM_ROWS:
            self._thin_gemm_fwd(x, y, y32, bias)
            return
        if self.thin and slope == 1.0 and not accumulate:
            PROFILER.call("thin_conv_fwd", 2.0 * x.B * Ho * Wo * 2 * 9 * self.Ci, f"{self.name}:thin-fwd",
                          "mireg_thin_conv_fwd", x.ptr, x.ld, self.packF.data_ptr(), self.Kf,
                          self.bias.data_ptr() if (bias and self.bias is not None) else None,
                          y.ptr if y is not None else None, y.ld if y is not None else 0,
                          y32.ptr if y32 is not None else None, y32.ld if y32 is not None else 0,
                          x.B, x.H, x.W, self.Cip, self.ws.code, _stream())
            return
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C = x.ptr, x.ld, x.H, x.W, self.Cip
        d.taps_y, d.taps_x = self.kh, self.kw
        d.mul_y = d.mul_x = self.s
        d.off_y = d.off_x = -self.p
        d.step_y = d.step_x = self.d
        d.g_H, d.g_W, d.n_img = Ho, Wo, x.B
        d.w, d.w_ld, d.N = self.packF.data_ptr(), self.Kf, self.Co
        d.x_bytes, d.w_bytes = x.bytes_left, self.packF.numel() * self.packF.element_size()
        self._fill_out(d, y, y32, Ho, Wo, 1, 1, 0, 0)
        d.bias = self.bias.data_ptr() if (bias and self.bias is not None) else None
        d.slope, d.accumulate = slope, int(accumulate)
        self._finish(d, x.B * Ho * Wo, self.Co, self.Kf, True, site="fwd" + ("+acc" if accumulate else ""))
        PROFILER.launch("mireg_conv_gemm", d, self._family(self.Co, d.split_k, d.tile_n, d.algo, d.tile_m),
                        2.0 * x.B * Ho * Wo * self.Co * self.kh * self.kw * self.Ci,
                        f"{self.name}:fwd M={x.B * Ho * Wo} N={self.Co} K={self.Kf} split={d.split_k}")

    @staticmethod
    def _family(N: int, split: int, tile_n: int = 0, algo: int = 1, tile_m: int = 0) -> str:
        bn = tile_n or (128 if N > 64 else (64 if N > 32 else 32))
        if algo == 2:
            return f"conv_halo_kernel<{tile_m},{max(bn, 64)}>"
        if algo == 3:
            return f"conv_wide_kernel<256,{bn}>" + ("+splitk" if split > 1 else "")
        return f"conv_gemm_kernel<128,{bn}>" + ("+splitk" if split > 1 else "")

    @staticmethod
    def _fill_out(d, y, y32, yH, yW, mul_y, mul_x, off_y, off_x):
        if y is not None:
            d.y, d.y_ld = y.ptr, y.ld
        if y32 is not None:
            d.y32, d.y32_ld = y32.ptr, y32.ld
        d.y_H, d.y_W, d.y_mul_y, d.y_mul_x, d.y_off_y, d.y_off_x = yH, yW, mul_y, mul_x, off_y, off_x

    def run_dgrad_form(self, g: View, out: Optional[View], *, y32: Optional[View] = None, slope: float = 1.0,
                       bias: bool = False, accumulate: bool = False) -> None:
        """out[(iy,ix), ci] = act(sum_{ky,kx,co} g[(iy+p-ky*d)/s, (ix+p-kx*d)/s, co] W[co][ci][ky][kx] + b[ci])
        (only exact divisions contribute).  out has the LARGER spatial size."""
        assert g.C <= self.Cop and g.c0 + self.Cop <= g.ld, (self.name, g.C, self.Co, g.ld)
        o = out if out is not None else y32
        if self.tiny and (TINY_MASK & 1) and o is not None and slope == 1.0 and not accumulate and (o.H, o.W) == (2 * g.H, 2 * g.W):
            _lib.call("mireg_tiny_deconv_fwd", g.ptr, g.ld, self.weight.data_ptr(),
                      self.bias.data_ptr() if (bias and self.bias is not None) else None,
                      out.ptr if out is not None else None, out.ld if out is not None else 0,
                      y32.ptr if y32 is not None else None, y32.ld if y32 is not None else 0, g.B, g.H, g.W,
                      self.ws.code, _stream())
            return
        if self.thin and out is not None and y32 is None and slope == 1.0 and not bias:
            PROFILER.call("thin_conv_dgrad", 2.0 * g.B * g.H * g.W * 2 * 9 * self.Ci, f"{self.name}:thin-dgrad",
                          "mireg_thin_conv_dgrad", g.ptr, g.ld, self.packF.data_ptr(), self.Kf, out.ptr, out.ld,
                          int(accumulate), g.B, g.H, g.W, self.Cip, self.ws.code, _stream())
            return
        live = []
        for c in self.classes:
            gH = (o.H - c["py"] + self.s - 1) // self.s
            gW = (o.W - c["px"] + self.s - 1) // self.s
            if gH > 0 and gW > 0:
                assert c["K"] > 0, "kernel smaller than stride is not supported"
                live.append((c, gH, gW))
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C = g.ptr, g.ld, g.H, g.W, self.Cop
        d.x_bytes = g.bytes_left
        d.mul_y = d.mul_x = 1
        d.step_y = d.step_x = -self.d if self.s == 1 else -1
        d.n_img, d.N = g.B, self.Ci
        d.bias = self.bias.data_ptr() if (bias and self.bias is not None) else None
        d.slope, d.accumulate = slope, int(accumulate)
        flops = 0.0
        for i, (c, gH, gW) in enumerate(live):
            k = ConvCls()
            k.taps_y, k.taps_x = c["nty"], c["ntx"]
            k.off_y, k.off_x = (self.p, self.p) if self.s == 1 else (c["cy"], c["cx"])
            k.g_H, k.g_W, k.y_off_y, k.y_off_x = gH, gW, c["py"], c["px"]
            k.w, k.w_ld, k.w_bytes = c["pack"].data_ptr(), c["pack"].shape[1], c["pack"].numel() * c["pack"].element_size()
            d.cls[i] = k
            flops += 2.0 * g.B * gH * gW * self.Ci * c["nty"] * c["ntx"] * self.Co
        c0, gH0, gW0 = live[0]
        # class 0 mirrored in the top-level fields (single-class launches read only those)
        d.taps_y, d.taps_x, d.off_y, d.off_x = d.cls[0].taps_y, d.cls[0].taps_x, d.cls[0].off_y, d.cls[0].off_x
        d.g_H, d.g_W = gH0, gW0
        d.w, d.w_ld, d.w_bytes = d.cls[0].w, d.cls[0].w_ld, d.cls[0].w_bytes
        self._fill_out(d, out, y32, o.H, o.W, self.s, self.s, c0["py"], c0["px"])
        d.n_cls = len(live) if len(live) > 1 else 0
        Mmax = max(g.B * gh * gw for _, gh, gw in live)
        Kmin = min(c["K"] for c, _, _ in live)
        self._finish(d, Mmax, self.Ci, Kmin, True, len(live), site="dgrad" + ("+acc" if accumulate else ""))
        PROFILER.launch("mireg_conv_gemm", d, self._family(self.Ci, d.split_k, d.tile_n, d.algo, d.tile_m), flops,
                        f"{self.name}:dgrad M={Mmax}x{len(live)} N={self.Ci} K={Kmin} split={d.split_k}")

    def plan_wgrad(self, x: View, dy: View) -> None:
        """Size the persistent split-K slab for dW[co][(ky,kx,ci_pad)] = sum_pix dy[pix][co] x[pix@tap][ci]."""
        bk = 32 if self.ws.code == DT_BF16 else 16
        tiles = ((self.Co + 127) // 128) * ((self.Kf + 127) // 128)
        nk = (dy.rows + bk - 1) // bk
        self.thin_gemm = self.thin and dy.rows >= THIN_GEMM_ROWS
        if self.tiny and (TINY_MASK & 4):
            self.wgrad_split = _lib.lib().mireg_tiny_deconv_blocks(dy.B, dy.H, dy.W)
        elif self.thin_gemm:                                   # 1x1 backward-weights GEMM on (dz, x): 128-column tiles x pixel splits
            self.wgrad_split = max(1, min(512 // ((self.Cip + 127) // 128), nk // 8, 192))
        elif self.thin:
            self.wgrad_split = _lib.lib().mireg_thin_conv_wgrad_tiles(dy.B, dy.H, dy.W, self.Cip, self.ws.code, None)
        elif self.stem:
            self.wgrad_split = _lib.lib().mireg_stem_conv_blocks(x.B, x.H, x.W)
        elif self._wgrad_key(x, dy) in self.ws.tuned_wgrad:
            tw = self.ws.tuned_wgrad[self._wgrad_key(x, dy)]
            self.wgrad_split, self.wgrad_algo = (tw, 0) if isinstance(tw, int) else (tw[0], tw[1])
            if self.wgrad_algo in (2, 3):                     # a cached kernel choice must still apply to these operands
                probe = self._wgrad_desc(x, dy, 1, 0, 0)
                ok = (_lib.lib().mireg_conv_wgrad_halo_eligible if self.wgrad_algo == 2 else _lib.lib().mireg_conv_wgrad_wide_eligible)
                if not ok(ctypes.byref(probe)):
                    self.wgrad_algo = 0
            self.wgrad_split = max(1, min(self.wgrad_split, max(nk // 8, 1)))
            self._wgrad_tuned = True
        else:
            self.wgrad_split = 1 if tiles >= NUM_CU else max(1, min((3 * NUM_CU) // tiles, max(nk // 8, 1), 192))   # <= 768 resident WGs
        if self.gpack is not None and self.n_slots * self.wgrad_split == 1:
            self.wgrad_slab = self.gpack.view(1, self.Co, self.Kf)
        else:
            self.wgrad_slab = torch.zeros(self.n_slots * self.wgrad_split, self.Co, self.Kf, device=self.ws.device, dtype=F32)
        if self.grad_w is None and self.gpack is None:
            self.grad_w = torch.zeros_like(self.weight, dtype=F32)
        if self.bias is not None and self.grad_b is None:
            self.grad_b = torch.zeros_like(self.bias, dtype=F32)

    def _wgrad_key(self, x: View, dy: View) -> str:
        """Cache key of a measured backward-weights launch shape: the layer AND the operands it was measured on."""
        return f"{self.name}|{dy.B}x{dy.H}x{dy.W}|{x.H}x{x.W}|{self.ws.code}"

    def _wgrad_desc(self, x: View, dy: View, split: int, slab_ptr: int, algo: Optional[int] = None) -> ConvDesc:
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C = x.ptr, x.ld, x.H, x.W, self.Cip
        d.taps_y, d.taps_x = self.kh, self.kw
        d.mul_y = d.mul_x = self.s
        d.off_y = d.off_x = -self.p
        d.step_y = d.step_x = self.d
        d.g_H, d.g_W, d.n_img = dy.H, dy.W, dy.B
        d.y, d.y_ld, d.N = dy.ptr, dy.ld, self.Co
        d.split_k, d.dtype = split, self.ws.code
        d.stages = self.ws.wgrad_stages
        d.slab = slab_ptr
        d.x_bytes, d.w_bytes = x.bytes_left, dy.bytes_left          # w_bytes carries the dy extent in WGRAD mode
        d.algo = (WGRAD_ALGO or self.wgrad_algo) if algo is None else algo
        return d

    def _tune_wgrad(self, x: View, dy: View) -> None:
        """Measured (kernel, split-K) of the backward-weights GEMM: stand-alone time + the time to write and re-read its fp32
        slabs.  (A chip-time model that favoured few-workgroup launches -- the launch shares the chip with the backward-data
        chain -- was measured and lost 10-35 % of the step: the two streams do not fill each other's gaps, the step is the sum
        of the kernels' stand-alone times, profiles/README.md round 2.)"""
        bk = 32 if self.ws.code == DT_BF16 else 16
        elems = self.Co * self.Kf
        nk = (dy.rows + bk - 1) // bk
        probe = self._wgrad_desc(x, dy, 1, 0)
        probe.algo = 0
        halo = WGRAD_ALGO != 1 and bool(_lib.lib().mireg_conv_wgrad_halo_eligible(ctypes.byref(probe)))
        cands = []
        if halo:
            tiles = ((self.Co + 127) // 128) * ((self.Cip + 31) // 32) * self.s * self.s
            cands += [(sp, 2) for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32) if sp <= max(dy.rows // 512, 1) and 96 <= tiles * sp <= 1536]
        if USE_WIDE and WGRAD_ALGO in (0, 3) and bool(_lib.lib().mireg_conv_wgrad_wide_eligible(ctypes.byref(probe))):
            tiles = ((self.Co + 255) // 256) * ((self.Kf + 255) // 256)     # the 256 x 256 8-wave tile (conv_wgrad_wide.hip)
            cands += [(sp, 3) for sp in (1, 2, 3, 4, 6, 8, 11, 16, 22, 32, 44, 64) if sp <= max(nk // 8, 1) and 64 <= tiles * sp <= 1024]
        if WGRAD_ALGO not in (2, 3):
            tiles = ((self.Co + 127) // 128) * ((self.Kf + 127) // 128)
            # one- and two-tile launches (PWC's 16- and 32-channel pyramid layers over 128x128 maps) get all their parallelism from
            # the pixel split: up to 768 ways
            cands += [(sp, 1) for sp in (1, 2, 3, 4, 6, 8, 11, 16, 22, 32, 44, 64, 96, 128, 192, 256, 384, 512, 768)
                      if sp <= max(nk // 8, 1) and 128 <= tiles * sp <= 2304]
        cands = [c for c in cands if c[0] * elems <= (1 << 26)] or [(self.wgrad_split, self.wgrad_algo)]
        tmp = torch.empty(max(c[0] for c in cands) * elems, device=self.ws.device, dtype=F32)
        st = _stream()
        best, best_c = cands[0], float("inf")
        for sp, algo in cands:
            d = self._wgrad_desc(x, dy, sp, tmp.data_ptr(), algo)
            _lib.call("mireg_conv_wgrad", ctypes.byref(d), st)
            t = float("inf")
            for _ in range(2):                              # best of two 3-launch timings
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    _lib.call("mireg_conv_wgrad", ctypes.byref(d), st)
                b_.record()
                b_.synchronize()
                t = min(t, a.elapsed_time(b_))
            c = t / 3 + (sp * elems * 4 / 3.0e9 if sp > 1 else 0.0)                            # ms; slabs re-read at ~3 TB/s
            if c < best_c:
                best, best_c = (sp, algo), c
        self._wgrad_tuned = True
        self.ws.tuned_wgrad[self._wgrad_key(x, dy)] = list(best)
        self.wgrad_algo = best[1]
        if best[0] != self.wgrad_split:
            self.wgrad_split = best[0]
            if self.gpack is not None and self.n_slots * best[0] == 1:
                self.wgrad_slab = self.gpack.view(1, self.Co, self.Kf)
            else:
                self.wgrad_slab = torch.zeros(self.n_slots * best[0], self.Co, self.Kf, device=self.ws.device, dtype=F32)

    def run_wgrad(self, x: View, dy: View, slot: int = 0) -> None:
        """x: tensor in this conv's INPUT space (C = Ci), dy: tensor in its OUTPUT space (C = Co).
        slot selects the slab group of a weight-shared layer (the unpack sums all groups)."""
        if self.wgrad_slab is None:
            self.plan_wgrad(x, dy)
        assert x.C <= self.Cip and x.c0 + self.Cip <= x.ld and dy.c0 + rup(self.Co, 8) <= dy.ld, self.name
        if self.tiny and (TINY_MASK & 4):
            _lib.call("mireg_tiny_deconv_bwd_weights", x.ptr, x.ld, dy.ptr, dy.ld, self.wgrad_slab[slot * self.wgrad_split].data_ptr(),
                      self.wgrad_split, self.Cip, dy.B, dy.H, dy.W, self.ws.code, _stream())
            return
        if self.thin and self.thin_gemm:
            self._thin_gemm_wgrad(x, dy, slot)
            return
        if self.thin:
            PROFILER.call("thin_conv_wgrad", 2.0 * dy.rows * 2 * 9 * self.Ci, f"{self.name}:thin-wgrad",
                          "mireg_thin_conv_wgrad", x.ptr, x.ld, dy.ptr, dy.ld,
                          self.wgrad_slab[slot * self.wgrad_split].data_ptr(), self.wgrad_split, dy.B, dy.H, dy.W,
                          self.Cip, self.ws.code, _stream())
            return
        if self.stem:
            PROFILER.call("stem_conv_wgrad", 2.0 * dy.rows * self.Co * 49 * self.Ci, f"{self.name}:stem-wgrad",
                          "mireg_stem_conv_wgrad", x.ptr, x.ld, dy.ptr, dy.ld, self.wgrad_slab[slot * self.wgrad_split].data_ptr(),
                          self.Kf, self.Cip, self.wgrad_split, x.B, x.H, x.W, self.Ci, self.Co, _stream())
            return
        if self.ws.tuning and not getattr(self, "_wgrad_tuned", False):
            self._tune_wgrad(x, dy)
        d = self._wgrad_desc(x, dy, self.wgrad_split, self.wgrad_slab[slot * self.wgrad_split].data_ptr())
        halo = d.algo not in (1, 3) and bool(_lib.lib().mireg_conv_wgrad_halo_eligible(ctypes.byref(d)))
        PROFILER.launch("mireg_conv_wgrad", d, "conv_wgrad_wide_kernel<256,256>" if d.algo == 3 else ("conv_wgrad_halo_kernel" if halo else "conv_wgrad_kernel<128,128>"),
                        2.0 * dy.rows * self.Co * self.kh * self.kw * self.Ci,
                        f"{self.name}:wgrad M={self.Co} N={self.Kf} K={dy.rows} split={d.split_k}")

    # ---- two-channel heads as 1x1 GEMMs at the fine levels (csrc/thin_conv.hip, GEMM formulation) -----------------------
    def _thin_gemm_fwd(self, x: View, y: Optional[View], y32: Optional[View], bias: bool) -> None:
        if getattr(self, "_z18", None) is None or self._z18.shape[0] != x.rows:
            self._z18 = torch.zeros(x.rows, 32, device=self.ws.device, dtype=F32)
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C = x.ptr, x.ld, x.H, x.W, self.Cip
        d.taps_y = d.taps_x = d.mul_y = d.mul_x = d.step_y = d.step_x = 1
        d.g_H, d.g_W, d.n_img = x.H, x.W, x.B
        d.w, d.w_ld, d.N = self.packF.data_ptr(), self.Cip, 18           # the FWD pack [2][9*Cip] read as [18][Cip]
        d.x_bytes, d.w_bytes = x.bytes_left, self.packF.numel() * self.packF.element_size()
        d.y32, d.y32_ld = self._z18.data_ptr(), 32
        d.y_H, d.y_W, d.y_mul_y, d.y_mul_x = x.H, x.W, 1, 1
        d.slope, d.dtype, d.split_k, d.algo = 1.0, self.ws.code, 1, 1
        PROFILER.launch("mireg_conv_gemm", d, "thin_conv_fwd", 2.0 * x.rows * 2 * 9 * self.Ci, f"{self.name}:thin-gemm-fwd")
        out = y if y is not None else y32
        _lib.call("mireg_thin_shift_sum", self._z18.data_ptr(), 32, self.bias.data_ptr() if (bias and self.bias is not None) else None,
                  y.ptr if y is not None else None, y.ld if y is not None else 0,
                  y32.ptr if y32 is not None else None, y32.ld if y32 is not None else 0, out.B, out.H, out.W, self.ws.code, _stream())

    def _thin_gemm_wgrad(self, x: View, dy: View, slot: int) -> None:
        if getattr(self, "_dz18", None) is None or self._dz18.rows != dy.rows:
            self._dz18 = self.ws.new(dy.B, dy.H, dy.W, 18, pad=32)
        dz = self._dz18
        _lib.call("mireg_thin_gather18", dy.ptr, dy.ld, dz.ptr, dz.ld, dy.B, dy.H, dy.W, self.ws.code, _stream())
        d = ConvDesc()
        d.x, d.x_ld, d.x_H, d.x_W, d.x_C = x.ptr, x.ld, x.H, x.W, self.Cip
        d.taps_y = d.taps_x = d.mul_y = d.mul_x = d.step_y = d.step_x = 1
        d.g_H, d.g_W, d.n_img = dy.H, dy.W, dy.B
        d.y, d.y_ld, d.N = dz.ptr, dz.ld, 18
        d.split_k, d.dtype, d.stages, d.algo = self.wgrad_split, self.ws.code, self.ws.wgrad_stages, 1
        d.slab = self.wgrad_slab[slot * self.wgrad_split].data_ptr()   # [split][18][Cip] == [split][2][9*Cip]
        d.slab_ld = self.Cip
        d.x_bytes, d.w_bytes = x.bytes_left, dz.bytes_left
        PROFILER.launch("mireg_conv_wgrad", d, "thin_conv_wgrad", 2.0 * dy.rows * 2 * 9 * self.Ci, f"{self.name}:thin-gemm-wgrad")

    def run_bias_grad(self, dy: View, accumulate: bool = False) -> None:
        if self.bias is None:
            return
        if self.grad_b is None:
            self.grad_b = torch.zeros_like(self.bias, dtype=F32)
        if self.ws.colsum_ws is None or self.ws.colsum_ws.numel() < 64 * dy.C:
            self.ws.colsum_ws = torch.empty(64 * max(dy.C, 1024), device=self.ws.device, dtype=F32)
        # bias_direct: grad_b IS the parameter's .grad for this backward (PredictorEngineBase.autograd_backward): always add into it
        _lib.call("mireg_colsum", dy.ptr, dy.ld, dy.rows, dy.C, self.grad_b.data_ptr(), int(accumulate or getattr(self, "bias_direct", False)),
                  self.ws.colsum_ws.data_ptr(), self.ws.code,
                  _stream())


class BatchNormAct:
    """Train/eval BatchNorm2d + LeakyReLU on an NHWC view (reference FlowNetS/util.py:17-30)."""
    MAX_BLOCKS = 512

    def __init__(self, bn: torch.nn.BatchNorm2d, ws: Workspace, slope: float = 0.1):
        self.bn, self.ws, self.slope = bn, ws, slope
        self.pending = 0                 # training forwards not yet added to bn.num_batches_tracked (flushed when a state_dict is taken)
        C = bn.num_features
        self.C = C
        self.partial = torch.zeros(self.MAX_BLOCKS * 2 * C, device=ws.device, dtype=F32)
        self.ss = torch.zeros(4 * C, device=ws.device, dtype=F32)
        self.red = torch.zeros(2 * C, device=ws.device, dtype=F32)
        self.grad_g = torch.zeros(C, device=ws.device, dtype=F32)
        self.grad_b = torch.zeros(C, device=ws.device, dtype=F32)

    def forward(self, y: View, out: View, training: bool) -> None:
        bn = self.bn
        if training and y.rows == 1:     # same refusal as torch.nn.functional.batch_norm (reference BatchNorm2d in train mode)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{y.B}, {self.C}, {y.H}, {y.W}])")
        mom = bn.momentum if bn.momentum is not None else 0.1
        _lib.call("mireg_bn_forward", y.ptr, y.ld, out.ptr, out.ld, y.rows, self.C, bn.weight.data_ptr(),
                  bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(mom), float(bn.eps),
                  int(training), self.slope, self.partial.data_ptr(), self.ss.data_ptr(), self.ws.code, _stream())

    def backward(self, y: View, da: View, dy: View, acc_param_grads: bool = False) -> None:
        _lib.call("mireg_bn_backward", y.ptr, y.ld, da.ptr, da.ld, dy.ptr, dy.ld, self.ss.data_ptr(),
                  self.partial.data_ptr(), self.red.data_ptr(), self.grad_g.data_ptr(), self.grad_b.data_ptr(),
                  int(acc_param_grads or getattr(self, "direct", False)), y.rows,
                  self.C, self.slope, self.ws.code, _stream())


def lrelu_bwd(g: View, a: View, slope: float, ws: Workspace) -> None:
    _lib.call("mireg_lrelu_bwd", g.ptr, g.ld, a.ptr, a.ld, g.rows, g.C, slope, ws.code, _stream())


def nchw_to_view(src: torch.Tensor, c0: int, nc: int, dst: View, code: Optional[int] = None) -> None:
    """src (B, Ctot, H, W) fp32 contiguous, channels [c0, c0+nc) -> dst channels [0, nc)."""
    B, Ctot, H, W = src.shape
    if code is None:
        code = DT_BF16 if dst.buf.dtype == torch.bfloat16 else DT_F32
    _lib.call("mireg_nchw_to_nhwc", src.data_ptr(), dst.ptr, B, Ctot, c0, nc, H * W, dst.ld, code, _stream())


def cast_from_f32(dst: View, src: View, alpha: float = 1.0, beta: float = 0.0) -> None:
    code = DT_BF16 if dst.buf.dtype == torch.bfloat16 else DT_F32
    _lib.call("mireg_cast_from_f32", dst.ptr, dst.ld, src.ptr, src.ld, dst.rows, dst.C, alpha, beta, code, _stream())
