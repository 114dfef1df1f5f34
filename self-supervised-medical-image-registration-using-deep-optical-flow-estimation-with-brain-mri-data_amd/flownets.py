"""FlowNetS predictor (reference FlowNetS/FlowNetS.py:10-93, FlowNetS/util.py:17-55) on the HIP engine.

The nn.Module keeps the reference's submodule names, constructor signature, init and train/eval
return arity, so state_dicts interchange key for key.  Its forward/backward are two hand-scheduled
kernel sequences over persistent NHWC buffers (no torch.cat, no ATen/MIOpen calls).
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import engine as engine_mod
from .engine import (F32, TINY_MASK, BatchNormAct, ConvLayer, View, WoptJob, Workspace, _stream, assign_tiles, cast_from_f32, zero_tensors,
                     lrelu_bwd, nchw_to_view, upload_table)

DIRECT_SMALL_GRADS = os.environ.get("MIREG_HANDOVER_SMALL_GRADS", "0") != "1"   # A/B switch: bias / BatchNorm gradients through AccumulateGrad
FORK_DECODER = os.environ.get("MIREG_SERIAL_DECODER", "0") != "1"   # A/B switch: decoder heads next to the deconvolutions

ENCODER = [  # name, cin, cout, k, stride   (FlowNetS/FlowNetS.py:17-26)
    ("conv1", 2, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2), ("conv3_1", 256, 256, 3, 1),
    ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1), ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1),
    ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1)]
DECONV = {5: (1024, 512), 4: (1026, 256), 3: (770, 128), 2: (386, 64)}   # FlowNetS.py:28-31
PREDICT = {6: 1024, 5: 1026, 4: 770, 3: 386, 2: 194}                      # FlowNetS.py:33-37
SLOPE = 0.1


def conv_block(bn: bool, cin: int, cout: int, k: int = 3, stride: int = 1) -> nn.Sequential:
    layers: List[nn.Module] = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=not bn)]
    if bn:
        layers.append(nn.BatchNorm2d(cout))
    layers.append(nn.LeakyReLU(SLOPE, inplace=True))
    return nn.Sequential(*layers)


def count_bn_batches(eng) -> None:
    """One training forward has run through every BatchNorm of the engine (torch increments num_batches_tracked per module
    call; siamese streams wrap one module twice and so count twice, as in the reference)."""
    for b in eng.bns.values():
        b.pending += 1


def drop_engines(module: nn.Module) -> None:
    """Forget the cached engines of a predictor (shape / device / dtype change) WITHOUT losing the training forwards their
    BatchNorms have counted: num_batches_tracked goes into every checkpoint (train.py:183-201)."""
    for eng in getattr(module, "_engines", {}).values():
        for b in getattr(eng, "bns", {}).values():
            if b.pending:
                b.bn.num_batches_tracked += b.pending
                b.pending = 0
    module._engines.clear()
    engine_mod._ZERO_TABLES.clear()                          # device tables keyed by the dropped engines' buffer addresses


def install_bn_counter_hook(module: nn.Module) -> None:
    """Keep BatchNorm2d.num_batches_tracked as torch would (train.py:183-201 writes it into every checkpoint) without a launch
    per layer per step: forwards are counted on the host and added when a state_dict is taken."""
    if getattr(module, "_bn_counter_hook", None) is not None:
        return

    def flush(mod, prefix, keep_vars):
        for eng in getattr(mod, "_engines", {}).values():
            for b in eng.bns.values():
                if b.pending:
                    b.bn.num_batches_tracked += b.pending
                    b.pending = 0
    module._bn_counter_hook = module.register_state_dict_pre_hook(flush)


class PackedOptimizerHook:
    """Mixin of the predictor modules (FlowNetS, FlowNetC, PWCDCNet, the FlowNet2 sub-networks): `mireg.Adam(params, fuse=model)`
    finds every sub-module with `fuse_optimizer` and from then on updates their convolution weights in the packed domain
    (`PredictorEngineBase.fused_adam`); those weights never get a `.grad`.  Single process, one backward per step."""
    _fopt = None
    _findex: Optional[Dict[int, int]] = None

    def fuse_optimizer(self, opt, index: Dict[int, int]) -> None:
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("mireg.Adam(fuse=...) updates the convolution weights from this rank's gradient slabs and leaves no `.grad` "
                               "for an all-reduce: single-process training only (RegistrationTrainer is the data-parallel path)")
        self._fopt, self._findex = opt, index

    def fused_pending(self) -> bool:
        return any(e.slab_pending for e in getattr(self, "_engines", {}).values())

    def fused_step(self, opt, tick: int) -> None:
        for e in getattr(self, "_engines", {}).values():
            if e.slab_pending:
                e.fused_adam(opt, tick)
                tick = 0


def grads_for_autograd(params, table: Dict[int, torch.Tensor]) -> tuple:
    """The engine's persistent gradient buffers as an autograd Function's return value.  A parameter that already has a `.grad`
    (mireg.Adam.zero_grad keeps and zeroes them) gets the buffer itself: AccumulateGrad adds it in place and keeps no reference, so
    the per-parameter clone (two extra small kernels per tensor and step, ~1,200 for FlowNet2) is not needed.  Without a `.grad`
    autograd may adopt the returned tensor as `.grad`, which must then not alias a buffer the next backward overwrites: clone."""
    return tuple(None if id(p) not in table else
                 (table[id(p)] if (p.grad is not None and p.grad.data_ptr() != table[id(p)].data_ptr()) else table[id(p)].clone())
                 for p in params)


class PredictorEngineBase:
    """Shared by the predictors: weight packing, gradient unpacking, parameter <-> grad bookkeeping."""

    def __init__(self, module: nn.Module, B: int, H: int, W: int, device, dtype: torch.dtype):
        self.module, self.B, self.H, self.W = module, B, H, W
        self.ws = Workspace(device, dtype)
        self.layers: Dict[str, ConvLayer] = {}
        self.bns: Dict[str, BatchNormAct] = {}
        self._pack_table = None
        self._pack_key = None
        self._unpack_table = None
        self._reduce_table = {}
        self.training_cache = False

    # -- parameters ---------------------------------------------------------------------------------
    def add_conv(self, name: str, conv: nn.Module, stride: int, pad: int, dil: int = 1, uses: int = 1) -> ConvLayer:
        lay = ConvLayer(name, conv.weight, conv.bias, stride, pad, dil, self.ws)
        lay.n_slots = uses                    # weight-shared layers (siamese streams) get one wgrad slab per use
        self.layers[name] = lay
        return lay

    # -- backward-weights on a side stream ---------------------------------------------------------------
    # wgrad(L) and dgrad(L) are independent and each alone under-fills 256 CUs on the deep layers, so wgrad is
    # forked onto a second HIP stream (captured as a parallel hipGraph branch) and joined before the unpack.
    use_side_stream = True
    # packed-domain optimizer under torch.autograd (mireg.Adam(fuse=model), PackedOptimizerHook below)
    fused_index: Optional[Dict[int, int]] = None
    slab_pending = False
    _fresh = None

    def mark(self):
        """Event on the current stream: 'the operands of a later wgrad_async(..., after=ev) are ready here'."""
        if not self.use_side_stream:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    # layer-name prefixes whose backward-weights launches stay on the main stream.  The flow heads': their gather + 18-column GEMM pairs
    # made the backward-weights stream the long pole of the decoder phase (the main chain then idled ~140 us at the phase join);
    # measured on FlowNetS: 2.695 -> 2.615 ms per step.  (Also tried: the 2 -> 2 upsamplers, the deconvolutions, conv6: each slower.)
    # Measured per predictor (default bench, one box): FlowNetS in the fused trainer gains, FlowNetC / PWC / the FlowNet2 stack lose 1-2 %,
    # so only FlowNetSEngine under the trainer sets it (main_only_default); MIREG_WGRAD_ON_MAIN overrides for experiments.
    _MAIN_ONLY = (tuple(t for t in os.environ["MIREG_WGRAD_ON_MAIN"].split(",") if t) if "MIREG_WGRAD_ON_MAIN" in os.environ else None)
    main_only_default: tuple = ()

    def wgrad_async(self, lay: ConvLayer, x: View, dy: View, slot: int = 0, after=None) -> None:
        main_only = self._MAIN_ONLY if self._MAIN_ONLY is not None else self.main_only_default
        if not self.use_side_stream or (main_only and lay.name.startswith(main_only)):
            lay.run_wgrad(x, dy, slot)
            return
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.ws.device)
        ev = after
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            lay.run_wgrad(x, dy, slot)

    def join_side(self) -> None:
        if getattr(self, "_side", None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    # "torch": backward ends in torch-layout gradients (autograd drop-in);  "packed": it ends in the packed-domain
    # flat gradient the fused optimizer consumes (RegistrationTrainer), and the optimizer keeps the packs fresh.
    grad_mode = "torch"
    packs_fresh = False
    # True: a backward phase ends WITHOUT joining the wgrad stream / reducing its slabs (the caller does both); kept
    # for experiments -- measured slower than the default, see RegistrationTrainer._fwd_bwd
    defer_grad_reduce = False

    def pack_dgrad_subset(self, names: Sequence[str]) -> None:
        """DGRAD packs of the named layers from their (fresh) FWD packs."""
        key = ("dgrad", tuple(names))
        if key not in self._reduce_table:
            jobs = [j for n in names for j in self.layers[n].pack_jobs()]
            _, dunits = assign_tiles(jobs, False)
            self._reduce_table[key] = (upload_table(jobs, self.ws.device), len(jobs), dunits) if dunits else None
        if self._reduce_table[key] is not None:
            tab, n, dunits = self._reduce_table[key]
            _lib.call("mireg_pack_weights", tab.data_ptr(), n, 0, dunits, self.ws.code, _stream())

    def pack_weights(self, force: bool = False, dgrad_only: bool = False) -> None:
        """torch-layout fp32 parameters -> GEMM packs (one table-driven launch)."""
        if self.packs_fresh and not force and not dgrad_only:
            return
        if self._fresh is not None and not force and not dgrad_only:      # fused_adam rewrote F and D packs with the new weights
            fresh, self._fresh = self._fresh, None
            if all(p._version == v for p, v in fresh):                    # ... and nobody has written a parameter since
                return
        key = tuple(l.weight.data_ptr() for l in self.layers.values())
        if self._pack_key != key:
            jobs = [j for l in self.layers.values() for j in l.pack_jobs()]
            self._pack_units, self._pack_dunits = assign_tiles(jobs, False)
            self._pack_table, self._pack_n, self._pack_key = upload_table(jobs, self.ws.device), len(jobs), key
        if dgrad_only and not self._pack_dunits:
            return
        _lib.call("mireg_pack_weights", self._pack_table.data_ptr(), self._pack_n, 0 if dgrad_only else self._pack_units,
                  self._pack_dunits, self.ws.code, _stream())

    def unpack_grads(self, names: Optional[Sequence[str]] = None) -> None:
        """wgrad slabs -> torch-layout gradients, for all layers or for the named subset (one backward phase)."""
        if self.grad_mode == "packed":
            return self.reduce_grads(names)
        if self.fused_index is not None:                       # the weights are updated straight from the slabs by fused_adam()
            return
        key = tuple(names) if names is not None else None
        if self._unpack_table is None:
            self._unpack_table = {}
        if key not in self._unpack_table:
            layers = [self.layers[n] for n in names] if names is not None else list(self.layers.values())
            # split-K / per-use slabs are first summed in place by the fully parallel reduce (fixed order), then ONE slab per layer is
            # transposed: the unpack kernel's own slab loop is a serial chain per lane (FlowNet2, batch 8, up to 44 slabs per layer:
            # 5.7 ms of a 23 ms step went there)
            red, r = [], 0
            for l in layers:
                ns = l.wgrad_split * l.n_slots if l.wgrad_slab is not None else 0
                if ns > 1:
                    j = WoptJob()
                    j.slab = j.g = l.wgrad_slab.data_ptr()
                    j.slab_stride, j.nsplit = l.Co * l.Kf, ns
                    j.Co, j.Ci, j.taps, j.Cpad, j.ld = l.Co, l.Ci, l.kh * l.kw, l.Cip, l.Kf
                    j.runit0 = r
                    r += (l.Co * l.Kf + 255) // 256
                    red.append(j)
            jobs = [l.unpack_job() for l in layers if l.wgrad_slab is not None]
            for j in jobs:
                j.nsplit = 1
            units, _ = assign_tiles(jobs, True)
            self._unpack_table[key] = (upload_table(jobs, self.ws.device), len(jobs), units,
                                       (upload_table(red, self.ws.device), len(red), r) if red else None)
        tab, n, units, reduce = self._unpack_table[key]
        if reduce is not None:
            _lib.call("mireg_wgrad_reduce", reduce[0].data_ptr(), reduce[1], reduce[2], _stream())
        _lib.call("mireg_unpack_wgrad", tab.data_ptr(), n, units, _stream())

    def reduce_grads(self, names: Optional[Sequence[str]] = None) -> None:
        """split-K wgrad slabs -> packed-domain gradients for all layers or one backward phase (one launch)."""
        key = tuple(names) if names is not None else None
        if key not in self._reduce_table:
            layers = [self.layers[n] for n in names] if names is not None else list(self.layers.values())
            jobs, units = [], 0
            for l in layers:
                if l.wgrad_slab is None:
                    continue
                j = l.wopt_job()
                if j.nsplit > 0:
                    j.runit0 = units
                    units += (l.Co * l.Kf + 255) // 256
                    jobs.append(j)
            self._reduce_table[key] = (upload_table(jobs, self.ws.device), len(jobs), units) if jobs else None
        if self._reduce_table[key] is not None:
            tab, n, units = self._reduce_table[key]
            _lib.call("mireg_wgrad_reduce", tab.data_ptr(), n, units, _stream())

    def packed_layout(self, params: Sequence[nn.Parameter]) -> Tuple[Dict[int, int], Dict[int, int], int]:
        """Offsets / slot sizes (floats) of every parameter's gradient in the packed flat buffer (parameter order;
        conv weights occupy Co*Kf, everything else numel; every slot starts 16-byte aligned)."""
        wl = {id(l.weight): l for l in self.layers.values()}
        off, size, o = {}, {}, 0
        for p in params:
            n = wl[id(p)].Co * wl[id(p)].Kf if id(p) in wl else p.numel()
            off[id(p)], size[id(p)] = o, (n + 3) // 4 * 4
            o += size[id(p)]
        return off, size, o

    def bind_packed_grads(self, params: Sequence[nn.Parameter], flat: torch.Tensor) -> None:
        """Packed-domain counterpart of bind_flat_grads: `flat` (packed_layout size) receives every gradient."""
        off, size, total = self.packed_layout(params)
        assert total == flat.numel()
        self.flat_off, self.flat_size = off, size
        byid = {id(p): p for p in params}

        def view(p):
            return flat[off[id(p)]:off[id(p)] + p.numel()].view(p.shape)
        for l in self.layers.values():
            l.bind_gpack(flat[off[id(l.weight)]:off[id(l.weight)] + l.Co * l.Kf].view(l.Co, l.Kf))
            l.grad_w = None
            if l.bias is not None:
                l.grad_b = view(l.bias)
        for b in self.bns.values():
            b.grad_g, b.grad_b = view(b.bn.weight), view(b.bn.bias)
        self._unpack_table, self._reduce_table = None, {}
        self.grad_mode = "packed"

    def flat_range(self, layer_names: Sequence[str], bn_names: Sequence[str] = (), extra: Sequence[nn.Parameter] = ()) -> Tuple[int, int]:
        """[start, end) of the flat gradient buffer covered by these layers' parameters (must be contiguous)."""
        ps = []
        for n in layer_names:
            l = self.layers[n]
            ps.append(l.weight)
            if l.bias is not None:
                ps.append(l.bias)
        for n in bn_names:
            ps += [self.bns[n].bn.weight, self.bns[n].bn.bias]
        ps += list(extra)                                   # parameters no layer of the engine uses (they sit inside the range)
        size = self.flat_size
        lo = min(self.flat_off[id(p)] for p in ps)
        hi = max(self.flat_off[id(p)] + size[id(p)] for p in ps)
        assert hi - lo == sum(size[i] for i in {id(p) for p in ps}), "phase parameters are not contiguous"
        return lo, hi

    def bind_flat_grads(self, params: Sequence[nn.Parameter], flat: torch.Tensor) -> None:
        """Make every gradient buffer a view of `flat` (parameter order), so one RCCL all-reduce / one Adam
        launch covers the whole model."""
        off, o = {}, 0
        for p in params:
            off[id(p)] = o
            o += p.numel()
        assert o == flat.numel()
        self.flat_off, self.flat_size = off, {id(p): p.numel() for p in params}
        self.grad_mode = "torch"

        def view(p):
            return flat[off[id(p)]:off[id(p)] + p.numel()].view(p.shape)
        for l in self.layers.values():
            l.grad_w = view(l.weight)
            if l.bias is not None:
                l.grad_b = view(l.bias)
        for b in self.bns.values():
            b.grad_g, b.grad_b = view(b.bn.weight), view(b.bn.bias)
        self._unpack_table = None

    def autograd_backward(self, g, fused: Optional[Dict[int, int]] = None) -> None:
        """backward() ending in torch-layout gradients even when a trainer has bound this engine to the packed domain.
        fused = {id(parameter): optimizer index} of a mireg.Adam(fuse=...): the convolution weights' gradients then stay in their
        backward-weights slabs (no unpack; `param_grads()` has no entry for them) until `fused_adam` consumes them."""
        if fused is not None and self.slab_pending:
            raise RuntimeError("a second backward before optimizer.step(): with mireg.Adam(fuse=...) the weight gradients live in the "
                               "backward-weights slabs, which one backward fills and one step consumes")
        mode, self.grad_mode, self.fused_index = self.grad_mode, "torch", fused
        bound = self._bind_small_grads() if (DIRECT_SMALL_GRADS and mode == "torch") else []
        try:
            self.backward(g)
            self.slab_pending = fused is not None
        finally:
            self.grad_mode = mode
            for obj, attr, val, flag in bound:                 # the engine's own buffers again (eval / trainer paths use them)
                setattr(obj, attr, val)
                setattr(obj, flag, False)
        if fused is not None:                                  # torch-layout weight gradients are not produced in this mode
            for l in self.layers.values():
                if id(l.weight) in fused:
                    l.grad_w = None

    def _bind_small_grads(self) -> list:
        """Bias and BatchNorm gradients of this backward go straight into the parameters' `.grad` where one exists (mireg.Adam.zero_grad
        keeps and zeroes them): the kernels add into it, `param_grads()` leaves those parameters out and autograd gets None for them.
        Otherwise every step pays AccumulateGrad's add for each of the few hundred small tensors of a stack like FlowNet2 (launch-sized
        kernels on the serial chain).  Single process only: DistributedDataParallel reduces a gradient when autograd hands it over."""
        self._direct_ids = set()
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return []
        dev, bound = self.ws.device, []

        def usable(p):
            g = p.grad
            return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == dev
        for l in self.layers.values():
            if l.bias is not None and usable(l.bias):
                bound.append((l, "grad_b", l.grad_b, "bias_direct"))
                l.grad_b, l.bias_direct = l.bias.grad, True
                self._direct_ids.add(id(l.bias))
        for b in self.bns.values():
            w, bi = b.bn.weight, b.bn.bias
            if usable(w) and usable(bi):                       # (siamese streams: two objects over one BatchNorm module, both add into its .grad)
                bound.append((b, "grad_g", b.grad_g, "direct"))
                bound.append((b, "grad_b", b.grad_b, "direct"))
                b.grad_g, b.grad_b, b.direct = w.grad, bi.grad, True
                self._direct_ids.update((id(w), id(bi)))
        return bound

    def fused_adam(self, opt, tick: int) -> None:
        """Slabs -> (in-place slab sum) -> Adam on the fp32 master weights + refreshed forward packs (`mireg_adam_pack`), then the
        backward-data packs from the fresh forward packs: the 2-D trainer's packed-domain optimizer for a model that trains through
        torch.autograd.  Biases and BatchNorm parameters keep the ordinary `.grad` / `mireg_adam_step` path."""
        index, st = self.fused_index, _stream()
        red, jobs, r, units, max_taps, done = [], [], 0, 0, 1, []
        for l in self.layers.values():
            if l.wgrad_slab is None or id(l.weight) not in index:
                continue
            ns = l.wgrad_split * l.n_slots
            j = WoptJob()
            j.slab = j.g = l.wgrad_slab.data_ptr()
            j.slab_stride, j.nsplit = l.Co * l.Kf, ns
            j.Co, j.Ci, j.taps, j.Cpad, j.ld = l.Co, l.Ci, l.kh * l.kw, l.Cip, l.Kf
            j.p, j.F = l.weight.data_ptr(), l.packF.data_ptr()
            j.m, j.v = opt.state_ptrs(index[id(l.weight)])
            j.unit0, j.runit0 = units, r
            units += l.Co * ((l.Cip + 63) // 64)
            max_taps = max(max_taps, l.kh * l.kw)
            jobs.append(j)
            done.append(l.weight)
            if ns > 1:
                red.append(WoptJob.from_buffer_copy(j))
                r += (l.Co * l.Kf + 255) // 256
        self.slab_pending = False
        if not jobs:
            return
        if red:
            self._ftab_r = upload_table(red, self.ws.device)
            _lib.call("mireg_wgrad_reduce", self._ftab_r.data_ptr(), len(red), r, st)
        self._ftab = upload_table(jobs, self.ws.device)
        _lib.call("mireg_adam_pack", self._ftab.data_ptr(), len(jobs), units, max_taps, opt.step_dev.data_ptr(), int(tick), opt.lr,
                  opt.betas[0], opt.betas[1], opt.eps, 1.0, self.ws.code, st)
        self.pack_weights(dgrad_only=True)
        # forward and backward-data packs are current for the next forward if every packed layer was just rewritten
        every = all(l.wgrad_slab is not None and id(l.weight) in index for l in self.layers.values())
        self._fresh = [(w, w._version) for w in done] if every else None

    def param_grads(self) -> Dict[int, torch.Tensor]:
        """id(parameter) -> persistent fp32 gradient buffer (torch layout)."""
        out = {}
        for l in self.layers.values():
            if l.grad_w is not None:
                out[id(l.weight)] = l.grad_w
            if l.bias is not None and l.grad_b is not None:
                out[id(l.bias)] = l.grad_b
        for b in self.bns.values():
            out[id(b.bn.weight)] = b.grad_g
            out[id(b.bn.bias)] = b.grad_b
        for i in getattr(self, "_direct_ids", ()):              # already added into .grad by the last autograd_backward
            out.pop(i, None)
        return out


class FlowNetDecoderMixin:
    """Refinement decoder shared by FlowNetS and FlowNetC (FlowNetS.py:60-80, flownet2/networks/FlowNetC.py:104-125):
    needs self.cat{2..5}, self.a61, self.skip_c, self.hs and the module's deconv / predict_flow / upsampler layers."""

    def setup_decoder(self, m: nn.Module) -> None:
        for lvl in DECONV:
            self.add_conv(f"deconv{lvl}", getattr(m, f"deconv{lvl}")[0], 2, 1)        # adjoint conv: Co=cin_deconv
        for lvl in PREDICT:
            self.add_conv(f"predict_flow{lvl}", getattr(m, f"predict_flow{lvl}"), 1, 1)
        for lvl in (6, 5, 4, 3):
            self.add_conv(f"up{lvl}", getattr(m, f"upsampled_flow{lvl}_to_{lvl - 1}"), 2, 1)
        new, B, hs = self.ws.new, self.B, self.hs
        self.flow32 = {lvl: new(B, *hs[lvl], 2, dtype=F32, pad=2) for lvl in PREDICT}
        self.flowT = {lvl: new(B, *hs[lvl], 2) for lvl in PREDICT}

    def decoder_forward(self) -> None:
        """Per level: deconv(feat) and [predict_flow(feat) -> flow upsampler] only share their input, and the forward pass has the chip
        to itself at concurrency 1 (profiles/round3_step_timeline.txt): the head branch (15-25 us of vector-ALU kernels) runs on the
        second stream next to the deconvolution GEMM and is joined before the level's concat is read.  Only when both head layers
        take their scratch-free kernels (a split-K GEMM on either branch would share the workspace slab) and not while launch shapes
        are being timed."""
        L, c = self.layers, self.cat
        feat = self.a61
        par = (FORK_DECODER and self.use_side_stream and not self.ws.tuning
               and all(L[f"predict_flow{l}"].thin for l in (6, 5, 4, 3)) and all(L[f"up{l}"].tiny for l in (6, 5, 4, 3)) and (TINY_MASK & 1))
        main = torch.cuda.current_stream()
        if par and getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.ws.device)
        for lvl in (5, 4, 3, 2):
            cs = self.skip_c[lvl]
            cd = DECONV[lvl][1]

            def heads(feat=feat, lvl=lvl, cs=cs, cd=cd):
                L[f"predict_flow{lvl + 1}"].run_fwd_form(feat, self.flowT[lvl + 1], y32=self.flow32[lvl + 1])
                L[f"up{lvl + 1}"].run_dgrad_form(self.flowT[lvl + 1], c[lvl].slice(cs + cd, 2), bias=True)
            if par:
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    heads()
            else:
                heads()
            L[f"deconv{lvl}"].run_dgrad_form(feat, c[lvl].slice(cs, cd), slope=SLOPE, bias=True)
            if par:
                main.wait_stream(self._side)
            feat = c[lvl]
        hook = getattr(self, "pre_tail", None)                                # the trainer's batch-only loss preparation (FusedRegLoss.prepare)
        if par and hook is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                hook()
        L["predict_flow2"].run_fwd_form(feat, self.flowT[2], y32=self.flow32[2])
        if par and hook is not None:
            main.wait_stream(self._side)

    def setup_decoder_grads(self) -> None:
        new, hs, B = self.ws.new, self.hs, self.B
        self.dcat = {lvl: new(B, *hs[lvl], v.C) for lvl, v in self.cat.items()}
        self.da61 = new(B, *hs[6], 1024)
        self.dflowT = {lvl: new(B, *hs[lvl], 2) for lvl in PREDICT}
        self.dflow32 = new(B, *hs[2], 2, dtype=F32, pad=2)

    def decoder_backward(self, glvl: Dict[int, Optional[torch.Tensor]], g0: Optional[torch.Tensor]) -> None:
        """glvl[l]: loss gradient of flow l (B,2,h,w fp32 or None); g0: gradient of the 256x256 upsampled flow2
        (FlowNetS only).  Leaves every decoder-side contribution in dcat[2..5] and da61."""
        L, c, dc, B, st = self.layers, self.cat, self.dcat, self.B, _stream()

        def load_loss_grad(lvl: int) -> None:
            gt = glvl.get(lvl)
            dst = self.dflowT[lvl]
            if lvl == 2 and g0 is not None:
                d32 = self.dflow32
                if gt is None:
                    zero_tensors([d32.buf])
                else:
                    nchw_to_view(gt.contiguous(), 0, 2, d32)
                g0c = g0.contiguous()
                _lib.call("mireg_resize_bilinear_bwd", g0c.data_ptr(), d32.ptr, B, 2, d32.H, d32.W, 256, 256,
                          d32.H * d32.W * 2, 1, 2, 2 * 256 * 256, 256 * 256, 1, 0, 1.0, st)
                cast_from_f32(dst, d32)
            elif gt is None:
                zero_tensors([dst.buf])
            else:
                nchw_to_view(gt.contiguous(), 0, 2, dst)

        load_loss_grad(2)
        pf = L["predict_flow2"]
        # everywhere below: mark() where a layer's wgrad operands are final, enqueue the critical-chain kernels first and
        # the side-stream wgrad (waiting only for that mark) after them -- hipGraph launches nodes in capture order
        ready = self.mark()
        pf.run_bias_grad(self.dflowT[2])
        pf.run_dgrad_form(self.dflowT[2], dc[2])                              # dcat2 <- (beta 0)
        self.wgrad_async(pf, c[2], self.dflowT[2], after=ready)
        for lvl in (2, 3, 4, 5):
            # dcat[lvl] holds every decoder-side contribution now; push it one level coarser
            cs, cd = self.skip_c[lvl], DECONV[lvl][1]
            feat_prev, dfeat_prev = (c[lvl + 1], dc[lvl + 1]) if lvl < 5 else (self.a61, self.da61)
            # flow upsampler (lvl+1 -> lvl): no activation
            gup = dc[lvl].slice(cs + cd, 2)
            up = L[f"up{lvl + 1}"]
            gt = glvl.get(lvl + 1)
            fused = (up.tiny_bwd_data_ok(gup, self.dflowT[lvl + 1]) and gt is not None and gt.dtype == torch.float32
                     and os.environ.get("MIREG_TINY_NOFUSE", "0") != "1")
            if not fused:
                load_loss_grad(lvl + 1)                                       # dflowT[lvl+1] <- loss grad
            m_up = self.mark()
            up.run_bias_grad(gup)
            if fused:                                                         # dflowT[lvl+1] <- loss grad + upsampler backward-data
                up.run_fwd_form(gup, self.dflowT[lvl + 1], bias=False, add_nchw=gt.contiguous())
            else:
                up.run_fwd_form(gup, self.dflowT[lvl + 1], bias=False, accumulate=True)
            self.wgrad_async(up, gup, self.flowT[lvl + 1], after=m_up)
            # feature deconv (lvl+1 -> lvl) + LeakyReLU
            gde = dc[lvl].slice(cs, cd)
            lrelu_bwd(gde, c[lvl].slice(cs, cd), SLOPE, self.ws)
            de = L[f"deconv{lvl}"]
            m_de = self.mark()
            de.run_bias_grad(gde)
            # predict_flow{lvl+1} writes dfeat_prev first (beta 0), the deconv then accumulates into it
            pfn = L[f"predict_flow{lvl + 1}"]
            pfn.run_bias_grad(self.dflowT[lvl + 1])
            pfn.run_dgrad_form(self.dflowT[lvl + 1], dfeat_prev)
            de.run_fwd_form(gde, dfeat_prev, bias=False, accumulate=True)
            self.wgrad_async(de, gde, feat_prev, after=m_de)
            self.wgrad_async(pfn, feat_prev, self.dflowT[lvl + 1], after=m_de)

    def chain_backward(self, name: str, src: View, dst: View, dsrc: Optional[View], acc: bool, ddst: View,
                       bn_key: Optional[str] = None, raw_key: Optional[str] = None, slot: int = 0,
                       acc_bn: bool = False) -> None:
        """Backward of one conv(+BN)+LeakyReLU block: ddst = grad wrt its activated output, dsrc = grad wrt its input."""
        lay = self.layers[name]
        if self.bn:
            rk = raw_key or name
            self.bns[bn_key or name].backward(self.raw[rk], ddst, self.draw[rk], acc_bn)
            dy = self.draw[rk]
        else:
            lrelu_bwd(ddst, dst, SLOPE, self.ws)
            dy = ddst
            lay.run_bias_grad(dy, accumulate=slot > 0)
        # the backward-data GEMM is on the critical chain: enqueue it first, the backward-weights GEMM (same inputs) after it
        ready = self.mark()
        if dsrc is not None:
            lay.run_dgrad_form(dy, dsrc, accumulate=acc)
        self.wgrad_async(lay, src, dy, slot, after=ready)


class FlowNetSEngine(PredictorEngineBase, FlowNetDecoderMixin):
    @property
    def main_only_default(self) -> tuple:
        # + conv1: the last launch of the backward-weights stream, which main would otherwise only wait for (2.637 -> 2.622 ms)
        return ("predict_flow", "conv1") if self.grad_mode == "packed" else ()

    def __init__(self, module: "FlowNetS", B: int, H: int, W: int, device, dtype: torch.dtype):
        super().__init__(module, B, H, W, device, dtype)
        if H % 64 or W % 64:
            raise RuntimeError(f"FlowNetS engine needs H, W divisible by 64, got {H}x{W}")
        ws, m = self.ws, module
        self.bn = m.batchNorm
        for name, cin, cout, k, s in ENCODER:
            seq = getattr(m, name)
            self.add_conv(name, seq[0], s, (k - 1) // 2)
            if self.bn:
                self.bns[name] = BatchNormAct(seq[1], ws, SLOPE)
        # ---- buffers -----------------------------------------------------------------------------
        hs = {lvl: (H >> lvl, W >> lvl) for lvl in range(1, 7)}
        self.hs = hs
        self.setup_decoder(m)
        new = ws.new
        self.cin = m.conv1[0].in_channels                    # 2 (FlowNetS/FlowNetS.py:17), 6 inside the FlowNet2 stack
        self.x8 = new(B, H, W, self.cin)
        self.a1 = new(B, *hs[1], 64)
        self.cat = {2: new(B, *hs[2], 194), 3: new(B, *hs[3], 386), 4: new(B, *hs[4], 770), 5: new(B, *hs[5], 1026)}
        self.skip_c = {2: 128, 3: 256, 4: 512, 5: 512}
        self.a3, self.a4, self.a5 = new(B, *hs[3], 256), new(B, *hs[4], 512), new(B, *hs[5], 512)
        self.a6, self.a61 = new(B, *hs[6], 1024), new(B, *hs[6], 1024)
        enc_out = {"conv1": (1, 64), "conv2": (2, 128), "conv3": (3, 256), "conv3_1": (3, 256), "conv4": (4, 512),
                   "conv4_1": (4, 512), "conv5": (5, 512), "conv5_1": (5, 512), "conv6": (6, 1024), "conv6_1": (6, 1024)}
        self.raw = {n: new(B, *hs[l], c) for n, (l, c) in enc_out.items()} if self.bn else {}
        self.flow0 = new(B, 256, 256, 2, dtype=F32, pad=2)
        # where each encoder conv reads / writes
        c = self.cat
        self.enc_io = {
            "conv1": (self.x8, self.a1), "conv2": (self.a1, c[2].slice(0, 128)), "conv3": (c[2].slice(0, 128), self.a3),
            "conv3_1": (self.a3, c[3].slice(0, 256)), "conv4": (c[3].slice(0, 256), self.a4),
            "conv4_1": (self.a4, c[4].slice(0, 512)), "conv5": (c[4].slice(0, 512), self.a5),
            "conv5_1": (self.a5, c[5].slice(0, 512)), "conv6": (c[5].slice(0, 512), self.a6), "conv6_1": (self.a6, self.a61)}
        self.grads_ready = False

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool) -> List[torch.Tensor]:
        L, c = self.layers, self.cat
        self.training_cache = training
        self.pack_weights()
        x = x.contiguous()
        nchw_to_view(x, 0, self.cin, self.x8)
        for name, *_ in ENCODER:
            src, dst = self.enc_io[name]
            if self.bn:
                L[name].run_fwd_form(src, self.raw[name])
                self.bns[name].forward(self.raw[name], dst, training)
            else:
                L[name].run_fwd_form(src, dst, slope=SLOPE)
        self.decoder_forward()
        f2 = self.flow32[2]
        _lib.call("mireg_resize_bilinear_fwd", f2.ptr, self.flow0.ptr, self.B, 2, f2.H, f2.W, 256, 256,
                  f2.H * f2.W * 2, 1, 2, 256 * 256 * 2, 1, 2, 0, _stream())
        flows = [self.flow0.nchw(), f2.nchw()]
        if training:
            flows += [self.flow32[l].nchw() for l in (3, 4, 5, 6)]
        return flows

    # ------------------------------------------------------------------------------------------------
    def _ensure_grad_buffers(self) -> None:
        if self.grads_ready:
            return
        new, hs, B = self.ws.new, self.hs, self.B
        self.setup_decoder_grads()
        self.da1, self.da3, self.da4 = new(B, *hs[1], 64), new(B, *hs[3], 256), new(B, *hs[4], 512)
        self.da5, self.da6 = new(B, *hs[5], 512), new(B, *hs[6], 1024)
        self.draw = {n: new(B, v.H, v.W, v.C) for n, v in self.raw.items()}
        dc = self.dcat
        self.dx8 = new(B, self.H, self.W, self.cin) if getattr(self, "want_dx", False) else None
        self.enc_dio = {  # (grad wrt conv input, accumulate?), grad wrt conv output (activated)
            "conv1": (self.dx8, False, self.da1), "conv2": (self.da1, False, dc[2].slice(0, 128)),
            "conv3": (dc[2].slice(0, 128), True, self.da3), "conv3_1": (self.da3, False, dc[3].slice(0, 256)),
            "conv4": (dc[3].slice(0, 256), True, self.da4), "conv4_1": (self.da4, False, dc[4].slice(0, 512)),
            "conv5": (dc[4].slice(0, 512), True, self.da5), "conv5_1": (self.da5, False, dc[5].slice(0, 512)),
            "conv6": (dc[5].slice(0, 512), True, self.da6), "conv6_1": (self.da6, False, self.da61)}
        self.grads_ready = True

    DEC_LAYERS = [f"deconv{l}" for l in DECONV] + [f"predict_flow{l}" for l in PREDICT] + [f"up{l}" for l in (6, 5, 4, 3)]
    # (measured, round 3: a third encoder phase -- conv4_1 conv4 conv3_1 | conv3 conv2 conv1, so that less optimizer work stays exposed
    # after the backward's last kernel -- is 1-2 % SLOWER: every phase boundary is a join of the backward-weights stream)
    PHASE_ENC = tuple(tuple(t.split()) for t in os.environ.get("MIREG_PHASE_ENC", "conv6_1 conv6 conv5_1 conv5|conv4_1 conv4 conv3_1 conv3 conv2 conv1").split("|"))

    def backward_phases(self, gflows: Sequence[Optional[torch.Tensor]]):
        """The backward pass cut where gradient buckets complete: decoder | conv6_1..conv5 | conv4_1..conv1.
        In parameter order these are the tail, the middle and the head of the flat gradient buffer, so the trainer
        can start the all-reduce of a finished bucket while the next phase computes."""
        self._ensure_grad_buffers()
        g = list(gflows) + [None] * (6 - len(gflows))

        def decoder():
            self.decoder_backward({2: g[1], 3: g[2], 4: g[3], 5: g[4], 6: g[5]}, g[0])
            if not self.defer_grad_reduce:
                self.join_side()
                self.unpack_grads(self.DEC_LAYERS)

        def encoder(names):
            def run():
                for name in names:
                    src, dst = self.enc_io[name]
                    dsrc, acc, ddst = self.enc_dio[name]
                    self.chain_backward(name, src, dst, dsrc, acc, ddst)
                if not self.defer_grad_reduce:
                    self.join_side()
                    self.unpack_grads(names)
            return run
        return [decoder] + [encoder(names) for names in self.PHASE_ENC]

    def phase_layers(self):
        """(layer names, BatchNorm names) of every backward phase, in phase order."""
        bn = lambda names: tuple(names) if self.bn else ()
        return [(tuple(self.DEC_LAYERS), ())] + [(names, bn(names)) for names in self.PHASE_ENC]

    def phase_ranges(self) -> List[Tuple[int, int]]:
        bn = lambda names: names if self.bn else ()
        return [self.flat_range(self.DEC_LAYERS)] + [self.flat_range(names, bn(names)) for names in self.PHASE_ENC]

    def input_grad(self) -> torch.Tensor:
        """d loss / d x as (B, cin, H, W) fp32 (only with `want_dx`, set before the first backward)."""
        if self.dx8 is None:
            raise RuntimeError("FlowNetSEngine: the input gradient was not requested before the gradient buffers were built")
        return self.dx8.nchw().float()

    def backward(self, gflows: Sequence[Optional[torch.Tensor]]) -> None:
        """gflows: gradients wrt (flow0, flow2, flow3, flow4, flow5, flow6) as (B,2,h,w) fp32 or None."""
        for phase in self.backward_phases(gflows):
            phase()


class _FlowNetSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine_for(x)
        flows = eng.forward(x, module.training)
        if module.training:
            count_bn_batches(eng)
        ctx.eng, ctx.module = eng, module
        return tuple(flows)

    @staticmethod
    def backward(ctx, *gflows):
        eng = ctx.eng
        if eng.training_cache:
            g = gflows
        else:  # eval arity (flow0, flow2)
            g = (gflows[0], gflows[1], None, None, None, None)
        eng.autograd_backward(g, ctx.module._findex)
        table = eng.param_grads()
        grads = grads_for_autograd(ctx.module.parameters(), table)
        return (None, None) + grads


class FlowNetS(nn.Module, PackedOptimizerHook):
    """Drop-in for reference FlowNetS.FlowNetS.FlowNetS (constructor `batchNorm=True`).

    precision: "bf16" (bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the throughput
    mode BASELINE.json names) or "fp32" (exact-fp32 MFMA, the parity mode)."""
    expansion = 1

    def __init__(self, batchNorm: bool = True, precision: str = "bf16"):
        super().__init__()
        self.batchNorm = batchNorm
        self.precision = precision
        for name, cin, cout, k, s in ENCODER:
            setattr(self, name, conv_block(batchNorm, cin, cout, k, s))
        for lvl, (cin, cout) in DECONV.items():
            setattr(self, f"deconv{lvl}", nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=False),
                                                       nn.LeakyReLU(SLOPE, inplace=True)))
        for lvl, cin in PREDICT.items():
            setattr(self, f"predict_flow{lvl}", nn.Conv2d(cin, 2, 3, 1, 1, bias=False))
        for lvl in (6, 5, 4, 3):
            setattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False))
        for m in self.modules():  # FlowNetS/FlowNetS.py:44-51 (second positional of kaiming_normal_ is a=0.1)
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, 0.1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._engines: Dict[tuple, FlowNetSEngine] = {}

    def engine_for(self, x: torch.Tensor) -> FlowNetSEngine:
        if not x.is_cuda:
            raise RuntimeError("mireg.FlowNetS runs on the MI355X only; there is no CPU fallback")
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        p0 = next(self.parameters())
        if p0.device != x.device:
            raise RuntimeError("model and input are on different devices")
        key = (tuple(x.shape), x.device, dtype, p0.data_ptr())
        if key not in self._engines:
            drop_engines(self)
            B, C, H, W = x.shape
            if C != 2:
                raise RuntimeError(f"FlowNetS expects (B,2,H,W) [fixed, moving], got {tuple(x.shape)}")
            self._engines[key] = FlowNetSEngine(self, B, H, W, x.device, dtype)
            install_bn_counter_hook(self)
        return self._engines[key]

    def forward(self, x):
        flows = _FlowNetSFn.apply(self, x.float(), *self.parameters())
        return tuple(flows)

    def weight_parameters(self):
        return [p for n, p in self.named_parameters() if "weight" in n]

    def bias_parameters(self):
        return [p for n, p in self.named_parameters() if "bias" in n]
