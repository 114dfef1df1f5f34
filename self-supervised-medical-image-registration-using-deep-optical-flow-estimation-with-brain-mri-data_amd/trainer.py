"""Fused training / evaluation step of the registration hot path (reference train.py:41-65 `epoch`).

One call = predictor forward -> stn warp at every scale (+ loss moments fused into the warp kernel)
-> OFEloss -> backward of all of it -> (RCCL gradient all-reduce) -> Adam, as one hand-scheduled
kernel sequence over persistent buffers; no autograd graph, no host synchronisation, optionally
replayed from a hipGraph.  Numerically it is the same computation as

    flows, warped, _, _ = model(imgs); p, c, s, loss = OFEloss(flows, warped, fixed)
    optimizer.zero_grad(); loss.backward(); optimizer.step()        # Adam(lr, (.9,.999), eps=1e-4)

(tests/test_trainer_gpu.py checks the two against each other and against the CPU oracle).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.nn as nn

from . import _lib
from . import dist as mdist
from .engine import PROFILER, AdamJob, F32, TailJob, _stream, upload_table, zero_tensors
from .ops import SLOTS


def flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-home every parameter as a view of one flat fp32 buffer (values preserved)."""
    params = list(module.parameters())
    dev = params[0].device
    flat = torch.empty(sum(p.numel() for p in params), device=dev, dtype=F32)
    o = 0
    for p in params:
        n = p.numel()
        flat[o:o + n].copy_(p.data.reshape(-1))
        p.data = flat[o:o + n].view(p.shape)
        o += n
    return flat


TAIL_PREPARE_EARLY = os.environ.get("MIREG_TAIL_SERIAL", "0") != "1"     # A/B switch


class FusedRegLoss:
    """stn warp + OFEloss forward/backward on engine buffers (K9-K13 fused; no autograd)."""

    def __init__(self, B: int, H: int, W: int, sizes: Sequence[tuple], device, lamb_da=0.5, gamma=100.0, zeta=100.0):
        self.B, self.H, self.W, self.sizes, self.dev = B, H, W, list(sizes), device
        self.hyper = (float(lamb_da), float(gamma), float(zeta))
        n = len(sizes)
        self.n = n
        z = lambda *s: torch.zeros(*s, device=device, dtype=F32)
        self.moving_r = [z(B, 1, h, w) for h, w in sizes]
        self.fixed_r = [z(B, 1, h, w) for h, w in sizes]
        self.warped = [z(B, 1, h, w) for h, w in sizes]
        self.gwarped = [z(B, 1, h, w) for h, w in sizes]
        self.gflow = [z(B, 2, h, w) for h, w in sizes]
        self.sums = torch.zeros(n, SLOTS, 8, device=device, dtype=torch.float64)
        self.npix = torch.tensor([B * h * w for h, w in sizes], dtype=torch.int64, device=device)
        self.out4 = torch.zeros(4, device=device, dtype=torch.float64)
        self.g4 = torch.tensor([0.0, 0.0, 0.0, 1.0], device=device, dtype=torch.float64)
        self.coef = torch.zeros(n, 8, device=device, dtype=F32)

    fused = True        # one launch per phase for all scales (mireg_tail_*); False = per-scale entry points

    def _table(self, flows: Sequence[torch.Tensor]):
        key = tuple((f.data_ptr(), f.stride()) for f in flows)
        if getattr(self, "_tab_key", None) != key:
            jobs, blk = [], 0
            for i, (h, w) in enumerate(self.sizes):
                f, g = flows[i], self.gflow[i]
                sb, sc, sy, sx = f.stride()
                j = TailJob()
                j.flow, j.fsb, j.fsc, j.fsp = f.data_ptr(), sb, sc, (sx if w > 1 else (sy if h > 1 else 1))
                j.moving_r, j.fixed_r, j.warped = self.moving_r[i].data_ptr(), self.fixed_r[i].data_ptr(), self.warped[i].data_ptr()
                j.gflow, j.gsb, j.gsc, j.gsp = g.data_ptr(), 2 * h * w, h * w, 1
                j.sums, j.coef = self.sums[i].data_ptr(), self.coef[i].data_ptr()
                j.h, j.w, j.blk0 = h, w, blk
                blk += (self.B * h * w + 1023) // 1024          # MIREG_TAIL_PIXELS_PER_BLOCK
                jobs.append(j)
            self._tab, self._tab_blocks, self._tab_key = upload_table(jobs, self.dev), blk, key
        return self._tab, self._tab_blocks

    def prepare(self, x: torch.Tensor) -> bool:
        """The part of forward() that depends on the batch only -- clearing the moment table and resizing fixed / moving to every flow
        scale -- on the current stream, so that it can run next to the last kernels of the predictor's forward pass.  Needs the job
        table of an earlier forward() (the flow buffers are the engine's, fixed after the first step); False when there is none yet."""
        if not self.fused or getattr(self, "_tab_key", None) is None:
            return False
        B, H, W = self.B, self.H, self.W
        zero_tensors([self.sums])
        npx = sum(B * h * w for h, w in self.sizes)
        PROFILER.call("tail_resize", 8.0 * B * H * W + 8.0 * npx, "tail:resize", "mireg_tail_resize", self._tab.data_ptr(), self.n,
                      self._tab_blocks, x.data_ptr(), B, H, W, _stream(), unit="B")
        self._prepared = True
        return True

    def forward(self, x: torch.Tensor, flows: Sequence[torch.Tensor]) -> torch.Tensor:
        """x: (B,2,H,W) contiguous fp32 [fixed, moving]; flows[i]: logical (B,2,h,w) fp32 (any pixel stride)."""
        B, H, W, st = self.B, self.H, self.W, _stream()
        HW = H * W
        fixed_ptr, moving_ptr = x.data_ptr(), x.data_ptr() + HW * 4
        prepared, self._prepared = getattr(self, "_prepared", False), False
        if self.fused:
            key = getattr(self, "_tab_key", None)
            tab, blocks = self._table(flows)
            npx = sum(B * h * w for h, w in self.sizes)          # algorithmic HBM bytes: DESIGN.md section 6
            if not prepared or key != self._tab_key:             # (a prepare() against a stale table is simply redone)
                zero_tensors([self.sums])                        # HIP fill (no ATen launch inside the step)
                PROFILER.call("tail_resize", 8.0 * B * H * W + 8.0 * npx, "tail:resize", "mireg_tail_resize", tab.data_ptr(), self.n,
                              blocks, x.data_ptr(), B, H, W, st, unit="B")
            PROFILER.call("tail_fwd", 20.0 * npx, "tail:fwd", "mireg_tail_fwd", tab.data_ptr(), self.n, blocks, B, st, unit="B")
            return self.sums
        zero_tensors([self.sums])
        for i, (h, w) in enumerate(self.sizes):
            _lib.call("mireg_resize_bilinear_fwd", moving_ptr, self.moving_r[i].data_ptr(), B, 1, H, W, h, w,
                      2 * HW, HW, 1, h * w, h * w, 1, 1, st)
            _lib.call("mireg_resize_bilinear_fwd", fixed_ptr, self.fixed_r[i].data_ptr(), B, 1, H, W, h, w,
                      2 * HW, HW, 1, h * w, h * w, 1, 0, st)
            f = flows[i]
            sb, sc, sy, sx = f.stride()
            sp = sx if w > 1 else (sy if h > 1 else 1)
            _lib.call("mireg_stn_warp_fwd", f.data_ptr(), sb, sc, sp, self.moving_r[i].data_ptr(),
                      self.fixed_r[i].data_ptr(), self.warped[i].data_ptr(), self.sums[i].data_ptr(), B, 1, h, w, st)
            _lib.call("mireg_smoothness_fwd", f.data_ptr(), sb, sc, sp, self.sums[i, 0, 6:].data_ptr(), B, h, w, st)
        return self.sums

    def _npix(self, B_global: Optional[int]) -> torch.Tensor:
        if not B_global or B_global == self.B:
            return self.npix
        if getattr(self, "_npix_g", None) is None:
            self._npix_g = self.npix * (B_global // self.B)
        return self._npix_g

    def finalize(self, B_global: Optional[int] = None) -> torch.Tensor:
        lamb_da, gamma, zeta = self.hyper
        _lib.call("mireg_ofe_finalize", self.sums.data_ptr(), self._npix(B_global).data_ptr(), self.n, B_global or self.B, lamb_da,
                  gamma, zeta, self.out4.data_ptr(), _stream())
        return self.out4

    def backward(self, flows: Sequence[torch.Tensor], B_global: Optional[int] = None) -> List[torch.Tensor]:
        B, st = self.B, _stream()
        lamb_da, gamma, zeta = self.hyper
        _lib.call("mireg_ofe_bwd_coef", self.sums.data_ptr(), self._npix(B_global).data_ptr(), self.n, B_global or B, lamb_da,
                  gamma, zeta, self.g4.data_ptr(), self.coef.data_ptr(), st)
        if self.fused:
            tab, blocks = self._table(flows)
            PROFILER.call("tail_bwd", 28.0 * sum(B * h * w for h, w in self.sizes), "tail:bwd", "mireg_tail_bwd", tab.data_ptr(),
                          self.n, blocks, B, st, unit="B")
            return self.gflow
        for i, (h, w) in enumerate(self.sizes):
            f = flows[i]
            sb, sc, sy, sx = f.stride()
            sp = sx if w > 1 else (sy if h > 1 else 1)
            _lib.call("mireg_loss_bwd", self.warped[i].data_ptr(), self.fixed_r[i].data_ptr(), self.coef[i].data_ptr(),
                      self.gwarped[i].data_ptr(), B * h * w, st)
            g = self.gflow[i]
            _lib.call("mireg_stn_warp_bwd", f.data_ptr(), sb, sc, sp, self.moving_r[i].data_ptr(),
                      self.gwarped[i].data_ptr(), g.data_ptr(), 2 * h * w, h * w, 1, 0.0, B, 1, h, w, st)
            _lib.call("mireg_smoothness_bwd", f.data_ptr(), sb, sc, sp, self.coef[i].data_ptr(), g.data_ptr(),
                      2 * h * w, h * w, 1, 1.0, B, h, w, st)
        return self.gflow


class RegistrationTrainer:
    """Owns the fused step for one opticalFlowReg on one GPU (one process per GPU under DP)."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-4,
                 lamb_da: float = 0.5, gamma: float = 100.0, zeta: float = 100.0, use_graph: bool = True,
                 process_group=None, sync_loss_stats: bool = False, overlap: bool = True, packed_optimizer: bool = True,
                 autotune: bool = True, tune_cache: Optional[str] = None, overlap_optimizer: bool = True):
        self.model = model
        self.predictor = model.predictor
        if not hasattr(self.predictor, "engine_for"):
            raise RuntimeError(f"RegistrationTrainer: predictor {type(self.predictor).__name__} has no fused engine "
                               "(engine_for); train it through autograd: loss.backward() + mireg.Adam")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.loss_hyper = (lamb_da, gamma, zeta)
        self.use_graph = use_graph
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.sync_loss_stats = sync_loss_stats and self.world > 1
        # Adam's gradient scale under data parallelism.  Per-rank losses are normalised by the LOCAL batch, so the summed
        # all-reduce is divided by world.  With sync_loss_stats the loss coefficients already carry the GLOBAL batch
        # (finalize / ofe_bwd_coef receive B_global): the summed all-reduce then IS the gradient of the concatenated batch.
        self.grad_scale = 1.0 if self.sync_loss_stats else 1.0 / self.world
        self.overlap = overlap
        self._graphs = None
        self._seg_ranges = [None]
        self.packed = packed_optimizer
        self.autotune = autotune
        self.overlap_optimizer = overlap_optimizer   # single GPU: Adam of a finished backward phase runs under the next phase
        self._tuning = False
        self._opt_stream = None
        self._phase_tabs = None
        self.tune_cache = tune_cache        # JSON file of measured launch shapes: loaded if present, written after tuning
        self._packs_fresh = False
        self.flat_p = flatten_parameters(model)
        dev = self.flat_p.device
        self.flat_g = None if self.packed else torch.zeros_like(self.flat_p)     # packed: sized in _setup (engine layout)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self._adam_tab = None
        model.register_load_state_dict_post_hook(lambda *_: self.refresh_packs())
        self.eng = None
        self.loss = None
        self.x_static = None
        self._graph_fb = None
        self._graph_opt = None
        self._warm = 0

    # ------------------------------------------------------------------------------------------
    def _setup(self, x: torch.Tensor) -> None:
        self.predictor.train()
        self.eng = self.predictor.engine_for(x)
        params, dev = list(self.model.parameters()), self.flat_p.device
        if self.packed and max(l.kh * l.kw for l in self.eng.layers.values()) > 49:
            self.packed = False
        if self.packed:
            # gradients live in the packed GEMM domain: conv weights as [Co][(ky,kx,ci_pad)], the rest as in torch
            off, _, total = self.eng.packed_layout(params)
            self.flat_g = torch.zeros(total, device=dev, dtype=F32)
            self.eng.bind_packed_grads(params, self.flat_g)
            wl = {id(l.weight) for l in self.eng.layers.values()}
            jobs, o = [], 0
            for p in params:
                if id(p) not in wl:
                    jobs.append(AdamJob(p.data_ptr(), self.flat_g.data_ptr() + 4 * off[id(p)], self.flat_m.data_ptr() + 4 * o,
                                        self.flat_v.data_ptr() + 4 * o, p.numel()))
                o += p.numel()
            self._adam_tab, self._adam_n = upload_table(jobs, dev), len(jobs)
            self._wopt_tab = None
            self._phase_tabs = None
        else:
            if self.flat_g is None:
                self.flat_g = torch.zeros_like(self.flat_p)
            self.eng.bind_flat_grads(params, self.flat_g)
            self._adam_tab, self._adam_n = upload_table([AdamJob(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                                                 self.flat_v.data_ptr(), self.flat_p.numel())], dev), 1
        self.x_static = torch.empty_like(x, memory_format=torch.contiguous_format)
        # size-discovery forward: its BatchNorm running-statistics update is undone (the first real step does it once)
        bufs = list(self.model.buffers())
        saved = [b.detach().clone() for b in bufs]
        flows = self.eng.forward(self.x_static.copy_(x), True)
        for b, v in zip(bufs, saved):
            b.copy_(v)
        sizes = [tuple(f.shape[2:]) for f in flows]
        B, _, H, W = x.shape
        self.loss = FusedRegLoss(B, H, W, sizes, x.device, *self.loss_hyper)
        if self.autotune:
            import os
            if self.tune_cache and os.path.exists(self.tune_cache):
                self.eng.ws.load_tuning(self.tune_cache)
            else:
                self._autotune()
                if self.tune_cache:
                    self.eng.ws.save_tuning(self.tune_cache)

    def refresh_packs(self) -> None:
        """Re-derive every GEMM pack from the fp32 master weights (after load_state_dict / manual edits)."""
        if self.eng is not None:
            self.eng.pack_weights(force=True)

    def _autotune(self) -> None:
        """One discarded forward+backward in which every contraction site times its candidate launch shapes
        (tile width x split-K) on the real buffers and keeps the fastest; BatchNorm running statistics are restored."""
        bufs = [b for b in self.model.buffers()]
        saved = [b.detach().clone() for b in bufs]
        ws, side = self.eng.ws, type(self.eng).use_side_stream
        ws.tuning, self.eng.use_side_stream, self._tuning = True, False, True
        try:
            self._fwd_bwd()
            torch.cuda.synchronize()
        finally:
            ws.tuning, self.eng.use_side_stream, self._tuning = False, side, False
            for b, v in zip(bufs, saved):
                b.copy_(v)
            self.eng._reduce_table, self.eng._unpack_table = {}, None     # wgrad slabs may have been re-sized

    def _forward_and_loss(self):
        self.eng.packs_fresh = self._packs_fresh          # the fused optimizer rewrote the packs with the weights
        # the loss tail's batch-only part rides the decoder's second stream next to the last head of the forward pass (decoder_forward)
        self.eng.pre_tail = (lambda: self.loss.prepare(self.x_static)) if (TAIL_PREPARE_EARLY and not self._tuning) else None
        flows = self.eng.forward(self.x_static, True)
        self.eng.packs_fresh, self.eng.pre_tail = False, None
        self.loss.forward(self.x_static, flows)
        Bg = self.loss.B * self.world if self.sync_loss_stats else None
        if self.sync_loss_stats:
            mdist.all_reduce_loss_moments_(self.loss.sums, self.pg)
        self.loss.finalize(Bg)
        return self.loss.backward(flows, Bg)

    @property
    def _phase_opt(self) -> bool:
        return (self.packed and self.world == 1 and self.overlap_optimizer and not self._tuning
                and hasattr(self.eng, "backward_phases") and hasattr(self.eng, "phase_layers")
                and getattr(self.eng, "phase_opt_single_gpu", True))

    @property
    def _phase_opt_dp(self) -> bool:
        """Data parallel: Adam of bucket k runs (on the optimizer stream) as soon as its all-reduce has landed."""
        return (self.packed and self.world > 1 and self.overlap and self.overlap_optimizer and not self._tuning
                and hasattr(self.eng, "backward_phases") and hasattr(self.eng, "phase_layers"))

    def _fwd_bwd(self) -> None:
        if not self._phase_opt:
            self.eng.backward(self._forward_and_loss())
            return
        # single GPU: as soon as a backward phase has reduced its gradients, its Adam + FWD re-pack (HBM-bound) runs on a
        # third stream underneath the next phase's contractions; a parallel hipGraph branch under capture
        # (measured: also moving the slab reduce and the wgrad-stream join off the main stream is 3 % SLOWER -- three
        # concurrent kernel streams thrash each other; so only the HBM-bound optimizer work leaves the main stream)
        phases = self.eng.backward_phases(self._forward_and_loss())
        names = [n for n, _ in self.eng.phase_layers()]
        # The optimizer work goes onto the engine's wgrad stream, right behind the phase's backward-weights GEMMs: the
        # hipGraph executor maps parallel branches onto few hardware queues, and a third branch was observed to share the
        # wgrad branch's queue and run only after ALL wgrads (i.e. at the very end of the step).  In-order on the wgrad
        # stream it runs between the phases, underneath the next phase's main-chain kernels.
        if getattr(self.eng, "_side", None) is None:
            self.eng._side = torch.cuda.Stream(device=self.flat_p.device)
        self._opt_stream = self.eng._side
        main = torch.cuda.current_stream()
        for k, phase in enumerate(phases):
            phase()                                         # ends with the wgrad-stream join and the slab reduce
            ev = torch.cuda.Event()
            ev.record(main)
            self._opt_stream.wait_event(ev)
            with torch.cuda.stream(self._opt_stream):
                self._optim_phase(k)
                self.eng.pack_dgrad_subset(names[k])
        main.wait_stream(self._opt_stream)

    def _build_phase_tab(self, k: int):
        """Tables of phase k, built when the phase has run once (its wgrad slabs exist then)."""
        base, dev = self.flat_p.data_ptr(), self.flat_p.device
        off = self.eng.flat_off
        phases = self.eng.phase_layers()
        assert {n for names, _ in phases for n in names} == set(self.eng.layers), "every layer must belong to one backward phase"
        for names, bn_names in phases[k:k + 1]:
            small, jobs, units = [], [], 0
            ps = []
            for n in names:
                l = self.eng.layers[n]
                if l.bias is not None:
                    ps.append(l.bias)
                if l.wgrad_slab is None:
                    continue
                o = l.weight.data_ptr() - base
                j = l.wopt_job(self.flat_m.data_ptr() + o, self.flat_v.data_ptr() + o)
                j.unit0 = units
                units += l.Co * ((l.Cip + 63) // 64)
                jobs.append(j)
            for n in bn_names:
                ps += [self.eng.bns[n].bn.weight, self.eng.bns[n].bn.bias]
            seen = set()
            for p in ps:
                if id(p) in seen:                           # siamese streams share their BatchNorm parameters
                    continue
                seen.add(id(p))
                o = p.data_ptr() - base
                small.append(AdamJob(p.data_ptr(), self.flat_g.data_ptr() + 4 * off[id(p)], self.flat_m.data_ptr() + o,
                                     self.flat_v.data_ptr() + o, p.numel()))
            return (upload_table(small, dev) if small else None, len(small),
                    upload_table(jobs, dev) if jobs else None, len(jobs), units, max([j.taps for j in jobs] or [1]),
                    sum(self.eng.layers[n].weight.numel() for n in names))

    def _optim_phase(self, k: int, scale: float = 1.0) -> None:
        if self._phase_tabs is None:
            self._phase_tabs = {}
        if k not in self._phase_tabs:
            self._phase_tabs[k] = self._build_phase_tab(k)
        st = _stream()
        small, ns, tab, n, units, max_taps, nparam = self._phase_tabs[k]
        tick = 1 if k == 0 else 0
        if tab is not None:
            PROFILER.call("adam_pack", 30.0 * nparam, f"optimizer:phase{k}", "mireg_adam_pack", tab.data_ptr(), n, units, max_taps,
                          self.step_dev.data_ptr(), tick, self.lr, self.betas[0], self.betas[1], self.eps, scale, self.eng.ws.code, st,
                          unit="B")
            tick = 0
        if small is not None:
            _lib.call("mireg_adam_step", small.data_ptr(), ns, self.step_dev.data_ptr(), tick, self.lr, self.betas[0], self.betas[1],
                      self.eps, scale, st)

    def _segments(self):
        """[(callable, flat-gradient range or None)]: the step cut where gradient buckets complete (DP overlap)."""
        if self.world > 1 and self.overlap and hasattr(self.eng, "backward_phases"):
            holder = {}

            def first():
                holder["phases"] = self.eng.backward_phases(self._forward_and_loss())
                holder["phases"][0]()
            ranges = self.eng.phase_ranges()
            segs = [(first, ranges[0])]
            for k in range(1, len(ranges)):
                segs.append(((lambda k=k: holder["phases"][k]()), ranges[k]))
            return segs
        return [(self._fwd_bwd, None)]

    def _optim(self) -> None:
        st = _stream()
        if self._phase_opt:                                 # Adam and both re-packs already ran phase by phase inside _fwd_bwd
            self._packs_fresh = True
            return
        if self._phase_opt_dp:                              # Adam ran per bucket inside _run
            self.eng.pack_weights(dgrad_only=True)
            self._packs_fresh = True
            return
        _lib.call("mireg_adam_step", self._adam_tab.data_ptr(), self._adam_n, self.step_dev.data_ptr(), 1, self.lr, self.betas[0],
                  self.betas[1], self.eps, self.grad_scale, st)
        if not self.packed:
            return
        if self._wopt_tab is None:
            jobs, units, base = [], 0, self.flat_p.data_ptr()
            for l in self.eng.layers.values():
                if l.wgrad_slab is None:                      # layer outside the backward graph: weights never move
                    continue
                o = l.weight.data_ptr() - base
                j = l.wopt_job(self.flat_m.data_ptr() + o, self.flat_v.data_ptr() + o)
                j.unit0 = units
                units += l.Co * ((l.Cip + 63) // 64)
                jobs.append(j)
            self._wopt_tab = (upload_table(jobs, self.flat_p.device), len(jobs), units, max(j.taps for j in jobs))
        tab, n, units, max_taps = self._wopt_tab
        nparam = sum(l.weight.numel() for l in self.eng.layers.values() if l.wgrad_slab is not None)
        PROFILER.call("adam_pack", 30.0 * nparam, "optimizer", "mireg_adam_pack", tab.data_ptr(), n, units, max_taps,
                      self.step_dev.data_ptr(), 0, self.lr, self.betas[0], self.betas[1], self.eps, self.grad_scale,
                      self.eng.ws.code, st, unit="B")
        self.eng.pack_weights(dgrad_only=True)
        self._packs_fresh = True

    def _run(self, runners) -> None:
        """runners[i]() executes segment i (eagerly or as a hipGraph replay); finished buckets are all-reduced
        asynchronously (RCCL runs on its own stream) while the next segment computes."""
        works = []
        dp_opt = self._phase_opt_dp and len(runners) == len(self.eng.phase_layers())
        if dp_opt and self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.flat_p.device)

        def optimise(k: int) -> None:                      # on the optimizer stream, behind bucket k's all-reduce
            with torch.cuda.stream(self._opt_stream):
                works[k].wait()
                self._optim_phase(k, self.grad_scale)
        for k, (run, rng) in enumerate(zip(runners, self._seg_ranges)):
            run()
            if self.world > 1:
                buf = self.flat_g if rng is None else self.flat_g[rng[0]:rng[1]]
                works.append(torch.distributed.all_reduce(buf, group=self.pg, async_op=True))
                if dp_opt and k > 0:
                    optimise(k - 1)
        if dp_opt:
            optimise(len(works) - 1)
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            return
        for w in works:
            w.wait()

    def step(self, x: torch.Tensor) -> torch.Tensor:
        """One optimizer step on batch x (B,2,H,W) fp32 on device.  Returns the device tensor
        (photo, corr, smooth, total) in float64 WITHOUT synchronising (call .tolist() when needed)."""
        if not x.is_cuda:
            raise RuntimeError("RegistrationTrainer.step needs a device batch (there is no CPU path)")
        if self.eng is not None and tuple(x.shape) != tuple(self.x_static.shape):
            # e.g. the short last batch of an epoch: rebuild the per-shape state (engine buffers, tables, graphs)
            torch.cuda.synchronize()
            self.eng, self._graphs, self._graph_opt, self._warm, self._packs_fresh = None, None, None, 0, False
        if self.eng is None:
            self._setup(x)
        self.x_static.copy_(x)
        graphable = self.use_graph and not self.sync_loss_stats
        if graphable and self._warm >= 2:
            if self._graphs is None:
                self._capture()
            self._run([g.replay for g in self._graphs])
            if self._graph_opt is not None:
                self._graph_opt.replay()
        else:
            segs = self._segments()
            self._seg_ranges = [r for _, r in segs]
            self._run([f for f, _ in segs])
            self._optim()
            self._warm += 1
        for b in self.eng.bns.values():                     # BatchNorm2d.num_batches_tracked, added when a state_dict is taken
            b.pending += 1
        return self.loss.out4

    def _capture(self) -> None:
        # A cyclic-garbage collection that fires while a stream is capturing can free device memory or another trainer's graphs
        # inside the capture (hipFree / hipGraphExecDestroy are illegal there and abort the process): collect now, then keep the
        # collector off until the capture is over.
        import gc
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_graphs()
        finally:
            if was_enabled:
                gc.enable()

    def _capture_graphs(self) -> None:
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        segs = self._segments()
        self._seg_ranges = [r for _, r in segs]
        self._graphs = []
        with torch.cuda.stream(s):
            for fn, _ in segs:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    fn()
                self._graphs.append(gr)
            if self._phase_opt:                             # nothing left to launch after the backward graph
                self._graph_opt = None
                self._optim()
            else:
                self._graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_opt, stream=s):
                    self._optim()
        torch.cuda.current_stream().wait_stream(s)
        self._graph_fb = self._graphs[0]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, segs: Optional[torch.Tensor] = None, metrics: bool = False) -> Dict[str, torch.Tensor]:
        """Eval forward + OFEloss (+ warped-segmentation Dice per sample), reference inference.py:43-68; metrics=True adds the
        per-sample MSE / PSNR / Pearson / mutual information of inference.py:69-75 (utils.py:41-59), all left on the device."""
        from . import ops
        self.model.eval()
        flows, warped, wseg, _ = self.model(x, segs)
        p, c, s, t = ops.OFEloss(flows, warped, x[:, 0:1].contiguous(), *self.loss_hyper)
        out = {"loss": torch.stack((p, c, s, t)), "flow": flows[0]}
        if segs is not None and tuple(wseg.shape[2:]) == tuple(segs.shape[2:]):
            # predictors whose finest flow is not at image resolution (FlowNetC: 64x64) have no Dice in the reference either
            out["dice"] = ops.dice_batch(segs[:, 0:1].float().contiguous(), wseg)
        if metrics and tuple(warped[0].shape) == tuple(x[:, 0:1].shape):
            from .metrics import pair_metrics
            out.update(pair_metrics(x[:, 0:1].contiguous(), warped[0]))
        self.model.train()
        return out

    def optimizer_state_dict(self) -> dict:
        """torch.optim.Adam-compatible state (reference checkpoints store optimizer_state_dict, train.py:183-188)."""
        state, o = {}, 0
        step = self.step_dev.to(torch.float32).cpu().reshape(())
        for i, p in enumerate(self.model.parameters()):
            n = p.numel()
            state[i] = {"step": step.clone(), "exp_avg": self.flat_m[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.flat_v[o:o + n].view(p.shape).clone()}
            o += n
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def set_hyper(self, lr: Optional[float] = None, betas=None, eps: Optional[float] = None) -> None:
        """Change Adam's hyper-parameters.  They are by-value kernel arguments, i.e. frozen into captured hipGraphs, so the
        graphs are dropped and re-captured on the next step (an lr schedule costs one capture per change)."""
        new = (self.lr if lr is None else float(lr), self.betas if betas is None else (float(betas[0]), float(betas[1])),
               self.eps if eps is None else float(eps))
        if new != (self.lr, self.betas, self.eps):
            self.lr, self.betas, self.eps = new
            if self._graphs is not None:
                torch.cuda.synchronize()
            self._graphs, self._graph_opt, self._graph_fb = None, None, None

    def load_optimizer_state_dict(self, sd: dict) -> None:
        groups = sd.get("param_groups") or []
        if groups:                                          # reference resume: optim.load_state_dict restores these too (train.py:154)
            g0 = groups[0]
            self.set_hyper(g0.get("lr"), g0.get("betas"), g0.get("eps"))
        o = 0
        for i, p in enumerate(self.model.parameters()):
            n = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_dev.fill_(int(st["step"]))
            o += n
