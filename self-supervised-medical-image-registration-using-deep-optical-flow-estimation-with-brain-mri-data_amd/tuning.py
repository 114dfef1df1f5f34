"""Launch-shape autotuning for models that train through torch.autograd (the fused RegistrationTrainer tunes by itself).

Every contraction site of the HIP engines can time its candidate launch shapes (kernel: ring / halo-staged / 8-wave tile, tile width,
split-K; backward-weights: kernel and pixel split) on the real buffers and keep the fastest -- `Workspace.tuned` / `tuned_wgrad`.
`autotune(model, run)` switches that on for one discarded `run()` (a forward + `loss.backward()` of the user's own step, without the
optimizer step) and puts everything the pass touched back: BatchNorm running statistics and batch counters, `.grad` of every parameter,
the pending-gradient state of a packed-domain optimizer.  Eager heuristics otherwise: the autograd-mode engines never tune on their own.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn

from .engine import Workspace
from .flownets import PredictorEngineBase


def _engines(model: nn.Module):
    for mod in model.modules():
        for e in getattr(mod, "_engines", {}).values():
            if isinstance(e, PredictorEngineBase):
                yield e


def autotune(model: nn.Module, run: Callable[[], object]) -> int:
    """One discarded run() with every engine of `model` timing its launch shapes; returns the number of tuned contraction sites."""
    params = list(model.parameters())
    grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
    bufs = list(model.buffers())
    saved = [b.detach().clone() for b in bufs]
    pending = {id(b): b.pending for e in _engines(model) for b in e.bns.values()}
    side = PredictorEngineBase.use_side_stream
    Workspace.TUNE_ALL, PredictorEngineBase.use_side_stream = True, False
    try:
        run()
        torch.cuda.synchronize()
    finally:
        Workspace.TUNE_ALL, PredictorEngineBase.use_side_stream = False, side
        with torch.no_grad():
            for b, v in zip(bufs, saved):
                b.copy_(v)
            for p, g in zip(params, grads):
                if g is None:
                    p.grad = None
                else:
                    p.grad.copy_(g)
        sites = 0
        for e in _engines(model):
            e.slab_pending = False                              # a packed-domain optimizer must not consume the discarded gradients
            e._reduce_table, e._unpack_table = {}, None          # backward-weights slabs may have been re-sized
            for b in e.bns.values():
                b.pending = pending.get(id(b), 0)
            sites += len(e.ws.tuned) + len(e.ws.tuned_wgrad)
    return sites
