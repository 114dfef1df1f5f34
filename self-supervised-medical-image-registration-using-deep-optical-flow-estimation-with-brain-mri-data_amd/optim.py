"""torch.optim.Adam on the HIP multi-tensor kernel (`mireg_adam_step`), for the models that train through torch.autograd
(the volume path).  Same update as torch.optim.Adam without weight decay / amsgrad; the reference builds its optimizer as
Adam(lr, betas=(0.9, 0.999), eps=1e-4) (train.py:129).  The 2-D trainer has its own packed-domain optimizer.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch

from . import _lib
from .engine import AdamJob, _stream, upload_table, zero_tensors


class Adam:
    CHUNK = 1 << 18

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, fuse: Optional[torch.nn.Module] = None):
        """fuse: a module (tree) whose HIP-engine sub-modules offer `fuse_optimizer` (mireg.FlowNetS3D): their convolution weights
        are then updated in the packed domain -- split-K gradient slabs -> Adam on the fp32 master weights -> refreshed bf16 forward
        packs in one pass (`mireg_adam_pack`), so the torch-layout gradient of those weights is never written (`.grad` stays None)
        and the next forward does not re-pack them.  Same update, same state; one backward per step."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("mireg.Adam updates contiguous float32 parameters on the MI355X only; there is no CPU fallback")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.offs, n = [], 0
        for p in self.params:                                  # 16-byte aligned moment slices keep the kernel on its float4 path
            self.offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        dev = self.params[0].device
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self._tab = None
        self._fused = []
        if fuse is not None:
            index = {id(p): i for i, p in enumerate(self.params)}
            for mod in fuse.modules():
                if hasattr(mod, "fuse_optimizer"):
                    mod.fuse_optimizer(self, index)
                    self._fused.append(mod)
            if not self._fused:
                raise ValueError("mireg.Adam(fuse=...): no sub-module with a packed-domain optimizer hook in that module")

    def state_ptrs(self, i: int) -> Tuple[int, int]:
        """Device addresses of parameter i's first / second moment slices (for the packed-domain kernels)."""
        return self.m.data_ptr() + 4 * self.offs[i], self.v.data_ptr() + 4 * self.offs[i]

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Default: keep the gradient tensors and zero them (one fill per tensor).  torch.optim's default drops them instead; with
        stable gradient storage the per-step job table of step() never changes (no host-to-device copy after the first step) and the
        whole training step can be captured into a hipGraph.  Pass set_to_none=True for torch's behaviour."""
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        grads = [p.grad for p in self.params if p.grad is not None and p.grad.is_cuda and p.grad.is_contiguous()]
        for i in range(0, len(grads), 200):                      # one table-driven launch per 200 tensors instead of a fill each
            zero_tensors(grads[i:i + 200])
        for p in self.params:
            if p.grad is not None and not (p.grad.is_cuda and p.grad.is_contiguous()):
                p.grad.zero_()

    def step(self) -> None:
        jobs, keep = [], []
        for p, off in zip(self.params, self.offs):
            n = p.numel()
            if p.grad is not None:
                g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                keep.append(g)
                # the kernel gives every table entry the same few workgroups: cut large tensors into CHUNK-element entries
                for c0 in range(0, n, self.CHUNK):
                    o = 4 * c0
                    jobs.append(AdamJob(p.data_ptr() + o, g.data_ptr() + o, self.m.data_ptr() + 4 * off + o,
                                        self.v.data_ptr() + 4 * off + o, min(self.CHUNK, n - c0)))
        pending = [mod for mod in self._fused if mod.fused_pending()]
        if not jobs and not pending:
            return
        tick = 1
        if jobs:
            self._tab, self._keep = upload_table(jobs, self.params[0].device), keep       # alive until the launch has run
            _lib.call("mireg_adam_step", self._tab.data_ptr(), len(jobs), self.step_dev.data_ptr(), tick, self.lr, self.betas[0],
                      self.betas[1], self.eps, 1.0, _stream())
            tick = 0
        for mod in pending:                                   # same step count: only the first launch of a step ticks it
            mod.fused_step(self, tick)
            tick = 0
