"""Data-parallel host logic (one process per GPU, RCCL over xGMI; `gloo` in the CPU tests).

The registration path shards by pairs: every rank holds a full replica and B pairs.
Two exchanges exist, both tiny next to the compute:
  1. gradients: ONE sum all-reduce over the flat fp32 gradient buffer (views of it are what the wgrad /
     BatchNorm kernels write), then Adam scales by 1/world inside the kernel;
  2. (optional, `sync_loss_stats=True`) the global-NCC term of OFEloss is a whole-batch statistic with an
     extra 1/B (reference loss.py:55-62): all-reducing the per-scale moment vectors
     {Sx,Sy,Sxy,Sxx,Syy,Scharb,Ssmooth} (n x 8 doubles) BEFORE finalisation reproduces the single-process
     loss of the concatenated batch exactly; without it each rank uses its local NCC (documented DP semantics).
BatchNorm batch statistics stay per rank (the reference never ran multi-GPU).
The functions below are backend-agnostic so the N>1 path is covered on CPU with gloo.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_reduce_gradients_(flat_grad: torch.Tensor, group=None, bucket_bytes: int = 0) -> None:
    """Sum all-reduce of the flat gradient buffer, optionally in buckets (xGMI is point-to-point: a few
    large messages keep all 7 links busy; buckets exist so the reduction can start before backward ends)."""
    if world_size(group) == 1:
        return
    if not bucket_bytes or flat_grad.numel() * flat_grad.element_size() <= bucket_bytes:
        dist.all_reduce(flat_grad, group=group)
        return
    step = max(1, bucket_bytes // flat_grad.element_size())
    for o in range(0, flat_grad.numel(), step):
        dist.all_reduce(flat_grad[o:o + step], group=group)


def all_reduce_loss_moments_(sums: torch.Tensor, group=None) -> None:
    """sums: (n_scales, 8) float64 moment table of the local pairs -> table of the global batch."""
    if world_size(group) > 1:
        dist.all_reduce(sums, group=group)


def finalize_ofe(sums: torch.Tensor, npix: Sequence[int], B: int, lamb_da: float = 0.5, gamma: float = 100.0,
                 zeta: float = 100.0) -> torch.Tensor:
    """Host mirror of csrc/warp_loss.hip:ofe_finalize_kernel (float64): moment table -> (p, c, s, total)."""
    s = sums.to(torch.float64).cpu()
    n = s.shape[0]
    p = c = sm = 0.0
    for i in range(n):
        N = float(npix[i])
        w = 0.05 * (i + 1)
        sx, sy, sxy, sxx, syy, sch, ssm = [float(v) for v in s[i, :7]]
        vxy, vxx, vyy = sxy - sx * sy / N, sxx - sx * sx / N, syy - sy * sy / N
        degenerate = not (vxx > 1e-12 * max(sxx, 1e-300)) or not (vyy > 1e-12 * max(syy, 1e-300))
        corr = 1.0 if degenerate else (1.0 / B) * vxy / (vxx ** 0.5 * vyy ** 0.5)
        p += w * sch / B
        c += w * (1.0 - corr)
        sm += w * ssm / 2.0 / B
    p, c, sm = gamma * p / n, zeta * c / n, lamb_da * sm / n
    return torch.tensor([p, c, sm, p + sm + c], dtype=torch.float64)


def broadcast_module_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica identical (parameters and BatchNorm buffers) before training starts."""
    if world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)
