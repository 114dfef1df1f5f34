"""CPU, world_size 2, gloo: the data-parallel host logic (mireg.dist) reproduces single-process results.

The per-rank arithmetic comes from the CPU oracle (these tests exercise the exchange logic and the
moment-table formulation of OFEloss, not the HIP kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nets, ops as oops


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _moments(flows, warped, fixed):
    """(n, 8) float64 table {Sx,Sy,Sxy,Sxx,Syy,Scharb,Ssmooth,0} exactly as the fused warp kernel accumulates it."""
    rows = []
    for f, w in zip(flows, warped):
        fr = oops.resize_bilinear(fixed, w.shape[2:], align_corners=False).double()
        x = w.double()
        down = torch.zeros_like(f); down[:, :, :-1] = f[:, :, 1:]
        right = torch.zeros_like(f); right[..., :-1] = f[..., 1:]
        ssm = (oops.charbonnier(f - down) + oops.charbonnier(f - right)).double().sum()
        rows.append(torch.stack([x.sum(), fr.sum(), (x * fr).sum(), (x * x).sum(), (fr * fr).sum(),
                                 oops.charbonnier((fr - x).float()).double().sum(), ssm, torch.zeros((), dtype=torch.float64)]))
    return torch.stack(rows)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mireg.dist as md
    torch.manual_seed(0)
    torch.set_num_threads(2)
    B, size = 4, 64
    x = nets.analytic_input((B, 2, size, size), seed=5)
    model = nets.OpticalFlowReg("pwc")            # BatchNorm-free predictor: DP must match single-process exactly
    nets.analytic_weights_(model)
    model.train()
    md.broadcast_module_(model)
    lo, hi = rank * B // world, (rank + 1) * B // world
    xs = x[lo:hi]
    flows, warped, _, _ = model(xs)
    fixed = xs[:, 0:1]
    # (1) exact global NCC through the moment table
    sums = _moments([f.detach() for f in flows], [w.detach() for w in warped], fixed)
    md.all_reduce_loss_moments_(sums)
    npix = [B * w.shape[2] * w.shape[3] for w in warped]
    got = md.finalize_ofe(sums, npix, B)
    # (2) gradient sum all-reduce over a flat buffer, bucketed
    from mireg.trainer import flatten_parameters
    flat_p = flatten_parameters(model)
    flows, warped, _, _ = model(xs)
    # local objective whose sum over ranks is the global photometric + smoothness loss (both are sums / B_global)
    local = sum(0.05 * (i + 1) * (100.0 * oops.photometric_loss(fixed, w) * (hi - lo) / B
                                  + 0.5 * oops.smoothness_loss(f) * (hi - lo) / B) for i, (f, w) in enumerate(zip(flows, warped)))
    local.backward()
    flat_g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    md.all_reduce_gradients_(flat_g, bucket_bytes=1 << 20)
    ret[rank] = (got, flat_g.double().norm().item(), flat_g[:1000].clone(), flat_p.numel())
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    # single-process reference on the full batch
    torch.manual_seed(0)
    B, size = 4, 64
    x = nets.analytic_input((B, 2, size, size), seed=5)
    model = nets.OpticalFlowReg("pwc")
    nets.analytic_weights_(model)
    model.train()
    flows, warped, _, _ = model(x)
    fixed = x[:, 0:1]
    ref = oops.ofe_loss([f.detach() for f in flows], [w.detach() for w in warped], fixed)
    for r in range(world):
        got = ret[r][0]
        for a, b in zip(got.tolist(), [v.item() for v in ref]):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (r, got, ref)
    obj = sum(0.05 * (i + 1) * (100.0 * oops.photometric_loss(fixed, w) + 0.5 * oops.smoothness_loss(f))
              for i, (f, w) in enumerate(zip(flows, warped)))
    obj.backward()
    flat_ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    for r in range(world):
        assert ret[r][3] == flat_ref.numel()
        assert abs(ret[r][1] / flat_ref.double().norm().item() - 1) < 1e-4
        assert (ret[r][2] - flat_ref[:1000]).abs().max() <= 1e-4 * flat_ref[:1000].abs().max() + 1e-7
    assert torch.equal(ret[0][2], ret[1][2])      # all ranks hold identical reduced gradients


def test_finalize_matches_oracle_single_rank():
    import mireg.dist as md
    fixed = nets.analytic_input((3, 1, 64, 64), seed=1)
    moving = nets.analytic_input((3, 1, 64, 64), seed=2)
    g = torch.Generator().manual_seed(1)
    flows = [torch.randn(3, 2, s, s, generator=g) for s in (64, 16, 4)]
    warped = [oops.stn(f, moving) for f in flows]
    ref = oops.ofe_loss(flows, warped, fixed)
    got = md.finalize_ofe(_moments(flows, warped, fixed), [w.numel() for w in warped], 3)
    for a, b in zip(got.tolist(), [v.item() for v in ref]):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b))
