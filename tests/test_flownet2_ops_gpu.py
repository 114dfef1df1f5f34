"""GPU parity of the FlowNet2 glue layers (flownet2/models.py:40-88,136-180) against the oracle's restatement of their
published definitions (the external CUDA sources are absent and unpinned: parity unpinned, stated in the oracle)."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("mag", [0.4, 3.0, 40.0])          # sub-pixel, a few pixels, far outside (border clamp)
def test_resample2d_fwd_bwd(mag):
    import mireg
    B, C, H, W = 2, 3, 20, 28
    src = nets.analytic_input((B, C, H, W), seed=1).requires_grad_(True)
    flow = (mag * nets.analytic_input((B, 2, H, W), seed=2, lo=-1.0, hi=1.0)).requires_grad_(True)
    g = nets.analytic_input((B, C, H, W), seed=3, lo=-1.0, hi=1.0)
    y = oops.resample2d(src, flow)
    y.backward(g)
    sd, fd = src.detach().to(DEV).requires_grad_(True), flow.detach().to(DEV).requires_grad_(True)
    yd = mireg.Resample2d()(sd, fd)
    yd.backward(g.to(DEV))
    assert (yd.cpu() - y.detach()).abs().max().item() < 2e-5
    assert _rel(fd.grad.cpu(), flow.grad) < 1e-4
    assert _rel(sd.grad.cpu(), src.grad) < 1e-4
    # the flownet2 call sites warp an input image (no gradient to it): only d/d flow is produced
    s2 = src.detach().to(DEV)
    f2 = flow.detach().to(DEV).requires_grad_(True)
    mireg.Resample2d()(s2, f2).backward(g.to(DEV))
    assert _rel(f2.grad.cpu(), flow.grad) < 1e-4


def test_channelnorm_and_upsample():
    import mireg
    x = nets.analytic_input((2, 5, 12, 10), seed=4, lo=-1.0, hi=1.0).requires_grad_(True)
    g = nets.analytic_input((2, 1, 12, 10), seed=5, lo=-1.0, hi=1.0)
    y = oops.channelnorm(x)
    y.backward(g)
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = mireg.ChannelNorm()(xd)
    yd.backward(g.to(DEV))
    assert yd.shape == (2, 1, 12, 10) and (yd.cpu() - y.detach()).abs().max().item() < 1e-6
    assert _rel(xd.grad.cpu(), x.grad) < 1e-5
    for mode in ("nearest", "bilinear"):
        ref = torch.nn.Upsample(scale_factor=4, mode=mode)
        xr = x.detach().clone().requires_grad_(True)
        yr = ref(xr)
        gu = nets.analytic_input(tuple(yr.shape), seed=6, lo=-1.0, hi=1.0)
        yr.backward(gu)
        xg = x.detach().to(DEV).requires_grad_(True)
        yg = mireg.Upsample(4, mode)(xg)
        yg.backward(gu.to(DEV))
        assert (yg.cpu() - yr.detach()).abs().max().item() < 1e-5, mode
        assert _rel(xg.grad.cpu(), xr.grad) < 1e-5, mode
