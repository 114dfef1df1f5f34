"""GPU parity of the volume (3-D) registration tail against torch on the CPU (pinned at op level)."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("src,dst", [((6, 9, 7), (12, 18, 14)), ((8, 8, 8), (3, 5, 2)), ((4, 6, 5), (16, 24, 20)), ((5, 4, 3), (5, 4, 3))])
def test_resize_trilinear_fwd_bwd(align, src, dst):
    from mireg.volume import resize_trilinear
    x = nets.analytic_input((2, 3, *src), seed=1, lo=-1.0, hi=1.0).requires_grad_(True)
    g = nets.analytic_input((2, 3, *dst), seed=2, lo=-1.0, hi=1.0)
    y = oops.resize_trilinear(x, dst, align)
    y.backward(g)
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = resize_trilinear(xd, dst, align)
    yd.backward(g.to(DEV))
    assert (yd.cpu() - y.detach()).abs().max().item() < 1e-5
    assert _rel(xd.grad.cpu(), x.grad) < 1e-5
    # channel-last input (what the predictor hands over) is read in place
    xcl = x.detach().to(DEV).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    assert (resize_trilinear(xcl, dst, align).cpu() - y.detach()).abs().max().item() < 1e-5


@pytest.mark.parametrize("size,full", [((6, 9, 7), (6, 9, 7)), ((4, 5, 3), (8, 10, 6)), ((1, 4, 4), (2, 8, 8))])
def test_stn3d_fwd_bwd(size, full):
    from mireg.volume import stn3d
    B = 2
    flow = (1.7 * nets.analytic_input((B, 3, *size), seed=3, lo=-1.0, hi=1.0)).requires_grad_(True)
    frame = nets.analytic_input((B, 1, *full), seed=4)
    g = nets.analytic_input((B, 1, *size), seed=5, lo=-1.0, hi=1.0)
    y = oops.stn3d(flow, frame)
    y.backward(g)
    fd = flow.detach().to(DEV).requires_grad_(True)
    yd = stn3d(fd, frame.to(DEV))
    yd.backward(g.to(DEV))
    assert (yd.cpu() - y.detach()).abs().max().item() < 2e-5
    assert _rel(fd.grad.cpu(), flow.grad) < 1e-4
    fcl = flow.detach().to(DEV).permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)        # channel-last flow
    assert (stn3d(fcl, frame.to(DEV)).cpu() - y.detach()).abs().max().item() < 2e-5


def test_ofeloss3d_value_and_gradients():
    from mireg.volume import OFEloss3d, smoothness_loss_3d, stn3d
    B, full = 2, (8, 12, 8)
    sizes = [(8, 12, 8), (4, 6, 4), (2, 3, 2)]
    x = nets.analytic_input((B, 2, *full), seed=6)
    fixed, moving = x[:, 0:1], x[:, 1:2]
    flows = [(0.8 * nets.analytic_input((B, 3, *s), seed=7 + i, lo=-1.0, hi=1.0)).requires_grad_(True) for i, s in enumerate(sizes)]
    warped = [oops.stn3d(f, moving) for f in flows]
    p, c, s, t = oops.ofe_loss_3d(flows, warped, fixed)
    (t + 0.25 * s).backward()
    fd = [f.detach().to(DEV).requires_grad_(True) for f in flows]
    wd = [stn3d(f, moving.to(DEV)) for f in fd]
    p2, c2, s2, t2 = OFEloss3d(fd, wd, fixed.to(DEV))
    (t2 + 0.25 * s2).backward()
    for a, b in ((p2, p), (c2, c), (s2, s), (t2, t)):
        assert abs(a.item() - b.item()) < 1e-5 * max(1.0, abs(b.item()))
    for a, b in zip(fd, flows):
        assert _rel(a.grad.cpu(), b.grad) < 2e-4
    assert abs(smoothness_loss_3d(fd[1].detach()).item() - oops.smoothness_loss_3d(flows[1].detach()).item()) < 1e-4
