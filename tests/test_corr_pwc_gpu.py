"""GPU parity: cost volume (K7/K8), PWC warp (K10), FlowNetC and PWC-DC-Net forward vs oracle / golden fixtures."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("cfg", [(20, 2, 256, 32, 32, 2), (4, 1, 196, 4, 4, 3), (4, 1, 32, 64, 64, 2), (4, 1, 96, 16, 24, 1),
                                 (20, 2, 8, 40, 72, 1)])
def test_correlation_module_fp32(cfg):
    """Drop-in Correlation module (NCHW fp32) vs the published-definition oracle (parity unpinned upstream)."""
    import mireg
    md, s2, C, H, W, B = cfg
    f1 = nets.analytic_input((B, C, H, W), seed=1) - 0.5
    f2 = nets.analytic_input((B, C, H, W), seed=2) - 0.5
    want = oops.correlation(f1, f2, md, 1, md, 1, s2, 1)
    got = mireg.Correlation(md, 1, md, 1, s2, 1)(f1.to(DEV), f2.to(DEV))
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-5


def test_correlation_bf16_views_with_lrelu():
    from mireg.correlation import correlation_views
    from mireg.engine import Workspace
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    B, C, H, W, md, s2 = 2, 256, 32, 32, 20, 2
    f1 = (nets.analytic_input((B, C, H, W), seed=1) - 0.5).bfloat16().float()
    f2 = (nets.analytic_input((B, C, H, W), seed=2) - 0.5).bfloat16().float()
    want = torch.nn.functional.leaky_relu(oops.correlation(f1, f2, md, 1, md, 1, s2, 1), 0.1)
    v1, v2 = ws.new(B, H, W, C), ws.new(B, H, W, C)
    v1.buf[..., :C] = f1.permute(0, 2, 3, 1).to(DEV)
    v2.buf[..., :C] = f2.permute(0, 2, 3, 1).to(DEV)
    out = ws.new(B, H, W, 473)
    correlation_views(v1, v2, out.slice(32, 441), C, md, s2, 0.1, ws.code)
    assert _rel(out.slice(32, 441).nchw().float(), want) < 1e-2
    assert float(out.slice(0, 32).nchw().abs().max()) == 0.0      # neighbours of the slice untouched


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg", [(4, 1, 32, 64, 64, 2), (4, 1, 32, 20, 72, 1), (2, 1, 16, 9, 33, 3), (4, 1, 8, 7, 130, 1)])
def test_correlation_few_channel_kernel(cfg, prec):
    """The vector-ALU forward (<= 32 channels, <= 81 displacements: PWC level 2) against the published-definition oracle, written
    into a channel slice of a wider buffer (ragged widths, more than one 64-pixel block per row)."""
    from mireg.correlation import correlation_views
    from mireg.engine import Workspace
    md, s2, C, H, W, B = cfg
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    ws = Workspace(torch.device(DEV), dt)
    f1 = nets.analytic_input((B, C, H, W), seed=1) - 0.5
    f2 = nets.analytic_input((B, C, H, W), seed=2) - 0.5
    if prec == "bf16":
        f1, f2 = f1.bfloat16().float(), f2.bfloat16().float()
    D = 2 * (md // s2) + 1
    want = torch.nn.functional.leaky_relu(oops.correlation(f1, f2, md, 1, md, 1, s2, 1), 0.1)
    v1, v2 = ws.new(B, H, W, C), ws.new(B, H, W, C)
    v1.buf[..., :C] = f1.permute(0, 2, 3, 1).to(DEV)
    v2.buf[..., :C] = f2.permute(0, 2, 3, 1).to(DEV)
    out = ws.new(B, H, W, 16 + D * D + 5)
    out.buf.fill_(7.0)
    correlation_views(v1, v2, out.slice(16, D * D), C, md, s2, 0.1, ws.code)
    assert _rel(out.slice(16, D * D).nchw().float(), want) < (2e-5 if prec == "fp32" else 1e-2)
    assert float((out.slice(0, 16).nchw().float() - 7.0).abs().max()) == 0.0 and float((out.slice(16 + D * D, 5).nchw().float() - 7.0).abs().max()) == 0.0


def test_pwc_warp_golden(golden):
    from mireg.correlation import pwc_warp_views
    from mireg.engine import Workspace
    g = golden("g6_pwc_warp")
    ws = Workspace(torch.device(DEV), torch.float32)
    for (C, H) in ((128, 8), (64, 32), (32, 64)):
        x = nets.analytic_input((2, C, H, H), seed=C)
        flo = torch.from_numpy(g[f"flo_{C}_{H}"])
        xv, ov = ws.new(2, H, H, C), ws.new(2, H, H, C)
        xv.buf[..., :C] = x.permute(0, 2, 3, 1).to(DEV)
        fv = ws.new(2, H, H, 2, dtype=torch.float32, pad=2)
        fv.buf[...] = flo.permute(0, 2, 3, 1).to(DEV)
        pwc_warp_views(xv, fv, 1.0, ov, ws.code)
        got = ov.nchw().cpu()[:, ::8]
        assert (got - torch.from_numpy(g[f"warp_{C}_{H}"])).abs().max().item() < 2e-5, (C, H)
        import mireg
        got2 = mireg.PWCDCNet(md=4, precision="fp32").warp(x.to(DEV), flo.to(DEV)).cpu()[:, ::8]      # the module method
        assert (got2 - torch.from_numpy(g[f"warp_{C}_{H}"])).abs().max().item() < 2e-5, (C, H)


def test_pwcnet_fp32_golden(golden):
    import mireg
    g = golden("g5_skeletons")
    m = mireg.PWCDCNet(md=4, precision="fp32")
    nets.analytic_weights_(m)
    m = m.to(DEV).eval()
    x = nets.analytic_input((1, 2, 256, 256), seed=8).to(DEV)
    with torch.no_grad():
        flows = m(x)
    assert [tuple(f.shape[2:]) for f in flows] == [(256, 256), (128, 128), (64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    for i, f in enumerate(flows):
        want = torch.from_numpy(g[f"pwc_flow{i}"])
        got = f.cpu() if f.shape[-1] <= 64 else f.cpu()[:, :, ::4, ::4]
        err, scale = (got - want).abs().max().item(), want.abs().max().item()
        assert err <= 2e-4 * max(1.0, scale), (i, err, scale)
    assert list(m.state_dict().keys()) == list(nets.PWCDCNet().state_dict().keys())


def test_flownetc_fp32_golden(golden):
    import mireg
    g = golden("g5_skeletons")
    m = mireg.FlowNetC(None, batchNorm=True, precision="fp32")
    nets.analytic_weights_(m)
    m = m.to(DEV)
    x = nets.analytic_input((2, 2, 256, 256), seed=9).to(DEV)
    for mode in ("train", "eval"):
        m.train(mode == "train")
        with torch.no_grad():
            flows = m(x)
        assert len(flows) == (5 if mode == "train" else 1)
        for i, f in enumerate(flows):
            want = torch.from_numpy(g[f"flownetc_{mode}_flow{i}"])
            err, scale = (f.cpu() - want).abs().max().item(), want.abs().max().item()
            assert err <= 5e-4 * max(1.0, scale), (mode, i, err, scale)
    sd_keys = [k for k in nets.FlowNetC().state_dict().keys()]
    assert [k for k in m.state_dict().keys()] == sd_keys


def test_pwc_bf16_close_and_registration_wrapper():
    import mireg
    torch.manual_seed(0)
    reg32 = mireg.opticalFlowReg("pwc", precision="fp32").to(DEV).eval()
    reg16 = mireg.opticalFlowReg("pwc", precision="bf16").to(DEV).eval()
    reg16.load_state_dict(reg32.state_dict())
    x = nets.analytic_input((2, 2, 256, 256), seed=4).to(DEV)
    with torch.no_grad():
        f32, w32, _, _ = reg32(x)
        f16, w16, _, _ = reg16(x)
    assert len(f32) == 7 and len(w32) == 7
    for a, b in zip(f32[:3], f16[:3]):
        assert ((a - b).double().norm() / a.double().norm().clamp_min(1e-9)).item() < 8e-2


def test_correlation_backward_vs_autograd():
    """dF1 / dF2 of the cost volume against torch autograd through the oracle's definition (fp32 + bf16 storage)."""
    from mireg import _lib
    from mireg.engine import Workspace, _stream
    for prec, tol in (("fp32", 3e-5), ("bf16", 2e-2)):
        dt = torch.float32 if prec == "fp32" else torch.bfloat16
        ws = Workspace(torch.device(DEV), dt)
        # 81 displacements and <= 128 channels take the vector-ALU kernels (PWC), the others the matrix-core ones
        for (md, s2, C, H, W, B) in ((20, 2, 256, 32, 32, 1), (4, 1, 96, 16, 40, 2), (4, 1, 200, 4, 4, 2), (4, 1, 32, 64, 64, 2),
                                     (4, 1, 32, 20, 72, 1), (2, 1, 16, 9, 33, 2)):
            f1 = (nets.analytic_input((B, C, H, W), seed=1) - 0.5)
            f2 = (nets.analytic_input((B, C, H, W), seed=2) - 0.5)
            D = 2 * (md // s2) + 1
            g = nets.analytic_input((B, D * D, H, W), seed=3) - 0.5
            if prec == "bf16":
                f1, f2, g = f1.bfloat16().float(), f2.bfloat16().float(), g.bfloat16().float()
            a, b = f1.clone().requires_grad_(), f2.clone().requires_grad_()
            (oops.correlation(a, b, md, 1, md, 1, s2, 1) * g).sum().backward()
            v1, v2, vg = ws.new(B, H, W, C), ws.new(B, H, W, C), ws.new(B, H, W, D * D)
            v1.buf[..., :C] = f1.permute(0, 2, 3, 1).to(DEV); v2.buf[..., :C] = f2.permute(0, 2, 3, 1).to(DEV)
            vg.buf[..., :D * D] = g.permute(0, 2, 3, 1).to(DEV)
            d1, d2 = ws.new(B, H, W, C), ws.new(B, H, W, C)
            _lib.call("mireg_correlation_bwd", vg.ptr, vg.ld, v1.ptr, v1.ld, v2.ptr, v2.ld, d1.ptr, d1.ld, d2.ptr, d2.ld,
                      B, H, W, C, C, md, s2, 0, 0, ws.code, _stream())
            assert _rel(d1.nchw().float(), a.grad) < tol, (prec, md, "dF1")
            assert _rel(d2.nchw().float(), b.grad) < tol, (prec, md, "dF2")


def test_flownetc_training_step_matches_oracle():
    """FlowNetC fp32: gradients through decoder, cost volume and both siamese streams vs the CPU oracle autograd."""
    import mireg
    torch.manual_seed(0)
    m = mireg.FlowNetC(None, batchNorm=True, precision="fp32")
    nets.analytic_weights_(m)
    o = nets.FlowNetC(None, batchNorm=True)
    o.load_state_dict(m.state_dict(), strict=False)
    m = m.to(DEV).train(); o.train()
    x = nets.analytic_input((4, 2, 128, 128), seed=9)

    def objective(fl, dev):
        return sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(dev)).sum() for f in fl)
    objective(o(x), "cpu").backward()
    objective(m(x.to(DEV)), DEV).backward()
    P, Q = dict(m.named_parameters()), dict(o.named_parameters())
    worst = 0.0
    for k in ("conv1.0.weight", "conv1.1.weight", "conv2.0.weight", "conv3.0.weight", "conv3.1.bias", "conv_redir.0.weight",
              "conv3_1.0.weight", "conv4.0.weight", "conv6_1.0.weight", "deconv5.0.weight", "deconv5.0.bias",
              "predict_flow4.weight", "predict_flow4.bias", "upsampled_flow4_to_3.weight", "upsampled_flow4_to_3.bias"):
        a, b = P[k].grad.detach().cpu().double(), Q[k].grad.double()
        rel = ((a - b).norm() / b.norm()).item()
        worst = max(worst, rel)
        assert rel < 2e-2, (k, rel)
    print("worst relative L2 gradient error", worst)


def test_pwc_warp_backward_vs_autograd():
    from mireg import _lib
    from mireg.engine import Workspace, _stream
    ws = Workspace(torch.device(DEV), torch.float32)
    B, C, H = 2, 32, 16
    x = nets.analytic_input((B, C, H, H), seed=1) - 0.5
    flo = (nets.analytic_input((B, 2, H, H), seed=2) - 0.5) * 3
    g = nets.analytic_input((B, C, H, H), seed=3) - 0.5
    a, f = x.clone().requires_grad_(), flo.clone().requires_grad_()
    (oops.pwc_warp(a, f * 1.25) * g).sum().backward()
    xv, gv = ws.new(B, H, H, C), ws.new(B, H, H, C)
    xv.buf[...] = x.permute(0, 2, 3, 1).to(DEV); gv.buf[...] = g.permute(0, 2, 3, 1).to(DEV)
    fv = ws.new(B, H, H, 2, dtype=torch.float32, pad=2); fv.buf[...] = flo.permute(0, 2, 3, 1).to(DEV)
    dx = ws.new(B, H, H, C, dtype=torch.float32); df = ws.new(B, H, H, 2, dtype=torch.float32, pad=2)
    _lib.call("mireg_pwc_warp_bwd", xv.ptr, xv.ld, fv.ptr, fv.ld, 1.25, gv.ptr, gv.ld, dx.ptr, dx.ld, df.ptr, df.ld, B, H, H, C,
              ws.code, _stream())
    assert _rel(dx.nchw(), a.grad) < 1e-5
    assert _rel(df.nchw(), f.grad) < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 32, 16, 16, 3.0), (3, 96, 24, 40, 9.0), (1, 128, 64, 64, 0.3), (2, 64, 8, 8, 40.0)])
def test_pwc_warp_backward_deterministic_vs_autograd(shape, prec):
    """mireg_pwc_warp_bwd_det (buckets by source pixel + ordered gather, no fp32 scatter atomics): equals autograd through the
    oracle's warp, equals the atomic kernel up to fp32 summation order, is bit-identical between two runs, leaves its counters
    zero, and survives flows that throw most pixels out of the image (scale 40) or pile them up (scale 9 on a 24x40 grid)."""
    from mireg import _lib
    from mireg.correlation import WarpBwdWorkspace
    from mireg.engine import Workspace, _stream
    B, C, H, W, mag = shape
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    ws = Workspace(torch.device(DEV), dt)
    x = nets.analytic_input((B, C, H, W), seed=1) - 0.5
    flo = (nets.analytic_input((B, 2, H, W), seed=2) - 0.5) * mag
    g = nets.analytic_input((B, C, H, W), seed=3) - 0.5
    if prec == "bf16":
        x, g = x.bfloat16().float(), g.bfloat16().float()
    a, f = x.clone().requires_grad_(), flo.clone().requires_grad_()
    (oops.pwc_warp(a, f * 1.25) * g).sum().backward()
    xv, gv = ws.new(B, H, W, C), ws.new(B, H, W, C)
    xv.buf[...] = x.permute(0, 2, 3, 1).to(DEV); gv.buf[...] = g.permute(0, 2, 3, 1).to(DEV)
    fv = ws.new(B, H, W, 2, dtype=torch.float32, pad=2); fv.buf[...] = flo.permute(0, 2, 3, 1).to(DEV)
    wsb = WarpBwdWorkspace(B * H * W, torch.device(DEV))
    outs = []
    for _ in range(2):
        dx = ws.new(B, H, W, C, dtype=torch.float32); dx.buf.fill_(7.0)              # dx32 is overwritten, not accumulated
        df = ws.new(B, H, W, 2, dtype=torch.float32, pad=2)
        _lib.call("mireg_pwc_warp_bwd_det", xv.ptr, xv.ld, fv.ptr, fv.ld, 1.25, gv.ptr, gv.ld, dx.ptr, dx.ld, df.ptr, df.ld,
                  wsb.cnt.data_ptr(), wsb.off.data_ptr(), wsb.entries.data_ptr(), B, H, W, C, ws.code, _stream())
        torch.cuda.synchronize()
        outs.append((dx.nchw().clone(), df.nchw().clone()))
        assert int(wsb.cnt.abs().max()) == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    scale_x, scale_f = max(a.grad.abs().max().item(), 1e-6), max(f.grad.abs().max().item(), 1e-6)
    assert (outs[0][0].cpu() - a.grad).abs().max().item() <= 2e-5 * scale_x
    assert (outs[0][1].cpu() - f.grad).abs().max().item() <= 2e-4 * scale_f
    dx0 = ws.new(B, H, W, C, dtype=torch.float32); df0 = ws.new(B, H, W, 2, dtype=torch.float32, pad=2)
    _lib.call("mireg_pwc_warp_bwd", xv.ptr, xv.ld, fv.ptr, fv.ld, 1.25, gv.ptr, gv.ld, dx0.ptr, dx0.ld, df0.ptr, df0.ld, B, H, W, C,
              ws.code, _stream())
    assert (dx0.nchw() - outs[0][0]).abs().max().item() <= 2e-5 * scale_x


def test_pwcnet_training_gradients_match_oracle():
    """PWC-DC-Net fp32: gradients through dense estimators, cost volumes, warps, context net and the siamese pyramid."""
    import mireg
    torch.manual_seed(0)
    m = mireg.PWCDCNet(md=4, precision="fp32")
    nets.analytic_weights_(m)
    o = nets.PWCDCNet(md=4)
    o.load_state_dict(m.state_dict())
    m = m.to(DEV).train(); o.train()
    x = nets.analytic_input((2, 2, 128, 128), seed=9)

    def objective(fl, dev):
        return sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(dev)).sum() for f in fl)
    objective(o(x), "cpu").backward()
    objective(m(x.to(DEV)), DEV).backward()
    P, Q = dict(m.named_parameters()), dict(o.named_parameters())
    worst = ("", 0.0)
    for k in ("conv1a.0.weight", "conv1a.0.bias", "conv2b.0.weight", "conv4aa.0.weight", "conv6b.0.weight", "conv6_0.0.weight",
              "conv6_4.0.bias", "conv5_2.0.weight", "conv3_0.0.weight", "conv2_4.0.weight", "predict_flow6.weight",
              "predict_flow2.bias", "deconv6.weight", "deconv3.bias", "upfeat5.weight", "upfeat3.bias", "dc_conv1.0.weight",
              "dc_conv4.0.weight", "dc_conv7.weight", "deconv2.weight", "deconv1.weight", "deconv1.bias"):
        a, b = P[k].grad.detach().cpu().double(), Q[k].grad.double()
        rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        if rel > worst[1]:
            worst = (k, rel)
        assert rel < 2e-2, (k, rel)
    print("worst relative L2 gradient error", worst)


def test_correlation_module_backward_matches_oracle_autograd():
    """mireg.Correlation is differentiable like the upstream correlation_package op it replaces."""
    import mireg
    for (md, s2, C, H, W, B) in ((20, 2, 48, 12, 20, 2), (4, 1, 30, 9, 11, 1)):
        f1 = (nets.analytic_input((B, C, H, W), seed=1) - 0.5)
        f2 = (nets.analytic_input((B, C, H, W), seed=2) - 0.5)
        D = 2 * (md // s2) + 1
        g = nets.analytic_input((B, D * D, H, W), seed=3) - 0.5
        a, b = f1.clone().requires_grad_(), f2.clone().requires_grad_()
        (oops.correlation(a, b, md, 1, md, 1, s2, 1) * g).sum().backward()
        ad, bd = f1.clone().to(DEV).requires_grad_(), f2.clone().to(DEV).requires_grad_()
        out = mireg.Correlation(md, 1, md, 1, s2, 1)(ad, bd)
        (out * g.to(DEV)).sum().backward()
        assert _rel(ad.grad, a.grad) < 3e-5 and _rel(bd.grad, b.grad) < 3e-5, (md, s2)
