"""GPU parity of the MFMA convolution family (K1-K4) and of the FlowNetS predictor built on it.

fp32 mode uses v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain): tolerance 2e-5 relative to the
output scale per layer, flows of the whole net 1e-4 (north_star "flow L2 vs reference < 1e-4").
bf16 mode (bf16 operands, fp32 accumulate): 2e-2 relative per layer.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _view_from(x, ws, pad_to=8):
    from mireg.engine import View
    B, C, H, W = x.shape
    v = ws.new(B, H, W, C)
    v.buf[..., :C] = x.permute(0, 2, 3, 1).to(v.buf.dtype)
    return v


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


CASES = [  # cin, cout, k, stride, pad, dil, H, W, B, bias
    (8, 64, 7, 2, 3, 1, 64, 64, 2, False),
    (64, 128, 5, 2, 2, 1, 32, 32, 2, False),
    (194, 2, 3, 1, 1, 1, 16, 16, 3, True),
    (1026, 256, 3, 1, 1, 1, 8, 8, 2, False),
    (128, 96, 3, 1, 8, 8, 24, 20, 1, True),     # dilated (PWC dc_conv4)
    (256, 32, 1, 1, 0, 1, 16, 16, 2, True),     # conv_redir 1x1
    (512, 1024, 3, 2, 1, 1, 8, 8, 3, False),    # split-K path
    (16, 16, 3, 2, 1, 1, 30, 26, 2, True),      # odd sizes, stride 2
    (1026, 2, 3, 1, 1, 1, 6, 6, 2, True),       # predict_flow head, wave-per-pixel thin kernels
    (34, 2, 3, 1, 1, 1, 96, 100, 2, True),      # predict_flow head, 8-lanes-per-pixel thin kernels (>= 16k pixels)
    (2, 64, 7, 2, 3, 1, 64, 96, 3, True),       # FlowNetS stem (patch-staged kernels in bf16)
    (1, 64, 7, 2, 3, 1, 40, 36, 2, False),      # FlowNetC stem, ragged tiles
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_three_forms(case, prec):
    from mireg.engine import ConvLayer, Workspace
    cin, cout, k, s, p, d, H, W, B, has_bias = case
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 3e-5 if prec == "fp32" else 3e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if has_bias else None
    if prec == "bf16":  # compare against the same rounded operands
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y_ref = F.conv2d(xr, wr, b, s, p, d)
    cot = torch.randn(y_ref.shape, generator=g)
    if prec == "bf16":
        cot = cot.bfloat16().float()
    (y_ref * cot).sum().backward()
    lay = ConvLayer("t", w.to(DEV), b.to(DEV) if has_bias else None, s, p, d, ws)
    import ctypes
    from mireg import _lib
    from mireg.engine import run_pack, run_unpack
    run_pack(lay.pack_jobs(), ws.code, DEV)
    xv = _view_from(x.to(DEV), ws)
    Ho, Wo = y_ref.shape[2:]
    yv = ws.new(B, Ho, Wo, cout)
    lay.run_fwd_form(xv, yv)
    assert _rel(yv.nchw().float(), y_ref.detach()) < tol, "fwd"
    # backward-data
    gv = _view_from(cot.to(DEV), ws)
    dxv = ws.new(B, H, W, cin)
    lay.run_dgrad_form(gv, dxv)
    assert _rel(dxv.nchw().float(), xr.grad) < tol, "dgrad"
    lay.run_dgrad_form(gv, dxv, accumulate=True)
    assert _rel(dxv.nchw().float(), 2 * xr.grad) < tol * 2, "dgrad accumulate"
    # backward-weights
    lay.run_wgrad(xv, gv)
    run_unpack([lay.unpack_job()], DEV)
    assert _rel(lay.grad_w, wr.grad) < tol, "wgrad"
    if has_bias:
        lay.run_bias_grad(gv)
        assert _rel(lay.grad_b, cot.sum((0, 2, 3))) < tol, "bias grad"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_deconv_forms(prec):
    """ConvTranspose2d(k4,s2,p1) == adjoint conv used backwards (FlowNetS/util.py:49-55)."""
    from mireg.engine import ConvLayer, Workspace, run_pack, run_unpack
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 3e-5 if prec == "fp32" else 3e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(5)
    for cin, cout, H, B in ((386, 64, 8, 2), (2, 2, 4, 3), (1026, 256, 4, 2)):
        x = torch.randn(B, cin, H, H, generator=g)
        w = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
        bias = torch.randn(cout, generator=g)
        if prec == "bf16":
            x, w = x.bfloat16().float(), w.bfloat16().float()
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        y_ref = F.leaky_relu(F.conv_transpose2d(xr, wr, bias, 2, 1), 0.1)
        cot = torch.randn(y_ref.shape, generator=g)
        if prec == "bf16":
            cot = cot.bfloat16().float()
        (y_ref * cot).sum().backward()
        lay = ConvLayer("d", w.to(DEV), bias.to(DEV), 2, 1, 1, ws)
        run_pack(lay.pack_jobs(), ws.code, DEV)
        xv = _view_from(x.to(DEV), ws)
        yv = ws.new(B, 2 * H, 2 * H, cout)
        lay.run_dgrad_form(xv, yv, slope=0.1, bias=True)
        assert _rel(yv.nchw().float(), y_ref.detach()) < tol, "deconv fwd"
        dz = cot * torch.where(y_ref.detach() > 0, 1.0, 0.1)
        if prec == "bf16":
            dz = dz.bfloat16().float()
        gv = _view_from(dz.to(DEV), ws)
        dxv = ws.new(B, H, H, cin)
        lay.run_fwd_form(gv, dxv, bias=False)
        assert _rel(dxv.nchw().float(), xr.grad) < 2 * tol, "deconv bwd-data"
        lay.run_wgrad(gv, xv)
        run_unpack([lay.unpack_job()], DEV)
        assert _rel(lay.grad_w, wr.grad) < 2 * tol, "deconv wgrad"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_tiny_upsampler_kernels(prec):
    """The pixel-parallel kernels of the 2 -> 2 channel ConvTranspose2d(4, 2, 1) flow upsamplers (FlowNetS/FlowNetS.py:37-40):
    forward, backward-data (plain / accumulating / with the fused planar addend) and backward-weights vs torch autograd."""
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack, run_unpack
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 3e-5 if prec == "fp32" else 1.5e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(11)
    old = engine.TINY_MASK
    engine.TINY_MASK = 7
    try:
        for H, W, B in ((4, 4, 3), (16, 8, 24), (5, 7, 1)):
            x = torch.randn(B, 2, H, W, generator=g)
            w = torch.randn(2, 2, 4, 4, generator=g) * 0.3
            bias = torch.randn(2, generator=g)
            cot = torch.randn(B, 2, 2 * H, 2 * W, generator=g)
            add = torch.randn(B, 2, H, W, generator=g)
            if prec == "bf16":
                x, cot = x.bfloat16().float(), cot.bfloat16().float()
            xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
            y_ref = F.conv_transpose2d(xr, wr, bias, 2, 1)
            (y_ref * cot).sum().backward()
            lay = ConvLayer("u", w.to(DEV), bias.to(DEV), 2, 1, 1, ws)
            assert lay.tiny
            run_pack(lay.pack_jobs(), ws.code, DEV)
            xv, gv = _view_from(x.to(DEV), ws), _view_from(cot.to(DEV), ws)
            yv = ws.new(B, 2 * H, 2 * W, 2)
            y32v = ws.new(B, 2 * H, 2 * W, 2, dtype=torch.float32, pad=2)
            lay.run_dgrad_form(xv, yv, y32=y32v, bias=True)
            assert _rel(yv.nchw().float(), y_ref.detach()) < tol, "forward"
            assert _rel(y32v.nchw(), y_ref.detach()) < min(tol, 1e-4) or prec == "bf16", "forward, fp32 copy"
            assert _rel(y32v.nchw(), y_ref.detach()) < tol
            lay.run_dgrad_form(xv, None, y32=y32v, bias=True)                      # fp32 copy only (PWC's flow0)
            assert _rel(y32v.nchw(), y_ref.detach()) < tol
            dxv = ws.new(B, H, W, 2)
            assert lay.tiny_bwd_data_ok(gv, dxv)
            lay.run_fwd_form(gv, dxv, bias=False)
            assert _rel(dxv.nchw().float(), xr.grad) < tol, "backward-data"
            lay.run_fwd_form(gv, dxv, bias=False, accumulate=True)
            assert _rel(dxv.nchw().float(), 2 * xr.grad) < 2 * tol, "backward-data, accumulating"
            lay.run_fwd_form(gv, dxv, bias=False, add_nchw=add.to(DEV))
            assert _rel(dxv.nchw().float(), xr.grad + add) < tol, "backward-data + planar addend"
            lay.run_wgrad(gv, xv)
            run_unpack([lay.unpack_job()], DEV)
            assert _rel(lay.grad_w, wr.grad) < tol, "backward-weights"
    finally:
        engine.TINY_MASK = old


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_batchnorm_lrelu(prec):
    from mireg.engine import BatchNormAct, Workspace
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 2e-5 if prec == "fp32" else 2e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(3)
    for C, H, B in ((64, 16, 3), (1024, 2, 2), (36, 5, 2), (64, 48, 1), (520, 6, 3)):   # one-launch (<= 2048 rows) and three-stage paths
        y = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
        if prec == "bf16":
            y = y.bfloat16().float()
        bn = torch.nn.BatchNorm2d(C)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
            bn.bias.copy_(0.1 * torch.randn(C, generator=g))
        bn_d = torch.nn.BatchNorm2d(C).to(DEV)
        bn_d.load_state_dict(bn.state_dict())
        yr = y.clone().requires_grad_()
        out_ref = F.leaky_relu(bn(yr), 0.1)
        cot = torch.randn(out_ref.shape, generator=g)
        if prec == "bf16":
            cot = cot.bfloat16().float()
        (out_ref * cot).sum().backward()
        op = BatchNormAct(bn_d, ws)
        yv, ov = _view_from(y.to(DEV), ws), ws.new(B, H, H, C)
        op.forward(yv, ov, True)
        assert _rel(ov.nchw().float(), out_ref.detach()) < tol
        assert _rel(bn_d.running_mean, bn.running_mean) < 1e-5 and _rel(bn_d.running_var, bn.running_var) < 1e-5
        dav, dyv = _view_from(cot.to(DEV), ws), ws.new(B, H, H, C)
        op.backward(yv, dav, dyv)
        assert _rel(dyv.nchw().float(), yr.grad) < 5 * tol
        assert _rel(op.grad_g, bn.weight.grad) < 5 * tol and _rel(op.grad_b, bn.bias.grad) < 5 * tol
        bn.eval(); bn_d.eval()
        op.forward(yv, ov, False)
        assert _rel(ov.nchw().float(), F.leaky_relu(bn(y), 0.1).detach()) < tol


def _check_flows(out, g, mode):
    """|mireg - reference| <= 1e-4 * max(1, scale) + 4 * noise, where noise is the reference's OWN fp32 rounding
    distance from a float64 evaluation (stored in the fixture): BatchNorm over the 4..32 samples of the deepest
    levels amplifies rounding, and no fp32 implementation can match another more tightly than that."""
    for i, f in enumerate(out):
        f = f.detach().cpu()
        noise = float(g[f"noise_flow{i}"]) if f"noise_flow{i}" in g.files else 0.0
        if f"{mode}_flow{i}" in g.files:
            want = torch.from_numpy(g[f"{mode}_flow{i}"])
        else:
            want, f = torch.from_numpy(g[f"{mode}_flow{i}_s8"]), f[:, :, ::8, ::8]
        err, scale = (f - want).abs().max().item(), want.abs().max().item()
        assert err <= 1e-4 * max(1.0, scale) + 4 * noise, (i, err, scale, noise)


def _check_grads(m, g):
    """relative L2 of each pinned parameter gradient vs the reference: <= 5e-3 + 8 * (reference's own fp32 noise
    vs float64).  The 5e-3 floor covers LeakyReLU kink flips: one pre-activation within rounding of 0 flips a
    0.1/1.0 slope and moves the L2 norm of a 1e5-element gradient by ~3e-3 in ANY fp32 implementation."""
    P = dict(m.named_parameters())
    keys = [k[5:] for k in g.files if k.startswith("grad_") and k != "grad_x_s4"]
    assert len(keys) >= 10
    for k in keys:
        want = torch.from_numpy(g["grad_" + k]).double()
        got = P[k].grad.detach().cpu().flatten()[:want.numel()].double()
        rel = ((got - want).norm() / want.norm()).item()
        tol = 5e-3 + 8 * float(g["gradnoise_" + k])
        assert rel <= tol, (k, rel, tol)
        assert abs(P[k].grad.double().norm().item() / float(g["gradnorm_" + k]) - 1) <= tol, k


def _flownets_pair(prec, shape, seed=3):
    import mireg
    m = mireg.FlowNetS(batchNorm=True, precision=prec)
    nets.analytic_weights_(m)
    m = m.to(DEV)
    x = nets.analytic_input(shape, seed=seed)
    return m, x


def test_flownets_fp32_golden_config1(golden):
    g = golden("g1_flownets_c1_4x64")
    m, x = _flownets_pair("fp32", (4, 2, 64, 64))
    m.train()
    xd = x.to(DEV)
    out = m(xd)
    assert [tuple(o.shape) for o in out] == [(4, 2, 256, 256), (4, 2, 16, 16), (4, 2, 8, 8), (4, 2, 4, 4), (4, 2, 2, 2), (4, 2, 1, 1)]
    _check_flows(out, g, "train")
    obj = sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(DEV)).sum() for f in out)
    obj.backward()
    _check_grads(m, g)
    assert _rel(m.conv2[1].running_mean, torch.from_numpy(g["bn_running_mean_conv2"])) < 1e-4
    m.eval()
    out = m(xd)
    assert len(out) == 2
    want = torch.from_numpy(g["eval_flow1"])
    assert (out[1].cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


def test_flownets_fp32_golden_256(golden):
    g = golden("g1_flownets_c2_2x256")
    m, x = _flownets_pair("fp32", (2, 2, 256, 256))
    m.train()
    out = m(x.to(DEV))
    _check_flows(out, g, "train")
    obj = sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(DEV)).sum() for f in out)
    obj.backward()
    _check_grads(m, g)


def test_flownets_bf16_close_to_fp32():
    """bf16 operands / fp32 accumulate vs the exact-fp32 engine on a realistically conditioned case
    (reference init = kaiming, B=8 so the deepest BatchNorm sees 128 samples): relative L2 per flow < 5e-2."""
    import mireg
    torch.manual_seed(0)
    m32 = mireg.FlowNetS(True, precision="fp32").to(DEV)
    m16 = mireg.FlowNetS(True, precision="bf16").to(DEV)
    m16.load_state_dict(m32.state_dict())
    x = nets.analytic_input((8, 2, 256, 256), seed=4).to(DEV)
    m32.train(); m16.train()
    a, b = m32(x), m16(x)
    for i, (fa, fb) in enumerate(zip(a, b)):
        rel = ((fb - fa).double().norm() / fa.double().norm()).item()
        print("bf16 vs fp32 flow", i, "rel L2", rel)
        assert rel < 5e-2, (i, rel)
