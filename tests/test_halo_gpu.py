"""GPU parity of the halo-staged implicit-GEMM kernel (csrc/conv_halo.hip) against torch's fp32 convolution on the same
(bf16-rounded) operands and against the ring kernel it replaces, for every tile shape, grid width and tap layout it accepts:
forward of stride-1 convolutions (FlowNetS/util.py:17-30), backward-data of stride-1 / stride-2 convolutions (per parity
class) and ConvTranspose2d forward (FlowNetS/util.py:49-55).

Tolerances: fp32 (exact-fp32 MFMA) 3e-5 of the output scale; bf16 operands 3e-2 vs the fp32 reference of the rounded
operands, and halo vs ring on identical inputs <= 1 bf16 ulp of the output scale (only the K order differs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _view_from(x, ws):
    B, C, H, W = x.shape
    v = ws.new(B, H, W, C)
    v.buf[..., :C] = x.permute(0, 2, 3, 1).to(v.buf.dtype)
    return v


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


FWD_CASES = [  # cin, cout, k, H, W, B, bias
    (256, 256, 3, 32, 32, 2, False),     # conv3_1 shape
    (72, 136, 3, 16, 16, 3, True),       # partial channel chunk (72 = 2 x 32 + 8), ragged N tile
    (64, 128, 3, 64, 64, 1, True),       # 64-wide grid
    (72, 64, 3, 32, 16, 2, False),       # 64-column tiles, partial channel chunk
    (512, 512, 3, 16, 16, 2, False),     # conv4_1 shape
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", FWD_CASES)
def test_halo_forward_and_backward_data(case, prec):
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack
    cin, cout, k, H, W, B, has_bias = case
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 3e-5 if prec == "fp32" else 3e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(cin + 3 * cout + k)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if has_bias else None
    cot = torch.randn(B, cout, H, W, generator=g)
    if prec == "bf16":
        x, w, cot = x.bfloat16().float(), w.bfloat16().float(), cot.bfloat16().float()
    y_ref = F.leaky_relu(F.conv2d(x, w, b, 1, k // 2), 0.1)
    dx_ref = F.conv_transpose2d(cot, w, None, 1, k // 2)
    lay = ConvLayer("t", w.to(DEV), b.to(DEV) if has_bias else None, 1, k // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, DEV)
    xv, gv = _view_from(x.to(DEV), ws), _view_from(cot.to(DEV), ws)
    outs = {}
    ran = 0
    try:
        for tag, force in (("ring", (1, 0)), ("halo128", (2, 128)), ("halo256", (2, 256))):
            engine.FORCE_ALGO = force
            yv, dxv = ws.new(B, H, W, cout), ws.new(B, H, W, cin)
            try:
                lay.run_fwd_form(xv, yv, slope=0.1)
                lay.run_dgrad_form(gv, dxv)
                lay.run_dgrad_form(gv, dxv, accumulate=True)
            except RuntimeError as e:                        # this tile size does not apply to the grid (cap / divisibility)
                assert "unsupported" in str(e) and tag != "ring", (tag, e)
                continue
            ran += tag != "ring"
            torch.cuda.synchronize()
            outs[tag] = (yv.nchw().float().cpu(), dxv.nchw().float().cpu())
            assert _rel(outs[tag][0], y_ref) < tol, (tag, "fwd")
            assert _rel(outs[tag][1], 2 * dx_ref) < 2 * tol, (tag, "dgrad + accumulate")
    finally:
        engine.FORCE_ALGO = None
    assert ran >= 1, "no halo tile size applied to this case"
    ulp = 1e-5 if prec == "fp32" else 2 ** -7
    for tag in outs:
        if tag != "ring":
            assert _rel(outs[tag][0], outs["ring"][0]) < ulp and _rel(outs[tag][1], outs["ring"][1]) < 2 * ulp, tag


S2_CASES = [  # conv cin, cout, k, pad, H (input), B  -- backward-data of a stride-2 conv: 4 parity classes, unit-stride gathers
    (64, 128, 5, 2, 64, 2),      # conv2 -> classes 3x3, 3x2, 2x3, 2x2 on a 32x32 grid
    (128, 256, 5, 2, 32, 3),     # conv3, 16x16 grid
    (96, 72, 4, 1, 128, 1),      # 4x4 stride 2 (the adjoint of the deconvolutions), 64x64 grid, ragged channels
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", S2_CASES)
def test_halo_stride2_backward_data_and_deconv_forward(case, prec):
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack
    cin, cout, k, p, H, B = case
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    tol = 3e-5 if prec == "fp32" else 3e-2
    ws = Workspace(torch.device(DEV), dt)
    g = torch.Generator().manual_seed(cin + cout)
    Ho = (H + 2 * p - k) // 2 + 1
    w = torch.randn(cout, cin, k, k, generator=g) / (cout * k * k / 4) ** 0.5
    dy = torch.randn(B, cout, Ho, Ho, generator=g)
    bias = torch.randn(cin, generator=g)
    if prec == "bf16":
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    # ConvTranspose2d(cout -> cin) forward with bias + LeakyReLU == backward-data of the conv, activated
    ref = F.leaky_relu(F.conv_transpose2d(dy, w, bias, 2, p, output_padding=H - ((Ho - 1) * 2 - 2 * p + k)), 0.1)
    lay = ConvLayer("t", w.to(DEV), None, 2, p, 1, ws)
    lay.bias = bias.to(DEV)
    run_pack(lay.pack_jobs(), ws.code, DEV)
    gv = _view_from(dy.to(DEV), ws)
    outs = {}
    try:
        for tag, force in (("ring", (1, 0)), ("halo128", (2, 128)), ("halo256", (2, 256)), ("halo128n64", (2, 128, 64))):
            engine.FORCE_ALGO = force
            out = ws.new(B, H, H, cin)
            try:
                lay.run_dgrad_form(gv, out, slope=0.1, bias=True)
            except RuntimeError as e:
                assert "unsupported" in str(e) and tag != "ring", (tag, e)
                continue
            torch.cuda.synchronize()
            outs[tag] = out.nchw().float().cpu()
            assert _rel(outs[tag], ref) < tol, tag
    finally:
        engine.FORCE_ALGO = None
    # classes of different tap shapes (5x5 / stride 2: 3x3, 3x2, 2x3, 2x2) stay on the ring kernel: the halo path refuses them
    assert (len(outs) >= 2) if k == 4 else (set(outs) == {"ring"})
    ulp = 1e-5 if prec == "fp32" else 2 ** -7
    for tag in outs:
        assert _rel(outs[tag], outs["ring"]) < ulp, tag


def test_halo_is_the_default_for_eligible_sites_and_refuses_others():
    import ctypes
    from mireg import _lib
    from mireg.engine import ConvDesc
    d = ConvDesc()
    d.x, d.y = 16, 16                                  # non-null placeholders: the query never dereferences them
    d.mul_y = d.mul_x = d.step_y = d.step_x = 1
    d.taps_y = d.taps_x = 3
    d.g_H = d.g_W = 32
    d.n_img, d.N, d.dtype, d.x_C = 24, 256, 1, 256
    t = (ctypes.c_long * 2)()
    assert _lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), t) == 1 and (t[0], t[1]) == (24 * 8, 24 * 4)
    d.mul_x = 2                                        # strided gather: ring kernel
    assert _lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), t) == 0
    d.mul_x, d.g_W = 1, 8                              # 8-wide grid: ring kernel
    assert _lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), t) == 0
    d.g_W, d.taps_y, d.taps_x = 32, 1, 1               # 1x1: nothing to reuse
    assert _lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), t) == 0


WGRAD_CASES = [  # cin, cout, k, stride, pad, H (conv input), W, B, split
    (256, 256, 3, 1, 1, 32, 32, 2, 1),     # conv3_1: one 3x3 class, 32-wide grid
    (72, 136, 3, 1, 1, 16, 16, 3, 3),      # ragged channels both sides, 16-wide grid, split with an idle tail
    (64, 128, 5, 2, 2, 64, 64, 2, 4),      # conv2-like: classes 3x3 / 3x2 / 2x3 / 2x2 over the parity sub-images, dy grid 32
    (24, 40, 5, 2, 2, 128, 128, 1, 2),     # dy grid 64 wide (256-pixel chunks)
    (96, 72, 4, 2, 1, 64, 64, 2, 5),       # 4x4 / stride 2 (the adjoint conv of the deconvolutions): four 2x2 classes
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_halo_backward_weights(case):
    """conv_wgrad_halo.hip vs torch autograd on the bf16-rounded operands and vs the ring kernel it replaces."""
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack, run_unpack
    cin, cout, k, s, p, H, W, B, split = case
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    cot = torch.randn(y.shape, generator=g).bfloat16().float()
    (y * cot).sum().backward()
    lay = ConvLayer("t", w.detach().to(DEV), None, s, p, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, DEV)
    xv, gv = _view_from(x.to(DEV), ws), _view_from(cot.to(DEV), ws)
    got = {}
    try:
        for tag, algo in (("ring", 1), ("halo", 2)):
            engine.WGRAD_ALGO = algo
            lay.plan_wgrad(xv, gv)
            lay.wgrad_split = split
            lay.wgrad_slab = torch.full((split, lay.Co, lay.Kf), 7.0, device=DEV)      # stale contents must be overwritten
            lay.wgrad_slab[:, :, :] = 7.0
            for t in range(k * k):                                                        # pad channel slots are never written:
                lay.wgrad_slab[:, :, t * lay.Cip + cin:(t + 1) * lay.Cip] = 0.0           # they are zero in a fresh slab
            lay.grad_w = None
            lay.run_wgrad(xv, gv)
            run_unpack([lay.unpack_job()], DEV)
            torch.cuda.synchronize()
            got[tag] = lay.grad_w.detach().cpu().clone()
            assert _rel(got[tag], w.grad) < 3e-2, tag
    finally:
        engine.WGRAD_ALGO = 0
    assert _rel(got["halo"], got["ring"]) < 1e-3        # fp32 accumulation in a different order over <= 6k pixels
