"""GPU parity of the 256-pixel 8-wave implicit-GEMM kernel (csrc/conv_wide.hip, mireg_conv_desc.algo = 3) against torch's fp32
convolution on the same bf16-rounded operands and against the ring kernel it competes with in the tuner: forward of strided /
unit-stride convolutions (FlowNetS/util.py:17-30), backward-data of stride-2 convolutions per parity class and ConvTranspose2d
forward (FlowNetS/util.py:49-55), both tile widths (256 / 128 columns), split-K, ragged rows / channels, K-steps that straddle
taps (channel counts that are not multiples of 64), fused bias + LeakyReLU, accumulate and the fp32 side output.

Tolerances: 3e-2 of the output scale against the fp32 reference of the rounded operands (bf16 output rounding), and wide vs ring
on identical inputs <= 1 bf16 ulp of the output scale (only the K order differs)."""
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


CASES = [  # cin, cout, k, stride, pad, H, W, B, bias
    (64, 128, 5, 2, 2, 64, 64, 2, False),      # conv2 shape: one tap per K-step
    (128, 256, 5, 2, 2, 32, 32, 3, True),      # conv3 shape, rows 768 = 3 tiles
    (256, 256, 3, 1, 1, 32, 32, 1, False),     # conv3_1 shape
    (72, 136, 3, 1, 1, 24, 20, 3, True),       # K-steps straddle taps (72 channels), ragged rows (1440) and channels (136)
    (200, 386, 3, 1, 1, 16, 16, 2, True),      # ragged N over two 256-column tiles / four 128-column tiles
    (512, 64, 1, 1, 0, 16, 16, 2, True),       # 1x1, 64 output channels
    (64, 48, 3, 2, 1, 30, 26, 2, True),        # odd sizes, stride 2, fewer columns than a tile
]


@pytest.mark.parametrize("case", CASES)
def test_wide_forward_and_backward_data(case):
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack
    cin, cout, k, s, p, H, W, B, has_bias = case
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    g = torch.Generator().manual_seed(cin + 3 * cout + k)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float()
    b = torch.randn(cout, generator=g) if has_bias else None
    cot = torch.randn(B, cout, Ho, Wo, generator=g).bfloat16().float()
    y_ref = F.leaky_relu(F.conv2d(x, w, b, s, p), 0.1)
    dx_ref = F.conv_transpose2d(cot, w, None, s, p, output_padding=(H - ((Ho - 1) * s - 2 * p + k), W - ((Wo - 1) * s - 2 * p + k)))
    lay = ConvLayer("t", w.to(DEV), b.to(DEV) if has_bias else None, s, p, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, DEV)
    xv, gv = _view_from(x.to(DEV), ws), _view_from(cot.to(DEV), ws)
    outs = {}
    try:
        for tag, force in (("ring", (1, 0)), ("wide256", (3, 256, 256)), ("wide128", (3, 256, 128)), ("wide128s2", (3, 256, 128, 2)),
                           ("wide256s3", (3, 256, 256, 3))):
            engine.FORCE_ALGO = force
            yv, dxv = ws.new(B, Ho, Wo, cout), ws.new(B, H, W, cin)
            y32 = ws.new(B, Ho, Wo, cout, dtype=torch.float32)
            lay.run_fwd_form(xv, yv, y32=y32, slope=0.1)
            if cout >= 64:                                   # backward-data walks the output channels: the wide kernel needs >= 64 of them
                lay.run_dgrad_form(gv, dxv)
                lay.run_dgrad_form(gv, dxv, accumulate=True)
            torch.cuda.synchronize()
            outs[tag] = (yv.nchw().float().cpu(), dxv.nchw().float().cpu(), y32.nchw().float().cpu())
            assert _rel(outs[tag][0], y_ref) < 3e-2, (tag, "fwd")
            assert _rel(outs[tag][2], y_ref) < (3e-2 if "s" in tag[4:] or tag == "ring" else 1e-2), (tag, "fwd fp32 side output")
            if cout >= 64:
                assert _rel(outs[tag][1], 2 * dx_ref) < 6e-2, (tag, "dgrad + accumulate")
    finally:
        engine.FORCE_ALGO = None
    for tag in outs:
        if tag != "ring":
            assert _rel(outs[tag][0], outs["ring"][0]) < 2 ** -7 and _rel(outs[tag][1], outs["ring"][1]) < 2 ** -6, tag


def test_wide_deconv_forward_with_bias_and_activation():
    """ConvTranspose2d(4, 2, 1) forward with bias + LeakyReLU (FlowNetS/util.py:49-55) = the backward-data form with four parity
    classes of 2x2 taps in one launch, 392 walked channels (386 real: deconv2 of FlowNetS)."""
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack
    cin, cout, H, B = 64, 386, 32, 2                         # conv view of the deconvolution weight: Co = 386 (its input), Ci = 64
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(cout, cin, 4, 4, generator=g) / (cout * 4) ** 0.5).bfloat16().float()
    dy = torch.randn(B, cout, H // 2, H // 2, generator=g).bfloat16().float()
    bias = torch.randn(cin, generator=g)
    ref = F.leaky_relu(F.conv_transpose2d(dy, w, bias, 2, 1), 0.1)
    lay = ConvLayer("t", w.to(DEV), None, 2, 1, 1, ws)
    lay.bias = bias.to(DEV)
    run_pack(lay.pack_jobs(), ws.code, DEV)
    gv = _view_from(dy.to(DEV), ws)
    outs = {}
    try:
        for tag, force in (("ring", (1, 0)), ("wide128", (3, 256, 128)), ("wide128s2", (3, 256, 128, 2))):
            engine.FORCE_ALGO = force
            out = ws.new(B, H, H, cin)
            lay.run_dgrad_form(gv, out, slope=0.1, bias=True)
            torch.cuda.synchronize()
            outs[tag] = out.nchw().float().cpu()
            assert _rel(outs[tag], ref) < 3e-2, tag
    finally:
        engine.FORCE_ALGO = None
    assert _rel(outs["wide128"], outs["ring"]) < 2 ** -7


def test_wide_kernel_is_refused_where_it_does_not_apply():
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack
    for dt, cin in ((torch.float32, 64), (torch.bfloat16, 32)):      # fp32 parity mode / fewer than 64 channels per tap
        ws = Workspace(torch.device(DEV), dt)
        w = torch.randn(64, cin, 3, 3)
        lay = ConvLayer("t", w.to(DEV), None, 1, 1, 1, ws)
        run_pack(lay.pack_jobs(), ws.code, DEV)
        xv, yv = ws.new(1, 16, 16, cin), ws.new(1, 16, 16, 64)
        engine.FORCE_ALGO = (3, 256, 128)
        try:
            with pytest.raises(RuntimeError, match="unsupported"):
                lay.run_fwd_form(xv, yv)
        finally:
            engine.FORCE_ALGO = None


WGRAD_CASES = [  # cin, cout, k, stride, pad, H, W, B
    (256, 256, 3, 1, 1, 32, 32, 2),       # conv3_1 shape: one 256-row tile, 9 column tiles, split over pixels
    (128, 256, 5, 2, 2, 32, 32, 3),       # conv3 shape (5x5 stride 2)
    (512, 1024, 3, 1, 1, 4, 4, 6),        # conv6_1-like: 96 pixels = 3 K-steps, no split (the slab is the gradient)
    (200, 386, 3, 1, 1, 12, 10, 2),       # ragged rows (386 = 256 + 130), ragged columns, ragged pixel count (240 = 7.5 K-steps)
    (72, 136, 4, 2, 1, 16, 16, 2),        # 4x4 stride 2 (the deconvolutions' adjoint), 136 output channels: one half-empty tile
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wide_backward_weights(case):
    """conv_wgrad_wide.hip (mireg_conv_wgrad, algo 3) against torch's weight gradient on the same bf16-rounded operands and against the
    128 x 128 ring kernel (algo 1): same slab layout, same pixel splits."""
    from mireg import engine
    from mireg.engine import ConvLayer, Workspace, run_pack, run_unpack
    cin, cout, k, s, p, H, W, B = case
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    g = torch.Generator().manual_seed(cin + cout + k)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    dy = torch.randn(B, cout, Ho, Wo, generator=g).bfloat16().float()
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x, w, None, s, p).backward(dy)
    ref = w.grad.clone()
    outs = {}
    for tag, algo, split in (("ring", 1, 1), ("wide", 3, 1), ("wide-split3", 3, 3), ("ring-split3", 1, 3)):
        lay = ConvLayer("t", torch.zeros(cout, cin, k, k, device=DEV), None, s, p, 1, ws)
        lay.wgrad_split, lay.wgrad_algo = split, algo
        lay.wgrad_slab = torch.zeros(split, lay.Co, lay.Kf, device=DEV, dtype=torch.float32)
        lay.grad_w = torch.zeros_like(lay.weight)
        xv, gv = _view_from(x.to(DEV), ws), _view_from(dy.to(DEV), ws)
        lay.run_wgrad(xv, gv)
        run_unpack([lay.unpack_job()], DEV)
        torch.cuda.synchronize()
        outs[tag] = lay.grad_w.float().cpu()
        assert _rel(outs[tag], ref) < 2e-3, (tag, _rel(outs[tag], ref))        # fp32 accumulation of exact bf16 products: order only
    assert _rel(outs["wide"], outs["ring"]) < 1e-4 and _rel(outs["wide-split3"], outs["ring-split3"]) < 1e-4


@pytest.mark.parametrize("case", [(64, 128, 3, 1, 12, 2), (128, 64, 5, 2, 16, 1), (72, 200, 3, 2, 10, 2)])   # cin, cout, k, stride, size, B
def test_wide_conv3d_forward_and_backward_data(case):
    """The depth axis of conv_wide.hip (Conv3d forward and backward-data per parity class, reference models.py:39-43 generalised by
    BASELINE configs[4]) against torch's conv3d on bf16-rounded operands and against the 128-row ring kernel."""
    from mireg import affine3d
    from mireg.affine3d import Conv3dLayer, Vol
    from mireg.engine import Workspace, assign_tiles, upload_table, _stream
    from mireg import _lib
    cin, cout, k, s, n, B = case
    ws = Workspace(torch.device(DEV), torch.bfloat16)
    g = torch.Generator().manual_seed(cin + cout)
    pad = (k - 1) // 2
    x = torch.randn(B, cin, n, n, n, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).bfloat16().float()
    y_ref = F.leaky_relu(F.conv3d(x, w, None, s, pad), 0.1)
    no = y_ref.shape[-1]
    cot = torch.randn(B, cout, no, no, no, generator=g).bfloat16().float()
    dx_ref = F.conv_transpose3d(cot, w, None, s, pad, output_padding=n - ((no - 1) * s - 2 * pad + k))
    outs = {}
    try:
        for tag, force in (("ring", ()), ("wide128", (128,)), ("wide256", (256,))):
            affine3d.FORCE_WIDE = force
            lay = Conv3dLayer(w.to(DEV), None, (s, s, s), (pad, pad, pad), ws)
            jobs = [lay.pack_job()]
            units, dunits = assign_tiles(jobs, False)
            tab = upload_table(jobs, DEV)
            _lib.call("mireg_pack_weights", tab.data_ptr(), 1, units, dunits, ws.code, _stream())
            mk = lambda t, c, m: Vol(torch.zeros(B, m, m, m, (c + 7) // 8 * 8, device=DEV, dtype=torch.bfloat16), (m, m, m), c)
            xv, yv, gv, dxv = mk(x, cin, n), mk(None, cout, no), mk(cot, cout, no), mk(None, cin, n)
            xv.buf[..., :cin] = x.permute(0, 2, 3, 4, 1).to(DEV)
            gv.buf[..., :cout] = cot.permute(0, 2, 3, 4, 1).to(DEV)
            lay.run(xv, (n, n, n), yv, 0.1)
            lay.dgrad(gv, (no, no, no), dxv, (n, n, n))
            torch.cuda.synchronize()
            outs[tag] = (yv.buf[..., :cout].permute(0, 4, 1, 2, 3).float().cpu(), dxv.buf[..., :cin].permute(0, 4, 1, 2, 3).float().cpu())
            assert _rel(outs[tag][0], y_ref) < 3e-2 and _rel(outs[tag][1], dx_ref) < 3e-2, tag
    finally:
        affine3d.FORCE_WIDE = None
    for tag in ("wide128", "wide256"):
        assert _rel(outs[tag][0], outs["ring"][0]) < 2 ** -7 and _rel(outs[tag][1], outs["ring"][1]) < 2 ** -6, tag
