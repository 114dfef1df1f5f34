"""GPU parity: warp / resize / losses / Dice HIP kernels vs the CPU oracle and the golden fixtures.

Tolerances (fp32 path): warped images 2e-5 abs (fp32 bilinear weights on coordinates up to 256),
loss scalars 1e-5 relative (device sums are accumulated in float64), gradients 1e-4 relative to the
tensor's max |g| away from the Charbonnier singularity.
"""
import numpy as np
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rand_flow(shape, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * sigma


def _close(a, b, tol, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), (what, err)


@pytest.mark.parametrize("align", [True, False])
def test_resize_fwd_bwd(align):
    import mireg
    x = nets.analytic_input((2, 3, 64, 48), seed=3)
    for size in ((256, 256), (16, 12), (64, 48), (1, 1), (5, 7)):
        xd = x.to(DEV).requires_grad_()
        y = mireg.resize_bilinear(xd, size, align)
        _close(y, oops.resize_bilinear(x, size, align), 2e-6, f"resize {size}")
        xr = x.clone().requires_grad_()
        yr = torch.nn.functional.interpolate(xr, size, mode="bilinear", align_corners=align)
        cot = nets.analytic_input(tuple(yr.shape), seed=9) - 0.5
        (yr * cot).sum().backward()
        (y * cot.to(DEV)).sum().backward()
        _close(xd.grad, xr.grad, 1e-5, f"resize bwd {size}")


def test_stn_golden_all_scales(golden):
    import mireg
    g = golden("g2_stn")
    frame = nets.analytic_input((3, 1, 256, 256), seed=5).to(DEV)
    for h in (256, 64, 32, 16, 8, 4, 1):
        for sigma in (0.5, 3.0, 20.0):
            flow = rand_flow((3, 2, h, h), sigma, seed=h * 7 + int(sigma))
            got = mireg.stn(flow.to(DEV), frame).cpu()
            if h > 64:
                got = got[:, :, ::4, ::4]
            _close(got, g[f"warp_h{h}_s{sigma}"], 2e-5, f"stn h={h} sigma={sigma}")
    frame64 = nets.analytic_input((2, 1, 64, 64), seed=9).to(DEV)
    flow = rand_flow((2, 2, 256, 256), 2.0, seed=77)
    _close(mireg.stn(flow.to(DEV), frame64).cpu()[:, :, ::4, ::4], g["warp_up64_s4"], 2e-5, "stn 64->256")


def test_stn_layouts_and_channels():
    """NHWC-interleaved flows (what the conv engine emits), multi-channel frames, odd sizes."""
    import mireg
    for (B, C, h, w) in ((2, 1, 32, 32), (1, 3, 7, 9), (2, 2, 16, 20)):
        flow = rand_flow((B, 2, h, w), 2.0, seed=h)
        frame = nets.analytic_input((B, C, h, w), seed=w)
        want = oops.stn(flow, frame)
        got = mireg.stn(flow.to(DEV), frame.to(DEV))
        _close(got, want, 2e-5, "planar")
        nhwc = flow.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)  # logical NCHW, NHWC storage
        assert not nhwc.is_contiguous() or h * w == 1
        _close(mireg.stn(nhwc, frame.to(DEV)), want, 2e-5, "interleaved")


def test_stn_backward_vs_torch():
    import mireg
    import torch.nn.functional as F
    B, h, w = 2, 32, 40
    frame = nets.analytic_input((B, 2, h, w), seed=1)
    flow = rand_flow((B, 2, h, w), 3.0, seed=2)
    cot = nets.analytic_input((B, 2, h, w), seed=3) - 0.5
    fr = flow.clone().requires_grad_()
    px, py = oops.stn_coords(fr)
    grid = torch.stack((px / (w - 1) * 2 - 1, py / (h - 1) * 2 - 1), -1)
    (F.grid_sample(frame, grid, mode="bilinear", padding_mode="zeros", align_corners=True) * cot).sum().backward()
    fd = flow.to(DEV).requires_grad_()
    (mireg.stn(fd, frame.to(DEV)) * cot.to(DEV)).sum().backward()
    _close(fd.grad, fr.grad, 1e-4, "stn bwd")


def test_ofeloss_golden_and_grads(golden):
    import mireg
    g = golden("g3_losses")
    fixed = nets.analytic_input((3, 1, 256, 256), seed=11)
    moving = nets.analytic_input((3, 1, 256, 256), seed=12)
    for n, sizes in ((2, (256, 64)), (6, (256, 64, 32, 16, 8, 4)), (7, (256, 128, 64, 32, 16, 8, 4))):
        flows = [rand_flow((3, 2, s, s), 1.5, seed=100 + s) for s in sizes]
        warped = [oops.stn(f, moving) for f in flows]
        fd = [f.to(DEV).requires_grad_() for f in flows]
        wd = [w.to(DEV).requires_grad_() for w in warped]
        vals = mireg.OFEloss(fd, wd, fixed.to(DEV))
        assert all(v.dtype == torch.float64 and v.dim() == 0 for v in vals)
        _close(torch.stack(vals), g[f"n{n}_values"], 1e-5, f"OFEloss n={n}")
        vals[3].backward()
        for i, sz in enumerate(sizes):
            st = 4 if sz > 64 else 1
            gw, want = wd[i].grad.cpu()[:, :, ::st, ::st], torch.from_numpy(g[f"n{n}_gwarp{i}"])
            # Charbonnier' ~ |d|^-0.5 blows up at d -> 0: compare where the reference gradient is tame
            tame = want.abs() < 50 * want.abs().median()
            assert tame.float().mean() > 0.9
            assert ((gw - want).abs()[tame] <= 2e-4 * want.abs()[tame].max()).all(), f"gwarp n={n} i={i}"
            gf, wantf = fd[i].grad.cpu()[:, :, ::st, ::st], torch.from_numpy(g[f"n{n}_gflow{i}"])
            assert ((gf - wantf).abs() <= 2e-4 * wantf.abs().max()).all(), f"gflow n={n} i={i}"


def test_ofeloss_through_stn_end_to_end():
    """model.stn -> criterion -> backward exactly as train.py:50-56 chains them."""
    import mireg
    B = 2
    fixed = nets.analytic_input((B, 1, 256, 256), seed=1)
    moving = nets.analytic_input((B, 1, 256, 256), seed=2)
    flows = [rand_flow((B, 2, s, s), 1.0, seed=s) for s in (256, 64, 16)]
    fr = [f.clone().requires_grad_() for f in flows]
    tot_r = oops.ofe_loss(fr, [oops.stn(f, moving) for f in fr], fixed)[3]
    tot_r.backward()
    fd = [f.to(DEV).requires_grad_() for f in flows]
    tot = mireg.OFEloss(fd, [mireg.stn(f, moving.to(DEV)) for f in fd], fixed.to(DEV))[3]
    tot.backward()
    _close(tot, tot_r, 1e-5, "total")
    for a, b in zip(fd, fr):
        want = b.grad
        tame = want.abs() < 50 * want.abs().median()
        assert ((a.grad.cpu() - want).abs()[tame] <= 5e-4 * want.abs()[tame].max()).all()


def test_ncc_degenerate_guard(golden):
    import mireg
    g = golden("g3_losses")
    fixed = nets.analytic_input((3, 1, 256, 256), seed=11).to(DEV)
    const = torch.full((3, 1, 64, 64), 0.25, device=DEV)
    flow = torch.zeros(3, 2, 64, 64, device=DEV)
    p, c, s, t = mireg.OFEloss([flow], [const], fixed, gamma=0.0, lamb_da=0.0, zeta=1.0 / 0.05)
    assert abs(c.item() - float(g["ncc_const"])) < 1e-12   # corr := 1 -> loss 0
    oob = torch.full((3, 2, 64, 64), 1e4, device=DEV)       # everything sampled out of bounds -> all-zero warp
    w = mireg.stn(oob, fixed)
    assert float(w.abs().max()) == 0.0
    assert mireg.OFEloss([oob], [w], fixed, gamma=0.0, lamb_da=0.0)[1].item() == 0.0


def test_dice_and_seg_round(golden):
    import mireg
    from oracle.gen_golden import make_labels
    g = golden("g4_dice")
    seg_f, seg_m = make_labels(2, 21), make_labels(2, 22)
    flow = rand_flow((2, 2, 256, 256), 2.0, seed=4)
    ws = mireg.seg_round(mireg.stn(flow.to(DEV), seg_m.to(DEV)))
    assert set(np.unique(ws.cpu().numpy())) <= {0.0, 1.0, 2.0, 3.0}
    hist = np.array([(ws == k).sum().item() for k in range(4)])
    assert np.abs(hist - g["ws_int_hist"]).max() <= 4
    d = mireg.dice_batch(seg_f.to(DEV), ws)
    _close(d, g["dice"], 1e-3, "dice batch")
    assert abs(mireg.dice_average(seg_f[0, 0].to(DEV), ws[0, 0]) - float(g["dice"][0])) < 1e-3
    half = torch.tensor([0.5, 1.5, 2.5, 3.5, -0.5, 4.2], device=DEV)   # round-half-even + clip
    assert mireg.seg_round(half).tolist() == [0.0, 2.0, 2.0, 3.0, 0.0, 3.0]
    empty = torch.zeros(1, 64, device=DEV)
    assert torch.isnan(mireg.dice_batch(empty, empty)).all()            # 0/0 like the reference


def test_full_size_properties():
    """BASELINE size (B=24, 256^2): size-independent properties instead of an oracle run."""
    import mireg
    B = 24
    ys, xs_ = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing="ij")
    smooth = 0.5 + 0.25 * torch.sin(xs_ / 17.0) * torch.cos(ys / 23.0)        # band-limited: |d/dx| < 0.02 / px
    fixed = smooth.view(1, 1, 256, 256).repeat(B, 1, 1, 1).contiguous().to(DEV)
    ident = torch.zeros(B, 2, 256, 256, device=DEV)
    w = mireg.stn(ident, fixed)
    # zero flow is NOT the identity: coordinate x*(w-1)/w -> shrinks towards 0 by < 1 px
    assert 1e-4 < (w - fixed).abs().max().item() < 0.05
    exact = torch.zeros(B, 2, 256, 256, device=DEV)
    xs = torch.arange(256, device=DEV, dtype=torch.float32)
    exact[:, 0] = (xs * 256 / 255 - xs).view(1, 1, 256)
    exact[:, 1] = (xs * 256 / 255 - xs).view(1, 256, 1)
    assert (mireg.stn(exact, fixed) - fixed).abs().max().item() < 2e-4  # flow that undoes the shrink
    p, c, s, t = mireg.OFEloss([exact], [fixed.clone()], fixed)
    assert c.item() < 1e-6 + (1 - 1 / B) * 100 * 0.05 + 1e-3 and c.item() > (1 - 1 / B) * 100 * 0.05 - 1e-3  # Q6
    assert abs(t.item() - (p + c + s).item()) < 1e-9


def test_dice_rejects_mismatched_shapes():
    import mireg
    a = torch.zeros(2, 1, 256, 256, device=DEV)
    b = torch.zeros(2, 1, 64, 64, device=DEV)
    with pytest.raises(RuntimeError, match="differ in size"):
        mireg.dice_batch(a, b)


def test_individual_loss_terms_match_oracle(golden):
    """loss.py's photometric_loss / correlation_loss / smoothness_loss as stand-alone drop-ins (values and gradients)."""
    import mireg
    g = torch.Generator().manual_seed(5)
    fixed = torch.rand(3, 1, 32, 40, generator=g)
    warped = torch.rand(3, 1, 16, 20, generator=g)
    flow = torch.randn(3, 2, 16, 20, generator=g)
    wd, wo = warped.clone().to(DEV).requires_grad_(), warped.clone().requires_grad_()
    fd, fo = flow.clone().to(DEV).requires_grad_(), flow.clone().requires_grad_()
    pairs = [(mireg.photometric_loss(fixed.to(DEV), wd), oops.photometric_loss(fixed, wo)),
             (mireg.correlation_loss(fixed.to(DEV), wd), oops.correlation_loss(fixed, wo)),
             (mireg.smoothness_loss(fd), oops.smoothness_loss(fo))]
    for a, b in pairs:
        assert abs(a.item() - b.item()) <= 1e-5 * max(1.0, abs(b.item())), (a.item(), b.item())
    (pairs[0][0] + pairs[1][0] + pairs[2][0]).backward()
    (pairs[0][1] + pairs[1][1] + pairs[2][1]).backward()
    assert (wd.grad.cpu() - wo.grad).abs().max().item() <= 2e-5 * max(1.0, wo.grad.abs().max().item())
    assert (fd.grad.cpu() - fo.grad).abs().max().item() <= 2e-5 * max(1.0, fo.grad.abs().max().item())
