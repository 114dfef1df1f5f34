"""GPU parity of the 3-D affine path (a14): Conv3d+ReLU chain, Linear, affine grid + trilinear sample, Affloss."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_affine_sample3d_golden(golden):
    from mireg import _lib
    from mireg.engine import _stream
    g = golden("g7_affine3d")
    vol = nets.analytic_input((2, 1, 6, 9, 7), seed=2).to(DEV)
    theta = torch.from_numpy(g["theta"]).to(DEV).contiguous()
    out = torch.empty_like(vol)
    _lib.call("mireg_affine_sample3d", vol.data_ptr(), theta.data_ptr(), out.data_ptr(), 2, 1, 6, 9, 7, _stream())
    assert (out.cpu() - torch.from_numpy(g["warped"])).abs().max().item() < 1e-5


def test_affmodel_fp32_vs_oracle_and_bf16():
    import mireg
    vol = (32, 32, 22)                                   # conv6 leaves 512 x (1,1,2)
    x = nets.analytic_input((2, 2, *vol), seed=3)
    o = nets.AffModel(fc_in=1024)
    nets.analytic_weights_(o)
    o.eval()
    with torch.no_grad():
        para_ref, warped_ref = o(x)
    for prec, tol in (("fp32", 2e-4), ("bf16", 5e-2)):
        m = mireg.affmodel(fc_in=1024, precision=prec)
        m.load_state_dict(o.state_dict())
        m = m.to(DEV).eval()
        with torch.no_grad():
            para, warped = m(x.to(DEV))
        assert para.shape == (2, 3, 4) and warped.shape == (2, 1, *vol)
        scale = para_ref.abs().max().item()
        assert (para.cpu() - para_ref).abs().max().item() <= tol * max(1.0, scale), prec
        if prec == "fp32":
            assert (warped.cpu() - warped_ref).abs().max().item() < 1e-3
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())


def test_affloss_golden(golden):
    import mireg
    g = golden("g3_losses")
    f3 = nets.analytic_input((2, 1, 8, 10, 6), seed=1).to(DEV)
    w3 = nets.analytic_input((2, 1, 8, 10, 6), seed=2).to(DEV)
    p, c, t = mireg.Affloss(w3, f3)
    got = torch.stack((p, c, t)).cpu()
    assert (got - torch.from_numpy(g["affloss"])).abs().max().item() < 1e-5 * max(1.0, float(g["affloss"].max()))


# ---- backward of the 3-D path (autograd of models.py:39-43,156-191 and loss.py:16-19,38-50,87-94) ----------------------
def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("cin,cout,k,stride,vol", [(16, 32, 5, (2, 2, 1), (9, 10, 7)), (32, 64, 3, (2, 2, 2), (7, 9, 6)),
                                                   (8, 16, 7, (2, 2, 1), (8, 9, 5)), (64, 12, 2, (1, 1, 1), (2, 2, 2))])
def test_conv3d_backward_vs_torch(cin, cout, k, stride, vol):
    """Backward-data (one launch per parity class) and backward-weights (one launch per depth tap) of Conv3d, fp32 MFMA."""
    import torch.nn.functional as F
    from mireg.affine3d import Conv3dLayer
    from mireg.engine import Workspace, run_pack, run_unpack
    B, pad = 2, (0, 0, 0) if k == 2 else ((k - 1) // 2,) * 3
    w = nets.analytic_input((cout, cin, k, k, k), seed=5, lo=-0.3, hi=0.3)
    b = nets.analytic_input((cout,), seed=6, lo=-0.1, hi=0.1)
    x = nets.analytic_input((B, cin, *vol), seed=7, lo=-1.0, hi=1.0).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv3d(x, wr, br, stride, pad)
    gy = nets.analytic_input(tuple(y.shape), seed=8, lo=-1.0, hi=1.0)
    y.backward(gy)

    ws = Workspace(torch.device(DEV), torch.float32)
    lay = Conv3dLayer(w.to(DEV), b.to(DEV), stride, pad, ws)
    run_pack([lay.pack_job()], ws.code, DEV)
    xd = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV)                   # NDHWC
    gyd = torch.zeros(B, *y.shape[2:], (cout + 7) // 8 * 8, device=DEV)
    gyd[..., :cout] = gy.permute(0, 2, 3, 4, 1).to(DEV)
    odims = tuple(y.shape[2:])
    assert lay.out_dims(*vol) == odims
    gx = torch.full((B, *vol, cin), 7.0, device=DEV)                               # every voxel must be overwritten
    lay.dgrad(gyd, odims, gx, vol)
    assert _rel(gx.permute(0, 4, 1, 2, 3).cpu(), x.grad) < 2e-5
    lay.wgrad(xd, vol, gyd, odims)
    gw, gb = torch.zeros_like(w, device=DEV), torch.zeros(cout, device=DEV)
    run_unpack([lay.unpack_job(gw)], DEV)
    lay.bias_grad(gyd, gb)
    assert _rel(gw.cpu(), wr.grad) < 2e-5
    assert _rel(gb.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("cin,cout,k,stride,pad", [(16, 32, 5, (2, 2, 2), (2, 2, 2)), (70, 12, 3, (1, 1, 1), (1, 1, 1)),
                                                  (3, 3, 4, (2, 2, 2), (1, 1, 1)), (130, 200, 3, (2, 2, 2), (1, 1, 1)),
                                                  (8, 16, 7, (2, 2, 1), (3, 3, 0))])
def test_conv3d_backward_data_packs_from_forward_packs_equal_the_direct_packs(prec, cin, cout, k, stride, pad):
    """mireg_pack_dgrad3d_fwd (class packs transposed out of the layer's forward pack, round 3) against mireg_pack_dgrad3d (gathered
    from the fp32 weights): bit-identical class packs, including the zero output-channel padding, for odd channel counts, channel
    tiles beyond 64 and every stride / padding the volume models use."""
    from mireg.affine3d import Conv3dLayer
    from mireg.engine import Workspace, run_pack
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    w = nets.analytic_input((cout, cin, k, k, k if stride[2] > 1 or k < 7 else 1), seed=5, lo=-0.3, hi=0.3).to(DEV)
    ws = Workspace(torch.device(DEV), dt)
    lay = Conv3dLayer(w, None, stride, pad, ws)
    run_pack([lay.pack_job()], ws.code, DEV)
    Conv3dLayer.pack_dgrad_table([lay], ws)
    direct = [c["pack"].clone() for c in lay.dgrad_classes()]
    for c in lay.dgrad_classes():
        c["pack"].fill_(7.0)
    Conv3dLayer.pack_dgrad_table([lay], ws, from_fwd=True)
    torch.cuda.synchronize()
    for a, c in zip(direct, lay.dgrad_classes()):
        assert torch.equal(a, c["pack"])


def test_affine_sample3d_backward_vs_oracle():
    from mireg import _lib
    from mireg.engine import _stream
    B, D, H, W = 2, 6, 9, 7
    vol = nets.analytic_input((B, 1, D, H, W), seed=2)
    theta = (torch.eye(3, 4).repeat(B, 1, 1) + 0.15 * nets.analytic_input((B, 3, 4), seed=4, lo=-1.0, hi=1.0)).requires_grad_(True)
    gout = nets.analytic_input((B, 1, D, H, W), seed=9, lo=-1.0, hi=1.0)
    oops.affine_grid_sample_3d(vol, theta).backward(gout)
    gt = torch.full((B, 12), 0.5, device=DEV)
    wsb = torch.empty(B * 512 * 12, device=DEV)
    vd, td, gd = vol.to(DEV), theta.detach().to(DEV).contiguous(), gout.to(DEV)     # keep the device copies alive
    _lib.call("mireg_affine_sample3d_bwd", vd.data_ptr(), td.data_ptr(), gd.data_ptr(), gt.data_ptr(), wsb.data_ptr(),
              1, B, 1, D, H, W, _stream())
    assert _rel(gt.cpu() - 0.5, theta.grad.reshape(B, 12)) < 1e-4                  # accumulate = 1 keeps the seed value


def test_affloss_backward_vs_oracle():
    import mireg
    f3 = nets.analytic_input((2, 1, 8, 10, 6), seed=1)
    w3 = nets.analytic_input((2, 1, 8, 10, 6), seed=2).requires_grad_(True)
    p, c, t = oops.aff_loss(w3, f3, 0.7, 1.3)
    (t + 0.5 * c).backward()
    wd = w3.detach().to(DEV).requires_grad_(True)
    p2, c2, t2 = mireg.Affloss(wd, f3.to(DEV), 0.7, 1.3)
    (t2 + 0.5 * c2).backward()
    assert abs(t2.item() - t.item()) < 1e-5 * max(1.0, abs(t.item()))
    assert _rel(wd.grad.cpu(), w3.grad) < 1e-4


def test_affmodel_training_grads_vs_oracle():
    """loss.backward() through mireg.affmodel + mireg.Affloss against autograd of the CPU oracle, fp32 MFMA and bf16."""
    import mireg
    vol = (32, 32, 22)
    x = nets.analytic_input((2, 2, *vol), seed=3)
    o = nets.AffModel(fc_in=1024)
    nets.analytic_weights_(o)
    para_ref, warped_ref = o(x)
    loss_ref = oops.aff_loss(warped_ref, x[:, 0:1])[2] + 0.3 * (para_ref * para_ref).sum()
    loss_ref.backward()
    ref = {k: p.grad.clone() for k, p in o.named_parameters()}
    for prec, tol in (("fp32", 1e-3), ("bf16", None)):
        m = mireg.affmodel(fc_in=1024, precision=prec)
        m.load_state_dict(o.state_dict())
        m = m.to(DEV).train()
        xd = x.to(DEV)
        para, warped = m(xd)
        loss = mireg.Affloss(warped, xd[:, 0:1])[2] + 0.3 * (para * para).sum()
        loss.backward()
        assert abs(loss.item() - loss_ref.item()) <= (1e-4 if prec == "fp32" else 3e-2) * abs(loss_ref.item())
        for k, p in m.named_parameters():
            assert p.grad is not None and p.grad.shape == ref[k].shape, k
            g, r = p.grad.cpu().double().flatten(), ref[k].double().flatten()
            if tol is not None:
                assert _rel(g, r) < tol, (prec, k, _rel(g, r))
            else:
                cos = torch.dot(g, r) / (g.norm() * r.norm() + 1e-300)
                assert cos > 0.98, (prec, k, cos.item())


def test_affmodel_adam_steps_reduce_loss():
    """A few torch.optim.Adam steps on the HIP gradients lower Affloss (the reference would train it with train.py's recipe)."""
    import mireg
    vol = (32, 32, 22)
    x = nets.analytic_input((2, 2, *vol), seed=3).to(DEV)
    m = mireg.affmodel(fc_in=1024, precision="bf16").to(DEV).train()
    torch.manual_seed(0)
    with torch.no_grad():
        m.fc.weight.mul_(0.01)
        m.fc.bias.copy_(torch.eye(3, 4).flatten() + 0.05)
    opt = torch.optim.Adam(m.parameters(), 1e-4, eps=1e-4)
    losses = []
    for _ in range(8):
        para, warped = m(x)
        loss = mireg.Affloss(warped, x[:, 0:1])[2]
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def test_affmodel_reference_volume_size_trains():
    """The reference's hard-wired shape (models.py:167: fc = Linear(176 * 512, 12) <=> 256 x 256 x 176 volumes), batch 1:
    forward + Affloss + backward run, the sampler output matches the oracle's sampler on the predicted parameters."""
    import mireg
    x = nets.analytic_input((1, 2, 256, 256, 176), seed=21).to(DEV)
    m = mireg.affmodel(precision="bf16").to(DEV).train()
    with torch.no_grad():
        m.fc.weight.mul_(0.01)
        m.fc.bias.copy_(torch.eye(3, 4).flatten() + 0.02)
    para, warped = m(x)
    assert para.shape == (1, 3, 4) and warped.shape == (1, 1, 256, 256, 176)
    loss = mireg.Affloss(warped, x[:, 0:1])[2]
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max().item() > 0, k
    ref = oops.affine_grid_sample_3d(x[:, 1:2].cpu(), para.detach().cpu())
    assert (warped.detach().cpu() - ref).abs().max().item() < 1e-4
