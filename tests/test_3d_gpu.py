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
