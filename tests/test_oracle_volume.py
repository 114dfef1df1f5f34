"""CPU: the volume (3-D) restatements in the oracle are tied to the golden-pinned 2-D conventions they generalise, and the
build's 3-D modules keep the oracle's parameter names (no reference counterpart exists)."""
import torch

from oracle import nets, ops as oops


def test_stn3d_reduces_to_the_pinned_2d_stn_on_a_z_constant_volume():
    """A volume that does not change along z, warped with a flow whose z component is zero, must give the 2-D `stn` result
    (pinned by golden G2) in every slice: same (i + f)(n - 1)/n convention per axis, same zero padding."""
    B, d, h, w = 2, 5, 12, 10
    img = nets.analytic_input((B, 1, h, w), seed=1)
    f2 = 1.5 * nets.analytic_input((B, 2, h, w), seed=2, lo=-1.0, hi=1.0)
    vol = img.unsqueeze(2).expand(B, 1, d, h, w).contiguous()
    f3 = torch.cat((f2.unsqueeze(2).expand(B, 2, d, h, w), torch.zeros(B, 1, d, h, w)), dim=1).contiguous()
    got = oops.stn3d(f3, vol)
    want = oops.stn(f2, img)
    for z in range(d):
        assert (got[:, :, z] - want).abs().max().item() < 2e-6, z


def test_volume_losses_reduce_to_the_2d_ones():
    """photometric_loss_3d / correlation_loss_3d are the 2-D formulas on flattened tensors (loss.py:16-19,38-50 vs 9-14,52-64)."""
    a = nets.analytic_input((2, 1, 6, 8, 5), seed=3)
    b = nets.analytic_input((2, 1, 6, 8, 5), seed=4)
    a2, b2 = a.reshape(2, 1, 6, 40), b.reshape(2, 1, 6, 40)
    assert abs(oops.photometric_loss_3d(a, b).item() - oops.photometric_loss(a2, b2).item()) < 1e-5
    assert abs(oops.correlation_loss_3d(a, b).item() - oops.correlation_loss(a2, b2).item()) < 1e-6
    # three-axis smoothness of a flow that is constant along z with a zero z channel, B = 1, written out term by term:
    #   in-plane differences of the two real channels: d slices x the 2-D raw sum (= 2 * smoothness_loss(f2));
    #   their z differences: charbonnier(0) inside, charbonnier(f - 0) at the zero-extended last slice;
    #   the all-zero channel: charbonnier(0) for every voxel and axis;   all of it / 3 channels
    f2 = nets.analytic_input((1, 2, 7, 9), seed=5, lo=-1.0, hi=1.0)
    d, npix = 4, 7 * 9
    f3 = torch.cat((f2.unsqueeze(2).expand(1, 2, d, 7, 9), torch.zeros(1, 1, d, 7, 9)), dim=1).contiguous()
    c0 = oops.charbonnier(torch.zeros(1)).item()
    want = (d * 2 * oops.smoothness_loss(f2).item() + (d - 1) * npix * 2 * c0 + oops.charbonnier(f2).sum().item() + 3 * d * npix * c0) / 3
    assert abs(oops.smoothness_loss_3d(f3).item() - want) < 1e-5 * max(1.0, want)


def test_build_modules_keep_the_oracle_parameter_names():
    import mireg
    o = nets.OpticalFlowReg3d(8)
    m = mireg.opticalFlowReg3d(width_div=8)                                  # construction needs no GPU
    assert list(o.state_dict().keys()) == list(m.state_dict().keys())
    assert [tuple(v.shape) for v in o.state_dict().values()] == [tuple(v.shape) for v in m.state_dict().values()]
    a = nets.AffModel(fc_in=1024)
    assert list(a.state_dict().keys()) == list(mireg.affmodel(fc_in=1024).state_dict().keys())
