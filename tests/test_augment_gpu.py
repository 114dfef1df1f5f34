"""GPU parity of the on-device elastic deformation against the torch ops the CPU generator uses (mireg/synth.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cpu_reference(img, seg, ctrl):
    B, C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    disp = F.interpolate(ctrl, size=(H, W), mode="bicubic", align_corners=True)
    grid = torch.stack((xs.unsqueeze(0) + disp[:, 0] * 2 / W, ys.unsqueeze(0) + disp[:, 1] * 2 / H), -1)
    mov = F.grid_sample(img, grid, mode="bicubic", padding_mode="zeros", align_corners=True).clamp(0, 1)
    sg = F.grid_sample(seg, grid, mode="nearest", padding_mode="zeros", align_corners=True)
    return disp, mov, sg


@pytest.mark.parametrize("size,spacing,mag", [(256, 16, 0.5), (64, 16, 1.0), (96, 8, 3.0)])
def test_elastic_deform_matches_the_cpu_generator_ops(size, spacing, mag):
    from mireg import _lib
    from mireg.engine import _stream
    from mireg.synth import elastic_deform, make_pairs
    x, s = make_pairs(3, size, seed=4)
    img, seg = x[:, 0:1].contiguous(), s[:, 0:1].contiguous()
    g = torch.Generator().manual_seed(9)
    cg = size // spacing + 1
    ctrl = (torch.rand(3, 2, cg, cg, generator=g) * 2 - 1) * mag * spacing
    disp_ref, mov_ref, seg_ref = _cpu_reference(img, seg, ctrl)
    cd = ctrl.to(DEV)
    disp = torch.empty(3, 2, size, size, device=DEV)
    _lib.call("mireg_resize_bicubic_fwd", cd.data_ptr(), disp.data_ptr(), 6, cg, cg, size, size, _stream())
    assert (disp.cpu() - disp_ref).abs().max().item() < 1e-4 * max(1.0, disp_ref.abs().max().item())
    mov, sg = elastic_deform(img.to(DEV), seg.to(DEV), cd)
    assert (mov.cpu() - mov_ref).abs().max().item() < 2e-4
    # nearest-neighbour ties (coordinate within float rounding of .5) may fall on either side: a handful of pixels at most
    assert (sg.cpu() != seg_ref).float().mean().item() < 2e-4
    only_img, none = elastic_deform(img.to(DEV), None, cd)
    assert none is None and torch.equal(only_img, mov)
    with pytest.raises(RuntimeError, match="does not match"):
        elastic_deform(img.to(DEV), None, cd[:2])


def test_affine_deform_matches_torch_affine_grid_sample():
    from mireg.synth import affine_deform, make_pairs
    x, s = make_pairs(3, 96, seed=5)
    img, seg = x[:, 0:1].contiguous(), s[:, 0:1].contiguous()
    g = torch.Generator().manual_seed(2)
    theta = torch.eye(2, 3).repeat(3, 1, 1) + 0.2 * (torch.rand(3, 2, 3, generator=g) - 0.5)
    grid = F.affine_grid(theta, list(img.shape), align_corners=False)
    mov_ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    seg_ref = F.grid_sample(seg, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    mov, sg = affine_deform(img.to(DEV), seg.to(DEV), theta.to(DEV))
    assert (mov.cpu() - mov_ref).abs().max().item() < 2e-5
    assert (sg.cpu() != seg_ref).float().mean().item() < 2e-4                   # nearest-neighbour ties only
    with pytest.raises(RuntimeError, match="does not match"):
        affine_deform(img.to(DEV), None, theta[:2].to(DEV))
