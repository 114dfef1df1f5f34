"""GPU parity of the on-device front of the reference's data pipeline (dataset.py:52-57,73-77,83 and
141-153) against the torch / numpy operations MONAI's transforms are made of (MONAI itself is absent from this image):
SpatialCropd = slicing, Resized = F.interpolate(mode, align_corners=False for linear modes), Rotate90d = torch.rot90 on the spatial
axes, ScaleIntensityd = (x - min) / (max - min); and of the on-device contour extraction + dist_hausdorff (utils.py:155-170,
201-211) against a numpy restatement of find_contours' vertex set for binary masks and scipy's cdist."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _volume(shape, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 1, *[max(s // 8, 2) for s in shape], generator=g)
    return F.interpolate(low, size=shape, mode="trilinear", align_corners=True)[0, 0].contiguous() * 1000.0


def test_slices_from_volume_matches_crop_resize_rot90():
    from mireg.synth import slices_from_volume
    vol = _volume((176, 176, 208), 1)                              # (Z, Y, X) after the reference's Transposed
    seg = torch.bucketize(vol, torch.tensor([300.0, 500.0, 700.0])).float()
    img, sg = slices_from_volume(vol.to(DEV), seg.to(DEV), z_range=(60, 140), yx_size=(176, 208), size=256, rot_k=1)
    assert tuple(img.shape) == (80, 1, 256, 256) and tuple(sg.shape) == (80, 1, 256, 256)
    crop, cseg = vol[60:140].unsqueeze(1), seg[60:140].unsqueeze(1)
    ref = torch.rot90(F.interpolate(crop, size=(256, 256), mode="bilinear", align_corners=False), 1, (2, 3))
    ref_seg = torch.rot90(F.interpolate(cseg, size=(256, 256), mode="nearest"), 1, (2, 3))
    assert (img.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert torch.equal(sg.cpu(), ref_seg)


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_resample_volume_rotations_and_views(k):
    """Every rot90 count, a transposed + cropped source view (strides only), ragged sizes, up- and down-scaling."""
    from mireg.synth import resample_volume
    base = _volume((9, 40, 52), 2 + k)
    view = base.permute(0, 2, 1)[1:8, 3:50, 2:37]                  # (7, 47, 35): Transposed + SpatialCropd as a view
    dview = base.to(DEV).permute(0, 2, 1)[1:8, 3:50, 2:37].unsqueeze(0)          # the same view of device memory (non-contiguous)
    out = resample_volume(dview, (7, 64, 24), "bilinear", k)
    ref = torch.rot90(F.interpolate(view.unsqueeze(1), size=(64, 24), mode="bilinear", align_corners=False), k, (2, 3))[:, 0]
    assert tuple(out.shape[1:]) == tuple(ref.shape) and (out[0].cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    outn = resample_volume(dview, (7, 64, 24), "nearest", k)
    refn = torch.rot90(F.interpolate(view.unsqueeze(1), size=(64, 24), mode="nearest"), k, (2, 3))[:, 0]
    assert torch.equal(outn[0].cpu(), refn)


def test_prepare_volume_and_affine_deform3d():
    """volume_ds: Resized to (s0, s1, s2) trilinear + Rotate90d(k=2, spatial_axes=(0, 1)), then the affine resampling step."""
    from mireg.synth import affine_deform3d, prepare_volume, scale_intensity
    vol = _volume((44, 36, 30), 7)                                 # (A0, A1, A2)
    out = prepare_volume(vol.to(DEV), size=(64, 64, 44), rot_k=2)
    ref = torch.rot90(F.interpolate(vol[None, None], size=(64, 64, 44), mode="trilinear", align_corners=False), 2, (2, 3))[0]
    assert tuple(out.shape) == (1, 64, 64, 44) and (out.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    g = torch.Generator().manual_seed(3)
    theta = torch.eye(3, 4).unsqueeze(0) + 0.1 * (torch.rand(1, 3, 4, generator=g) - 0.5)
    x5 = ref.unsqueeze(0)
    mov_ref = F.grid_sample(x5, F.affine_grid(theta, list(x5.shape), align_corners=False), mode="bilinear", padding_mode="zeros", align_corners=False)
    mov = affine_deform3d(out.unsqueeze(0), theta.to(DEV))
    assert (mov.cpu() - mov_ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    pair = torch.cat((out.unsqueeze(0), mov), 1).contiguous()     # ConcatItemsd + ScaleIntensityd over the 2-channel item
    want = (pair.cpu() - pair.cpu().amin()) / (pair.cpu().amax() - pair.cpu().amin())
    got = scale_intensity(pair.clone())
    assert (got.cpu() - want).abs().max().item() <= 1e-6


def test_scale_intensity_per_item_and_constant_item():
    from mireg.synth import scale_intensity
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 2, 37, 41, generator=g) * 300 + 50
    x[3] = 7.0                                                     # constant item: MONAI returns x * minv
    got = scale_intensity(x.to(DEV).clone(), 0.0, 1.0).cpu()
    for i in range(5):
        lo, hi = x[i].min(), x[i].max()
        want = (x[i] - lo) / (hi - lo) if hi > lo else x[i] * 0.0
        assert (got[i] - want).abs().max().item() <= 1e-6, i
    got2 = scale_intensity(x.to(DEV).clone(), -1.0, 3.0).cpu()
    assert abs(got2[0].min().item() + 1.0) <= 1e-6 and abs(got2[0].max().item() - 3.0) <= 1e-5


def _np_boundary(mask):
    """The vertices skimage.measure.find_contours(mask, 0.5) produces for a binary mask (midpoints of 4-neighbour pairs with differing
    values), truncated to int as the reference does -- without the repeated closing vertex of closed contours."""
    m = np.asarray(mask) > 0.5
    pts = []
    H, W = m.shape
    for r in range(H):
        for c in range(W):
            if c + 1 < W and m[r, c] != m[r, c + 1]:
                pts.append((r, c))
            if r + 1 < H and m[r, c] != m[r + 1, c]:
                pts.append((r, c))
    return np.array(pts, dtype=np.float64).reshape(-1, 2)


def test_boundary_points_and_dist_hausdorff_on_device():
    from scipy.spatial.distance import cdist
    import mireg
    from mireg.synth import make_pairs
    _, segs = make_pairs(2, 128, seed=11, magnitude=(0.5, 1.0))
    for b in range(2):
        s1, s2 = segs[b, 0], segs[b, 1]
        pts = mireg.extract_boundary_points((s1 == 2).float().to(DEV)).cpu().numpy()
        want = _np_boundary((s1 == 2).numpy())
        assert pts.shape == want.shape and np.array_equal(pts, want)          # same points, same (raster) order
        dists = []
        for lab in (1, 2, 3):
            A, Bp = _np_boundary((s1 == lab).numpy()), _np_boundary((s2 == lab).numpy())
            D = cdist(A, Bp)
            dists.append(max(np.mean(np.min(D, axis=0)), np.mean(np.min(D, axis=1))))
        got = float(mireg.dist_hausdorff(s1.to(DEV), s2.to(DEV)))
        assert abs(got - float(np.mean(dists))) <= 1e-5 * max(1.0, float(np.mean(dists))), (got, np.mean(dists))
    same = float(mireg.dist_hausdorff(segs[0, 0].to(DEV), segs[0, 0].to(DEV)))
    assert same == 0.0
