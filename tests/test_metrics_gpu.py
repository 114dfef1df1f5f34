"""GPU parity of the per-sample evaluation metrics (reference utils.py:41-59, called per sample at inference.py:66-75)."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(B, size, seed):
    f = nets.analytic_input((B, 1, size, size), seed=seed)
    w = (0.8 * f + 0.2 * nets.analytic_input((B, 1, size, size), seed=seed + 1)).clamp(0, 1)
    return f, w


def test_pair_metrics_batch_vs_oracle():
    import mireg
    f, w = _pair(5, 256, 3)
    res = mireg.pair_metrics(f.to(DEV), w.to(DEV))
    for b in range(5):
        fb, wb = f[b, 0], w[b, 0]
        assert abs(res["mse"][b].item() - oops.mse(fb, wb).item()) < 1e-6 * max(1.0, oops.mse(fb, wb).item()) + 1e-9
        assert abs(res["psnr"][b].item() - float(oops.psnr(fb, wb))) < 1e-4
        assert abs(res["corr"][b].item() - oops.pearson(fb, wb).item()) < 1e-6
        assert abs(res["mi"][b].item() - oops.mutual_info(fb, wb)) < 1e-9 * 1e3
    sk = pytest.importorskip("sklearn.metrics")
    a = torch.round(f[0, 0] * 1500).int().reshape(-1).numpy()
    b = torch.round(w[0, 0] * 1500).int().reshape(-1).numpy()
    assert abs(res["mi"][0].item() - sk.mutual_info_score(a, b)) < 1e-6        # the reference's own dependency


def test_reference_named_single_sample_calls_and_edge_cases():
    import mireg
    f, w = _pair(1, 64, 7)
    fd, wd = f[0, 0].to(DEV), w[0, 0].to(DEV)
    assert abs(mireg.MSE(fd, wd).item() - oops.mse(f[0, 0], w[0, 0]).item()) < 1e-8
    assert abs(mireg.PSNR(fd, wd).item() - float(oops.psnr(f[0, 0], w[0, 0]))) < 1e-4
    assert abs(mireg.CORR(fd, wd).item() - oops.pearson(f[0, 0], w[0, 0]).item()) < 1e-6
    assert abs(mireg.MI(fd, wd).item() - oops.mutual_info(f[0, 0], w[0, 0])) < 1e-6
    assert mireg.PSNR(fd, fd).item() == 100.0                                  # utils.py:47-48: identical images
    assert abs(mireg.MI(fd, fd).item() - oops.mutual_info(f[0, 0], f[0, 0])) < 1e-6   # entropy of the label histogram
    with pytest.raises(RuntimeError, match="differ"):
        mireg.pair_metrics(fd.unsqueeze(0), wd[:32].unsqueeze(0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mireg.MSE(f[0, 0], w[0, 0])


def test_ssim_images_and_label_maps():
    """inference.py:70-71 calls SSIM on the image pair and on the (float) segmentation pair."""
    import mireg
    f, w = _pair(3, 96, 11)
    got = mireg.ssim_batch(f.to(DEV), w.to(DEV))
    for b in range(3):
        assert abs(got[b].item() - oops.ssim(f[b, 0], w[b, 0])) < 2e-5
    seg_a = torch.floor(f[0, 0] * 4).clamp(0, 3)
    seg_b = torch.floor(w[0, 0] * 4).clamp(0, 3)
    assert abs(mireg.structural_similarity(seg_a.to(DEV), seg_b.to(DEV)).item() - oops.ssim(seg_a, seg_b)) < 2e-5
    assert abs(mireg.structural_similarity(seg_a.to(DEV), seg_a.to(DEV)).item() - 1.0) < 1e-6          # float32 maps
    r = mireg.pair_metrics(f.to(DEV), w.to(DEV))
    assert "ssim" in r and abs(r["ssim"][1].item() - got[1].item()) < 1e-12
    with pytest.raises(RuntimeError, match="invalid argument"):
        mireg.ssim_batch(torch.zeros(1, 5, 5, device=DEV), torch.zeros(1, 5, 5, device=DEV))


def test_modified_hausdorff_of_point_sets_vs_scipy():
    """utils.py:187-199 on the reference's own dependency (scipy.spatial.distance.cdist)."""
    import numpy as np
    import mireg
    from scipy.spatial.distance import cdist
    rng = np.random.default_rng(5)
    for na, nb in ((1, 1), (37, 5), (700, 913)):
        A = rng.integers(0, 256, size=(na, 2)).astype(np.float32)
        Bp = rng.integers(0, 256, size=(nb, 2)).astype(np.float32)
        D = cdist(A, Bp)
        want = max(np.mean(np.min(D, axis=0)), np.mean(np.min(D, axis=1)))
        got = mireg.modified_hausdorff(torch.from_numpy(A).to(DEV), torch.from_numpy(Bp).to(DEV)).item()
        assert abs(got - want) < 1e-5 * max(1.0, want), (na, nb, got, want)
    with pytest.raises(RuntimeError, match="non-empty"):
        mireg.modified_hausdorff(torch.zeros(0, 2, device=DEV), torch.zeros(3, 2, device=DEV))
