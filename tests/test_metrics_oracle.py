"""CPU: the oracle's evaluation-metric restatements (reference utils.py:41-59) against the reference's own dependencies where
they are importable here (sklearn) and against numpy for the published definitions."""
import numpy as np
import pytest
import torch

from oracle import nets, ops as oops


def test_metric_restatements():
    f = nets.analytic_input((1, 1, 48, 40), seed=1)[0, 0]
    w = (0.7 * f + 0.3 * nets.analytic_input((1, 1, 48, 40), seed=2)[0, 0]).clamp(0, 1)
    assert abs(oops.pearson(f, w).item() - np.corrcoef(f.reshape(-1).double().numpy(), w.reshape(-1).double().numpy())[0, 1]) < 1e-12
    assert abs(float(oops.psnr(f, w)) - 10 * np.log10(1.0 / float(((w - f) ** 2).mean()))) < 1e-5
    assert float(oops.psnr(f, f)) == 100.0
    sk = pytest.importorskip("sklearn.metrics")
    a = torch.round(f * 1500).int().reshape(-1).numpy()
    b = torch.round(w * 1500).int().reshape(-1).numpy()
    assert abs(oops.mutual_info(f, w) - sk.mutual_info_score(a, b)) < 1e-12
