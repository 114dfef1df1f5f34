"""GPU edge cases of the drop-in surface: ragged / minimal shapes, batch 1, error behaviour (no silent fallbacks)."""
import ctypes

import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_c_abi_rejects_bad_arguments_without_launching():
    from mireg import _lib
    lib = _lib.lib()
    buf = torch.zeros(64, device=DEV)
    p = buf.data_ptr()
    assert lib.mireg_stn_warp_fwd(None, 0, 0, 0, p, None, p, None, 1, 1, 4, 4, None) == -1          # null flow
    assert lib.mireg_stn_warp_fwd(p, 32, 16, 1, p, p, p, None, 1, 1, 4, 4, None) == -1              # fixed without sums
    assert lib.mireg_resize_bilinear_fwd(p, p, 1, 1, 0, 4, 4, 4, 0, 0, 1, 0, 0, 1, 1, None) == -1   # empty input plane
    assert lib.mireg_dice(p, p, p, p, 0, 16, None) == -1                                            # empty batch
    assert lib.mireg_correlation_fwd(p, 8, p, 8, p, 8, 1, 2, 2, 8, 8, 3, 2, 0.1, 1, None) == -1     # md % stride2 != 0
    assert lib.mireg_thin_conv_fwd(p, 6, p, 72, None, p, 2, None, 0, 1, 2, 2, 6, 1, None) == -1     # Cpad not a granule multiple
    d = (ctypes.c_byte * 1024)()
    assert lib.mireg_conv_gemm(ctypes.cast(d, ctypes.c_void_p), None) == -1                         # zeroed descriptor
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.call("mireg_seg_round", None, p, 4, None)


def test_predictors_reject_unsupported_image_sizes_loudly():
    import mireg
    m = mireg.opticalFlowReg("flownets", precision="fp32").to(DEV)
    with pytest.raises(RuntimeError, match="divisible by 64"):
        m(torch.zeros(1, 2, 96, 100, device=DEV))
    # reference models.py:208-252: any other string selects FlowNetS; the predictor outside the hot path (RAFT) says so
    assert type(mireg.opticalFlowReg("anything-else").predictor).__name__ == "FlowNetS"
    with pytest.raises(NotImplementedError):
        mireg.opticalFlowReg("raft")


@pytest.mark.parametrize("B", [1, 3])
def test_flownets_batch_one_and_odd_batches_eval_and_train(B):
    """B=1 is what the reference's inference loop feeds (inference.py:43); BatchNorm over B*h*w rows, deepest level 1x1."""
    import mireg
    torch.manual_seed(2)
    m = mireg.opticalFlowReg("flownets", precision="fp32")
    nets.analytic_weights_(m)
    o = nets.OpticalFlowReg("flownets")
    o.load_state_dict(m.state_dict())
    x = nets.analytic_input((B, 2, 64, 64), seed=5)
    m = m.to(DEV)
    for train in (False, True):
        m.train(train), o.train(train)
        if train and B == 1:                 # 1x1 deepest level: torch's BatchNorm refuses a single value per channel
            with pytest.raises(ValueError, match="more than 1 value per channel"):
                o(x)
            with pytest.raises(ValueError, match="more than 1 value per channel"):
                m(x.to(DEV))
            continue
        with torch.no_grad():
            flows, warped, _, _ = m(x.to(DEV))
            rf, rw, _, _ = o(x)
        assert len(flows) == len(rf) and [tuple(f.shape) for f in flows] == [tuple(f.shape) for f in rf]
        for a, b in zip(flows, rf):
            assert (a.cpu() - b).abs().max().item() <= 2e-3 * max(1.0, b.abs().max().item())
        for a, b in zip(warped, rw):
            assert (a.cpu() - b).abs().max().item() <= 2e-3


def test_ragged_scales_through_ofeloss_match_oracle():
    """Non-square, odd and 1x1 scales (the 64x64 config reaches a 1x1 flow): values and gradients vs the CPU oracle."""
    import mireg
    g = torch.Generator().manual_seed(9)
    B = 2
    fixed = torch.rand(B, 1, 24, 40, generator=g)
    moving = torch.rand(B, 1, 24, 40, generator=g)
    shapes = [(24, 40), (7, 13), (3, 5), (1, 1)]
    flows = [torch.randn(B, 2, h, w, generator=g) for h, w in shapes]
    fd = [f.clone().to(DEV).requires_grad_() for f in flows]
    fo = [f.clone().requires_grad_() for f in flows]
    wd = [mireg.ops.stn(f, moving.to(DEV)) for f in fd]
    wo = [oops.stn(f, moving) for f in fo]
    vd = mireg.OFEloss(fd, wd, fixed.to(DEV))
    vo = oops.ofe_loss(fo, wo, fixed)
    for a, b in zip(vd, vo):
        assert abs(a.item() - b.item()) <= 1e-5 * max(1.0, abs(b.item()))
    vd[3].backward(), vo[3].backward()
    for a, b in zip(fd, fo):
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 2e-4 * max(1.0, b.grad.abs().max().item())


def test_trainer_rejects_cpu_batches_and_refreshes_packs_after_load_state_dict():
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("flownets", precision="bf16")
    nets.analytic_weights_(m)
    m = m.to(DEV)
    x, _ = make_pairs(2, 64, seed=3)
    tr = mireg.RegistrationTrainer(m, use_graph=False, autotune=False)
    with pytest.raises(RuntimeError):
        tr.step(x)                                           # host tensor: no CPU path
    xd = x.to(DEV)
    l0 = tr.step(xd).tolist()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for _ in range(3):
        tr.step(xd)
    m.load_state_dict(sd)                                    # hook re-derives every GEMM pack from the restored masters
    tr.flat_m.zero_(), tr.flat_v.zero_(), tr.step_dev.fill_(1)
    l1 = tr.step(xd).tolist()
    # same weights as after the first step (BN running stats differ only in eval mode): the training loss matches the
    # second step of a fresh run to bf16 accuracy, which it would not with stale packs
    m2 = mireg.opticalFlowReg("flownets", precision="bf16")
    nets.analytic_weights_(m2)
    tr2 = mireg.RegistrationTrainer(m2.to(DEV), use_graph=False, autotune=False)
    tr2.step(xd)
    l1_ref = tr2.step(xd).tolist()
    assert abs(l1[3] - l1_ref[3]) <= 2e-2 * abs(l1_ref[3]), (l0, l1, l1_ref)


def test_trainer_survives_a_short_last_batch():
    """train.py's DataLoader has no drop_last: the final batch of an epoch is smaller.  The trainer rebuilds its
    per-shape state and keeps the optimizer state; losses stay finite and the weights keep moving."""
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("flownets", precision="bf16")
    nets.analytic_weights_(m)
    m = m.to(DEV)
    x, _ = make_pairs(6, 64, seed=3)
    xd = x.to(DEV)
    tr = mireg.RegistrationTrainer(m, use_graph=True, autotune=False)
    for _ in range(4):
        la = tr.step(xd[:4]).tolist()
    w0 = tr.flat_p.clone()
    lb = tr.step(xd[4:6]).tolist()                           # batch of 2 after batches of 4
    lc = tr.step(xd[:4]).tolist()
    assert all(v == v and abs(v) < 1e9 for v in la + lb + lc)
    assert (tr.flat_p - w0).abs().max().item() > 0
    assert int(tr.step_dev.item()) == 6


def test_full_size_properties_batch24_256():
    """BASELINE configs[1] shape (24 pairs of 256x256, bf16), where the CPU oracle is too slow for a direct comparison:
    size-independent properties instead -- run-to-run bit determinism, batch-permutation equivariance in eval mode,
    identity behaviour of the warp at zero flow, and two identically seeded trainers staying bit-identical."""
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("flownets", precision="bf16")
    nets.analytic_weights_(m)
    m = m.to(DEV).eval()
    x, _ = make_pairs(24, 256, seed=6)
    xd = x.to(DEV)
    with torch.no_grad():
        f1, w1, _, _ = m(xd)
        f1 = [t.clone() for t in f1]
        f2, w2, _, _ = m(xd)
        assert all(torch.equal(a, b) for a, b in zip(f1, f2))                      # deterministic kernels
        perm = torch.randperm(24, generator=torch.Generator().manual_seed(2)).to(DEV)
        f3, _, _, _ = m(xd[perm].contiguous())
        for a, b in zip(f1, f3):                                                    # eval-mode BN: samples independent
            assert (a[perm] - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
        zero = torch.zeros(24, 2, 256, 256, device=DEV)
        wz = m.stn(zero, xd[:, 1:2].contiguous())
        # (x+0)(w-1)/w sampling of the reference: interior pixels move by < 1 px; mean intensity is preserved to 1 %
        assert abs(wz.mean().item() - xd[:, 1:2].mean().item()) <= 1e-2 * xd[:, 1:2].mean().item()
    outs = []
    for _ in range(2):
        torch.manual_seed(1)
        mm = mireg.opticalFlowReg("flownets", precision="bf16")
        nets.analytic_weights_(mm)
        tr = mireg.RegistrationTrainer(mm.to(DEV), use_graph=True, autotune=False)
        for _ in range(4):
            loss = tr.step(xd)
        outs.append((loss.clone(), tr.flat_p.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) or (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-9 * outs[0][0].abs().max().item()
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-7          # f64 moment atomics are the only unordered sums
