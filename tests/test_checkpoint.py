"""CPU: checkpoint / weight-format interop with the reference (train.py:150-156,183-201; models.py:247,305-309; .flo files)."""
import struct

import numpy as np
import pytest
import torch

from oracle import nets


def test_reference_layout_checkpoints_round_trip(tmp_path):
    import mireg
    from mireg import checkpoint as ck
    ref = nets.OpticalFlowReg("flownets")                     # the reference's module tree (same state_dict keys)
    nets.analytic_weights_(ref)
    opt = torch.optim.Adam(ref.parameters(), 1e-4, eps=1e-4)
    # a file exactly as reference train.py:183-188 writes it
    path = str(tmp_path / "training_state.pt")
    torch.save({"epoch": 6, "model_state_dict": ref.state_dict(), "best_loss": 123.5, "optimizer_state_dict": opt.state_dict()}, path)
    m = mireg.opticalFlowReg("flownets")
    start, best = ck.load_training_state(path, m)
    assert (start, best) == (7, 123.5)
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    # and written by the build, read back by plain torch the way the reference does (train.py:150-156)
    path2 = str(tmp_path / "ts2.pt")
    ck.save_training_state(path2, m, opt.state_dict(), 3, 9.0)
    got = torch.load(path2, map_location="cpu", weights_only=False)
    assert sorted(got.keys()) == ["best_loss", "epoch", "model_state_dict", "optimizer_state_dict"]
    ref.load_state_dict(got["model_state_dict"])
    opt.load_state_dict(got["optimizer_state_dict"])
    path3 = str(tmp_path / "best_weight.pt")
    ck.save_best_weight(path3, m, dict(loss=1.0, photo_loss=2.0, corr_loss=3.0, smooth_loss=4.0),
                        dict(loss=5.0, photo_loss=6.0, corr_loss=7.0, smooth_loss=8.0))
    bw = torch.load(path3, map_location="cpu", weights_only=False)
    assert bw["loss_val"] == 1.0 and bw["smooth_loss"] == 8.0 and "model_state_dict" in bw      # train.py:195-201 keys


def test_rgb_folding_matches_reference_recipe():
    from mireg import checkpoint as ck
    torch.manual_seed(0)
    sd = {"conv1.0.weight": torch.randn(64, 6, 7, 7), "conv1a.0.weight": torch.randn(16, 3, 3, 3), "other": torch.ones(2)}
    out = ck.fold_rgb_pretrained(sd)                                                               # models.py:305-309
    w = sd["conv1.0.weight"]
    want = torch.cat([w[:, :3].sum(dim=1, keepdim=True), w[:, 3:].sum(dim=1, keepdim=True)], dim=1)
    assert torch.equal(out["conv1.0.weight"], want) and out["conv1.0.weight"].shape == (64, 2, 7, 7)
    out2 = ck.fold_rgb_pretrained(sd, "conv1a.0.weight", images=1)                                 # models.py:247
    assert torch.equal(out2["conv1a.0.weight"], sd["conv1a.0.weight"].sum(1, keepdim=True))
    assert out["other"] is sd["other"] and sd["conv1.0.weight"].shape[1] == 6                      # input untouched
    with pytest.raises(ValueError):
        ck.fold_rgb_pretrained(out)


def test_flo_files(tmp_path):
    from mireg import checkpoint as ck
    rng = np.random.default_rng(0)
    uv = rng.standard_normal((5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    ck.write_flo(p, uv)
    raw = open(p, "rb").read()
    assert raw[:4] == b"PIEH" and struct.unpack("<ii", raw[4:12]) == (7, 5) and len(raw) == 12 + 5 * 7 * 2 * 4
    assert struct.unpack("<f", raw[:4])[0] == 202021.25                                           # the Middlebury tag value
    assert np.array_equal(np.frombuffer(raw[12:20], np.float32), uv[0, 0])                        # u then v, row-major
    assert np.array_equal(ck.read_flo(p), uv)
    ck.write_flo(p, torch.from_numpy(uv).permute(2, 0, 1))                                        # (2, H, W) tensors as the predictors emit
    assert np.array_equal(ck.read_flo(p), uv)
    open(p, "wb").write(b"nope" + raw[4:])
    with pytest.raises(ValueError):
        ck.read_flo(p)


def test_pretrained_loader_accepts_bare_and_prefixed_checkpoints(tmp_path):
    """opticalFlowReg(pretrained=...) (reference models.py:244-252,305-309): bare predictor checkpoints (un-prefixed keys, RGB
    first layer) and registration checkpoints (`predictor.` keys) both load; a file that matches nothing raises instead of
    silently loading zero tensors."""
    import mireg
    torch.manual_seed(0)
    donor = mireg.FlowNetS(batchNorm=True)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    rgb = torch.randn(64, 6, 7, 7)
    sd["conv1.0.weight"] = rgb                                                  # FlyingChairs layout: two RGB frames
    bare = str(tmp_path / "bare.pth.tar")
    torch.save({"state_dict": sd}, bare)
    reg = mireg.opticalFlowReg("flownets", pretrained=bare)
    want = torch.cat([rgb[:, :3].sum(1, keepdim=True), rgb[:, 3:].sum(1, keepdim=True)], 1)
    assert torch.equal(reg.predictor.conv1[0].weight.detach(), want)
    assert torch.equal(reg.predictor.conv6_1[0].weight.detach(), sd["conv6_1.0.weight"])
    full = str(tmp_path / "full.pt")
    torch.save({"model_state_dict": reg.state_dict()}, full)
    reg2 = mireg.opticalFlowReg("flownets", pretrained=full)
    assert all(torch.equal(a, b) for a, b in zip(reg.state_dict().values(), reg2.state_dict().values()))
    junk = str(tmp_path / "junk.pt")
    torch.save({"something.else": torch.zeros(3)}, junk)
    with pytest.raises(RuntimeError, match="no tensor of the checkpoint matches"):
        mireg.opticalFlowReg("flownets", pretrained=junk)


def test_fused_trainer_refuses_a_predictor_without_an_engine():
    """opticalFlowReg('flownet2') is constructible, but the fused RegistrationTrainer only drives predictors with a buffer engine;
    it must say so at construction (before any flat optimizer buffer exists), not die in the first step."""
    import pytest
    import torch
    import mireg

    class NoEngine(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.predictor = torch.nn.Conv2d(2, 2, 3)
    with pytest.raises(RuntimeError, match="no fused engine"):
        mireg.RegistrationTrainer(NoEngine())


def test_measured_wgrad_shapes_are_cached_per_operand_shape():
    """A backward-weights launch shape measured at one batch / image size must not be replayed on another (the short last batch of
    an epoch, another resolution): the cache key carries the operand shapes and the dtype."""
    import torch
    from mireg.engine import ConvLayer, View, Workspace
    ws = Workspace(torch.device("cpu"), torch.bfloat16)
    lay = ConvLayer("conv3_1", torch.zeros(256, 256, 3, 3), None, 1, 1, 1, ws)
    mk = lambda B, H: View(torch.zeros(1, 1, 1, 256, dtype=torch.bfloat16), B, H, H, 256)
    k24, k7, k64 = lay._wgrad_key(mk(24, 32), mk(24, 32)), lay._wgrad_key(mk(7, 32), mk(7, 32)), lay._wgrad_key(mk(24, 64), mk(24, 64))
    assert len({k24, k7, k64}) == 3 and k24.startswith("conv3_1|")
