"""GPU: the FlowNet2 stack on the HIP engine (mireg/flownet2.py) against fixture G9 -- the reference's own FlowNet2 /
FlowNetSD / FlowNetFusion / flownet2-FlowNetS classes run on the CPU with the oracle's Correlation / Resample2d / ChannelNorm
injected (oracle/gen_golden.py g9; flownet2/models.py:30-191, networks/FlowNetSD.py, FlowNetFusion.py, FlowNetS.py).

Tolerances (fp32 = exact-fp32 MFMA): single networks 5e-4 of the output scale in eval mode (as FlowNetC's G5 test), 3e-3 with
batch statistics; the five-network
chain 2e-3 of the fused flow's scale -- every stage's flow error is multiplied by div_flow = 20 and moves the warp that
feeds the next stage.  bf16: relative L2 < 0.2 against the fp32 engine (three chained bf16 pyramids whose flow errors are
scaled by div_flow before each warp; measured 0.12)."""
import numpy as np
import pytest
import torch

from oracle import nets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _err(got, want):
    want = torch.as_tensor(np.asarray(want))
    return (got.float().cpu() - want).abs().max().item() / max(1.0, want.abs().max().item())


@pytest.mark.parametrize("which", ["FlowNetSD", "FlowNetFusion", "FlowNet2S"])
def test_flownet2_subnetworks_fp32_golden(golden, which):
    import mireg
    g = golden("g9_flownet2")
    key, shape, seed, ocls = {"FlowNetSD": ("flownetsd", (2, 2, 64, 64), 21, nets.FlowNetSD),
                              "FlowNetFusion": ("flownetfusion", (2, 9, 64, 64), 22, nets.FlowNetFusion),
                              "FlowNet2S": ("flownets", (2, 6, 64, 64), 23, nets.FlowNet2S)}[which]
    m = getattr(mireg, which)(None, batchNorm=True, precision="fp32")
    assert list(m.state_dict().keys()) == list(ocls(None, batchNorm=True).state_dict().keys())
    nets.analytic_weights_(m)
    m = m.to(DEV)
    x = nets.analytic_input(shape, seed=seed, lo=-1.0, hi=1.0).to(DEV)
    for mode in ("train", "eval"):
        m.train(mode == "train")
        with torch.no_grad():
            out = m(x)
        out = out if isinstance(out, tuple) else (out,)
        assert len(out) == sum(k.startswith(f"{key}_{mode}_") for k in g.files)
        # train mode: the deep BatchNorms normalise over 2..32 samples here (64x64 inputs, batch 2), which amplifies fp32
        # summation-order noise exactly as G1 records for FlowNetS
        tol = 3e-3 if mode == "train" else 5e-4
        for i, o in enumerate(out):
            assert _err(o, g[f"{key}_{mode}_{i}"]) <= tol, (mode, i, _err(o, g[f"{key}_{mode}_{i}"]))
    with pytest.raises(NotImplementedError):               # no silent untrainable flows: autograd on trainable parameters raises
        m(x)


def test_flownet2_chain_fp32_golden_and_bf16(golden):
    import mireg
    g = golden("g9_flownet2")
    m = mireg.FlowNet2(None, batchNorm=True, precision="fp32")
    assert list(m.state_dict().keys()) == list(nets.FlowNet2(None, batchNorm=True).state_dict().keys())
    nets.analytic_weights_(m)
    m = m.to(DEV).eval()
    x = nets.analytic_input((1, 2, 256, 256), seed=24).to(DEV)
    with torch.no_grad():
        st = m.stages(x)
        a, b = m(x)
    assert a.shape == (1, 2, 256, 256) and torch.equal(a, b)              # flownet2/models.py:189 returns the fused flow twice
    for name, t in zip(("flownetc_flow2", "flownets1_flow2", "flownets2_flow2", "flownetsd_flow2"), st[:4]):
        assert _err(t, g[f"flownet2_{name}"]) <= 1e-3, (name, _err(t, g[f"flownet2_{name}"]))
    assert _err(st[-1][:, :, ::2, ::2], g["flownet2_fused"]) <= 2e-3, _err(st[-1][:, :, ::2, ::2], g["flownet2_fused"])
    m16 = mireg.FlowNet2(None, batchNorm=True, precision="bf16")
    m16.load_state_dict(m.state_dict())
    m16 = m16.to(DEV).eval()
    with torch.no_grad():
        f16 = m16(x)[0]
    rel = ((f16 - a).norm() / a.norm()).item()
    assert rel < 0.2, rel                                  # measured 0.12 with these random weights


def test_registration_wrapper_with_flownet2():
    """opticalFlowReg('flownet2') (reference models.py:212-225): two identical full-resolution flows, each warped."""
    import mireg
    reg = mireg.opticalFlowReg("flownet2", precision="bf16").to(DEV).eval()
    x = nets.analytic_input((2, 2, 256, 256), seed=5).to(DEV)
    with torch.no_grad():
        flows, warped, _, _ = reg(x)
    assert len(flows) == 2 and len(warped) == 2 and flows[0].shape == (2, 2, 256, 256) and warped[0].shape == (2, 1, 256, 256)
    assert torch.isfinite(flows[0]).all() and torch.isfinite(warped[0]).all()
