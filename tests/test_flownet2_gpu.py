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
    # the reference's training loop on it (train.py:48-57): forward, OFEloss, backward, Adam(eps=1e-4)
    reg.train()
    opt = mireg.Adam(reg.parameters(), 1e-4, eps=1e-4)
    before = [p.detach().clone() for p in reg.parameters()]
    flows, warped, _, _ = reg(x)
    loss = mireg.OFEloss(flows, warped, x[:, 0:1])[3]
    opt.zero_grad()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in reg.parameters())
    opt.step()
    moved = sum(int(not torch.equal(a, b)) for a, b in zip(before, reg.parameters()))
    assert moved > 0.9 * len(before) and torch.isfinite(loss)


def _grad_report(m_hip, m_cpu, m_ref):
    """Against m_ref (the oracle run in float64): relative L2 over all parameter gradients together and the worst per-parameter
    relative L2 among the gradients that carry at least 1e-3 of the total norm, for the HIP model and for the fp32 oracle itself
    (= the fp32 noise of this computation: LeakyReLU kink flips, few-sample BatchNorms)."""
    gr = {k: p.grad.detach().double().flatten() for k, p in m_ref.named_parameters()}
    out = []
    for mod in (m_hip, m_cpu):
        g = {k: p.grad.detach().double().cpu().flatten() for k, p in mod.named_parameters()}
        assert set(g) == set(gr)
        tot = torch.cat([gr[k] for k in gr]).norm().item()
        err = torch.cat([g[k] - gr[k] for k in gr]).norm().item()
        worst = max(((g[k] - gr[k]).norm() / gr[k].norm()).item() for k in gr if gr[k].norm().item() > 1e-3 * tot)
        out.append((err / tot, worst))
    return out


@pytest.mark.parametrize("which", ["FlowNetSD", "FlowNetFusion", "FlowNet2S"])
def test_flownet2_subnetworks_backward_vs_cpu_autograd(which):
    """HIP backward of the stack's sub-networks (every parameter gradient, and the input gradient where the chain needs one)
    against torch autograd through the CPU restatement, fp32, train-mode BatchNorm.  Bounds as for FlowNetS (G1): LeakyReLU
    kink flips and few-sample BatchNorms move any fp32 implementation by a few 1e-3 of a gradient's norm."""
    import mireg
    shape, seed, ocls = {"FlowNetSD": ((2, 2, 128, 128), 31, nets.FlowNetSD), "FlowNetFusion": ((2, 9, 64, 64), 32, nets.FlowNetFusion),
                         "FlowNet2S": ((2, 6, 128, 128), 33, nets.FlowNet2S)}[which]
    o = ocls(None, batchNorm=True)
    nets.analytic_weights_(o)
    m = getattr(mireg, which)(None, batchNorm=True, precision="fp32")
    m.load_state_dict(o.state_dict())
    m = m.to(DEV)
    o.train(); m.train()
    x = nets.analytic_input(shape, seed=seed, lo=-1.0, hi=1.0)
    need_dx = which != "FlowNetSD"
    xc = x.clone().requires_grad_(need_dx)
    xg = x.clone().to(DEV).requires_grad_(need_dx)
    oc, og = o(xc), m(xg)
    oc, og = (oc if isinstance(oc, tuple) else (oc,)), (og if isinstance(og, tuple) else (og,))
    gen = torch.Generator().manual_seed(seed)
    cots = [torch.randn(t.shape, generator=gen) for t in oc]
    sum((a * c).sum() for a, c in zip(oc, cots)).backward()
    sum((a * c.to(DEV)).sum() for a, c in zip(og, cots)).backward()
    # the same computation in float64 is the reference; the fp32 oracle's distance from it is the noise unit of the bound (G1's form)
    import copy
    o64 = copy.deepcopy(o).double()
    o64.zero_grad()
    x64 = x.double().clone().requires_grad_(need_dx)
    o64o = o64(x64)
    o64o = o64o if isinstance(o64o, tuple) else (o64o,)
    sum((a * c.double()).sum() for a, c in zip(o64o, cots)).backward()
    (total, worst), (n_total, n_worst) = _grad_report(m, o, o64)
    assert total <= 5e-3 + 8 * n_total and worst <= 5e-3 + 8 * n_worst, (which, total, worst, n_total, n_worst)
    print(which, "gradient rel L2 vs fp64 oracle: HIP", (total, worst), "fp32 oracle", (n_total, n_worst))
    if need_dx:
        rel = ((xg.grad.cpu().double() - x64.grad).norm() / x64.grad.norm()).item()
        noise = ((xc.grad.double() - x64.grad).norm() / x64.grad.norm()).item()
        assert rel <= 5e-3 + 8 * noise, (rel, noise)


def test_flownet2_trains_end_to_end():
    """loss.backward() through the whole chain (five sub-networks, three warps): gradients reach every sub-network and agree
    in direction with torch autograd through the CPU restatement (cosine per sub-network; the chain multiplies flow errors by
    div_flow at every stage, so element-wise bounds are not meaningful here)."""
    import mireg
    o = nets.FlowNet2(None, batchNorm=True)
    nets.analytic_weights_(o)
    m = mireg.FlowNet2(None, batchNorm=True, precision="fp32")
    m.load_state_dict(o.state_dict())
    m = m.to(DEV)
    o.train(); m.train()
    x = nets.analytic_input((2, 2, 256, 256), seed=41)
    cot = torch.randn(2, 2, 256, 256, generator=torch.Generator().manual_seed(42))
    (o(x)[0] * cot).sum().backward()
    (m(x.to(DEV))[0] * cot.to(DEV)).sum().backward()
    # float64 run of the oracle = reference; per sub-network: rel L2 <= 5e-3 + 8 * (the fp32 oracle's own distance from it)
    import copy
    o64 = copy.deepcopy(o).double()
    o64.zero_grad()
    (o64(x.double())[0] * cot.double()).sum().backward()
    report = {}
    for sub in ("flownetc", "flownets_1", "flownets_2", "flownets_d", "flownetfusion"):
        gh = torch.cat([p.grad.detach().double().cpu().flatten() for p in getattr(m, sub).parameters()])
        gc = torch.cat([p.grad.detach().double().flatten() for p in getattr(o, sub).parameters()])
        gr = torch.cat([p.grad.detach().flatten() for p in getattr(o64, sub).parameters()])
        assert gr.norm().item() > 0
        rel, noise = ((gh - gr).norm() / gr.norm()).item(), ((gc - gr).norm() / gr.norm()).item()
        report[sub] = (rel, noise)
        assert rel <= 5e-3 + 8 * noise, (sub, rel, noise)
    print("FlowNet2 end-to-end gradient rel L2 vs fp64 oracle per sub-network (HIP, oracle-fp32 noise):", report)


@pytest.mark.parametrize("name", ["flownets", "flownetc", "pwc", "flownet2"])
def test_packed_domain_adam_under_autograd_matches_the_plain_optimizer(name):
    """mireg.Adam(fuse=model) for the models that train through torch.autograd (round 3): convolution weights updated from their
    backward-weights slabs by `mireg_adam_pack` (no gradient unpack, no re-pack; no `.grad` on those weights), everything else through
    `.grad` / `mireg_adam_step`.  Four steps of both optimizers from the same seed: same losses, same parameters, and the forward
    after the last step (which runs on the packs the optimizer rewrote) gives the same flows."""
    import mireg
    from mireg.synth import make_pairs
    x, _ = make_pairs(2, 128 if name != "flownet2" else 256, seed=4)
    x = x.to(DEV)

    def build(fuse):
        torch.manual_seed(11)
        m = mireg.opticalFlowReg(name, precision="fp32").to(DEV).train()
        return m, mireg.Adam(m.parameters(), 1e-4, eps=1e-4, fuse=m if fuse else None)

    def step(m, opt):
        flows, warped, _, _ = m(x)
        loss = mireg.OFEloss(flows, warped, x[:, 0:1])[3]
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.detach()

    (ma, oa), (mb, ob) = build(False), build(True)
    for _ in range(4):
        la, lb = step(ma, oa), step(mb, ob)
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), (float(la), float(lb))
    torch.cuda.synchronize()
    fused_any = False
    for (k, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.isfinite(b).all(), k
        assert (a - b).abs().max().item() <= 1e-6 + 2e-5 * a.abs().max().item(), k
        if b.dim() == 4 and b.grad is None and a.grad is not None:
            fused_any = True
    assert fused_any
    ma.eval(), mb.eval()
    with torch.no_grad():
        fa, fb = ma(x)[0][0], mb(x)[0][0]
    assert (fa - fb).abs().max().item() <= 1e-5 + 1e-4 * fa.abs().max().item()
    # one backward per step
    mb.train()
    flows, warped, _, _ = mb(x)
    mireg.OFEloss(flows, warped, x[:, 0:1])[3].backward()
    flows, warped, _, _ = mb(x)
    with pytest.raises(RuntimeError, match="second backward"):
        mireg.OFEloss(flows, warped, x[:, 0:1])[3].backward()


def test_autotune_leaves_the_training_state_untouched_and_training_equivalent():
    """mireg.autotune(model, run): one discarded forward + backward in which every engine times its launch shapes.  Afterwards the
    parameters, their `.grad`, the BatchNorm running statistics / batch counters and a packed-domain optimizer's pending state are what
    they were, sites are recorded, and the next training step gives the loss of an untuned twin (fp32: different split-K orders only)."""
    import mireg
    from mireg.synth import make_pairs
    x, _ = make_pairs(2, 128, seed=9)
    x = x.to(DEV)

    def build():
        torch.manual_seed(21)
        m = mireg.opticalFlowReg("flownetc", precision="fp32").to(DEV).train()
        return m, mireg.Adam(m.parameters(), 1e-4, eps=1e-4, fuse=m)

    def fwd_bwd(m):
        flows, warped, _, _ = m(x)
        loss = mireg.OFEloss(flows, warped, x[:, 0:1])[3]
        loss.backward()
        return loss.detach()

    def step(m, opt):
        opt.zero_grad()
        loss = fwd_bwd(m)
        opt.step()
        return float(loss)

    (ma, oa), (mb, ob) = build(), build()
    step(ma, oa), step(mb, ob)
    before = {k: v.detach().clone() for k, v in mb.state_dict().items()}
    gbefore = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in mb.named_parameters()}
    n = mireg.autotune(mb, lambda: fwd_bwd(mb))
    assert n > 20
    for k, v in mb.state_dict().items():
        assert torch.equal(v, before[k]), k
    for k, p in mb.named_parameters():
        assert (p.grad is None) == (gbefore[k] is None) and (p.grad is None or torch.equal(p.grad, gbefore[k])), k
    for i in range(3):
        la, lb = step(ma, oa), step(mb, ob)
        # first step: the same weights, other launch shapes (summation orders) only.  Later steps: Adam's +-lr moves on noise-level gradient
        # entries amplify those last-bit differences step by step (seen: 1e-7, 1e-5, 1.6e-2 relative), as between any two summation orders
        assert abs(la - lb) <= (2e-6 if i == 0 else 5e-2) * abs(la), (i, la, lb)
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in sa:
        if "num_batches_tracked" in k:
            assert int(sa[k]) == int(sb[k]), k
