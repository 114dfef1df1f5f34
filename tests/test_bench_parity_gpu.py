"""GPU parity AT THE SIZES bench.py measures (BASELINE.json configs[1]..[4]) against the CPU oracle.

configs[1]  FlowNetS, 24 pairs of 256x256: one RegistrationTrainer.step, autotuned launch shapes + hipGraph replay, bf16
            (the benchmarked configuration) and fp32 (parity mode): losses and all six flows of the first steps.
configs[2]  FlowNetC, 24 pairs;  configs[3]  PWC-DC-Net, 48 pairs: eval-mode forward (fp32 and bf16) and, since round 3, the training
            step with autotune + hipGraph in fp32: loss scalars of three steps and first-step weight gradients vs the oracle.
configs[4]  FlowNetS-3D at FULL width on a 128^3 volume pair: training-mode forward, OFEloss3d, a handful of parameter gradients.

Tolerances.  fp32 (exact-fp32 MFMA): |flow error| <= 1e-4 * max(1, scale) + 4 * noise, noise = the oracle's own fp32 distance
from its float64 run on the same batch (measured here, ~1e-5): north_star's "flow L2 vs reference < 1e-4".  bf16 operands:
relative L2 per flow <= max(5e-2, 10 * eps_op), eps_op = the oracle's relative L2 move when only its weights and input are
rounded to bf16 (the GPU also rounds every activation, ~14 layers deep); losses 1e-2 relative."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _engine_flows(tr):
    e = tr.eng
    return [e.flow0.nchw().float().cpu().clone()] + [e.flow32[l].nchw().float().cpu().clone() for l in (2, 3, 4, 5, 6)]


def _oracle_steps(sd, x, n, dtype=torch.float32):
    om = nets.OpticalFlowReg("flownets")
    om.load_state_dict(sd)
    om = om.to(dtype).train()
    opt = torch.optim.Adam(om.parameters(), 1e-4, betas=(0.9, 0.999), eps=1e-4)
    out = []
    for _ in range(n):
        flows, warped, _, _ = om(x.to(dtype))
        vals = oops.ofe_loss(flows, warped, x[:, 0:1].to(dtype))
        opt.zero_grad(); vals[3].backward(); opt.step()
        out.append(([f.detach().clone() for f in flows], [float(v) for v in vals]))
    return out


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_flownets_step_batch24_256_graph_autotune_vs_oracle(prec):
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    model = mireg.opticalFlowReg("flownets", precision=prec)
    nets.analytic_weights_(model)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x, _ = make_pairs(24, 256, seed=6)
    ref = _oracle_steps(sd, x, 4)
    tr = mireg.RegistrationTrainer(model.to(DEV), use_graph=True, autotune=True)
    xd = x.to(DEV)
    got = []
    for _ in range(4):                                   # steps 1-2 eager with the tuned shapes, 3 captures, 4 replays the graph
        losses = tr.step(xd).tolist()
        got.append((_engine_flows(tr), losses))
    assert tr._graphs is not None and len(tr.eng.ws.tuned) > 10
    if prec == "fp32":
        with torch.no_grad():                            # the oracle's own fp32 noise on this batch
            o64 = nets.OpticalFlowReg("flownets"); o64.load_state_dict(sd); o64 = o64.double().train()
            f64, _, _, _ = o64(x.double())
        noise = [float((a.double() - b).abs().max()) for a, b in zip(ref[0][0], f64)]
        for i, (a, b) in enumerate(zip(got[0][0], ref[0][0])):
            err, scale = (a - b).abs().max().item(), b.abs().max().item()
            assert err <= 1e-4 * max(1.0, scale) + 4 * noise[i], (i, err, scale, noise[i])
        for a, b in zip(got[0][1], ref[0][1]):
            assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (got[0][1], ref[0][1])
        # later steps: Adam (eps 1e-4) moves every weight whose gradient is at rounding level by +-lr, in a direction that depends
        # on the summation order, i.e. on the launch shapes the autotuner happened to measure fastest in this run; the
        # trajectories separate by ~0.2 % of a loss term per step (the smoothness term most: seen 0.05-0.7 % at the third step)
        for k, tol in ((1, 5e-3), (2, 2e-2), (3, 2e-2)):
            for a, b in zip(got[k][1], ref[k][1]):
                assert abs(a - b) <= tol * abs(b) + 1e-6, (k, got[k][1], ref[k][1])
        return
    with torch.no_grad():                                # operand-rounding noise of the oracle
        ob = nets.OpticalFlowReg("flownets")
        ob.load_state_dict({k: (v.bfloat16().float() if v.dim() >= 3 else v) for k, v in sd.items()})
        fb, _, _, _ = ob.train()(x.bfloat16().float())
    for i, (a, b) in enumerate(zip(got[0][0], ref[0][0])):
        eps_op = _rel_l2(fb[i], b)
        assert _rel_l2(a, b) <= max(5e-2, 10 * eps_op), (i, _rel_l2(a, b), eps_op)
    for k in range(4):
        for a, b in zip(got[k][1], ref[k][1]):
            assert abs(a - b) <= 1e-2 * abs(b) + 1e-6, (k, got[k][1], ref[k][1])
    # (flows of the later steps are not compared: Adam moves every weight by ~lr per step whatever its gradient's size, so weights
    # whose bf16 gradient is noise-level walk differently from the fp32 oracle's and the coarse flows decorrelate within three
    # steps -- the same happens between two fp32 summation orders, tests/test_trainer_gpu.py; the losses above stay within 1e-2,
    # and graph replay == eager is pinned bit-level by tests/test_trainer_gpu.py::test_graph_replay_equals_eager)


@pytest.mark.parametrize("name,B", [("flownetc", 24), ("pwc", 48)])
def test_flownetc_batch24_and_pwc_batch48_eval_forward_vs_oracle(name, B):
    import mireg
    x = nets.analytic_input((B, 2, 256, 256), seed=21)
    om = nets.OpticalFlowReg(name)
    nets.analytic_weights_(om)
    om.eval()
    with torch.no_grad():
        fref, wref, _, _ = om(x)
    for prec in ("fp32", "bf16"):
        m = mireg.opticalFlowReg(name, precision=prec)
        m.load_state_dict(om.state_dict())
        m = m.to(DEV).eval()
        with torch.no_grad():
            flows, warped, _, _ = m(x.to(DEV))
        assert len(flows) == len(fref)
        for i, (a, b) in enumerate(zip(flows, fref)):
            if prec == "fp32":
                err, scale = (a.cpu() - b).abs().max().item(), b.abs().max().item()
                assert err <= 5e-4 * max(1.0, scale), (name, i, err, scale)
            else:
                assert _rel_l2(a, b) <= 8e-2, (name, i, _rel_l2(a, b))
        if prec == "fp32":
            for a, b in zip(warped, wref):
                assert (a.cpu() - b).abs().max().item() <= 1e-3
        del m
        torch.cuda.empty_cache()


def _packed_grad_of(eng, flat_g, lay):
    o = eng.flat_off[id(lay.weight)]
    return flat_g[o:o + lay.Co * lay.Kf].view(lay.Co, lay.kh * lay.kw, lay.Cip)[..., :lay.Ci].double().cpu()


@pytest.mark.parametrize("name,B", [("flownetc", 24), ("pwc", 48)])
def test_flownetc_batch24_and_pwc_batch48_train_step_graph_autotune_vs_oracle(name, B):
    """BASELINE configs[2] / [3] as bench.py runs them: RegistrationTrainer.step with autotuned launch shapes (768-way pixel splits, the
    vector-ALU cost-volume backward at B=48, the 256-pixel tiles) and hipGraph replay, fp32 parity mode, against the CPU oracle's
    training steps: the four loss scalars of three steps and, at the first step, the weight gradients of layers spread over the
    network (packed-domain gradient of the fused trainer vs torch autograd through the oracle), bounded by the oracle's own fp32
    noise measured against its float64 run (G1's form: rel L2 <= 5e-3 + 8 * noise; cosine >= 0.999).  The oracle's side of this
    (three fp32 training steps and a float64 pass: 3-5 minutes of CPU per model) is the committed fixture
    tests/golden/g10_train_parity_<name>_b<B>.npz, written by tests/golden/make_g10_train_parity.py from the same seeded weights
    and batch; gradients are compared on its 2048 sampled positions per layer."""
    import os
    import numpy as np
    import mireg
    from mireg.synth import make_pairs
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"g10_train_parity_{name}_b{B}.npz"))
    torch.manual_seed(1)
    x, _ = make_pairs(B, 256, seed=6)
    om = nets.OpticalFlowReg(name)
    nets.analytic_weights_(om)
    model = mireg.opticalFlowReg(name, precision="fp32")
    model.load_state_dict(om.state_dict())
    del om
    ref_losses = gold["losses"].tolist()

    tr = mireg.RegistrationTrainer(model.to(DEV), use_graph=True, autotune=True)
    xd = x.to(DEV)
    got_losses = [tr.step(xd).tolist()]
    torch.cuda.synchronize()
    eng = tr.eng
    pname = {id(p): k for k, p in model.named_parameters()}
    lays = [l for l in eng.layers.values() if l.wgrad_slab is not None and l.Co * l.Kf >= 4096 and f"{pname[id(l.weight)]}|idx" in gold]
    picks = [lays[i] for i in sorted({0, 1, len(lays) // 4, len(lays) // 2, (3 * len(lays)) // 4, len(lays) - 2, len(lays) - 1})]
    report = {}
    for lay in picks:
        k = pname[id(lay.weight)]
        a = _packed_grad_of(eng, tr.flat_g, lay).permute(0, 2, 1).reshape(-1)          # [Co][taps][Ci] -> torch layout, flat
        idx = torch.from_numpy(gold[f"{k}|idx"])
        a, b64, b32 = a[idx], torch.from_numpy(gold[f"{k}|g64"]), torch.from_numpy(gold[f"{k}|g32"])
        n64, nd = gold[f"{k}|norms"]
        noise = max(float(nd / n64), ((b32 - b64).norm() / b64.norm()).item())          # whole tensor and the sample
        rel = ((a - b64).norm() / b64.norm()).item()
        cos = (torch.dot(a, b64) / (a.norm() * b64.norm())).item()
        report[k] = (rel, noise, cos)
        assert rel <= min(5e-3 + 8 * noise, 5e-2) and cos >= 0.999, (k, rel, noise, cos)
    print(name, "gradient (rel L2 vs fp64 oracle, oracle-fp32 noise, cosine):", report)
    for _ in range(2):                                       # step 2 eager, step 3 captures the graphs and replays them
        got_losses.append(tr.step(xd).tolist())
    assert tr._graphs is not None and len(tr.eng.ws.tuned) > 10
    for a, b in zip(got_losses[0], ref_losses[0]):
        assert abs(a - b) <= 5e-5 * abs(b) + 1e-7, (got_losses[0], ref_losses[0])
    for k, tol in ((1, 5e-3), (2, 2e-2)):                    # later steps: Adam's +-lr moves of noise-level gradients (see the FlowNetS test above)
        for a, b in zip(got_losses[k], ref_losses[k]):
            assert abs(a - b) <= tol * abs(b) + 1e-6, (k, got_losses[k], ref_losses[k])


def test_flownets3d_full_width_128cubed_vs_oracle():
    """BASELINE configs[4] at its real width (64..1024 channels) on one 128^3 pair, fp32: training-mode forward (batch statistics),
    OFEloss3d and parameter gradients at the top, the middle and the bottom of the network against torch autograd on the CPU."""
    import mireg
    x = nets.analytic_input((1, 2, 128, 128, 128), seed=31)
    o = nets.OpticalFlowReg3d(1)
    nets.analytic_weights_(o)
    sd = {k: v.detach().clone() for k, v in o.state_dict().items()}
    o.train()
    flows_ref, warped_ref = o(x)
    vals_ref = oops.ofe_loss_3d(flows_ref, warped_ref, x[:, 0:1])
    vals_ref[3].backward()
    ref = {k: p.grad.detach().clone() for k, p in o.named_parameters()}
    m = mireg.opticalFlowReg3d(precision="fp32")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    xd = x.to(DEV)
    flows, warped = m(xd)
    vals = mireg.OFEloss3d(flows, warped, xd[:, 0:1])
    vals[3].backward()
    assert [tuple(f.shape) for f in flows] == [tuple(f.shape) for f in flows_ref]
    for i, (a, b) in enumerate(zip(flows, flows_ref)):
        err, scale = (a.detach().cpu() - b.detach()).abs().max().item(), b.detach().abs().max().item()
        assert err <= 2e-3 * max(1.0, scale), (i, err, scale)       # BatchNorm3d over one sample's voxels, 10 layers deep
    for a, b in zip(vals, vals_ref):
        assert abs(a.item() - b.item()) <= 2e-3 * abs(b.item()) + 1e-6
    # parameter gradients against the oracle's own float64 run, bounded by the oracle's measured fp32 noise (as fixture G1 does for
    # FlowNetS): rel L2 <= 5e-3 + 8 * noise_k, noise_k = |oracle fp32 - oracle fp64| / |oracle fp64| for that parameter.  The deepest
    # layers see 2^3 .. 4^3 voxels of ONE sample behind BatchNorm3d, so their fp32 noise is orders above the shallow layers'.
    import copy
    o64 = nets.OpticalFlowReg3d(1)
    o64.load_state_dict(sd)
    o64 = o64.double().train()
    f64, w64 = o64(x.double())
    oops.ofe_loss_3d(f64, w64, x[:, 0:1].double())[3].backward()
    ref64 = {k: p.grad.detach().clone() for k, p in o64.named_parameters()}
    P = dict(m.named_parameters())
    names = [k for k in ref if k.endswith("weight") and ref[k].dim() == 5]
    picks = [names[0], names[len(names) // 3], names[len(names) // 2], names[-4], names[-1]]
    report = {}
    for k in picks:
        a, b32, b64 = P[k].grad.double().flatten().cpu(), ref[k].double().flatten(), ref64[k].flatten()
        noise = ((b32 - b64).norm() / b64.norm()).item()
        rel = ((a - b64).norm() / b64.norm()).item()
        report[k] = (rel, noise)
        assert rel <= 5e-3 + 8 * noise, (k, rel, noise)
    print("FlowNetS-3D gradient rel L2 vs fp64 oracle (HIP, oracle-fp32 noise):", report)


def test_model_forward_with_segs_is_the_reference_4_tuple():
    """models.py:270-289: model(x, segs) -> (flows, warped, rounded warped segmentation in {0..3}, warped deformation grid), and
    models.py:195-204 generate_grid / utils.py:15-23 grid_generator, against the oracle."""
    import mireg
    from mireg.synth import make_pairs
    x, segs = make_pairs(3, 256, seed=8, magnitude=(0.5, 1.0))
    om = nets.OpticalFlowReg("flownets")
    nets.analytic_weights_(om)
    om.eval()
    with torch.no_grad():
        fref, wref, sref, gref = om(x, segs)
    m = mireg.opticalFlowReg("flownets", precision="fp32")
    m.load_state_dict(om.state_dict())
    m = m.to(DEV).eval()
    with torch.no_grad():
        flows, warped, wseg, wgrid = m(x.to(DEV), segs.to(DEV))
    assert len(flows) == 2 and len(warped) == 2 and wseg.shape == (3, 1, 256, 256) and wgrid.shape == (3, 1, 256, 256)
    for a, b in zip(flows, fref):
        assert (a.cpu() - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
    for a, b in zip(warped, wref):
        assert (a.cpu() - b).abs().max().item() <= 2e-4
    assert (wgrid.cpu() - gref).abs().max().item() <= 2e-3                      # lines are 0/1 steps: a 1e-4 flow error moves them by 1e-4 px
    assert wseg.dtype == torch.float32 and set(wseg.unique().cpu().tolist()) <= {0.0, 1.0, 2.0, 3.0}
    assert (wseg.cpu() != sref).float().mean().item() < 2e-4                    # rint() flips only where the bilinear value sits on .5
    g = mireg.generate_grid(2, 5, 7, torch.device(DEV))
    assert torch.equal(g.cpu(), oops.generate_grid(2, 5, 7))
    assert torch.equal(mireg.grid_generator(torch.device(DEV)).cpu(), oops.grid_generator())


def test_reference_layout_checkpoint_resumes_on_the_gpu(tmp_path):
    """train.py:150-156,183-188 / inference.py:147-148: a training_state.pt written the way the reference writes it (torch module
    + torch.optim.Adam on the CPU oracle, whose state_dict keys are the reference's) loads into the HIP model, reproduces the
    oracle's flows, and the trainer resumes from its Adam moments: the next step equals the oracle's next step."""
    import mireg
    from mireg import checkpoint
    from mireg.synth import make_pairs
    x, _ = make_pairs(4, 128, seed=5)
    om = nets.OpticalFlowReg("flownets")
    nets.analytic_weights_(om)
    om.train()
    opt = torch.optim.Adam(om.parameters(), 2e-4, betas=(0.9, 0.999), eps=1e-4)
    for _ in range(2):
        flows, warped, _, _ = om(x)
        loss = oops.ofe_loss(flows, warped, x[:, 0:1])[3]
        opt.zero_grad(); loss.backward(); opt.step()
    path = str(tmp_path / "training_state.pt")
    torch.save({"epoch": 7, "model_state_dict": om.state_dict(), "best_loss": 123.5, "optimizer_state_dict": opt.state_dict()}, path)
    # the oracle continues for one more step
    flows, warped, _, _ = om(x)
    vals_ref = oops.ofe_loss(flows, warped, x[:, 0:1])
    flows_ref = [f.detach().clone() for f in flows]
    p_before = torch.cat([p.detach().reshape(-1).clone() for p in om.parameters()])
    opt.zero_grad(); vals_ref[3].backward(); opt.step()
    delta_ref = torch.cat([p.detach().reshape(-1) for p in om.parameters()]) - p_before
    # HIP side
    m = mireg.opticalFlowReg("flownets", precision="fp32").to(DEV)
    tr = mireg.RegistrationTrainer(m, lr=1e-4, eps=1e-4, use_graph=False, autotune=False)
    epoch, best = checkpoint.load_training_state(path, m, trainer=tr)
    assert (epoch, best) == (8, 123.5) and tr.lr == 2e-4                          # optimizer hyper-parameters come from the file
    assert int(tr.step_dev.item()) == 2
    p0 = tr.flat_p.detach().cpu().clone()
    assert (p0 - p_before).abs().max().item() == 0.0
    losses = tr.step(x.to(DEV)).tolist()
    for a, b in zip(losses, vals_ref):
        assert abs(a - float(b)) <= 2e-4 * abs(float(b)) + 1e-6, (losses, [float(v) for v in vals_ref])
    for a, b in zip(_engine_flows(tr), flows_ref):
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
    delta = tr.flat_p.detach().cpu() - p0
    cos = torch.nn.functional.cosine_similarity(delta.double(), delta_ref.double(), dim=0).item()
    assert cos > 0.98, cos                                                         # third Adam step continues the oracle's trajectory
    assert abs(delta.abs().mean().item() / delta_ref.abs().mean().item() - 1.0) < 0.05
    # and the file the trainer writes back has the reference's layout
    out = str(tmp_path / "out.pt")
    checkpoint.save_training_state(out, m, tr.optimizer_state_dict(), epoch, best)
    ck = torch.load(out, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "best_loss", "optimizer_state_dict"}
    assert list(ck["model_state_dict"].keys()) == list(om.state_dict().keys())
    assert ck["optimizer_state_dict"]["param_groups"][0]["lr"] == 2e-4
