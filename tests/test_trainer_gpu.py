"""GPU: the fused training step == the drop-in autograd path == the CPU oracle (fp32 parity mode)."""
import pytest
import torch

from oracle import nets, ops as oops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(prec, B=2, size=64):
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    model = mireg.opticalFlowReg("flownets", precision=prec)
    nets.analytic_weights_(model)
    x, seg = make_pairs(B, size, seed=3)
    return model.to(DEV), x, seg


def test_fused_step_matches_autograd_and_oracle():
    import mireg
    model, x, _ = _setup("fp32", B=4, size=128)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # (a) CPU oracle: reference semantics end to end
    om = nets.OpticalFlowReg("flownets")
    om.load_state_dict(sd)
    om.train()
    opt = torch.optim.Adam(om.parameters(), 1e-4, betas=(0.9, 0.999), eps=1e-4)
    ref_losses = []
    for _ in range(2):
        flows, warped, _, _ = om(x)
        vals = oops.ofe_loss(flows, warped, x[:, 0:1])
        opt.zero_grad(); vals[3].backward(); opt.step()
        ref_losses.append([v.item() for v in vals])
    # (b) drop-in autograd path with torch's own Adam on the HIP model
    m2 = mireg.opticalFlowReg("flownets", precision="fp32").to(DEV)
    m2.load_state_dict(sd)
    m2.train()
    opt2 = torch.optim.Adam(m2.parameters(), 1e-4, betas=(0.9, 0.999), eps=1e-4)
    xd = x.to(DEV)
    auto_losses = []
    for _ in range(2):
        flows, warped, _, _ = m2(xd)
        vals = mireg.OFEloss(flows, warped, xd[:, 0:1].contiguous())
        opt2.zero_grad(); vals[3].backward(); opt2.step()
        auto_losses.append([v.item() for v in vals])
    # (c) fused trainer (no autograd, own Adam), eager then graph replay
    tr = mireg.RegistrationTrainer(model, use_graph=False, autotune=False)   # same launch shapes as the autograd path
    fused_losses = [tr.step(xd).tolist() for _ in range(2)]
    for a, b, c in zip(ref_losses, auto_losses, fused_losses):
        for va, vb, vc in zip(a, b, c):
            assert abs(vb - va) <= 2e-4 * abs(va) + 1e-6, (a, b)
            assert abs(vc - va) <= 2e-4 * abs(va) + 1e-6, (a, c)
    # parameters after two steps: fused == autograd path (same kernels, same order) and ~ oracle
    P2, PF = dict(m2.named_parameters()), dict(model.named_parameters())
    for k in ("conv1.0.weight", "conv6_1.0.weight", "predict_flow2.weight", "deconv3.0.weight", "conv3.1.bias"):
        k = "predictor." + k
        a, b = P2[k].detach(), PF[k].detach()
        # same kernels, different reduction grouping of the loss moments.  Adam moves a weight whose gradient is
        # noise-level (|g| ~ eps) by up to lr per step in either direction, so compare the update DIRECTION and
        # bound the element-wise gap by the two steps' worth of lr.
        da_, db_ = (a.cpu() - sd[k]).flatten().double(), (b.cpu() - sd[k]).flatten().double()
        assert torch.nn.functional.cosine_similarity(da_, db_, dim=0).item() > 0.999, k
        assert (a - b).abs().max().item() <= 2.5e-4, k
        o = dict(om.named_parameters())[k].detach()
        # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE direction
        delta_ref = o - sd[k]
        delta = b.cpu() - sd[k]
        cos = torch.nn.functional.cosine_similarity(delta.flatten().double(), delta_ref.flatten().double(), dim=0).item()
        assert cos > 0.98, (k, cos)


def test_graph_replay_equals_eager():
    import mireg
    model_a, x, _ = _setup("bf16", B=2, size=64)
    model_b, _, _ = _setup("bf16", B=2, size=64)
    xd = x.to(DEV)
    ta = mireg.RegistrationTrainer(model_a, use_graph=False, autotune=False)
    tb = mireg.RegistrationTrainer(model_b, use_graph=True, autotune=False)
    for i in range(5):
        la, lb = ta.step(xd).tolist(), tb.step(xd).tolist()
        assert all(abs(p - q) <= 1e-6 * abs(p) + 1e-9 for p, q in zip(la, lb)), (i, la, lb)
    assert tb._graph_fb is not None
    pa, pb = ta.flat_p, tb.flat_p
    assert (pa - pb).abs().max().item() <= 1e-6


def test_evaluate_dice_matches_oracle():
    import mireg
    model, x, seg = _setup("fp32", B=2, size=256)
    tr = mireg.RegistrationTrainer(model, use_graph=False)
    ev = tr.evaluate(x.to(DEV), seg.to(DEV))
    om = nets.OpticalFlowReg("flownets")
    om.load_state_dict(model.state_dict())
    om.eval()
    with torch.no_grad():
        flows, warped, wseg, wgrid = om(x, seg)
    want = [oops.dice_average(seg[j, 0], wseg[j, 0]) for j in range(2)]
    got = ev["dice"].cpu().tolist()
    assert all(abs(a - b) < 2e-3 for a, b in zip(got, want)), (got, want)
    vals = oops.ofe_loss(flows, warped, x[:, 0:1])
    assert abs(ev["loss"][3].item() - vals[3].item()) <= 1e-4 * abs(vals[3].item())
    # optimizer state round-trips in torch.optim.Adam's format
    sd = tr.optimizer_state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(list(model.parameters()))


def _dp_worker(rank, world, port, ret, name="flownets", size=64):
    """Two ranks share cuda:0 over gloo (RCCL refuses two ranks on one device): exercises the bucketed,
    phase-overlapped gradient exchange of the trainer, eager and under hipGraph replay."""
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mireg
    from mireg.synth import make_pairs
    torch.cuda.set_device(0)
    torch.manual_seed(1)
    model = mireg.opticalFlowReg(name, precision="fp32")
    nets.analytic_weights_(model)
    model = model.to(DEV)
    x, _ = make_pairs(4, size, seed=3)
    tr = mireg.RegistrationTrainer(model, use_graph=True, autotune=False)
    assert tr.world == 2
    xs = x[rank * 2:(rank + 1) * 2].to(DEV)
    for _ in range(4):
        tr.step(xs)
    torch.cuda.synchronize()
    assert tr._graphs is not None and len(tr._graphs) == len(tr.eng.phase_layers())
    torch.save(tr.flat_p.detach().cpu().clone(), os.path.join(ret, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_bucketed_overlap_on_one_gpu():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import tempfile, os
    tmp = tempfile.mkdtemp(prefix="mireg_dp_")           # results travel as files: no manager process to lose
    _dp_compare("flownets", 64, port, tmp)


def test_dp2_bucketed_overlap_flownetc_on_one_gpu():
    """Same exchange for FlowNetC: three buckets (decoder | conv6_1..conv4 | conv3_1, conv_redir, siamese conv3..conv1)."""
    import socket, tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    _dp_compare("flownetc", 128, port, tempfile.mkdtemp(prefix="mireg_dp_"))


def test_dp2_bucketed_overlap_pwc_on_one_gpu():
    """PWC-DC-Net buckets (dc_conv + level 2 | levels 3..6 | pyramid).  Its warp backward is the bucketed, ordered gather
    (mireg_pwc_warp_bwd_det): no fp32 scatter atomics are left on the path, so the comparison is as strict as FlowNetS's."""
    import socket, tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    _dp_compare("pwc", 128, port, tempfile.mkdtemp(prefix="mireg_dp_"))


def _dp_compare(name, size, port, tmp, strict=True):
    import os
    import torch.multiprocessing as mp
    mp.spawn(_dp_worker, args=(2, port, tmp, name, size), nprocs=2, join=True)
    ret = {r: torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)}
    assert torch.equal(ret[0], ret[1])                       # replicas stay identical
    # reference: same two half-batches, gradients averaged by hand, non-overlapped single process
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    x, _ = make_pairs(4, size, seed=3)
    flats = []
    model = mireg.opticalFlowReg(name, precision="fp32")
    nets.analytic_weights_(model)
    model = model.to(DEV)
    tr = mireg.RegistrationTrainer(model, use_graph=False, autotune=False, overlap_optimizer=False)
    # emulate DP: per step, grads of both halves summed, Adam scale 1/2  (BN stats per half, like per rank)
    tr._setup(x[:2].to(DEV))
    for _ in range(4):
        acc = None
        bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
        for r in range(2):
            # both "ranks" must see the same pre-step BN running stats
            model.load_state_dict(bn_state, strict=False)
            tr.x_static.copy_(x[r * 2:(r + 1) * 2].to(DEV))
            tr._fwd_bwd()
            acc = tr.flat_g.clone() if acc is None else acc + tr.flat_g
        tr.flat_g.copy_(acc)
        tr.grad_scale = 0.5                                # the 1/world of the summed all-reduce
        tr._optim()
        tr.grad_scale = 1.0
    if strict:
        diff = (tr.flat_p.cpu() - ret[0]).abs().max().item()
        assert diff < 5e-6, diff
    else:
        torch.manual_seed(1)
        m0 = mireg.opticalFlowReg(name, precision="fp32")
        nets.analytic_weights_(m0)
        p0 = torch.cat([q.detach().reshape(-1) for q in m0.parameters()])
        da, db = (tr.flat_p.cpu() - p0).double(), (ret[0] - p0).double()
        assert torch.nn.functional.cosine_similarity(da, db, dim=0).item() > 0.9


def _dp_sync_worker(rank, world, port, ret):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mireg
    from mireg.flownets import FlowNetS
    from mireg.synth import make_pairs
    torch.cuda.set_device(0)
    model = mireg.opticalFlowReg("flownets", precision="fp32")
    model.predictor = FlowNetS(batchNorm=False, precision="fp32")
    nets.analytic_weights_(model)
    model = model.to(DEV)
    x, _ = make_pairs(4, 64, seed=3)
    tr = mireg.RegistrationTrainer(model, use_graph=False, autotune=False, sync_loss_stats=True)
    assert tr.world == 2 and tr.sync_loss_stats and tr.grad_scale == 1.0
    losses1 = tr.step(x[rank * 2:(rank + 1) * 2].to(DEV)).tolist()
    torch.cuda.synchronize()
    m1 = tr.flat_m.detach().cpu().clone()                  # first moment after one step = (1 - beta1) * applied gradient
    losses = tr.step(x[rank * 2:(rank + 1) * 2].to(DEV)).tolist()
    torch.cuda.synchronize()
    torch.save((tr.flat_p.detach().cpu().clone(), losses, m1, losses1), os.path.join(ret, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_sync_loss_stats_is_the_single_process_step_on_the_concatenated_batch():
    """sync_loss_stats=True all-reduces the loss moments and normalises by the GLOBAL batch, so the summed gradient
    all-reduce is already the concatenated-batch gradient: Adam must not divide by world again (a BatchNorm-free predictor
    makes the two runs the same arithmetic; reference loss.py:55-62 whole-batch NCC)."""
    import os, socket, tempfile
    import torch.multiprocessing as mp
    import mireg
    from mireg.flownets import FlowNetS
    from mireg.synth import make_pairs
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tmp = tempfile.mkdtemp(prefix="mireg_dp_")
    # same kernels for B=2 per rank and B=4 in one process: the heads switch to the 1x1-GEMM form at 1024 pixels, which would
    # put another summation order (and, after one Adam step, +-lr differences) between the two runs
    from mireg import engine
    old_rows, old_env = engine.THIN_GEMM_ROWS, os.environ.get("MIREG_THIN_GEMM_ROWS")
    engine.THIN_GEMM_ROWS, os.environ["MIREG_THIN_GEMM_ROWS"] = 10 ** 9, str(10 ** 9)
    try:
        mp.spawn(_dp_sync_worker, args=(2, port, tmp), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2))
        assert torch.equal(r0[0], r1[0]) and r0[1] == r1[1]
        model = mireg.opticalFlowReg("flownets", precision="fp32")
        model.predictor = FlowNetS(batchNorm=False, precision="fp32")
        nets.analytic_weights_(model)
        model = model.to(DEV)
        p0 = torch.cat([q.detach().reshape(-1).cpu() for q in model.parameters()])
        x, _ = make_pairs(4, 64, seed=3)
        tr = mireg.RegistrationTrainer(model, use_graph=False, autotune=False)
        losses1 = tr.step(x.to(DEV)).tolist()
        losses = tr.step(x.to(DEV)).tolist()
    finally:
        engine.THIN_GEMM_ROWS = old_rows
        if old_env is None:
            os.environ.pop("MIREG_THIN_GEMM_ROWS", None)
        else:
            os.environ["MIREG_THIN_GEMM_ROWS"] = old_env
    # same weights, other launch shapes at B=2 vs B=4, f64 moments over another partition
    assert all(abs(a - b) <= 1e-6 * abs(b) + 1e-12 for a, b in zip(r0[3], losses1)), (r0[3], losses1)
    # after one Adam step the weights differ by +-lr wherever a gradient was at rounding level (see below)
    assert all(abs(a - b) <= 1e-4 * abs(b) + 1e-12 for a, b in zip(r0[1], losses)), (r0[1], losses)
    # parameters after two steps: Adam turns noise-level gradients (|g| ~ fp32 rounding of a two-half sum) into +-lr moves, so
    # compare the update direction, as the other trainer tests do
    da, db = (tr.flat_p.cpu() - p0).double(), (r0[0] - p0).double()
    assert da.abs().max().item() > 1e-4
    assert torch.nn.functional.cosine_similarity(da, db, dim=0).item() > 0.999


def test_fused_multiscale_tail_equals_per_scale_kernels():
    """mireg_tail_{resize,fwd,bwd} (one launch for all scales) == the per-scale entry points, op for op."""
    from mireg.trainer import FusedRegLoss
    from mireg.synth import make_pairs
    B, H, W = 3, 64, 64
    sizes = [(64, 64), (16, 16), (8, 8), (5, 7), (1, 1)]
    x, _ = make_pairs(B, H, seed=11)
    x = x.to(DEV).contiguous()
    g = torch.Generator().manual_seed(4)
    flows = [(torch.randn(B, 2, h, w, generator=g) * 1.5).to(DEV) for h, w in sizes]
    flows[1] = flows[1].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # NHWC-strided like the engine's
    outs = []
    for fused in (False, True):
        L = FusedRegLoss(B, H, W, sizes, torch.device(DEV))
        L.fused = fused
        L.forward(x, flows)
        out4 = L.finalize().clone()
        gf = [t.clone() for t in L.backward(flows)]
        outs.append((L.sums.sum(1).clone(), out4, gf, [w.clone() for w in L.warped]))
    (s0, o0, g0, w0), (s1, o1, g1, w1) = outs
    assert torch.allclose(s0, s1, rtol=1e-6, atol=1e-9)
    assert torch.allclose(o0, o1, rtol=1e-6)
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())


def test_autotuned_launch_shapes_keep_the_numbers():
    """The tuning pass only changes tile width / split-K per contraction site: same losses, BN statistics untouched."""
    import mireg
    model_a, x, _ = _setup("bf16", B=2, size=64)
    model_b, _, _ = _setup("bf16", B=2, size=64)
    xd = x.to(DEV)
    ta = mireg.RegistrationTrainer(model_a, use_graph=False, autotune=False)
    tb = mireg.RegistrationTrainer(model_b, use_graph=True, autotune=True)
    # only the first step is comparable: Adam moves noise-level-gradient weights by +-lr, so trajectories of two
    # different (equally valid) summation orders drift apart afterwards (they also do between two fp32 split-K shapes)
    la, lb = ta.step(xd).tolist(), tb.step(xd).tolist()
    assert all(abs(p - q) <= 5e-3 * abs(p) + 1e-6 for p, q in zip(la, lb)), (la, lb)   # bf16 operands
    sa, sb = model_a.state_dict(), model_b.state_dict()
    for k in sa:                                       # the discarded tuning pass left the BatchNorm statistics alone
        if "running" in k and ("conv1." in k or "conv2." in k):     # shallow layers: enough rows for a bf16 comparison
            assert torch.allclose(sa[k], sb[k], rtol=2e-2, atol=5e-3), k
    for _ in range(3):
        tb.step(xd)                                    # tuned shapes survive graph capture / replay
    assert tb._graphs is not None and len(tb.eng.ws.tuned) > 10


def _rccl_worker(rank, world, port, ret):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import mireg
    from mireg.synth import make_pairs
    assert dist.get_backend() == "nccl"
    torch.manual_seed(1)
    model = mireg.opticalFlowReg("flownets", precision="fp32")
    nets.analytic_weights_(model)
    model = model.to(dev)
    x, _ = make_pairs(4, 64, seed=3)
    tr = mireg.RegistrationTrainer(model, use_graph=True, autotune=False)
    xs = x[rank * 2:(rank + 1) * 2].to(dev)
    for _ in range(4):
        tr.step(xs)
    torch.cuda.synchronize()
    torch.save(tr.flat_p.detach().cpu().clone(), os.path.join(ret, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_dp2_over_rccl_on_two_gpus():
    """One process per GPU over the `nccl` backend (= RCCL on ROCm, xGMI between the devices): the bucketed, phase-overlapped
    gradient all-reduce under hipGraph replay keeps the replicas bit-identical and equals the hand-averaged single-process run
    (the same reference the gloo rehearsals on one GPU use).  Runs wherever a second GPU is visible, e.g. the driver's 8-GPU node."""
    import os, socket, tempfile
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tmp = tempfile.mkdtemp(prefix="mireg_rccl_")
    mp.spawn(_rccl_worker, args=(2, port, tmp), nprocs=2, join=True)
    ret = {r: torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)}
    assert torch.equal(ret[0], ret[1])
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(1)
    x, _ = make_pairs(4, 64, seed=3)
    model = mireg.opticalFlowReg("flownets", precision="fp32")
    nets.analytic_weights_(model)
    model = model.to(DEV)
    tr = mireg.RegistrationTrainer(model, use_graph=False, autotune=False, overlap_optimizer=False)
    tr._setup(x[:2].to(DEV))
    for _ in range(4):
        acc = None
        bn_state = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
        for r in range(2):
            model.load_state_dict(bn_state, strict=False)
            tr.x_static.copy_(x[r * 2:(r + 1) * 2].to(DEV))
            tr._fwd_bwd()
            acc = tr.flat_g.clone() if acc is None else acc + tr.flat_g
        tr.flat_g.copy_(acc)
        tr.grad_scale = 0.5
        tr._optim()
        tr.grad_scale = 1.0
    assert (tr.flat_p.cpu() - ret[0]).abs().max().item() < 5e-6


def test_batchnorm_num_batches_tracked_follows_torch():
    """BatchNorm2d.num_batches_tracked in a state_dict (train.py:183-201 checkpoints it): one per training forward per module
    call, through the fused trainer (eager warm-up, capture, hipGraph replays) and through the autograd path."""
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(0)
    m = mireg.opticalFlowReg("flownets", precision="bf16").to(DEV)
    x = make_pairs(2, 64, seed=1)[0].to(DEV)
    tr = mireg.RegistrationTrainer(m, use_graph=True, autotune=False)
    for _ in range(5):
        tr.step(x)
    sd = m.state_dict()
    assert all(int(v) == 5 for k, v in sd.items() if k.endswith("num_batches_tracked")), {k: int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")}
    m.train()
    flows, warped, _, _ = m(x)
    with torch.no_grad():
        m.eval()
        m(x)
    assert all(int(v) == 6 for k, v in m.state_dict().items() if k.endswith("num_batches_tracked"))


@pytest.mark.parametrize("name,batch", [("pwc", 4), ("flownetc", 4)])
def test_backward_is_run_to_run_deterministic_with_biased_two_channel_layers(name, batch):
    """Same weights, same batch, forward + backward eight times: the packed gradient (conv weights, biases, BatchNorm) is bit-identical
    every time.  PWC / FlowNetC carry biases on their 2-channel heads and upsamplers (PWCNet.py:31-34, submodules.py:32-38); their
    bias gradients are column sums over up to 256x256xB rows, which used to go through fp32 atomics."""
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(3)
    m = mireg.opticalFlowReg(name, precision="bf16")
    nets.analytic_weights_(m)
    tr = mireg.RegistrationTrainer(m.to(DEV), use_graph=False, autotune=False, overlap_optimizer=False)
    x = make_pairs(batch, 256, seed=2)[0].to(DEV)
    tr.step(x)
    torch.cuda.synchronize()
    ref = None
    for it in range(8):
        tr._fwd_bwd()
        torch.cuda.synchronize()
        g = tr.flat_g.detach().clone()
        if ref is None:
            ref = g
        assert torch.equal(g, ref), (it, int((g != ref).sum()))


def test_bench_self_launch_two_ranks_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it (what a driver that calls it like `--gpus 1` does): bench.py starts its
    own two rank processes before touching the GPU, both train (gloo rehearsal: two ranks share this card, the exchange is the same
    bucketed all-reduce RCCL carries on a multi-GPU node), rank 0 prints ONE contract line with dist.world == 2 and a whole-job value."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MIREG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "3", "--batch", "4",
                        "--size", "64", "--no-3d", "--no-other-models", "--no-cpu-baseline", "--no-autotune"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dist"] == {"world": 2, "backend": "gloo"} and d["n_gpus"] == 2 and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"] and d["scaling"] == "weak"


def test_batchnorm_counter_survives_a_batch_size_change():
    """The engine is rebuilt when the batch shape changes (the short last batch of an epoch); the training forwards its BatchNorms
    had counted must reach num_batches_tracked all the same (train.py:183-201 checkpoints it)."""
    import mireg
    from mireg.synth import make_pairs
    torch.manual_seed(0)
    m = mireg.opticalFlowReg("flownets", precision="bf16").to(DEV)
    tr = mireg.RegistrationTrainer(m, use_graph=False, autotune=False)
    x4, x2 = make_pairs(4, 64, seed=1)[0].to(DEV), make_pairs(2, 64, seed=2)[0].to(DEV)
    for _ in range(3):
        tr.step(x4)
    for _ in range(2):
        tr.step(x2)
    tr.step(x4)
    counts = {k: int(v) for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")}
    assert counts and all(v == 6 for v in counts.values()), counts
