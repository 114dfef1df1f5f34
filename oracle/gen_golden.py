#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU.

Run in the build container only (needs /root/reference; the GPU box never sees
it):   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

What it does
  1. imports the reference modules from /root/reference without editing them.
     Imports of third-party packages that are absent from this image and are
     NOT exercised on the hot path (monai, torchvision, skimage, torchmetrics,
     cv2, RAFT, tensorboard...) are satisfied with empty placeholder modules.
     The external correlation_package has no source in the reference tree: the
     oracle's published-definition Correlation is injected in its place, so the
     PWC / FlowNetC fixtures pin everything AROUND the correlation arithmetic
     (convs, warps, concats, scales) but not that arithmetic itself
     ("parity unpinned" for K7/K8, see oracle/__init__.py).  The same holds for
     resample2d_package / channelnorm_package in the FlowNet2 fixture (G9): the
     reference's FlowNet2 / FlowNetSD / FlowNetFusion / FlowNetS classes run with
     the oracle's Resample2d / ChannelNorm / Correlation in place of the three
     external layers.
  2. gives reference and oracle models the same analytic weights and inputs,
     asserts the oracle restatement equals the reference (<= 1e-5 abs / rel),
  3. writes inputs-by-formula + expected outputs as small fixtures.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import nets, ops  # noqa: E402


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


def import_reference():
    sys.path.insert(0, REF)
    _placeholder("monai")
    _placeholder("monai.data", MetaTensor=object)
    _placeholder("torchvision")
    _placeholder("torchvision.transforms")
    sk = _placeholder("skimage", transform=None)
    sk.transform = _placeholder("skimage.transform")
    _placeholder("skimage.measure", find_contours=None)
    _placeholder("skimage.metrics", structural_similarity=None)
    _placeholder("torchmetrics")
    _placeholder("torchmetrics.functional")
    _placeholder("torchmetrics.functional.clustering", mutual_info_score=None)
    _placeholder("torchmetrics.functional.regression", pearson_corrcoef=None)
    _placeholder("cv2")
    _placeholder("RAFT")
    _placeholder("RAFT.core")
    _placeholder("RAFT.core.raft", RAFT=None)
    corr_mod = dict(Correlation=nets.Correlation)
    _placeholder("correlation_package")
    _placeholder("correlation_package.correlation", **corr_mod)
    for pkg in ("flownet2.networks.correlation_package", "flownet2.networks.resample2d_package",
                "flownet2.networks.channelnorm_package"):
        _placeholder(pkg)
    _placeholder("flownet2.networks.correlation_package.correlation", **corr_mod)
    _placeholder("flownet2.networks.resample2d_package.resample2d", Resample2d=nets.Resample2d)
    _placeholder("flownet2.networks.channelnorm_package.channelnorm", ChannelNorm=nets.ChannelNorm)
    ref = types.SimpleNamespace()
    ref.FlowNetS = importlib.import_module("FlowNetS.FlowNetS").FlowNetS
    ref.loss = importlib.import_module("loss")
    ref.loss.device = torch.device("cpu")
    ref.utils = importlib.import_module("utils")
    ref.models = importlib.import_module("models")
    ref.models.device = torch.device("cpu")
    ref.FlowNetC = importlib.import_module("flownet2.networks.FlowNetC").FlowNetC
    ref.pwc = importlib.import_module("PWC.models.PWCNet")
    ref.flownet2 = importlib.import_module("flownet2.models")
    return ref


def close(a, b, tol=1e-5, what=""):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    assert err <= tol * max(1.0, scale), f"{what}: restatement differs from reference: {err} (scale {scale})"
    return err


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {os.path.relpath(path, REPO)} ({os.path.getsize(path) / 1024:.0f} KiB)")


GRAD_KEYS = ("conv1.0.weight", "conv2.0.weight", "conv3_1.1.weight", "conv3_1.1.bias", "conv4_1.0.weight",
             "conv5_1.0.weight", "conv6_1.0.weight", "deconv5.0.weight", "deconv2.0.weight", "predict_flow6.weight",
             "predict_flow2.weight", "upsampled_flow6_to_5.weight", "upsampled_flow3_to_2.weight")


def rand_flow(shape, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * sigma


# ---------------------------------------------------------------------------
def g1_flownets(ref):
    print("G1 FlowNetS (reference FlowNetS/FlowNetS.py)")
    for tag, shape in (("c1_4x64", (4, 2, 64, 64)), ("c2_2x256", (2, 2, 256, 256))):
        r = ref.FlowNetS(batchNorm=True)
        o = nets.FlowNetS(batchNorm=True)
        nets.analytic_weights_(r)
        o.load_state_dict(r.state_dict())
        x = nets.analytic_input(shape, seed=3)
        out = {}
        for mode in ("train", "eval"):
            r.train(mode == "train")
            o.train(mode == "train")
            xr = x.clone().requires_grad_(mode == "train")
            fr = r(xr)
            fo = o(x)
            for i, (a, b) in enumerate(zip(fo, fr)):
                close(a, b.detach(), 2e-5, f"FlowNetS {tag} {mode} flow[{i}]")
                arr = b.detach().numpy()
                if arr.shape[-1] == 256:  # keep the fixture small: strided sample + moments
                    out[f"{mode}_flow{i}_s8"] = arr[:, :, ::8, ::8]
                    out[f"{mode}_flow{i}_sum"] = arr.astype(np.float64).sum()
                    out[f"{mode}_flow{i}_abssum"] = np.abs(arr.astype(np.float64)).sum()
                else:
                    out[f"{mode}_flow{i}"] = arr
            if mode == "train":
                # fp32 rounding noise of the REFERENCE itself: distance of its fp32 flows from the same
                # module evaluated in float64 (deep BatchNorms over a handful of samples are ill-conditioned,
                # so parity bounds are stated as 1e-4*scale + 4*noise)
                import copy
                r64 = copy.deepcopy(r).double()
                r64.load_state_dict({k: v.double() for k, v in o.state_dict().items()})  # pre-forward weights/stats
                r64.train()
                f64 = r64(x.double())
                for i, (a, b) in enumerate(zip(fr, f64)):
                    out[f"noise_flow{i}"] = (a.detach().double() - b.detach()).abs().max().item()
                # scalar objective with analytic per-scale cotangents; the float64 twin gives the reference's
                # own gradient rounding noise (relative L2), which bounds how tightly anything can match it
                def objective(fl, dt):
                    return sum((f * torch.cos(torch.arange(f.numel(), dtype=dt).reshape(f.shape) * 0.01)).sum()
                               for f in fl)
                objective(fr, torch.float32).backward()
                r64.zero_grad()
                objective(r64(x.double()), torch.float64).backward()
                out["grad_x_s4"] = xr.grad.numpy()[:, :, ::4, ::4]
                P, P64 = dict(r.named_parameters()), dict(r64.named_parameters())
                for k in GRAD_KEYS:
                    g = P[k].grad
                    out["gradnorm_" + k] = g.double().norm().item()
                    out["gradnoise_" + k] = ((g.double() - P64[k].grad).norm() / P64[k].grad.norm()).item()
                    out["grad_" + k] = g.flatten()[:16384].numpy()
                out["bn_running_mean_conv2"] = r.conv2[1].running_mean.numpy()
                out["bn_running_var_conv2"] = r.conv2[1].running_var.numpy()
        save("g1_flownets_" + tag, shape=np.array(shape), **out)


def g2_stn(ref):
    print("G2 stn (reference models.py:256-268)")
    out = {}
    frame = nets.analytic_input((3, 1, 256, 256), seed=5)
    for h in (256, 64, 32, 16, 8, 4, 1):
        for sigma in (0.5, 3.0, 20.0):
            flow = rand_flow((3, 2, h, h), sigma, seed=h * 7 + int(sigma))
            want = ref.models.opticalFlowReg.stn(None, flow, frame)
            got = ops.stn(flow, frame)
            close(got, want, 2e-5, f"stn h={h} sigma={sigma}")
            key = f"h{h}_s{sigma}"
            out["flow_" + key] = flow.numpy() if h <= 64 else flow.numpy()[:, :, ::4, ::4]
            out["warp_" + key] = want.numpy() if h <= 64 else want.numpy()[:, :, ::4, ::4]
    # rectangular + 64x64 frame upsampled to 256 (config 1 behaviour)
    frame64 = nets.analytic_input((2, 1, 64, 64), seed=9)
    flow = rand_flow((2, 2, 256, 256), 2.0, seed=77)
    want = ref.models.opticalFlowReg.stn(None, flow, frame64)
    close(ops.stn(flow, frame64), want, 2e-5, "stn 64->256")
    out["warp_up64_s4"] = want.numpy()[:, :, ::4, ::4]
    g = ref.models.generate_grid(2, 5, 7, torch.device("cpu"))
    close(ops.generate_grid(2, 5, 7), g, 0, "generate_grid")
    out["generate_grid_2x5x7"] = g.numpy()
    save("g2_stn", **out)


def g3_losses(ref):
    print("G3 losses (reference loss.py)")
    L = ref.loss
    out = {}
    fixed = nets.analytic_input((3, 1, 256, 256), seed=11)
    moving = nets.analytic_input((3, 1, 256, 256), seed=12)
    for n, sizes in ((2, (256, 64)), (6, (256, 64, 32, 16, 8, 4)), (7, (256, 128, 64, 32, 16, 8, 4))):
        flows = [rand_flow((3, 2, s, s), 1.5, seed=100 + s).requires_grad_() for s in sizes]
        warped = [ops.stn(f.detach(), moving).requires_grad_() for f in flows]
        p, c, s, tot = L.OFEloss(flows, warped, fixed)
        po, co, so, to = ops.ofe_loss([f.detach() for f in flows], [w.detach() for w in warped], fixed)
        for a, b, nm in ((po, p, "p"), (co, c, "c"), (so, s, "s"), (to, tot, "total")):
            close(a, b.detach(), 1e-5, f"OFEloss n={n} {nm}")
        assert tot.dtype == torch.float64
        tot.backward()
        out[f"n{n}_values"] = np.array([p.item(), c.item(), s.item(), tot.item()])
        for i, sz in enumerate(sizes):
            gf, gw = flows[i].grad.numpy(), warped[i].grad.numpy()
            st = 4 if sz > 64 else 1
            out[f"n{n}_gflow{i}"] = gf[:, :, ::st, ::st]
            out[f"n{n}_gwarp{i}"] = gw[:, :, ::st, ::st]
    # single terms incl. the degenerate-NCC guard
    w = ops.stn(rand_flow((3, 2, 64, 64), 2.0, seed=5), moving)
    out["photo_64"] = L.photometric_loss(fixed, w).item()
    out["ncc_64"] = L.correlation_loss(fixed, w).item()
    out["smooth_64"] = L.smoothness_loss(rand_flow((3, 2, 64, 64), 2.0, seed=5)).item()
    close(ops.photometric_loss(fixed, w), out["photo_64"], 1e-5, "photometric")
    close(ops.correlation_loss(fixed, w), out["ncc_64"], 1e-5, "ncc")
    close(ops.smoothness_loss(rand_flow((3, 2, 64, 64), 2.0, seed=5)), out["smooth_64"], 1e-5, "smooth")
    const = torch.full((3, 1, 64, 64), 0.25)
    out["ncc_const"] = float(L.correlation_loss(fixed, const))
    close(ops.correlation_loss(fixed, const), out["ncc_const"], 0, "ncc guard")
    # 3-D losses
    f3 = nets.analytic_input((2, 1, 8, 10, 6), seed=1)
    w3 = nets.analytic_input((2, 1, 8, 10, 6), seed=2)
    p3, c3, t3 = L.Affloss(w3, f3)
    po, co, to = ops.aff_loss(w3, f3)
    close(po, p3, 1e-6, "Affloss p"); close(co, c3, 1e-6, "Affloss c")
    out["affloss"] = np.array([p3.item(), c3.item(), t3.item()])
    save("g3_losses", **out)


def make_labels(B, seed):
    img = nets.analytic_input((B, 1, 256, 256), seed=seed)
    return torch.bucketize(img, torch.tensor([0.3, 0.5, 0.7])).to(torch.float32)


def g4_dice(ref):
    print("G4 dice / seg-round (reference utils.py:72-91, models.py:278,286)")
    out = {}
    seg_f = make_labels(2, 21)
    seg_m = make_labels(2, 22)
    flow = rand_flow((2, 2, 256, 256), 2.0, seed=4)
    ws = ref.models.opticalFlowReg.stn(None, flow, seg_m)
    ws_int = torch.from_numpy(np.clip(np.rint(ws.numpy()), 0, 3))
    assert torch.equal(ops.seg_round(ops.stn(flow, seg_m)), ws_int) or \
        (ops.seg_round(ops.stn(flow, seg_m)) != ws_int).float().mean() < 1e-4
    d = [float(ref.utils.dice_average(seg_f[j, 0], ws_int[j, 0])) for j in range(2)]
    do = [ops.dice_average(seg_f[j, 0], ws_int[j, 0]) for j in range(2)]
    close(do, d, 1e-6, "dice")
    out["flow_s4"] = flow.numpy()[:, :, ::4, ::4]
    out["dice"] = np.array(d)
    out["ws_int_hist"] = np.array([(ws_int == k).sum().item() for k in range(4)])
    close(ops.grid_generator(), ref.utils.grid_generator(), 0, "grid_generator")
    out["grid_generator_rowsum"] = ref.utils.grid_generator().sum(1).numpy()
    save("g4_dice", **out)


def g5_skeletons(ref):
    print("G5 PWC / FlowNetC skeletons (reference convs + injected oracle Correlation)")
    torch.Tensor.cuda = lambda self, *a, **k: self  # PWCNet.py:169 hard-codes .cuda()
    out = {}
    r = ref.pwc.PWCDCNet(md=4)
    o = nets.PWCDCNet(md=4)
    nets.analytic_weights_(r)
    o.load_state_dict(r.state_dict())
    x = nets.analytic_input((1, 2, 256, 256), seed=8)
    r.eval(); o.eval()
    with torch.no_grad():
        fr, fo = r(x), o(x)
    for i, (a, b) in enumerate(zip(fo, fr)):
        close(a, b, 5e-5, f"PWC flow{i}")
        out[f"pwc_flow{i}"] = b.numpy() if b.shape[-1] <= 64 else b.numpy()[:, :, ::4, ::4]
    args = types.SimpleNamespace(fp16=False)
    r = ref.FlowNetC(args, batchNorm=True)
    o = nets.FlowNetC(args, batchNorm=True)
    nets.analytic_weights_(r)
    o.load_state_dict(r.state_dict(), strict=False)
    x = nets.analytic_input((2, 2, 256, 256), seed=9)
    for mode in ("train", "eval"):
        r.train(mode == "train"); o.train(mode == "train")
        with torch.no_grad():
            fr, fo = r(x), o(x)
        assert len(fr) == len(fo)
        for i, (a, b) in enumerate(zip(fo, fr)):
            close(a, b, 5e-5, f"FlowNetC {mode} flow{i}")
            out[f"flownetc_{mode}_flow{i}"] = b.numpy()
    save("g5_skeletons", **out)


def g6_pwc_warp(ref):
    print("G6 PWCDCNet.warp (reference PWC/models/PWCNet.py:143-179)")
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    net = ref.pwc.PWCDCNet(md=4)
    for (C, H), sigma in (((128, 8), 0.7), ((64, 32), 2.0), ((32, 64), 6.0)):
        x = nets.analytic_input((2, C, H, H), seed=C)
        flo = rand_flow((2, 2, H, H), sigma, seed=C + H)
        with torch.no_grad():
            want = net.warp(x, flo)
        close(ops.pwc_warp(x, flo), want, 2e-5, f"pwc warp {C}x{H}")
        out[f"flo_{C}_{H}"] = flo.numpy()
        out[f"warp_{C}_{H}"] = want.numpy()[:, ::8]
    # threshold edge: integer shifts land exactly on taps -> mask must stay 1 inside
    x = nets.analytic_input((1, 4, 8, 8), seed=1)
    flo = torch.zeros(1, 2, 8, 8)
    with torch.no_grad():
        want = net.warp(x, flo)
    close(ops.pwc_warp(x, flo), want, 2e-5, "pwc warp zero flow")
    out["warp_zero"] = want.numpy()
    save("g6_pwc_warp", **out)


def g7_affine3d(ref):
    print("G7 3-D affine grid + trilinear sample (models.py:187-188, torch defaults)")
    import torch.nn.functional as F
    vol = nets.analytic_input((2, 1, 6, 9, 7), seed=2)
    theta = torch.tensor([[[1.05, 0.02, -0.03, 0.1], [0.04, 0.95, 0.01, -0.05], [0.0, 0.03, 1.1, 0.02]],
                          [[0.9, -0.1, 0.0, 0.0], [0.1, 0.9, 0.05, 0.2], [0.02, 0.0, 1.0, -0.3]]])
    want = F.grid_sample(vol, F.affine_grid(theta, vol.size()))
    close(ops.affine_grid_sample_3d(vol, theta), want, 1e-5, "affine3d")
    save("g7_affine3d", theta=theta.numpy(), warped=want.numpy())


def g8_adam(ref):
    print("G8 one Adam step as configured at train.py:129")
    p = [nets.analytic_input((5, 7), seed=1) - 0.5, nets.analytic_input((11,), seed=2) - 0.5]
    g = [nets.analytic_input((5, 7), seed=3) * 1e-3 - 5e-4, nets.analytic_input((11,), seed=4) - 0.5]
    tp = [t.clone().requires_grad_() for t in p]
    opt = torch.optim.Adam(tp, 1e-4, betas=(0.9, 0.999), eps=1e-4)
    mine = [t.clone() for t in p]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    for step in (1, 2, 3):
        for t, gg in zip(tp, g):
            t.grad = gg.clone() * step
        opt.step()
        ops.adam_step(mine, [gg * step for gg in g], m, v, step)
    for a, b in zip(mine, tp):
        close(a, b.detach(), 1e-6, "adam")
    save("g8_adam", p0=tp[0].detach().numpy(), p1=tp[1].detach().numpy())


def g9_flownet2(ref):
    print("G9 FlowNet2 stack (reference flownet2/models.py:30-191 + networks/FlowNetS|SD|Fusion.py, oracle Correlation / "
          "Resample2d / ChannelNorm injected for the absent custom layers)")
    args = types.SimpleNamespace(fp16=False, rgb_max=255.0)
    out = {}
    # the two new sub-networks on their own, train-mode BatchNorm and eval
    for cls_name, shape, seed in (("FlowNetSD", (2, 2, 64, 64), 21), ("FlowNetFusion", (2, 9, 64, 64), 22), ("FlowNetS", (2, 6, 64, 64), 23)):
        mod = importlib.import_module(f"flownet2.networks.{cls_name}")
        r = getattr(mod, cls_name)(args, batchNorm=True)
        o = {"FlowNetSD": nets.FlowNetSD, "FlowNetFusion": nets.FlowNetFusion, "FlowNetS": nets.FlowNet2S}[cls_name](args, batchNorm=True)
        nets.analytic_weights_(r)
        o.load_state_dict(r.state_dict())
        x = nets.analytic_input(shape, seed=seed, lo=-1.0, hi=1.0)
        for mode in ("train", "eval"):
            r.train(mode == "train"); o.train(mode == "train")
            with torch.no_grad():
                fr, fo = r(x), o(x)
            fr, fo = (fr, fo) if isinstance(fr, tuple) else ((fr,), (fo,))
            assert len(fr) == len(fo)
            for i, (a, b) in enumerate(zip(fo, fr)):
                close(a, b, 5e-5, f"{cls_name} {mode} out{i}")
                out[f"{cls_name.lower()}_{mode}_{i}"] = b.numpy()
    # the whole stack as models.py:225 builds it
    r = ref.flownet2.FlowNet2(args, batchNorm=True)
    o = nets.FlowNet2(args, batchNorm=True)
    nets.analytic_weights_(r)
    o.load_state_dict(r.state_dict())
    x = nets.analytic_input((1, 2, 256, 256), seed=24)
    r.eval(); o.eval()
    with torch.no_grad():
        fr = r(x)
        st = o.stages(x)
    assert len(fr) == 2 and torch.equal(fr[0], fr[1])
    close(st[-1], fr[0], 5e-4, "FlowNet2 fused flow")
    for name, t in zip(("flownetc_flow2", "flownets1_flow2", "flownets2_flow2", "flownetsd_flow2"), st[:4]):
        out[f"flownet2_{name}"] = t.numpy()
    out["flownet2_fused"] = fr[0].numpy()[:, :, ::2, ::2]
    save("g9_flownet2", **out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    only = set(sys.argv[1:])
    for fn in (g1_flownets, g2_stn, g3_losses, g4_dice, g5_skeletons, g6_pwc_warp, g7_affine3d, g8_adam, g9_flownet2):
        if not only or fn.__name__.split("_")[0] in only:
            fn(ref)
    print("all restatement-vs-reference checks passed")


if __name__ == "__main__":
    main()
