"""CPU: the oracle restatement reproduces the reference-generated fixtures.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py from the
reference's own Python (see that script).  These tests re-derive every input by
formula and check the oracle against the stored reference outputs, so the
oracle stays pinned on machines where /root/reference does not exist.
"""
import numpy as np
import torch

from oracle import nets, ops

TOL = 2e-5


def rand_flow(shape, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * sigma


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_flownets_config1(golden):
    g = golden("g1_flownets_c1_4x64")
    m = nets.FlowNetS(batchNorm=True)
    nets.analytic_weights_(m)
    x = nets.analytic_input(tuple(g["shape"]), seed=3)
    m.train()
    out = m(x)
    assert [tuple(o.shape) for o in out] == [(4, 2, 256, 256), (4, 2, 16, 16), (4, 2, 8, 8), (4, 2, 4, 4),
                                              (4, 2, 2, 2), (4, 2, 1, 1)]
    _close(out[0].detach()[:, :, ::8, ::8], g["train_flow0_s8"])
    _close(out[0].detach().double().sum(), g["train_flow0_sum"], 1e-4)
    for i in range(1, 6):
        _close(out[i].detach(), g[f"train_flow{i}"])
    _close(m.conv2[1].running_mean, g["bn_running_mean_conv2"])
    m.eval()
    out = m(x)
    assert len(out) == 2
    _close(out[1].detach(), g["eval_flow1"])


def test_stn_all_scales(golden):
    g = golden("g2_stn")
    frame = nets.analytic_input((3, 1, 256, 256), seed=5)
    for h in (256, 64, 32, 16, 8, 4, 1):
        for sigma in (0.5, 3.0, 20.0):
            flow = rand_flow((3, 2, h, h), sigma, seed=h * 7 + int(sigma))
            got = ops.stn(flow, frame)
            want = g[f"warp_h{h}_s{sigma}"]
            if h > 64:
                got = got[:, :, ::4, ::4]
            _close(got, want)
    _close(ops.generate_grid(2, 5, 7), g["generate_grid_2x5x7"], 0)


def test_losses(golden):
    g = golden("g3_losses")
    fixed = nets.analytic_input((3, 1, 256, 256), seed=11)
    moving = nets.analytic_input((3, 1, 256, 256), seed=12)
    for n, sizes in ((2, (256, 64)), (6, (256, 64, 32, 16, 8, 4))):
        flows = [rand_flow((3, 2, s, s), 1.5, seed=100 + s) for s in sizes]
        warped = [ops.stn(f, moving) for f in flows]
        vals = ops.ofe_loss(flows, warped, fixed)
        assert vals[3].dtype == torch.float64
        _close([v.item() for v in vals], g[f"n{n}_values"], 1e-5)
    _close(ops.correlation_loss(fixed, torch.full((3, 1, 64, 64), 0.25)).item(), g["ncc_const"], 0)


def test_dice(golden):
    g = golden("g4_dice")
    from oracle.gen_golden import make_labels
    seg_f, seg_m = make_labels(2, 21), make_labels(2, 22)
    flow = rand_flow((2, 2, 256, 256), 2.0, seed=4)
    ws = ops.seg_round(ops.stn(flow, seg_m))
    hist = np.array([(ws == k).sum().item() for k in range(4)])
    assert np.abs(hist - g["ws_int_hist"]).max() <= 4
    d = [ops.dice_average(seg_f[j, 0], ws[j, 0]) for j in range(2)]
    _close(d, g["dice"], 1e-3)


def test_pwc_warp(golden):
    g = golden("g6_pwc_warp")
    for (C, H), sigma in (((128, 8), 0.7), ((64, 32), 2.0), ((32, 64), 6.0)):
        x = nets.analytic_input((2, C, H, H), seed=C)
        flo = torch.from_numpy(g[f"flo_{C}_{H}"])
        _close(ops.pwc_warp(x, flo)[:, ::8], g[f"warp_{C}_{H}"])


def test_correlation_closed_form():
    """K7/K8 are parity-unpinned third-party ops: check the oracle against the
    independent closed form mean_c(f1 * shift(f2)) written with torch.roll-free
    slicing, at both parameterisations the reference uses."""
    for (md, s2, C, H) in ((20, 2, 8, 12), (4, 1, 6, 9)):
        f1 = nets.analytic_input((2, C, H, H), seed=1) - 0.5
        f2 = nets.analytic_input((2, C, H, H), seed=2) - 0.5
        out = ops.correlation(f1, f2, md, 1, md, 1, s2, 1)
        R = md // s2
        D = 2 * R + 1
        assert out.shape == (2, D * D, H, H)
        for (dy, dx) in ((0, 0), (-R, R), (1, -2), (R, R)):
            want = torch.zeros(2, H, H)
            for y in range(H):
                for x in range(H):
                    yy, xx = y + s2 * dy, x + s2 * dx
                    if 0 <= yy < H and 0 <= xx < H:
                        want[:, y, x] = (f1[:, :, y, x] * f2[:, :, yy, xx]).mean(1)
            _close(out[:, (dy + R) * D + (dx + R)], want, 1e-6)


def test_skeleton_fixtures_shapes(golden):
    g = golden("g5_skeletons")
    assert g["pwc_flow6"].shape == (1, 2, 4, 4)
    assert g["flownetc_train_flow0"].shape == (2, 2, 64, 64)
    assert "flownetc_eval_flow1" not in g.files  # eval returns (flow2,) only


def test_affine3d_and_adam(golden):
    g = golden("g7_affine3d")
    vol = nets.analytic_input((2, 1, 6, 9, 7), seed=2)
    _close(ops.affine_grid_sample_3d(vol, torch.from_numpy(g["theta"])), g["warped"], 1e-5)
    g = golden("g8_adam")
    p = [nets.analytic_input((5, 7), seed=1) - 0.5, nets.analytic_input((11,), seed=2) - 0.5]
    gr = [nets.analytic_input((5, 7), seed=3) * 1e-3 - 5e-4, nets.analytic_input((11,), seed=4) - 0.5]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    for step in (1, 2, 3):
        ops.adam_step(p, [x * step for x in gr], m, v, step)
    _close(p[0], g["p0"], 1e-6)
    _close(p[1], g["p1"], 1e-6)


def test_flownet2_stack(golden):
    """G9: the FlowNet2 stack of flownet2/models.py:30-191 (reference classes run with the oracle's Correlation / Resample2d /
    ChannelNorm injected for the absent custom layers): the two new sub-networks and the 6-channel FlowNetS in train and eval
    mode, and the whole chain's intermediate and fused flows."""
    g = golden("g9_flownet2")
    for cls, key, shape, seed in ((nets.FlowNetSD, "flownetsd", (2, 2, 64, 64), 21), (nets.FlowNetFusion, "flownetfusion", (2, 9, 64, 64), 22),
                                  (nets.FlowNet2S, "flownets", (2, 6, 64, 64), 23)):
        m = cls(None, batchNorm=True)
        nets.analytic_weights_(m)
        x = nets.analytic_input(shape, seed=seed, lo=-1.0, hi=1.0)
        for mode in ("train", "eval"):
            m.train(mode == "train")
            with torch.no_grad():
                out = m(x)
            out = out if isinstance(out, tuple) else (out,)
            assert len(out) == sum(k.startswith(f"{key}_{mode}_") for k in g.files)
            for i, o in enumerate(out):
                _close(o, g[f"{key}_{mode}_{i}"], 5e-5)
    m = nets.FlowNet2(None, batchNorm=True)
    nets.analytic_weights_(m)
    m.eval()
    with torch.no_grad():
        st = m.stages(nets.analytic_input((1, 2, 256, 256), seed=24))
    for name, t in zip(("flownetc_flow2", "flownets1_flow2", "flownets2_flow2", "flownetsd_flow2"), st[:4]):
        _close(t, g[f"flownet2_{name}"], 5e-5)
    _close(st[-1][:, :, ::2, ::2], g["flownet2_fused"], 5e-4)
