"""Upsampler backward-data hazard: two identically seeded FlowNetS trainers, B=24 256x256, four steps each, with
the pixel-parallel backward-data kernel of the 2->2 upsamplers ON (MIREG_TINY_MASK=7).  Prints, per differing element of dflowT[l],
what the kernel's output equals: the recomputation from the snapshot operands (expected), bf16(loss gradient) alone (= the kernel's
own store missing / overtaken by its predecessor's), bf16(conv) alone (= the accumulate operand read as zero), or neither.
usage: MIREG_TINY_MASK=7 python3 scratch/hazard_probe.py [graph|eager] [dot-prefix]"""
import os, sys, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
dot = sys.argv[2] if len(sys.argv) > 2 else None
import mireg
if os.environ.get("ALT_LIB"):                                # A/B of an alternative build of the library
    import mireg._lib as _L
    assert _L._lib is None
    _L.LIB_PATH = os.environ["ALT_LIB"]
from mireg import engine
from mireg.synth import make_pairs
import torch.nn.functional as F
print("mode", mode, "TINY_MASK", engine.TINY_MASK, flush=True)
if dot and mode == "graph":                                  # keep the captured graphs' DOT dumps (hipGraphDebugDotPrint)
    _G = torch.cuda.CUDAGraph
    made = []
    class DbgGraph(_G):
        def __new__(cls, *a, **k):
            g = super().__new__(cls, *a, **k)
            try:
                g.enable_debug_mode()
            except Exception as e:                           # noqa: BLE001
                print("enable_debug_mode failed:", e)
            made.append(g)
            return g
    torch.cuda.CUDAGraph = DbgGraph
x, _ = make_pairs(24, 256, seed=6); xd = x.cuda()
NST = 3
upname = {3: "upsampled_flow3_to_2", 4: "upsampled_flow4_to_3", 5: "upsampled_flow5_to_4", 6: "upsampled_flow6_to_5"}
snaps, hist, wsnap, gsnap = [], [], [], []
for rep in range(2):
    torch.manual_seed(1)
    mm = mireg.opticalFlowReg("flownets", precision="bf16")
    tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=(mode == "graph"), autotune=False)
    sd = dict(mm.named_parameters())
    rows, ss, ww, gg = [], [], [], []
    for st in range(NST):
        ww.append({lv: sd[[n for n in sd if upname[lv] in n and n.endswith("weight")][0]].detach().float().cpu().clone() for lv in upname})
        loss = tr.step(xd).clone(); torch.cuda.synchronize()
        rows.append((loss.cpu(), tr.flat_p.detach().cpu().clone()))
        gg.append(tr.flat_g.detach().cpu().clone())
        e = tr.eng
        snap = {}
        for nm in ("dflowT", "dcat"):
            for k, v in getattr(e, nm).items():
                snap[f"{nm}[{k}]"] = v.buf.detach().float().cpu().clone()
        for i, g in enumerate(tr.loss.gflow):
            snap[f"gflow[{i}]"] = g.detach().cpu().clone()
        ss.append(snap)
    hist.append(rows); snaps.append(ss); wsnap.append(ww); gsnap.append(gg)
for st in range(NST):
    a, b = hist[0][st], hist[1][st]
    print("step", st, "loss equal", torch.equal(a[0], b[0]), "max |dp|", (a[1] - b[1]).abs().max().item(),
          "packed-gradient elements differing", int((gsnap[0][st] != gsnap[1][st]).sum()), flush=True)
e = tr.eng
first = next((st for st in range(NST) if not torch.equal(gsnap[0][st], gsnap[1][st])), None)
print("first step whose gradients differ:", first)
if first is not None:
    d = (gsnap[0][first] - gsnap[1][first]).abs()
    for n, l in e.layers.items():
        o = e.flat_off[id(l.weight)]
        seg = d[o:o + l.Co * l.Kf]
        if (seg > 0).any():
            print(f"  layer {n}: {int((seg > 0).sum())} of {seg.numel()} gradient elements differ, max {seg.max().item():.3e}")
    for k in snaps[0][first]:
        aa, bb = snaps[0][first][k], snaps[1][first][k]
        dd = (aa - bb).abs()
        if (dd > 0).any():
            print(f"  buffer {k}: {int((dd > 0).sum())} of {dd.numel()} elements differ, max {dd.max().item():.3e}")
st0 = first if first is not None else 0
for lv in (3, 4, 5, 6):
    key = f"dflowT[{lv}]"
    a, b = snaps[0][st0][key][..., :2], snaps[1][st0][key][..., :2]
    d = (a - b).abs()
    print(f"{key}: {int((d > 0).sum())} of {d.numel()} elements differ between the two trainers (step {st0})")
    nz = torch.nonzero(d > 0)
    rowsets = {}
    for bb, yy, xx, cc in nz.tolist():
        rowsets.setdefault((bb, yy, cc), []).append(xx)
    for (bb, yy, cc), xs in list(rowsets.items())[:8]:
        print(f"   image {bb} row {yy} channel {cc}: x = {xs}")
    cs, cd = e.skip_c[lv - 1], {2: 64, 3: 128, 4: 256, 5: 512}[lv - 1]
    for t in (0, 1):
        W = wsnap[t][st0][lv]
        sn = snaps[t][st0]
        gup = sn[f"dcat[{lv - 1}]"][..., cs + cd: cs + cd + 2].permute(0, 3, 1, 2)
        conv = F.conv2d(gup, W, None, 2, 1)
        gl = [g for k2, g in sn.items() if k2.startswith("gflow") and g.shape[-1] == a.shape[2] and g.shape[-2] == a.shape[1]][0]
        got = sn[key][..., :2]
        exp_nf = (gl.bfloat16().float() + conv).bfloat16().float().permute(0, 2, 3, 1)     # loss gradient cast first, kernel accumulates
        exp_f = (gl + conv).bfloat16().float().permute(0, 2, 3, 1)                          # fused planar fp32 addend
        exp = exp_f if int(((got - exp_f).abs() > 0).sum()) <= int(((got - exp_nf).abs() > 0).sum()) else exp_nf
        print(f"   (expectation: {'fused add_nchw' if exp is exp_f else 'cast + accumulate'})")
        only_loss = gl.bfloat16().float().permute(0, 2, 3, 1)
        only_conv = conv.bfloat16().float().permute(0, 2, 3, 1)
        bad = torch.nonzero((got - exp).abs() > 0)
        kinds = {"= bf16(loss gradient) alone": 0, "= bf16(conv) alone": 0, "neither": 0}
        for p in bad.tolist():
            bb, yy, xx, cc = p
            g_ = got[bb, yy, xx, cc].item()
            if g_ == only_loss[bb, yy, xx, cc].item():
                kinds["= bf16(loss gradient) alone"] += 1
            elif g_ == only_conv[bb, yy, xx, cc].item():
                kinds["= bf16(conv) alone"] += 1
            else:
                kinds["neither"] += 1
        print(f"   trainer {t}: {bad.shape[0]} elements off the recomputation: {kinds}")
        for p in bad[:6].tolist():
            bb, yy, xx, cc = p
            print(f"      {p}: got {got[bb, yy, xx, cc].item():.6f} expected {exp[bb, yy, xx, cc].item():.6f} lossgrad {gl[bb, cc, yy, xx].item():.6f} conv {conv[bb, cc, yy, xx].item():.6f}")

if os.environ.get("MIREG_HAZ_DBG") and int(os.environ["MIREG_HAZ_DBG"]) & 8:
    import ctypes, numpy as np
    from mireg import _lib
    n = 16 + 300 * 208
    buf = (ctypes.c_float * n)()
    _lib.lib().mireg_dbg_tiny_records.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if _lib.lib().mireg_dbg_tiny_records(buf, n):
        a = np.frombuffer(buf, dtype=np.float32).copy()
        cnt = int(a[:1].view(np.int32)[0])
        print(f"in-kernel double evaluation: {cnt} threads saw different operands ~5 us apart (over all launches of both trainers)")
        rec = a[16:16 + min(cnt, 300) * 208].reshape(-1, 208)
        names = ("f0", "f1", "w00", "w01", "w10", "w11")
        for r in rec[:24]:
            print(f"   img {int(r[0])} y {int(r[1])} x {int(r[2])} Hc {int(r[3])} block {int(r[12])} lane {int(r[15])}: first (conv0, conv1, add0, add1) = {r[4]:.6f} {r[5]:.6f} {r[6]:.6f} {r[7]:.6f}   late = {r[8]:.6f} {r[9]:.6f} {r[10]:.6f} {r[11]:.6f}")
            t1, t2 = r[16:112].reshape(16, 6), r[112:208].reshape(16, 6)
            for t in range(16):
                d = [f"{names[i]} {t1[t, i]:.6g} -> {t2[t, i]:.6g}" for i in range(6) if t1[t, i] != t2[t, i]]
                if d:
                    print(f"        tap ky {t // 4} kx {t % 4}: " + "; ".join(d))
