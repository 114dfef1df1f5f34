"""Two identically seeded FlowNetS trainers, four steps: where do they diverge?  Snapshots every engine buffer after step 0 of each
trainer, lists the differing ones and recomputes the upsampler backward-data outputs on the host (the tool that found the LDS
write-after-read race and the upsampler backward-data hazard of round 2; MIREG_TINY_MASK=7 re-enables that kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
for name, cin, cout, k, s, H in [
                                 ("conv3", 128, 256, 5, 2, 64)][:0]:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    for form in ("fwd", "dgrad", "wgrad"):
        ref = None
        bad = 0
        for it in range(30):
            if form == "fwd":
                if s != 1: break
                out = ws.new(B, Ho, Ho, cout); lay.run_fwd_form(x, out); got = out.buf
            elif form == "dgrad":
                out = ws.new(B, H, H, cin); lay.run_dgrad_form(y, out); got = out.buf
            else:
                lay.wgrad_slab = None; lay.plan_wgrad(x, y); lay.wgrad_slab.fill_(3.0); lay.run_wgrad(x, y); got = lay.wgrad_slab.sum(0)
            torch.cuda.synchronize()
            if ref is None: ref = got.clone()
            elif not torch.equal(ref, got): bad += 1
        if ref is not None:
            print(name, form, "mismatching runs:", bad, "of 29", flush=True)
# whole trainer: where do two identical trainers diverge?
from mireg.synth import make_pairs
x, _ = make_pairs(24, 256, seed=6); xd = x.cuda()
hist = []
snaps = []
for rep in range(2):
    torch.manual_seed(1)
    mm = mireg.opticalFlowReg("flownets", precision="bf16");    tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=os.environ.get('NOGRAPH') != '1', autotune=False,
                                   overlap_optimizer=os.environ.get('NOOVL') != '1')
    if os.environ.get('NOSIDE') == '1':
        from mireg import flownets as _fs
        _fs.PredictorEngineBase.use_side_stream = False
    rows = []
    for st in range(4):
        loss = tr.step(xd).clone(); torch.cuda.synchronize()
        rows.append((loss.cpu(), tr.flat_g.detach().cpu().clone() if st == 0 else None, tr.flat_p.detach().cpu().clone()))
        if st == 0:
            e = tr.eng
            snap = {}
            for nm in ("dflowT", "dcat", "flowT", "cat", "flow32", "raw", "draw"):
                for k, v in getattr(e, nm).items():
                    snap[f"{nm}[{k}]"] = v.buf.detach().float().cpu().clone()
            for nm in ("da61", "da6", "da5", "da4", "da3", "da1", "dflow32", "a61"):
                snap[nm] = getattr(e, nm).buf.detach().float().cpu().clone()
            for i, g in enumerate(tr.loss.gflow):
                snap[f"gflow[{i}]"] = g.detach().cpu().clone()
            for n, l in e.layers.items():
                if l.wgrad_slab is not None:
                    snap[f"slab:{n}"] = l.wgrad_slab.detach().float().cpu().clone()
                if getattr(l, "_dz18", None) is not None:
                    snap[f"dz18:{n}"] = l._dz18.buf.detach().float().cpu().clone()
            snaps.append(snap)
    hist.append(rows)
for st in range(4):
    a, b = hist[0][st], hist[1][st]
    print("step", st, "loss equal", torch.equal(a[0], b[0]), "max |dp|", (a[2] - b[2]).abs().max().item(), flush=True)
ga, gb = hist[0][0][1], hist[1][0][1]
d = (ga - gb).abs()
print("step-0 packed gradient: differing elements", int((d > 0).sum()), "of", d.numel(), "max", d.max().item())
eng = tr.eng
for n, l in eng.layers.items():
    o = eng.flat_off[id(l.weight)]
    seg = d[o:o + l.Co * l.Kf]
    if (seg > 0).any():
        print(f"  layer {n}: {int((seg > 0).sum())} of {seg.numel()} differ, max {seg.max().item():.3e}, split {l.wgrad_split} algo {l.wgrad_algo} tiny {l.tiny} thin {l.thin} thin_gemm {l.thin_gemm}")

print("---- buffers after step 0 ----")
import itertools
for k in snaps[0]:
    a, b = snaps[0][k], snaps[1][k]
    if a.shape != b.shape:
        print(k, "shape differs", a.shape, b.shape); continue
    d = (a - b).abs()
    if (d > 0).any():
        nz = torch.nonzero(d > 0)
        chans = sorted(set(nz[:, -1].tolist()))[:12] if d.dim() == 4 else []
        if k.startswith(("dflowT", "gflow", "flowT", "flow32", "cat", "raw")) or k in ("a61",):
            print(f"{k}: {int((d > 0).sum())} of {d.numel()} differ, max {d.max().item():.3e}, last-dim indices {chans}, first positions {nz[:6].tolist()}")

# which trainer is wrong, and by what: recompute dflowT[l+1] = bf16(bf16(loss grad) + conv_s2(gup, W_up)) from the snapshots
import torch.nn.functional as F
torch.manual_seed(1)
mref = mireg.opticalFlowReg("flownets", precision="bf16")            # same seed as the two trainers: their step-0 weights
sd = dict(mref.named_parameters())
upname = {4: "upsampled_flow4_to_3", 5: "upsampled_flow5_to_4", 6: "upsampled_flow6_to_5"}
for lv in (4, 5, 6):
    key = f"dflowT[{lv}]"
    a, b = snaps[0][key], snaps[1][key]
    d = (a - b).abs()
    if not (d > 0).any():
        continue
    wname = [n for n in sd if upname[lv] in n and n.endswith("weight")]
    print(lv, "weight", wname)
    W = sd[wname[0]].detach().float()
    e = tr.eng
    cs, cd = e.skip_c[lv - 1], {2: 64, 3: 128, 4: 256, 5: 512}[lv - 1]
    for t in (0, 1):
        gup = snaps[t][f"dcat[{lv - 1}]"][..., cs + cd: cs + cd + 2].permute(0, 3, 1, 2)
        conv = F.conv2d(gup, W, None, 2, 1)
        gl = [g for k2, g in snaps[t].items() if k2.startswith("gflow") and g.shape[-1] == a.shape[2] and g.shape[-2] == a.shape[1]][0]
        exp = (gl.bfloat16().float() + conv).bfloat16().float().permute(0, 2, 3, 1)
        got = snaps[t][key][..., :2]
        dd = (got - exp).abs()
        print(f"  trainer {t}: {int((dd > 0).sum())} elements off the recomputation, max {dd.max().item():.3e}")
        nz = torch.nonzero(dd > 0)[:10]
        for p in nz.tolist():
            bb, yy, xx, cc = p
            print(f"    {p}: got {got[bb, yy, xx, cc].item():.6f} expected {exp[bb, yy, xx, cc].item():.6f} lossgrad {gl[bb, cc, yy, xx].item():.6f} conv {conv[bb, cc, yy, xx].item():.6f}")
