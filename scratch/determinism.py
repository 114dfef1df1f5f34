"""Run-to-run bit determinism of the halo kernels (same inputs, many launches, interleaved with other launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
for name, cin, cout, k, s, H in [("conv3_1", 256, 256, 3, 1, 32), ("conv4_1", 512, 512, 3, 1, 16), ("deconv3", 128, 770, 4, 2, 32), ("deconv2", 64, 386, 4, 2, 64),
                                 ("conv3", 128, 256, 5, 2, 64)]:
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
from oracle import nets
from mireg.synth import make_pairs
x, _ = make_pairs(24, 256, seed=6); xd = x.cuda()
hist = []
for rep in range(2):
    torch.manual_seed(1)
    mm = mireg.opticalFlowReg("flownets", precision="bf16"); nets.analytic_weights_(mm)
    tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=True, autotune=False)
    rows = []
    for st in range(4):
        loss = tr.step(xd).clone(); torch.cuda.synchronize()
        rows.append((loss.cpu(), tr.flat_g.detach().cpu().clone() if st == 0 else None, tr.flat_p.detach().cpu().clone()))
    hist.append(rows)
for st in range(4):
    a, b = hist[0][st], hist[1][st]
    print("step", st, "loss equal", torch.equal(a[0], b[0]), "max |dp|", (a[2] - b[2]).abs().max().item(), flush=True)
ga, gb = hist[0][0][1], hist[1][0][1]
d = (ga - gb).abs()
print("step-0 packed gradient: differing elements", int((d > 0).sum()), "of", d.numel(), "max", d.max().item())
if (d > 0).any():
    idx = torch.nonzero(d > 0).flatten()
    print("first differing offsets", idx[:10].tolist(), "last", idx[-10:].tolist())
    import bisect
    eng = tr.eng
    offs = sorted((eng.flat_off[id(l.weight)], n) for n, l in eng.layers.items())
    for i in (idx[0].item(), idx[len(idx) // 2].item(), idx[-1].item()):
        j = bisect.bisect_right([o for o, _ in offs], i) - 1
        print("  offset", i, "in layer", offs[j][1])
