"""Same weights, same batch, forward + backward repeated N times (no optimizer): the packed gradient must be bit-identical every
time.  On a mismatch, the layers whose gradient slices differ are listed.  python scratch/determinism_repeat.py [model] [reps] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.synth import make_pairs
model = sys.argv[1] if len(sys.argv) > 1 else "pwc"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
if os.environ.get("NOSIDE") == "1":
    from mireg import flownets as _fs
    _fs.PredictorEngineBase.use_side_stream = False
x = make_pairs(B, 256, seed=6)[0].cuda()
torch.manual_seed(1)
mm = mireg.opticalFlowReg(model, precision="bf16")
tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=False, autotune=False, overlap_optimizer=False)
for _ in range(2):
    tr.step(x)
torch.cuda.synchronize()
eng = tr.eng
ref, ref_bufs = None, None
bad = 0
names = {}
for n, l in eng.layers.items():
    names[n] = (eng.flat_off[id(l.weight)], l.Co * l.Kf)
for it in range(reps):
    if it % 7 == 3:
        torch.cuda.synchronize()                       # perturb the timing now and then
    tr._fwd_bwd()
    torch.cuda.synchronize()
    g = tr.flat_g.detach().clone()
    loss = tr.loss.out4.clone()
    if ref is None:
        ref, ref_loss = g, loss
        continue
    if not torch.equal(g, ref) or not torch.equal(loss, ref_loss):
        bad += 1
        d = (g - ref).abs()
        lay = [(n, int((d[o:o + k] > 0).sum())) for n, (o, k) in names.items() if (d[o:o + k] > 0).any()]
        print(f"repeat {it}: loss equal {bool(torch.equal(loss, ref_loss))}, {int((d > 0).sum())} gradient elements differ; layers: {lay[:12]}{' ...' if len(lay) > 12 else ''} ({len(lay)} layers)", flush=True)
print(f"{model} B={B}: {bad} of {reps - 1} repeats differ from the first")
