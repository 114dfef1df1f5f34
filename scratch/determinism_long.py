"""Two identically seeded trainers, N hipGraph-replayed steps each: parameters must stay bit-identical (autotune off: fixed launch
shapes).  python scratch/determinism_long.py [model] [steps] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
if os.environ.get('ALT_LIB'):                            # A/B of an alternative build of the library
    import mireg._lib as _L
    assert _L._lib is None
    _L.LIB_PATH = os.environ['ALT_LIB']
from mireg.synth import make_pairs
model = sys.argv[1] if len(sys.argv) > 1 else "flownets"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 24
x = make_pairs(B, 256, seed=6)[0].cuda()
if os.environ.get('NOSIDE') == '1':
    from mireg import flownets as _fs
    _fs.PredictorEngineBase.use_side_stream = False
import time
finals, losses = [], []
for rep in range(2):
    torch.manual_seed(1)
    mm = mireg.opticalFlowReg(model, precision="bf16")
    tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=True, autotune=False)
    ls = []
    for st in range(steps):
        if st == 10:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ls.append(tr.step(x).clone())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(steps - 10, 1) * 1e3
    finals.append(tr.flat_p.detach().clone())
    losses.append(torch.stack(ls)[:, 3].cpu())
same = torch.equal(finals[0], finals[1])
first_bad = next((i for i in range(steps) if losses[0][i] != losses[1][i]), None)
print(f"{model} B={B}: {steps} steps, parameters bit-identical: {same}, first differing loss at step {first_bad}; "
      f"loss {float(losses[0][0]):.3f} -> {float(losses[0][-1]):.3f}, finite: {bool(torch.isfinite(losses[0]).all())}, {ms:.3f} ms/step (autotune off)")
