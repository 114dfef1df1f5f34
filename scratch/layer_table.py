"""Per-launch table of the MFMA contractions of one train step (eager, optional single stream)."""
import sys, torch
sys.path.insert(0, ".")
import mireg
from mireg.engine import PROFILER
from mireg.synth import make_pairs
model_name = sys.argv[1] if len(sys.argv) > 1 else "flownets"
side = (sys.argv[2] == "side") if len(sys.argv) > 2 else False
dev = torch.device("cuda:0")
torch.manual_seed(6)
model = mireg.opticalFlowReg(model_name, precision="bf16").to(dev)
tr = mireg.RegistrationTrainer(model, lr=1e-4, eps=1e-4, use_graph=False)
from mireg import flownets
flownets.PredictorEngineBase.use_side_stream = side
x, _ = make_pairs(24, 256, seed=6)
x = x.to(dev)
for _ in range(3):
    tr.step(x)
PROFILER.enabled, PROFILER.records = True, []
R = 5
for _ in range(R):
    tr._fwd_bwd()
torch.cuda.synchronize()
agg = {}
order = []
for fam, fl, a, b, tag in PROFILER.records:
    if tag not in agg:
        agg[tag] = [fam, fl, 0.0, 0]
        order.append(tag)
    agg[tag][2] += a.elapsed_time(b)
    agg[tag][3] += 1
tot = 0
for tag in order:
    fam, fl, ms, n = agg[tag]
    us = ms / n * 1e3
    tot += us * (n / R)
    print(f"{tag:70s} {fam:34s} x{n//R} {us:8.1f} us {fl / (us * 1e-6) / 1e12:7.1f} TF  {fl/1e9:6.1f} GF")
print("total us/step", tot)
