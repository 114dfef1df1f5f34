"""Per-site isolated kernel times of one FlowNetS train step (no side stream, no overlap): where the contraction time goes."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
if os.environ.get('ALT_LIB'):                            # A/B of an alternative build of the library (same process layout)
    import mireg._lib as _L
    assert _L._lib is None
    _L.LIB_PATH = os.environ['ALT_LIB']
from mireg.engine import PROFILER
from mireg.synth import make_pairs
B = int(os.environ.get("B", "24"))
x, _ = make_pairs(B, 256, seed=6); xd = x.cuda()
torch.manual_seed(1)
mm = mireg.opticalFlowReg(os.environ.get("MODEL", "flownets"), precision="bf16")
tr = mireg.RegistrationTrainer(mm.cuda(), use_graph=False, autotune=os.environ.get("TUNE", "1") == "1")
for _ in range(3):
    tr.step(xd)
tr.eng.use_side_stream, tr.overlap_optimizer = False, False
PROFILER.enabled, PROFILER.records, PROFILER.byte_records = True, [], []
N = 3
for _ in range(N):
    tr._fwd_bwd()
torch.cuda.synchronize()
rows = collections.OrderedDict()
for fam, fl, a, b, tag in PROFILER.records + PROFILER.byte_records:
    d = rows.setdefault((tag, fam), [0, 0.0, 0.0])
    d[0] += 1; d[1] += fl; d[2] += a.elapsed_time(b)
PROFILER.enabled = False
tot = sum(v[2] for v in rows.values()) / N
print(f"total isolated {tot:.3f} ms/step")
for (tag, fam), (n, fl, ms) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    print(f"{tag:34s} {fam:34s} n/step {n // N:2d} us/step {ms / N * 1e3:7.1f}  {fl / N / 1e9:8.2f} G(FLOP|B)  {fl / (ms * 1e-3) / 1e12 if ms else 0:7.1f} T/s")
