"""One contraction shape, a fixed number of launches of the ring kernel, the halo kernel and the 8-wave tile (for rocprofv3 --pmc runs):
python3 scratch/mb_one.py [cin cout k s H form]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
a = sys.argv[1:]
cin, cout, k, s, H = [int(v) for v in (a[:5] if len(a) >= 5 else (256, 256, 3, 1, 32))]
form = a[5] if len(a) > 5 else "fwd"
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
lay = ConvLayer("t", w, None, s, (k - 1) // 2, 1, ws)
run_pack(lay.pack_jobs(), ws.code, "cuda:0")
Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
x = ws.new(B, H, H, cin); x.buf.normal_()
y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
dx = ws.new(B, H, H, cin)
run = (lambda: lay.run_fwd_form(x, y)) if form == "fwd" else (lambda: lay.run_dgrad_form(y, dx))
for force in ((1, 0), (2, 128), (2, 256), (3, 256, 128), (3, 256, 256)):      # ring, halo 128 / 256 rows, 8-wave tile 128 / 256 columns
    engine.FORCE_ALGO = force
    try:
        for _ in range(12):
            run()
    except RuntimeError:
        pass
    torch.cuda.synchronize()
print("done")
