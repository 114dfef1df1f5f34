"""micro-bench of the HBM-bound kernels: python scratch/mb_tail.py [warp|warpbwd|corr|corrpwc|smooth] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import _lib
from mireg.engine import Workspace, _stream
from mireg.correlation import correlation_views
which = sys.argv[1] if len(sys.argv) > 1 else "warp"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
B, H = 24, 256
torch.manual_seed(0)
if which in ("warp", "warpbwd", "smooth"):
    flow = torch.randn(B, H, H, 2, device=dev)          # NHWC interleaved like the conv engine emits
    fl = flow.permute(0, 3, 1, 2)
    sb, sc, sy, sx = fl.stride()
    moving, fixed = torch.rand(B, 1, H, H, device=dev), torch.rand(B, 1, H, H, device=dev)
    warped, g = torch.empty_like(moving), torch.rand(B, 1, H, H, device=dev)
    gflow = torch.empty(B, 2, H, H, device=dev)
    sums = torch.zeros(32, 8, device=dev, dtype=torch.float64)
    if which == "warp":
        run = lambda: _lib.call("mireg_stn_warp_fwd", fl.data_ptr(), sb, sc, sx, moving.data_ptr(), fixed.data_ptr(), warped.data_ptr(), sums.data_ptr(), B, 1, H, H, _stream())
        nbytes = B * H * H * 20
    elif which == "warpbwd":
        run = lambda: _lib.call("mireg_stn_warp_bwd", fl.data_ptr(), sb, sc, sx, moving.data_ptr(), g.data_ptr(), gflow.data_ptr(), 2 * H * H, H * H, 1, 0.0, B, 1, H, H, _stream())
        nbytes = B * H * H * 28
    else:
        run = lambda: _lib.call("mireg_smoothness_fwd", fl.data_ptr(), sb, sc, sx, sums.data_ptr(), B, H, H, _stream())
        nbytes = B * H * H * 8
else:
    ws = Workspace(dev, torch.bfloat16)
    if which == "corr":
        C, Hc, md, s2, nd = 256, 32, 20, 2, 441
    else:
        C, Hc, md, s2, nd = 32, 64, 4, 1, 81
        B = 48
    f1, f2 = ws.new(B, Hc, Hc, C), ws.new(B, Hc, Hc, C)
    f1.buf.normal_(); f2.buf.normal_()
    out = ws.new(B, Hc, Hc, nd)
    run = lambda: correlation_views(f1, f2, out, C, md, s2, 0.1, ws.code)
    nbytes = B * Hc * Hc * (2 * C + nd) * 2
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): run()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / iters * 1e3
print(f"{which}: {us:.1f} us/launch, algorithmic {nbytes/1e6:.1f} MB -> {nbytes/us/1e3:.0f} GB/s ({nbytes/us/1e3/8000*100:.1f}% of 8 TB/s)")
