"""Timing of the cost-volume kernels at the benchmarked sizes (events around 20 launches each)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.engine import Workspace, DT_BF16
from mireg.correlation import correlation_views, correlation_bwd_views
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
for B, C, H, md, s2 in ((24, 256, 32, 20, 2), (48, 32, 64, 4, 1), (48, 64, 32, 4, 1), (48, 196, 4, 4, 1)):
    D = 2 * (md // s2) + 1
    f1, f2 = ws.new(B, H, H, C), ws.new(B, H, H, C)
    f1.buf.normal_(); f2.buf.normal_()
    out, g = ws.new(B, H, H, D * D), ws.new(B, H, H, D * D)
    g.buf.normal_()
    d1, d2 = ws.new(B, H, H, C), ws.new(B, H, H, C)
    def fwd(): correlation_views(f1, f2, out, C, md, s2, 0.1, DT_BF16)
    def bwd(): correlation_bwd_views(g, f1, f2, d1, d2, (C + 7) // 8 * 8, C, md, s2, 0, 0, DT_BF16)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print(f"B={B} C={C} {H}x{H} md={md} s2={s2} {name}: {a.elapsed_time(b) / 20 * 1e3:.1f} us")
