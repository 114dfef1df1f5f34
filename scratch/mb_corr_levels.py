"""Cost-volume kernels per size, back-to-back launches (events): FlowNetC's and every PWC level; bytes = SURVEY 8d algorithmic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.engine import Workspace, DT_BF16
from mireg.correlation import correlation_views, correlation_bwd_views
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for tag, B, C, H, md, s2 in (("flownetc", 24, 256, 32, 20, 2), ("pwc6", 48, 196, 4, 4, 1), ("pwc5", 48, 128, 8, 4, 1), ("pwc4", 48, 96, 16, 4, 1),
                             ("pwc3", 48, 64, 32, 4, 1), ("pwc2", 48, 32, 64, 4, 1)):
    D = 2 * (md // s2) + 1
    f1, f2 = ws.new(B, H, H, C), ws.new(B, H, H, C)
    f1.buf.normal_(); f2.buf.normal_()
    out, g = ws.new(B, H, H, D * D), ws.new(B, H, H, D * D)
    g.buf.normal_()
    d1, d2 = ws.new(B, H, H, C), ws.new(B, H, H, C)
    px = B * H * H
    bf, bb = px * (2 * C + D * D) * 2, px * (D * D + 4 * C) * 2
    tf = timed(lambda: correlation_views(f1, f2, out, C, md, s2, 0.1, DT_BF16))
    tb = timed(lambda: correlation_bwd_views(g, f1, f2, d1, d2, (C + 7) // 8 * 8, C, md, s2, 0, 0, DT_BF16))
    print(f"{tag:9s} B={B} C={C:3d} {H}x{H} D={D}: fwd {tf:6.1f} us ({bf / 1e6:5.1f} MB, {bf / tf / 1e6:5.2f} TB/s)   bwd {tb:6.1f} us ({bb / 1e6:5.1f} MB, {bb / tb / 1e6:5.2f} TB/s)", flush=True)
