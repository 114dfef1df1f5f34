"""Bit determinism of one kernel while another kernel runs on a second stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
def mk(name, cin, cout, k, s, H):
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    return lay, x, y, Ho
A = mk("conv3_1", 256, 256, 3, 1, 32)
C3 = mk("conv3", 128, 256, 5, 2, 64)
C4 = mk("conv4", 256, 512, 3, 2, 32)
side = torch.cuda.Stream()
def victim(kind):
    lay, x, y, Ho = A
    if kind == "halo_dgrad":
        out = ws.new(B, 32, 32, 256); lay.run_dgrad_form(y, out); return out.buf
    if kind == "halo_fwd":
        out = ws.new(B, 32, 32, 256); lay.run_fwd_form(x, out); return out.buf
    if kind == "ring_dgrad":
        engine.FORCE_ALGO = (1, 0); out = ws.new(B, 32, 32, 256); lay.run_dgrad_form(y, out); engine.FORCE_ALGO = None; return out.buf
    if kind == "halo_wgrad":
        if lay.wgrad_slab is None: lay.plan_wgrad(x, y)
        lay.run_wgrad(x, y); return lay.wgrad_slab
def aggressor(kind):
    if kind == "wgrad_halo_3x3":
        lay, x, y, _ = A
        if lay.wgrad_slab is None: lay.plan_wgrad(x, y)
        lay.run_wgrad(x, y)
    elif kind == "wgrad_halo_5x5s2":
        lay, x, y, _ = C3
        if lay.wgrad_slab is None: lay.plan_wgrad(x, y)
        lay.run_wgrad(x, y)
    elif kind == "wgrad_ring":
        lay, x, y, _ = C4
        if lay.wgrad_slab is None: lay.plan_wgrad(x, y)
        lay.run_wgrad(x, y)
    elif kind == "halo_fwd":
        lay, x, y, _ = A
        out = ws.new(B, 32, 32, 256); lay.run_fwd_form(x, out)
for v in ("halo_dgrad", "halo_fwd", "ring_dgrad", "halo_wgrad"):
    for a in ("none", "wgrad_halo_3x3", "wgrad_halo_5x5s2", "wgrad_ring", "halo_fwd"):
        if v == "halo_wgrad" and a == "wgrad_halo_3x3": continue
        ref, bad = None, 0
        for it in range(25):
            if a != "none":
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3): aggressor(a)
            got = victim(v).clone()
            torch.cuda.synchronize()
            if ref is None: ref = got
            elif not torch.equal(ref, got): bad += 1
        print(f"victim {v:11s} aggressor {a:17s}: {bad} of 24 runs differ", flush=True)
