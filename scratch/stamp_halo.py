"""Where a halo-kernel launch spends its cycles: prologue / K loop / epilogue per workgroup (three s_memtime stamps) for the
full kernel and for ablations of the K loop (scratch/build_stamp.py builds the diagnostic libraries).
python scratch/stamp_halo.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine, _lib
from mireg.engine import ConvLayer, Workspace, run_pack
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ["full"]
libs = {}
for v in VARIANTS:
    l = ctypes.CDLL(os.path.join(HERE, f"libhalo_{v}.so"))
    l.stamp_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
    libs[v] = l
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
grab = {}
orig_call = _lib.call
def spy(name, *a):
    if name == "mireg_conv_gemm":
        grab["d"] = a[0]._obj
        return
    return orig_call(name, *a)
for name, form, cin, cout, k, s, H in [("conv3_1", "fwd", 256, 256, 3, 1, 32), ("conv3", "dgrad", 128, 256, 5, 2, 64)]:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    dx = ws.new(B, H, H, cin)
    for bm in (128, 256):
        engine.FORCE_ALGO = (2, bm)
        engine._lib.call = spy
        try:
            (lay.run_fwd_form(x, y) if form == "fwd" else lay.run_dgrad_form(y, dx))
        finally:
            engine._lib.call = orig_call
            engine.FORCE_ALGO = None
        d = grab["d"]
        tiles = (ctypes.c_long * 2)()
        assert _lib.lib().mireg_conv_halo_eligible(ctypes.byref(d), tiles)
        tm = tiles[1 if bm == 256 else 0]
        ncls = max(d.n_cls, 1)
        nblk = tm * ((d.N + 127) // 128) * ncls
        dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
        d.slab = dbg.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        for var in VARIANTS:
            lib = libs[var]
            for _ in range(20):
                lib.stamp_launch(ctypes.byref(d), bm, tm, st)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.stamp_launch(ctypes.byref(d), bm, tm, st)
            b.record(); torch.cuda.synchronize()
            v = dbg.view(nblk, 4, 8).double()
            v = v[v[:, 0, 5] > 0]
            med = v.median(0).values.mean(0)
            steps = med[5].item()
            clk = (v[:, :, 3] / v[:, :, 4].clamp_min(1)).median().item() * 100.0
            span = (v[:, :, 7].max() - v[:, :, 6].min()).item()      # first start -> last end over the grid (one XCD clock domain assumed)
            late = (v[:, 0, 6] - v[:, 0, 6].min())
            print(f"{name} {form} halo{bm} {var:10s}: {a.elapsed_time(b) * 100:6.1f} us/launch | WGs {v.shape[0]}, clock {clk:.0f} MHz | per WG: prologue {med[0].item():.0f}, "
                  f"K loop {med[1].item():.0f} = {med[1].item() / steps:.0f}/step, epilogue {med[2].item():.0f}, total {med[3].item():.0f} cyc | "
                  f"start skew median {late.median().item():.0f} max {late.max().item():.0f} cyc, grid span {span:.0f} cyc", flush=True)
