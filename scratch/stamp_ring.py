"""Where a split-K ring-kernel launch of the deep layers spends its time: prologue / K loop / epilogue cycles per wave (s_memtime) and
the wall-clock window of every workgroup (s_memrealtime, 100 MHz).  Needs scratch/libring_stamp.so (python scratch/build_stamp_ring.py).
python scratch/stamp_ring.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine, _lib
from mireg.engine import ConvLayer, Workspace, run_pack
HERE = os.path.dirname(os.path.abspath(__file__))
ctypes.CDLL(_lib.lib()._name, mode=ctypes.RTLD_GLOBAL)       # the copy still refers to the halo entry points of the real library
lib = ctypes.CDLL(os.path.join(HERE, "libring_stamp.so"))
lib.stamp_launch_ring.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
grab = {}
orig_call = _lib.call


def spy(name, *a):
    if name == "mireg_conv_gemm":
        grab["d"] = a[0]._obj
        return
    return orig_call(name, *a)


for name, cin, cout, k, s, H, bn, split in [("conv6_1", 1024, 1024, 3, 1, 4, 64, 8), ("conv5_1", 512, 512, 3, 1, 8, 64, 4),
                                             ("conv6_1", 1024, 1024, 3, 1, 4, 128, 8), ("conv4_1", 512, 512, 3, 1, 16, 64, 2)]:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout)
    engine.FORCE_ALGO = (1, 0, bn)
    engine._lib.call = spy
    try:
        lay.run_fwd_form(x, y)
    finally:
        engine._lib.call = orig_call
        engine.FORCE_ALGO = None
    d = grab["d"]
    M, N = B * Ho * Ho, cout
    d.split_k, d.tile_n = split, bn
    slab = torch.empty(split * M * N, device="cuda")
    d.slab, d.slab_cls_stride = slab.data_ptr(), split * M * N
    nwg = ((M + 127) // 128) * ((N + bn - 1) // bn) * split
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
    d.y32 = dbg.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        lib.stamp_launch_ring(ctypes.byref(d), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.stamp_launch_ring(ctypes.byref(d), st)
    b.record(); torch.cuda.synchronize()
    v = dbg.view(nwg, 4, 8).double()
    ok = v[:, 0, 7] > 0
    v = v[ok]
    pro, loop, epi, nk = v[:, :, 0].mean(1), v[:, :, 1].mean(1), v[:, :, 2].mean(1), v[:, 0, 3]
    r0, r1 = v[:, :, 4].min(1).values, v[:, :, 5].max(1).values
    t0 = r0.min()
    dur = (r1 - r0) / 100.0                                   # us per workgroup (100 MHz wall clock)
    start = (r0 - t0) / 100.0
    end = (r1 - t0) / 100.0
    print(f"{name} fwd <128,{bn}> split {split}: {a.elapsed_time(b) * 100:6.1f} us/launch (GEMM only) | {int(ok.sum())} WGs, {nk.median().item():.0f} K-steps each | per wave cycles: "
          f"prologue {pro.median().item():.0f}, K loop {loop.median().item():.0f} = {(loop / nk).median().item():.0f}/step, epilogue {epi.median().item():.0f} | "
          f"WG duration median {dur.median().item():.1f} us (min {dur.min().item():.1f}, max {dur.max().item():.1f}); starts: median {start.median().item():.1f} us, "
          f"90 % by {start.quantile(0.9).item():.1f} us, last {start.max().item():.1f} us; last end {end.max().item():.1f} us", flush=True)
