"""The deep (8x8 / 4x4 grid) layers alone, back-to-back launches: forward, backward-data and backward-weights over splits.
python scratch/mb_deep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
SH = [("conv5", 512, 512, 3, 2, 16), ("conv5_1", 512, 512, 3, 1, 8), ("conv6", 512, 1024, 3, 2, 8), ("conv6_1", 1024, 1024, 3, 1, 4)]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, cin, cout, k, s, H in SH:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    dx = ws.new(B, H, H, cin)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    wbytes = cout * cin * k * k
    line = f"{name:8s} {fl / 1e9:5.2f} GF, weights {wbytes * 2 / 1e6:5.1f} MB bf16 / {wbytes * 4 / 1e6:5.1f} MB fp32 |"
    t = timed(lambda: lay.run_fwd_form(x, y)); line += f" fwd {t:5.1f} us {fl / t / 1e6:4.0f} TF |"
    t = timed(lambda: lay.run_dgrad_form(y, dx)); line += f" dgrad {t:5.1f} us {fl / t / 1e6:4.0f} TF |"
    lay.plan_wgrad(x, y)
    engine.WGRAD_ALGO = 1
    for sp in (1, 2, 3, 4, 6):
        lay.wgrad_split = sp
        lay.wgrad_slab = torch.zeros(sp, lay.Co, lay.Kf, device="cuda")
        try:
            t = timed(lambda: lay.run_wgrad(x, y))
            line += f" wgrad/{sp} {t:5.1f} us {fl / t / 1e6:4.0f} TF {sp * wbytes * 4 / t / 1e6:5.2f} TB/s |"
        except RuntimeError as e:
            line += f" wgrad/{sp} {str(e)[:20]} |"
    engine.WGRAD_ALGO = 0
    print(line, flush=True)
# reference points on the same box: a plain fp32 fill and copy of 37.7 MB
buf = torch.empty(1024 * 9216, device="cuda"); src = torch.randn(1024 * 9216, device="cuda")
print(f"fill 37.7 MB: {timed(lambda: buf.fill_(1.0)):.1f} us   copy 37.7 MB: {timed(lambda: buf.copy_(src)):.1f} us")
