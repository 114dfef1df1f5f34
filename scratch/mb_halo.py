"""A/B of the halo-staged kernel against the ring kernel on the FlowNetS contraction shapes (B=24, bf16), one process,
interleaved rounds: python scratch/mb_halo.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 24
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
# name, form, cin, cout, k, s, H(of the conv input)
SHAPES = [("conv3_1", "fwd", 256, 256, 3, 1, 32), ("conv4_1", "fwd", 512, 512, 3, 1, 16),
          ("conv3_1", "dgrad", 256, 256, 3, 1, 32), ("conv4_1", "dgrad", 512, 512, 3, 1, 16),
          ("conv2", "dgrad", 64, 128, 5, 2, 128), ("conv3", "dgrad", 128, 256, 5, 2, 64), ("conv4", "dgrad", 256, 512, 3, 2, 32),
          ("deconv2", "dgrad", 64, 386, 4, 2, 64), ("deconv3", "dgrad", 128, 770, 4, 2, 32), ("deconv4", "dgrad", 256, 1026, 4, 2, 16)]
VARIANTS = [("auto", None), ("ring", (1, 0)), ("halo128", (2, 128)), ("halo256", (2, 256)), ("halo128n64", (2, 128, 64)), ("halo256n64", (2, 256, 64))]


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, form, cin, cout, k, s, H in SHAPES:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    dx = ws.new(B, H, H, cin)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    run = (lambda: lay.run_fwd_form(x, y)) if form == "fwd" else (lambda: lay.run_dgrad_form(y, dx))
    res = {}
    for tag, force in VARIANTS:
        engine.FORCE_ALGO = force
        try:
            run(); torch.cuda.synchronize()
        except RuntimeError:
            continue
        res[tag] = []
    for _ in range(rounds):
        for tag, force in VARIANTS:
            if tag in res:
                engine.FORCE_ALGO = force
                res[tag].append(timed(run))
    engine.FORCE_ALGO = None
    line = f"{name:8s} {form:5s} {fl/1e9:6.1f} GF |"
    for tag in res:
        t = sorted(res[tag])[len(res[tag]) // 2]
        line += f" {tag} {t:6.1f} us {fl/t/1e6:6.0f} TF |"
    print(line, flush=True)


# ---- backward-weights: ring vs halo over a few splits -------------------------------------------------------------------
WSHAPES = [("conv3_1", 256, 256, 3, 1, 32), ("conv4_1", 512, 512, 3, 1, 16), ("conv2", 64, 128, 5, 2, 128), ("conv3", 128, 256, 5, 2, 64),
           ("deconv2", 64, 386, 4, 2, 64), ("deconv3", 128, 770, 4, 2, 32)]
for name, cin, cout, k, s, H in WSHAPES:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, (k - 1) // 2, 1, ws)
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    lay.plan_wgrad(x, y)
    heur = lay.wgrad_split
    line = f"{name:8s} wgrad {fl/1e9:6.1f} GF |"
    for algo, tag, splits in ((1, "ring", (heur,)), (2, "halo", (2, 4, 8, 16, 24))):
        engine.WGRAD_ALGO = algo
        for sp in splits:
            lay.wgrad_split = sp
            lay.wgrad_slab = torch.zeros(sp, lay.Co, lay.Kf, device="cuda")
            try:
                lay.run_wgrad(x, y); torch.cuda.synchronize()
            except RuntimeError:
                continue
            t = sorted(timed(lambda: lay.run_wgrad(x, y)) for _ in range(rounds))[rounds // 2]
            line += f" {tag} z{sp} {t:6.1f} us {fl/t/1e6:5.0f} TF |"
    engine.WGRAD_ALGO = 0
    print(line, flush=True)
