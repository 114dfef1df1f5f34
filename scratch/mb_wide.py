"""Stand-alone timing of the contraction sites of one FlowNetS step (B=24, 256x256) on every kernel that can run them:
ring (algo 1), halo (algo 2), the 256-pixel 8-wave tile (algo 3: 256 / 128 columns x split-K).  Random bf16 operands.
python3 scratch/mb_wide.py [filter]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import engine
from mireg.engine import ConvLayer, Workspace, run_pack
B = int(os.environ.get("B", "24"))
SITES = [  # name, cin, cout, k, stride, pad, H(in), form
    ("conv2", 64, 128, 5, 2, 2, 128, "fwd"), ("conv2", 64, 128, 5, 2, 2, 128, "dgrad"),
    ("conv3", 128, 256, 5, 2, 2, 64, "fwd"), ("conv3", 128, 256, 5, 2, 2, 64, "dgrad"),
    ("conv3_1", 256, 256, 3, 1, 1, 32, "fwd"), ("conv3_1", 256, 256, 3, 1, 1, 32, "dgrad"),
    ("conv4", 256, 512, 3, 2, 1, 32, "fwd"), ("conv4", 256, 512, 3, 2, 1, 32, "dgrad"),
    ("conv4_1", 512, 512, 3, 1, 1, 16, "fwd"), ("conv4_1", 512, 512, 3, 1, 1, 16, "dgrad"),
    ("conv5", 512, 512, 3, 2, 1, 16, "fwd"), ("conv5_1", 512, 512, 3, 1, 1, 8, "fwd"),
    ("conv6_1", 1024, 1024, 3, 1, 1, 4, "fwd"),
    ("deconv2", 64, 386, 4, 2, 1, 64, "fwd"), ("deconv2", 64, 386, 4, 2, 1, 64, "dgrad"),
    ("deconv3", 128, 770, 4, 2, 1, 32, "fwd"), ("deconv3", 128, 770, 4, 2, 1, 32, "dgrad"),
    ("deconv4", 256, 1026, 4, 2, 1, 16, "fwd"), ("deconv4", 256, 1026, 4, 2, 1, 16, "dgrad"),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
VARIANTS = [("ring", (1, 0)), ("ring64", (1, 0, 64)), ("halo128", (2, 128)), ("halo256", (2, 256)),
            ("wide256", (3, 256, 256)), ("wide256s2", (3, 256, 256, 2)), ("wide256s3", (3, 256, 256, 3)), ("wide256s4", (3, 256, 256, 4)),
            ("wide128", (3, 256, 128)), ("wide128s2", (3, 256, 128, 2)), ("wide128s4", (3, 256, 128, 4))]
for name, cin, cout, k, s, p, H, form in SITES:
    if flt and flt not in name + ":" + form:
        continue
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    lay = ConvLayer(name, w, None, s, p, 1, ws)
    run_pack(lay.pack_jobs(), ws.code, "cuda:0")
    Ho = (H + 2 * p - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
    dx = ws.new(B, H, H, cin)
    run = (lambda: lay.run_fwd_form(x, y, slope=0.1)) if form == "fwd" else (lambda: lay.run_dgrad_form(y, dx))
    flops = 2.0 * B * Ho * Ho * cout * k * k * cin
    res = []
    for tag, force in VARIANTS:
        engine.FORCE_ALGO = force
        try:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    run()
                b.record(); b.synchronize()
                best = min(best, a.elapsed_time(b) / 10)
            res.append((tag, best))
        except RuntimeError:
            pass
        finally:
            engine.FORCE_ALGO = None
    res.sort(key=lambda r: r[1])
    print(f"{name}:{form:5s} {flops / 1e9:6.2f} GF  " + "  ".join(f"{t}={us * 1e3:.1f}us({flops / us / 1e9:.0f}T)" for t, us in res), flush=True)
