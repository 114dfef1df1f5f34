"""Stand-alone timing of the backward-weights GEMMs of one FlowNetS step (B=24, 256x256) on the ring kernel (algo 1), the halo kernel
(algo 2) and the 256 x 256 8-wave tile (algo 3) over pixel splits; time = kernel + the slab reduce the split implies (re-read at 3 TB/s,
as the tuner prices it).  python3 scratch/mb_wgrad.py [filter]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import _lib
from mireg.engine import ConvLayer, Workspace
B = int(os.environ.get("B", "24"))
SITES = [  # name, cin (conv view), cout, k, stride, pad, H(in)
    ("conv2", 64, 128, 5, 2, 2, 128), ("conv3", 128, 256, 5, 2, 2, 64), ("conv3_1", 256, 256, 3, 1, 1, 32),
    ("conv4", 256, 512, 3, 2, 1, 32), ("conv4_1", 512, 512, 3, 1, 1, 16), ("conv5", 512, 512, 3, 2, 1, 16),
    ("conv5_1", 512, 512, 3, 1, 1, 8), ("conv6", 512, 1024, 3, 2, 1, 8), ("conv6_1", 1024, 1024, 3, 1, 1, 4),
    ("deconv5", 512, 1024, 4, 2, 1, 8), ("deconv4", 256, 1026, 4, 2, 1, 16), ("deconv3", 128, 770, 4, 2, 1, 32), ("deconv2", 64, 386, 4, 2, 1, 64),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
ws = Workspace(torch.device("cuda:0"), torch.bfloat16)
for name, cin, cout, k, s, p, H in SITES:
    if flt and flt not in name:
        continue
    lay = ConvLayer(name, torch.zeros(cout, cin, k, k, device="cuda"), None, s, p, 1, ws)
    Ho = (H + 2 * p - k) // s + 1
    x = ws.new(B, H, H, cin); x.buf.normal_()
    dy = ws.new(B, Ho, Ho, cout); dy.buf.normal_()
    flops = 2.0 * B * Ho * Ho * cout * k * k * cin
    elems = lay.Co * lay.Kf
    nk = (dy.rows + 31) // 32
    res = []
    for algo, tag in ((1, "ring"), (2, "halo"), (3, "wide")):
        for sp in (1, 2, 3, 4, 6, 8, 11, 16, 22, 32, 44, 64):
            if sp > max(nk // 8, 1) or sp * elems > (1 << 26):
                continue
            tiles = ((lay.Co + 255) // 256) * ((lay.Kf + 255) // 256) if algo == 3 else ((lay.Co + 127) // 128) * ((lay.Kf + 127) // 128)
            if algo != 2 and not (64 <= tiles * sp <= 2304):
                continue
            slab = torch.empty(sp * elems, device="cuda", dtype=torch.float32)
            d = lay._wgrad_desc(x, dy, sp, slab.data_ptr(), algo)
            try:
                for _ in range(2):
                    _lib.call("mireg_conv_wgrad", ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
            except RuntimeError:
                break
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    _lib.call("mireg_conv_wgrad", ctypes.byref(d), torch.cuda.current_stream().cuda_stream)
                b.record(); b.synchronize()
                best = min(best, a.elapsed_time(b) / 10)
            cost = best + (sp * elems * 4 / 3.0e9 if sp > 1 else 0.0)
            res.append((cost, f"{tag}/{sp}={best * 1e3:.1f}us(+reduce {cost * 1e3:.1f})"))
    res.sort()
    print(f"{name:8s} {flops / 1e9:6.2f} GF  " + "  ".join(t for _, t in res[:7]), flush=True)
