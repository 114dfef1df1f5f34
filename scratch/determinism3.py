"""Stress of the tiny 2->2 upsampler backward-data kernel (RMW on its output) under a busy second stream."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg import _lib, engine
dev = "cuda:0"
B, Hc = 24, 16
ldf, ldc = 392, 8
side = torch.cuda.Stream()
w = (torch.randn(2, 2, 4, 4, device=dev) * 0.3).contiguous()
mode = sys.argv[1] if len(sys.argv) > 1 else "mm"
A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
big = torch.empty(64 << 20, device=dev)
for trial in range(3):
    bad_total = 0
    for it in range(300):
        gf = torch.randn(B, 2 * Hc, 2 * Hc, ldf, device=dev).bfloat16()
        g = torch.randn(B, Hc, Hc, ldc, device=dev).bfloat16()
        d = torch.full((B, Hc, Hc, ldc), 7.0, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if mode == "mm":
                for _ in range(3): A @ A
            elif mode == "fill":
                for _ in range(6): big.fill_(1.0)
        d.copy_(g)                                                  # the loss gradient lands ...
        _lib.call("mireg_tiny_deconv_bwd_data", gf.data_ptr() + 2 * 384, ldf, w.data_ptr(), d.data_ptr(), ldc, 1, B, Hc, Hc,
                  engine.DT_BF16, torch.cuda.current_stream().cuda_stream)
        out = d.clone()
        torch.cuda.synchronize()
        conv = F.conv2d(gf[..., 384:386].float().permute(0, 3, 1, 2), w, None, 2, 1).permute(0, 2, 3, 1)
        exp = (g[..., :2].float() + conv).bfloat16().float()
        diff = (out[..., :2].float() - exp).abs()
        tol = 2.0 ** -7 * exp.abs().clamp_min(1e-3)
        bad = int((diff > tol).sum())
        bad_total += bad
        if bad and bad_total == bad:
            nz = torch.nonzero(diff > tol)[:8].tolist()
            print("first bad iteration", it, "positions", nz, flush=True)
    print(mode, "trial", trial, "bad elements over 300 launches:", bad_total, flush=True)
