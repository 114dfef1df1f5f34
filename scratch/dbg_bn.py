import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from mireg.engine import BatchNormAct, Workspace, View
DEV = "cuda:0"
ws = Workspace(torch.device(DEV), torch.float32)
g = torch.Generator().manual_seed(3)
for C, H, B, wide in ((512, 8, 4, 1032), (512, 16, 4, 776), (256, 32, 4, 392), (512, 8, 4, 512), (128, 8, 2, 200)):
    y = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C)
    bn_d = torch.nn.BatchNorm2d(C).to(DEV)
    yr = y.clone().requires_grad_()
    out_ref = F.leaky_relu(bn(yr), 0.1)
    cot = torch.randn(out_ref.shape, generator=g)
    (out_ref * cot).sum().backward()
    op = BatchNormAct(bn_d, ws)
    yv = ws.new(B, H, H, C); yv.buf[..., :C] = y.permute(0, 2, 3, 1).to(DEV)
    wide_o = ws.new(B, H, H, wide - 6)
    ov = wide_o.slice(0, C)
    op.forward(yv, ov, True)
    e1 = ((ov.nchw().cpu() - out_ref.detach()).abs().max() / out_ref.abs().max()).item()
    wide_d = ws.new(B, H, H, wide - 6)
    dav = wide_d.slice(0, C)
    dav.buf[..., :C] = cot.permute(0, 2, 3, 1).to(DEV)
    dyv = ws.new(B, H, H, C)
    op.backward(yv, dav, dyv)
    e2 = ((dyv.nchw().cpu() - yr.grad).abs().max() / yr.grad.abs().max()).item()
    e3 = ((op.grad_b.cpu() - bn.bias.grad).abs().max() / bn.bias.grad.abs().max()).item()
    print(C, H, B, wide, "fwd", e1, "bwd", e2, "dbeta", e3)
print("---- sums check")
for C, H, B in ((256, 32, 4), (256, 16, 4), (256, 32, 2), (64, 64, 4), (512, 32, 4)):
    y = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    bn_d = torch.nn.BatchNorm2d(C).to(DEV)
    op = BatchNormAct(bn_d, ws)
    yv = ws.new(B, H, H, C); yv.buf[..., :C] = y.permute(0, 2, 3, 1).to(DEV)
    ov = ws.new(B, H, H, C)
    op.forward(yv, ov, True)
    cot = torch.randn(B, C, H, H, generator=g)
    dav = ws.new(B, H, H, C); dav.buf[..., :C] = cot.permute(0, 2, 3, 1).to(DEV)
    dyv = ws.new(B, H, H, C)
    op.backward(yv, dav, dyv)
    yd = yv.buf.double().reshape(-1, C); dad = dav.buf.double().reshape(-1, C)
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    xh = (yd - mean) / (var + 1e-5).sqrt()
    dz = dad * torch.where(xh > 0, 1.0, 0.1)
    s0, s1 = dz.sum(0), (dz * xh).sum(0)
    got = op.sums.reshape(2, C)
    print(C, H, B, "M", yd.shape[0], "sum dz err", ((got[0] - s0).abs().max() / s0.abs().max()).item(), "sum dz*xh err", ((got[1] - s1).abs().max() / s1.abs().max()).item(),
          "worst channel", (got[0] - s0).abs().argmax().item())
