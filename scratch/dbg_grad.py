import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from oracle import nets
torch.set_num_threads(16)
shape = (2, 2, 256, 256) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))
init = sys.argv[2] if len(sys.argv) > 2 else "analytic"
torch.manual_seed(0)
o = nets.FlowNetS(True)
if init == "analytic": nets.analytic_weights_(o)
o.train()
x = nets.analytic_input(shape, seed=3)
def objective(fl, dev="cpu", dt=torch.float32):
    return sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(dev).to(dt)).sum() for f in fl)
sd = {k: v.clone() for k, v in o.state_dict().items()}
objective(o(x)).backward()
o64 = nets.FlowNetS(True); o64.load_state_dict(sd); o64 = o64.double(); o64.train()
objective(o64(x.double()), dt=torch.float64).backward()
m = mireg.FlowNetS(True, precision="fp32"); m.load_state_dict(sd); m = m.cuda(); m.train()
objective(m(x.cuda()), "cuda").backward()
P32, P64, PM = dict(o.named_parameters()), dict(o64.named_parameters()), dict(m.named_parameters())
for k in P32:
    g64 = P64[k].grad
    s = g64.abs().max().item() + 1e-30
    e32 = ((P32[k].grad.double() - g64).norm() / g64.norm()).item()
    em = ((PM[k].grad.cpu().double() - g64).norm() / g64.norm()).item()
    flag = "  <<<<" if em > 10 * e32 + 1e-4 else ""
    print(f"{k:34s} scale {s:.3e} torch32 {e32:.2e} mireg32 {em:.2e}{flag}")
