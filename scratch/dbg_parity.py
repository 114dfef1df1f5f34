import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from oracle import nets
torch.set_num_threads(16)
shape = (2, 2, 256, 256)
o = nets.FlowNetS(True); nets.analytic_weights_(o); o.train()
x = nets.analytic_input(shape, seed=3)
f32 = [f.detach() for f in o(x)]
o64 = nets.FlowNetS(True); o64.load_state_dict(o.state_dict()); o64 = o64.double(); o64.train()
import oracle.ops as oops
f64 = [f.detach() for f in o64(x.double())]
m = mireg.FlowNetS(True, precision="fp32"); m.load_state_dict(o.state_dict()); m = m.cuda(); m.train()
fm = [f.detach().cpu() for f in m(x.cuda())]
for i, (a, b, c) in enumerate(zip(f32, f64, fm)):
    s = b.abs().max().item()
    print(f"flow[{i}] scale {s:.4f}  torch32-vs-64 {(a.double()-b).abs().max().item():.3e}  mireg32-vs-64 {(c.double()-b).abs().max().item():.3e}  mireg-vs-torch32 {(c-a).abs().max().item():.3e}")
