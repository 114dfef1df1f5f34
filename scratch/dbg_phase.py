import sys, torch
sys.path.insert(0, ".")
import mireg
from oracle import nets
from mireg.synth import make_pairs
DEV = "cuda:0"
def run(ov, prec="fp32"):
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("flownets", precision=prec); nets.analytic_weights_(m); m = m.to(DEV)
    x, _ = make_pairs(4, 128, seed=3); xd = x.to(DEV)
    tr = mireg.RegistrationTrainer(m, use_graph=False, autotune=False, overlap_optimizer=ov)
    l1 = tr.step(xd).tolist(); torch.cuda.synchronize()
    p1 = {k: v.detach().clone() for k, v in m.named_parameters()}
    l2 = tr.step(xd).tolist()
    return l1, l2, p1, tr
la1, la2, pa, _ = run(False)
lb1, lb2, pb, trb = run(True)
print(la1[3], lb1[3], la2[3], lb2[3])
for k in pa:
    d = (pa[k] - pb[k]).abs().max().item()
    if d > 1e-7: print(k, d, pa[k].abs().max().item())
print("step", trb.step_dev.item())
