import sys, torch
sys.path.insert(0, ".")
import mireg
from oracle import nets
from mireg.synth import make_pairs
DEV="cuda:0"
def run(graph):
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("pwc", precision="fp32"); nets.analytic_weights_(m)
    p0 = torch.cat([q.detach().reshape(-1) for q in m.parameters()]).clone()
    m = m.to(DEV)
    x, _ = make_pairs(4, 128, seed=3)
    tr = mireg.RegistrationTrainer(m, use_graph=graph, autotune=False)
    for _ in range(4): tr.step(x.to(DEV))
    return (tr.flat_p.cpu() - p0).double()
a, b, c = run(False), run(False), run(True)
cs = torch.nn.functional.cosine_similarity
print("eager vs eager", cs(a, b, dim=0).item(), "eager vs graph", cs(a, c, dim=0).item(), (a-b).abs().max().item())
