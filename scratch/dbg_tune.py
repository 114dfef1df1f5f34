import sys, torch
sys.path.insert(0, ".")
import mireg
from oracle import nets
from mireg.synth import make_pairs
DEV="cuda:0"
def setup(prec):
    torch.manual_seed(1)
    m = mireg.opticalFlowReg("flownets", precision=prec); nets.analytic_weights_(m)
    return m.to(DEV)
x,_ = make_pairs(2, 64, seed=3); xd = x.to(DEV)
for prec in ("fp32", "bf16"):
    for mode in ("none", "tune", "notune_twice"):
        m = setup(prec)
        tr = mireg.RegistrationTrainer(m, use_graph=False, autotune=(mode == "tune"))
        if mode == "notune_twice":
            tr._setup(xd); tr._autotune()      # pass runs but nothing is tuned (tuning flag is set inside) -> same as tune
        out = [tr.step(xd).tolist()[3] for _ in range(3)]
        print(prec, mode, out, len(tr.eng.ws.tuned))
