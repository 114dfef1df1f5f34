import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.synth import make_pairs
DEV = "cuda:0"
x, _ = make_pairs(2, 128, seed=9); x = x.to(DEV)
def build():
    torch.manual_seed(21)
    m = mireg.opticalFlowReg("flownetc", precision="fp32").to(DEV).train()
    return m, mireg.Adam(m.parameters(), 1e-4, eps=1e-4, fuse=m if os.environ.get("FUSE", "1") == "1" else None)
def fwd_bwd(m):
    flows, warped, _, _ = m(x)
    loss = mireg.OFEloss(flows, warped, x[:, 0:1])[3]
    loss.backward()
    return loss.detach()
def step(m, opt):
    opt.zero_grad(); loss = fwd_bwd(m); opt.step(); return float(loss)
(ma, oa), (mb, ob) = build(), build()
print("step0", step(ma, oa), step(mb, ob))
if os.environ.get("TUNE", "1") == "1":
    print("sites", mireg.autotune(mb, lambda: fwd_bwd(mb)))
else:
    fwd_bwd(mb)            # a discarded backward without tuning: grads accumulate, then get zeroed by the next step
    for e in [e for mod in mb.modules() for e in getattr(mod, "_engines", {}).values()]:
        e.slab_pending = False
for i in range(3):
    print("step", i + 1, step(ma, oa), step(mb, ob))
