"""Two identically seeded FlowNet2 registration models trained N steps through torch.autograd: parameters must stay bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.synth import make_pairs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = make_pairs(4, 256, seed=6)[0].cuda()
finals = []
for rep in range(2):
    torch.manual_seed(1)
    reg = mireg.opticalFlowReg("flownet2", precision="bf16")
    reg = reg.cuda().train()
    opt = mireg.Adam(reg.parameters(), 1e-4, eps=1e-4)
    for st in range(steps):
        flows, warped, _, _ = reg(x)
        loss = mireg.OFEloss(flows, warped, x[:, 0:1])[3]
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    finals.append(torch.cat([p.detach().flatten() for p in reg.parameters()]).clone())
    del reg, opt
    torch.cuda.empty_cache()
d = (finals[0] - finals[1]).abs()
print(f"flownet2 B=4: {steps} steps, parameters bit-identical: {bool(torch.equal(finals[0], finals[1]))}, differing {int((d > 0).sum())} of {d.numel()}, max {float(d.max()):.3e}, loss {float(loss):.3f}")
