import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from oracle import nets
torch.set_num_threads(16)
shape = (4, 2, 256, 256)
torch.manual_seed(0)
o = nets.FlowNetS(True); o.train()
sd = {k: v.clone() for k, v in o.state_dict().items()}
x = nets.analytic_input(shape, seed=3)
def objective(fl, dev="cpu", dt=torch.float32):
    return sum((f * torch.cos(torch.arange(f.numel(), dtype=torch.float32).reshape(f.shape) * 0.01).to(dev).to(dt)).sum() for f in fl)
o64 = nets.FlowNetS(True); o64.load_state_dict(sd); o64 = o64.double(); o64.train()
grads = {}
acts = {}
def hook(name):
    def f(mod, inp, out):
        acts[name] = out.detach()
        out.register_hook(lambda g: grads.__setitem__(name, g.detach()))
    return f
for n in ("conv1","conv2","conv3","conv3_1","conv4","conv4_1","conv5","conv5_1","conv6","conv6_1","deconv5","deconv4","deconv3","deconv2"):
    getattr(o64, n).register_forward_hook(hook(n))
for n in ("conv5_1","conv4_1"):
    getattr(o64, n)[0].register_forward_hook(hook(n + ".raw"))
objective(o64(x.double()), dt=torch.float64).backward()
m = mireg.FlowNetS(True, precision="fp32"); m.load_state_dict(sd); m = m.cuda(); m.train()
objective(m(x.cuda()), "cuda").backward()
e = next(iter(m._engines.values()))
def cmp(name, view, ref):
    a = view.nchw().double().cpu()
    s = ref.abs().max().item()
    print(f"{name:28s} scale {s:.3e} relerr {(a-ref).abs().max().item()/s:.3e}")
cmp("act conv5_1 (cat5[0:512])", e.cat[5].slice(0,512), acts["conv5_1"])
cmp("raw conv5_1", e.raw["conv5_1"], acts["conv5_1.raw"])
cmp("d act conv6_1", e.da61, grads["conv6_1"])
cmp("d act conv6", e.da6, grads["conv6"])
cmp("d act conv5_1 dcat5[0:512]", e.dcat[5].slice(0,512), grads["conv5_1"])
cmp("d raw conv5_1", e.draw["conv5_1"], grads["conv5_1.raw"])
cmp("d act conv5", e.da5, grads["conv5"])
cmp("d act conv4_1 dcat4[0:512]", e.dcat[4].slice(0,512), grads["conv4_1"])
cmp("d raw conv4_1", e.draw["conv4_1"], grads["conv4_1.raw"])
cmp("d act conv4", e.da4, grads["conv4"])
cmp("d act conv3_1", e.dcat[3].slice(0,256), grads["conv3_1"])
cmp("d act conv2", e.dcat[2].slice(0,128), grads["conv2"])
cmp("d act conv1", e.da1, grads["conv1"])
