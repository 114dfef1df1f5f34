"""Diagnosis of the round-2 'segmentation fault while capturing the 3-D step into a hipGraph' (ADVICE): two eager steps, then
one capture attempt with faulthandler on, reporting the first call that is not capturable.  python3 scratch/capture3d.py [small]"""
import faulthandler, sys, os, gc, traceback
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mireg
dev = torch.device("cuda:0")
small = len(sys.argv) > 1
n = 64 if small else 128
g = torch.Generator(device="cpu").manual_seed(6)
low = torch.rand(2 if small else 8, 2, 8, 8, 8, generator=g)
vol = torch.nn.functional.interpolate(low, size=(n, n, n), mode="trilinear", align_corners=False).to(dev)
reg3 = mireg.opticalFlowReg3d(precision="bf16", width_div=4 if small else 1).to(dev).train()
opt = mireg.Adam(reg3.parameters(), 1e-4, eps=1e-4)

def step():
    flows, warped = reg3(vol)
    loss = mireg.OFEloss3d(flows, warped, vol[:, 0:1])[3]
    opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    return loss

for _ in range(2):
    print("eager loss", float(step()), flush=True)
torch.cuda.synchronize()
gc.disable()
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
print("capturing ...", flush=True)
try:
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            loss = step()
    print("capture OK", flush=True)
    gr.replay()
    torch.cuda.synchronize()
    print("first replay: loss", float(loss), flush=True)
    e = list(reg3.predictor._eng.values())[0]
    def chk(tag, t):
        t = t.float()
        n = int((~torch.isfinite(t)).sum())
        if n:
            print(f"   non-finite: {tag}: {n} of {t.numel()}", flush=True)
    for grp in ("raw", "act", "flow32", "gflow", "gcat", "graw", "gact"):
        for k, v in e[grp].items():
            chk(f"{grp}[{k}]", v.buf)
    for k, l in e["layers"].items():
        if getattr(l, "slab", None) is not None:
            chk(f"slab[{k}]", l.slab)
    for k, pair in e.get("gbuf", {}).items():
        for i, t in enumerate(pair):
            if t is not None:
                chk(f"gbuf[{k}][{i}]", t)
    for k, p_ in reg3.named_parameters():
        if p_.grad is not None:
            chk(f"grad {k}", p_.grad)
        chk(f"param {k}", p_)
    print("checked", flush=True)
    for i in range(3):
        gr.replay(); torch.cuda.synchronize()
        bad = [k for k, p in reg3.named_parameters() if not torch.isfinite(p).all()]
        print("replay", i, "loss", float(loss), "non-finite parameters:", bad[:4], flush=True)
except BaseException as e:                                    # noqa: BLE001
    print("capture failed with", type(e).__name__, ":", str(e)[:600], flush=True)
    traceback.print_exc()
