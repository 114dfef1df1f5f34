"""FlowNet2 training step (B=8, 256^2, bf16): eager time, then an attempt to capture the whole step into one hipGraph and replay it.
python3 scratch/capture_fn2.py [nograph]"""
import faulthandler, gc, os, sys, time
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mireg
from mireg.synth import make_pairs
dev = torch.device("cuda:0")
x2, _ = make_pairs(8, 256, seed=3)
x2 = x2.to(dev)
reg2 = mireg.opticalFlowReg("flownet2", precision="bf16").to(dev).train()
opt2 = mireg.Adam(reg2.parameters(), 1e-4, eps=1e-4, fuse=None if os.environ.get("NOFUSE") else reg2)

def step():
    flows, warped, _, _ = reg2(x2)
    loss = mireg.OFEloss(flows, warped, x2[:, 0:1])[3]
    opt2.zero_grad()
    loss.backward()
    opt2.step()
    return loss.detach()

def timed(fn, n=5, warm=2):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
if os.environ.get("TUNE"):
    def fb():
        flows, warped, _, _ = reg2(x2)
        mireg.OFEloss(flows, warped, x2[:, 0:1])[3].backward()
    step()
    t0 = time.perf_counter()
    print("tuned sites:", mireg.autotune(reg2, fb), f"in {time.perf_counter() - t0:.1f} s", flush=True)
t, l = timed(step)
print(f"eager {t:.2f} ms/step  loss {float(l):.3f}", flush=True)
if len(sys.argv) < 2:
    gc.disable()
    gr, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    print("capturing ...", flush=True)
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            l = step()
    gc.enable()
    print("captured", flush=True)
    t, _ = timed(gr.replay)
    print(f"hipGraph replay {t:.2f} ms/step  loss {float(l):.3f}", flush=True)
    bad = [k for k, p in reg2.named_parameters() if not torch.isfinite(p).all()]
    print("non-finite parameters:", bad[:5], flush=True)
