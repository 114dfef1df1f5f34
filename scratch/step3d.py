"""Time of the full-size 3-D FlowNetS train step (B=8, 128^3, bf16), eager and replayed from one hipGraph.
python3 scratch/step3d.py [nograph]    (env MIREG_3D_WGRAD_MAIN=1: backward-weights on the main stream)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mireg
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(6)
low = torch.rand(8, 2, 8, 8, 8, generator=g)
vol = torch.nn.functional.interpolate(low, size=(128, 128, 128), mode="trilinear", align_corners=False).to(dev)
reg3 = mireg.opticalFlowReg3d(precision="bf16").to(dev).train()
opt = mireg.Adam(reg3.parameters(), 1e-4, eps=1e-4, fuse=None if os.environ.get("NOFUSE") else reg3)

def step():
    flows, warped = reg3(vol)
    loss = mireg.OFEloss3d(flows, warped, vol[:, 0:1])[3]
    opt.zero_grad()
    loss.backward()
    opt.step()
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
t, l = timed(step)
print(f"eager {t:.2f} ms/step  loss {float(l):.1f}", flush=True)
if len(sys.argv) < 2:
    gc.disable()
    gr, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            l = step()
    gc.enable()
    t, _ = timed(gr.replay)
    print(f"hipGraph replay {t:.2f} ms/step  loss {float(l):.1f}", flush=True)
