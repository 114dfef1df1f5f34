"""Three train steps of FlowNetS over 128^3 volume pairs (batch 8) for rocprofv3 --kernel-trace --stats."""
import sys, time
sys.path.insert(0, ".")
import torch
import mireg

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(6)
low = torch.rand(8, 2, 8, 8, 8, generator=g)
vol = torch.nn.functional.interpolate(low, size=(128, 128, 128), mode="trilinear", align_corners=False).to(dev)
reg3 = mireg.opticalFlowReg3d(precision="bf16").to(dev).train()
opt = mireg.Adam(reg3.parameters(), 1e-4, eps=1e-4, fuse=reg3)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flows, warped = reg3(vol)
    loss = mireg.OFEloss3d(flows, warped, vol[:, 0:1])[3]
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms  loss {loss.item():.1f}", flush=True)
