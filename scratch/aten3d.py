"""Which ATen ops run inside one FlowNetS-3D training step (the HIP kernels are ours; these are the residue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(6)
low = torch.rand(8, 2, 8, 8, 8, generator=g)
vol = torch.nn.functional.interpolate(low, size=(128, 128, 128), mode="trilinear", align_corners=False).to(dev)
reg3 = mireg.opticalFlowReg3d(precision="bf16").to(dev).train()
opt = mireg.Adam(reg3.parameters(), 1e-4, eps=1e-4, fuse=reg3)
def step():
    flows, warped = reg3(vol)
    loss = mireg.OFEloss3d(flows, warped, vol[:, 0:1])[3]
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:40s} calls {e.count:4d}  device us {e.device_time_total:9.1f}")
