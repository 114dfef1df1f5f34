"""micro-bench of the conv contraction kernels: python scratch/mb_conv.py [fwd|dgrad|wgrad] cin cout k s H B [prec]"""
import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mireg
from mireg.engine import ConvLayer, Workspace, upload_table, _stream
from mireg import _lib
if os.environ.get('MIREG_LIB'): _lib.LIB_PATH = os.environ['MIREG_LIB']
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cin, cout, k, s, H, B = [int(v) for v in (sys.argv[2:8] if len(sys.argv) > 7 else (256, 256, 3, 1, 32, 24))]
prec = sys.argv[8] if len(sys.argv) > 8 else "bf16"
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
dt = torch.bfloat16 if prec == "bf16" else torch.float32
ws = Workspace(torch.device("cuda:0"), dt)
w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
lay = ConvLayer("t", w, None, s, (k - 1) // 2, 1, ws)
from mireg.engine import run_pack
run_pack(lay.pack_jobs(), ws.code, "cuda:0")
x = ws.new(B, H, H, cin); x.buf.normal_()
Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
y = ws.new(B, Ho, Ho, cout); y.buf.normal_()
dx = ws.new(B, H, H, cin)
import os
ABL = int(os.environ.get("ABL", "0"))
def run():
    if mode == "fwd": lay.run_fwd_form(x, y, accumulate=ABL)
    elif mode == "dgrad": lay.run_dgrad_form(y, dx)
    else: lay.run_wgrad(x, y)
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
fl = 2.0 * B * Ho * Ho * cout * cin * k * k
print(f"{mode} cin{cin} cout{cout} k{k} s{s} H{H} B{B} {prec}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  split={lay.wgrad_split if mode=='wgrad' else '-'}")
