"""Diagnostic build (never shipped): a copy of csrc/conv_gemm.hip whose forward / backward-data ring kernel writes four s_memtime
stamps per wave (start, K loop begin, K loop end, end) and the wall-clock start / end into the buffer handed over in desc.y32
(split-K launches only; the reduce launch is disabled).  python scratch/build_stamp_ring.py  ->  scratch/libring_stamp.so"""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd", "csrc")
s = open(os.path.join(SRC, "conv_gemm.hip")).read()
s = s.replace('#include "mireg_common.h"', f'#include "{SRC}/mireg_common.h"').replace('#include "../../include/mireg.h"', f'#include "{ROOT}/include/mireg.h"')
for name in ("mireg_conv_gemm", "mireg_conv_wgrad"):
    s = s.replace(f"int {name}(", f"int stampcopy_{name[6:]}(")


def rep(s, a, b):
    assert s.count(a) == 1, (s.count(a), a[:60])
    return s.replace(a, b)


s = rep(s, "  mireg_conv_desc p = pd;\n  const int cls = blockIdx.y;\n  if (pd.n_cls > 1) {\n    const mireg_conv_cls k = pd.cls[cls];\n    p.taps_y",
        "  const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();\n  mireg_conv_desc p = pd;\n  const int cls = blockIdx.y;\n  if (pd.n_cls > 1) {\n    const mireg_conv_cls k = pd.cls[cls];\n    p.taps_y")
s = rep(s, "  int it = 0;\n  const int steady = nk - (STAGES - 1);\n  if (loads_per_tile == A_PW + B_PW) {",
        "  int it = 0;\n  const int steady = nk - (STAGES - 1);\n  const unsigned long long st_l0 = __builtin_amdgcn_s_memtime();\n  if (loads_per_tile == A_PW + B_PW) {")
s = rep(s, "  // ---- epilogue: accumulators -> LDS (fp32 [BM][BN], reusing the ring) -> 16-byte coalesced row stores ------\n  __syncthreads();",
        "  const unsigned long long st_l1 = __builtin_amdgcn_s_memtime();\n  __syncthreads();")
s = rep(s, "      else { const float vv[4] = {v.x, v.y, v.z, v.w}; for (int q = 0; q < 4 && n + q < p.N; ++q) d[q] = vv[q]; }\n    }\n    return;\n  }\n",
        "      else { const float vv[4] = {v.x, v.y, v.z, v.w}; for (int q = 0; q < 4 && n + q < p.N; ++q) d[q] = vv[q]; }\n    }\n"
        "    { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n"
        "      if (lane == 0 && pd.y32) { unsigned long long* dbg = reinterpret_cast<unsigned long long*>(pd.y32) + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wid) * 8;\n"
        "        dbg[0] = st_l0 - st_k0; dbg[1] = st_l1 - st_l0; dbg[2] = t1 - st_l1; dbg[3] = nk; dbg[4] = st_r0; dbg[5] = r1; dbg[6] = bid; dbg[7] = 1; } }\n"
        "    return;\n  }\n")
s = rep(s, "  if (z > 1) {\n    const long total = M * p.N;", "  if (false) {\n    const long total = M * p.N;")
s += '\nextern "C" int stamp_launch_ring(const mireg_conv_desc* p, hipStream_t stream) {\n  return launch_fwd<__bf16>(*p, stream);\n}\n'
out = os.path.join(ROOT, "scratch", "ring_gen_stamp.hip")
open(out, "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-w", out, "-o",
                       os.path.join(ROOT, "scratch", "libring_stamp.so")])
print("built scratch/libring_stamp.so")
