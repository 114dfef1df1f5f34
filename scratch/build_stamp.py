"""Diagnostic builds (never shipped, never imported by the package): copies of csrc/conv_halo.hip with three s_memtime stamps
(prologue / K loop / epilogue cycles per workgroup, written into desc.slab) and optional ablations of the K loop, compiled to
scratch/libhalo_<variant>.so with one entry point each.  Usage: python scratch/build_stamp.py  (CPU; hipcc cross-compiles)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd", "csrc")
base = open(os.path.join(SRC, "conv_halo.hip")).read()
base = base.replace('#include "mireg_common.h"', f'#include "{SRC}/mireg_common.h"').replace('#include "../../include/mireg.h"', f'#include "{ROOT}/include/mireg.h"')
base = base.replace("mireg_conv_halo_eligible", "stamp_unused_eligible").replace("mireg_conv_halo_try", "stamp_unused_try")

def rep(s, a, b):
    assert s.count(a) == 1, a
    return s.replace(a, b)

def make(variant):
    s = base
    s = rep(s, "  mireg_conv_desc p = pd;\n  const int cls = blockIdx.y;\n",
            "  const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();\n  mireg_conv_desc p = pd;\n  const int cls = blockIdx.y;\n")
    s = rep(s, "    for (int c = 0; c < nchunks; ++c) {\n      const bool more", "    const unsigned long long st_l0 = __builtin_amdgcn_s_memtime();\n    for (int c = 0; c < nchunks; ++c) {\n      const bool more")
    s = rep(s, "    wait_vmcnt<0>();                                                 // padding / past-the-end DMAs still target the ring\n  }\n",
            "    wait_vmcnt<0>();\n  }\n  asm volatile(\"s_nop 0\" ::: \"memory\");\n  const unsigned long long st_l1 = __builtin_amdgcn_s_memtime();\n  const int nsteps = nchunks * TY * TX;\n  const unsigned long long st_l0 = st_l0_;\n")
    s = s.replace("const unsigned long long st_l0 = __builtin_amdgcn_s_memtime();\n    for (int c = 0;", "st_l0_ = __builtin_amdgcn_s_memtime();\n    for (int c = 0;")
    s = rep(s, "  // ---- pipeline: everything per step", "  unsigned long long st_l0_ = 0;\n  // ---- pipeline: everything per step")
    # end of kernel: the last closing of the hp loop
    tail = "        }\n      }\n    }\n  }\n}\n\n// geometry of one class"
    s = rep(s, tail, "        }\n      }\n    }\n  }\n"
            "  { const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n"
            "    if (lane == 0 && pd.slab) { unsigned long long* dbg = reinterpret_cast<unsigned long long*>(pd.slab) + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid) * 8;\n"
            "      dbg[0] = st_l0 - st_k0; dbg[1] = st_l1 - st_l0; dbg[2] = t1 - st_l1; dbg[3] = t1 - st_k0; dbg[4] = r1 - st_r0; dbg[5] = nsteps; dbg[6] = st_k0; dbg[7] = t1; } }\n"
            "}\n\n// geometry of one class")
    s += '\nextern "C" int stamp_launch(const mireg_conv_desc* p, int bm, long tiles_m, hipStream_t stream) {\n  return launch_halo_t<__bf16>(*p, bm, tiles_m, stream);\n}\n'
    out = os.path.join(ROOT, "scratch", f"halo_gen_{variant}.hip")
    open(out, "w").write(s)
    return subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-w", out, "-o",
                             os.path.join(ROOT, "scratch", f"libhalo_{variant}.so")])

VARIANTS = ["full"]
if __name__ == "__main__":
    procs = [make(v) for v in VARIANTS]
    assert all(p.wait() == 0 for p in procs)
    print("built", VARIANTS)
