"""CPU: the C-ABI library loads and exports every symbol include/mireg.h declares (no compute)."""
import ctypes
import os

from mireg import _lib


def test_library_exports_all_declared_symbols():
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    syms = _lib.declared_symbols()
    assert len(syms) >= 14
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in syms:
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.mireg_version() >= 2          # bumped with the depth / volume / metrics additions to the ABI
    assert lib.mireg_arch() == b"gfx950"


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import mireg
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mireg.stn(torch.zeros(1, 2, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mireg.OFEloss([torch.zeros(1, 2, 8, 8)], [torch.zeros(1, 1, 8, 8)], torch.zeros(1, 1, 8, 8))


def test_ctypes_mirrors_match_the_c_struct_layouts(tmp_path):
    """The job / descriptor structs cross the ABI by value or as device tables: the ctypes mirrors in mireg/engine.py must have
    the C compiler's sizes and field offsets (compiled here with gcc from include/mireg.h)."""
    import shutil
    import subprocess
    import pytest
    from mireg import engine as e
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the HIP headers are not available")
    checks = [("mireg_conv_desc", e.ConvDesc, ["x_ld", "w", "y32", "slab", "n_cls", "cls", "slab_cls_stride", "x_D", "tile_n", "stages", "slab_ld", "algo", "tile_m"]),
              ("mireg_conv_cls", e.ConvCls, ["w", "w_bytes"]),
              ("mireg_pack_job", e.PackJob, ["Cpad", "ld", "cls", "nsplit", "dunit0"]),
              ("mireg_pack3d_job", e.Pack3dJob, ["dst", "Co", "sz", "px", "unit0"]),
              ("mireg_wopt_job", e.WoptJob, ["slab_stride", "g", "F", "Co", "ld", "unit0"]),
              ("mireg_adam_job", e.AdamJob, ["g", "v", "n"]),
              ("mireg_tail_job", e.TailJob, []),
              ("mireg_zero_job", e.ZeroJob, [])]
    lines = []
    for cname, _, fields in checks:
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf(" %zu", offsetof({cname}, {f}));')
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mireg.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    inc = os.path.dirname(_lib.HEADER_PATH)
    subprocess.run(["gcc", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{inc}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (cname, ct, fields), line in zip(checks, out):
        got = [int(v) for v in line.split()[1:]]
        want = [ctypes.sizeof(ct)] + [getattr(ct, f).offset for f in fields]
        assert got == want, (cname, got, want)
