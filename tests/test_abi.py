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
    assert lib.mireg_version() >= 1
    assert lib.mireg_arch() == b"gfx950"


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    import mireg
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mireg.stn(torch.zeros(1, 2, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mireg.OFEloss([torch.zeros(1, 2, 8, 8)], [torch.zeros(1, 1, 8, 8)], torch.zeros(1, 1, 8, 8))
