"""ctypes binding of libmireg_hip.so (the C ABI declared in include/mireg.h).

There is NO fallback: if the shared object is missing or a symbol is absent the
first op call fails with a RuntimeError that says how to build it.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmireg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mireg.h")

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_CTYPE = {"int": _I, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
          "hipStream_t": _P, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "unsigned": ctypes.c_uint}


def declared_symbols(header: str = HEADER_PATH):
    """Parse include/mireg.h -> {name: (restype, [argtypes])}; every pointer becomes c_void_p."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(mireg_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(_P)
                else:
                    types.append(_CTYPE[a.replace("const ", "").split()[0]])
        out[name] = (ctypes.c_char_p if "char" in ret else _I, types)
    return out


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"mireg: {LIB_PATH} not found. The HIP extension is mandatory (no CPU/ATen fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C <pkg>/csrc`.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (ret, types) in declared_symbols().items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"mireg: symbol {name} declared in include/mireg.h is missing from {LIB_PATH}") from e
            fn.restype = ret
            fn.argtypes = types
        _lib = handle
    return _lib


_ERR = {-1: "invalid argument (shape / stride / alignment / null pointer)", -2: "kernel launch failed",
        -3: "unsupported configuration"}


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"mireg.{name} failed: {_ERR.get(rc, rc)}")
