"""ctypes loader for libtvts_hip.so (the C-ABI HIP kernel library).

The prototypes are read from ``include/tvts_hip.h`` so the binding can never drift from the
header.  There is NO fallback: if the library is missing or a symbol is absent this module raises,
and every op in ``tvts_amd`` goes through it (the product path must fail loudly without the HIP
extension).
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtvts_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tvts_hip.h")
COMM_LIB_PATH = os.path.join(_HERE, "libtvts_comm.so")
COMM_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tvts_comm.h")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "hipStream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long|void)\s+(tvts_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            argnames.append(re.split(r"[\s\*]+", a)[-1])
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            else:
                ty = a.replace("const", "").split()[0]
                argtypes.append(_CTYPES[ty])
        protos[name] = ({"int": ctypes.c_int, "long": ctypes.c_long}.get(ret), argtypes, argnames)
    return protos


class HipLibraryMissing(RuntimeError):
    pass


_lib = None
_protos = None


def load():
    """Load the library (once) and bind every prototype of the header.  Raises if anything is missing."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- load PyTorch-ROCm's HIP runtime first so both sides share ONE libamdhip64
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(tvts_amd has no CPU / PyTorch fallback path)")
    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    for name, (res, argtypes, _) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing(f"{LIB_PATH} does not export {name} declared in {HEADER_PATH}") from e
        fn.restype = res
        fn.argtypes = argtypes
    _lib, _protos = lib, protos
    return lib


def prototypes():
    load()
    return _protos


_comm = None


def load_comm():
    """libtvts_comm.so (include/tvts_comm.h): the RCCL exchange steps on a library-owned side stream.  Loaded on demand."""
    global _comm
    if _comm is not None:
        return _comm
    import torch  # noqa: F401
    if not os.path.exists(COMM_LIB_PATH):
        raise HipLibraryMissing(f"{COMM_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(COMM_LIB_PATH)
    for name, (res, argtypes, _) in parse_header(COMM_HEADER_PATH).items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing(f"{COMM_LIB_PATH} does not export {name} declared in {COMM_HEADER_PATH}") from e
        fn.restype = res
        fn.argtypes = argtypes
    _comm = lib
    return lib
