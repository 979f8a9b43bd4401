"""ctypes binding of libsifsr_hip.so.

The signatures are parsed from ``include/sifsr_hip.h`` -- the header is the single source of truth
for the C ABI, and ``tests/test_capi_symbols.py`` checks that the library exports every declared
symbol.  There is NO fallback: if the library is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch  # noqa: F401  (must be imported first: the library binds to the libamdhip64 torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "sifsr_hip.h")
# SIFSR_LIB: another build of the same C ABI (same-device A/B of kernel variants, tools/ab/); default: the in-tree library
LIB_PATH = os.environ.get("SIFSR_LIB") or os.path.join(_HERE, "libsifsr_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
}


class SifsrError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype, [(argname, ctype)])} for every SIFSR_API declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"SIFSR_API\s+([\w\s]+?)\s+(\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                if "*" in typ:
                    parsed.append((an, ctypes.c_void_p))
                else:
                    parsed.append((an, _CTYPES[typ.replace("const ", "")]))
        decls[name] = (_CTYPES[ret], parsed)
    return decls


_lib = None
_decls = None


def lib():
    """Load (once) and return the ctypes handle; raises SifsrError if the HIP library is absent."""
    global _lib, _decls
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SifsrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        _decls = parse_header()
        for name, (ret, args) in _decls.items():
            fn = getattr(handle, name)     # AttributeError if the library lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for _, t in args]
        _lib = handle
    return _lib


def declared_symbols():
    return sorted(parse_header().keys())


def _conv(v):
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        return v.data_ptr()
    return v


def call(name: str, *args):
    """Call a C-ABI function; tensors are passed as device pointers.  Raises on a non-zero status."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv(a) for a in args])
    if fn.restype is ctypes.c_int and rc != 0 and not name.startswith(("sifsr_abi", "sifsr_num", "sifsr_layer", "sifsr_huber_partial", "sifsr_model_workspace_regions", "sifsr_conv3x3_stat", "sifsr_conv_in_stat", "sifsr_conv3x3_bwd16_stat", "sifsr_profile_add", "sifsr_up2x_bwd_stat")):
        raise SifsrError(f"{name} failed with status {rc}")
    return rc


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise SifsrError(f"{what} is on {t.device}: this package only runs on a ROCm GPU (gfx950); "
                         "there is no CPU path.")
    if t.dtype != torch.float32:
        raise SifsrError(f"{what} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise SifsrError(f"{what} must be contiguous")
