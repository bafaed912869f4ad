"""ctypes binding of libpolygon_hip.so (the C ABI declared in include/polygon_hip.h).

The library is built in-tree by ``polygon_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no
CPU fallback: if the shared object is missing, ``lib()`` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpolygon_hip.so")
PG_F32, PG_BF16 = 0, 1

_lib = None


class PolygonHipError(RuntimeError):
    pass


def lib():
    """Loads (once) and returns the ctypes handle; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise PolygonHipError(
                "libpolygon_hip.so not found at %s -- build it with `make -C polygon_amd/csrc` "
                "(or __graft_entry__.build()); polygon_amd has no CPU fallback" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.pg_last_error.restype = ctypes.c_char_p
        _lib.pg_version.restype = ctypes.c_int
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().pg_last_error().decode("utf-8", "replace")
        raise PolygonHipError("%s failed (%d): %s" % (what or "libpolygon_hip call", status, msg))


def ptr(t):
    """Raw device pointer of a torch tensor (or NULL for None) as c_void_p."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return PG_F32
    if torch_dtype == torch.bfloat16:
        return PG_BF16
    raise ValueError("unsupported compute dtype %r (float32 or bfloat16)" % (torch_dtype,))
