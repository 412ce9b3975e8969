"""TEST INFRASTRUCTURE: ctypes binding of include/layerskip_hip_test.h (liblayerskip_hip_test.so), the single kernels of the
engine on caller-owned device buffers.  The package never loads this library; the isolated kernel tests and the GPU diagnostics
under tools/ do."""
from __future__ import annotations

import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_float, c_int32, c_size_t, c_uint64, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from layerskip_amd import _lib  # noqa: E402

_CSRC = os.path.join(ROOT, "layerskip_amd", "csrc")
LIB_PATH = os.path.join(_CSRC, "liblayerskip_hip_test.so")
LIB_PATH_F16 = os.path.join(_CSRC, "liblayerskip_hip_test_f16.so")

# name -> (restype, argtypes); exactly the symbols include/layerskip_hip_test.h declares
PROTOTYPES = {
    "lsk_test_last_error": (c_char_p, []),
    "lsk_test_abi_version": (c_int32, []),
    "lsk_test_elem_dtype": (c_int32, []),
    "lsk_test_accept_sampled": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                          c_uint64, c_uint64, c_void_p, c_void_p]),
    "lsk_test_gemm": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_float, c_void_p, c_int32, c_void_p]),
    "lsk_test_accept": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "lsk_test_qkv": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_float, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsk_test_swiglu": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_float, c_int32, c_void_p, c_void_p]),
    "lsk_test_resid": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "lsk_test_head_scratch_bytes": (c_int32, [c_int32, POINTER(c_size_t)]),
    "lsk_test_head": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_float, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                c_void_p, c_void_p]),
    "lsk_test_attention_scratch_bytes": (c_int32, [c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "lsk_test_attention": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                     c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_int32, c_void_p]),
}

_LIBS: dict = {}


def load(dtype: str = "bf16", path: str | None = None) -> ctypes.CDLL:
    """Load the test library for a model dtype (once) and type every exported symbol.  Raises if anything is missing."""
    if path is None and dtype in _LIBS:
        return _LIBS[dtype]
    explicit = path is not None
    path = path or (LIB_PATH if dtype == "bf16" else LIB_PATH_F16)
    import torch  # noqa: F401  -- loads the ROCm runtime the library links against
    if not os.path.exists(path):
        raise _lib.LskError(f"test library not built: {path} is missing (python -m layerskip_amd.build)")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise _lib.LskError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lsk_test_abi_version() != _lib.LSK_ABI_VERSION:
        raise _lib.LskError(f"ABI mismatch: test library {lib.lsk_test_abi_version()} vs binding {_lib.LSK_ABI_VERSION}")
    if not explicit:
        _LIBS[dtype] = lib
    return lib


def check(status: int, lib: ctypes.CDLL | None = None) -> None:
    if status != 0:
        msg = (lib or load()).lsk_test_last_error()
        raise _lib.LskError(msg.decode("utf-8", "replace") if msg else f"liblayerskip_hip_test status {status}")
