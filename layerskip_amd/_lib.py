"""ctypes binding of include/layerskip_hip.h (the C-ABI drop-in boundary).

The shared library is built in-tree by ``layerskip_amd/build.py`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing or a symbol is absent, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_size_t, c_uint64, c_void_p

LSK_MAX_ROWS = 16
LSK_MAX_SPEC = 15
LSK_MAX_EOS = 1024
LSK_ABI_VERSION = 4
LSK_OPT_BIG_THRESHOLD = 1
LSK_OPT_TARGET_WGS = 2
LSK_OPT_FUSED_ATTN = 3
LSK_OPT_FLASH_PREFILL = 5
LSK_OPT_GRAPH_STEPS = 7       # steady-state greedy steps replayed from hipGraphs (default off)

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "liblayerskip_hip.so")            # bf16 build (BASELINE configs)
LIB_PATH_F16 = os.path.join(_CSRC, "liblayerskip_hip_f16.so")    # fp16 build (same sources, -DLSK_ELEM_F16)
DTYPE_CODES = {"bf16": 0, "fp16": 1}


class LskConfig(ctypes.Structure):
    _fields_ = [
        ("num_layers", c_int32), ("hidden", c_int32), ("intermediate", c_int32), ("n_heads", c_int32),
        ("n_kv_heads", c_int32), ("head_dim", c_int32), ("vocab", c_int32), ("rms_eps", c_float),
        ("max_ctx", c_int32), ("page_size", c_int32), ("max_prompt", c_int32), ("target_wgs", c_int32),
    ]


class LskStepResult(ctypes.Structure):
    _fields_ = [
        ("num_matches", c_int32), ("num_drafts", c_int32), ("num_emitted", c_int32), ("next_token", c_int32),
        ("kv_len", c_int32),
        ("emitted", c_int32 * (LSK_MAX_ROWS + 1)),
        ("draft_tokens", c_int32 * LSK_MAX_ROWS),
        ("verified_tokens", c_int32 * (LSK_MAX_ROWS + 1)),
    ]


# name -> (restype, argtypes); exactly the symbols include/layerskip_hip.h declares
PROTOTYPES = {
    "lsk_last_error": (c_char_p, []),
    "lsk_abi_version": (c_int32, []),
    "lsk_elem_dtype": (c_int32, []),
    "lsk_workspace_bytes": (c_int32, [POINTER(LskConfig), POINTER(c_size_t)]),
    "lsk_kv_pool_bytes": (c_int32, [POINTER(LskConfig), POINTER(c_size_t)]),
    "lsk_packed_bytes": (c_int32, [c_int32, c_int32, POINTER(c_size_t)]),
    "lsk_pack_linear": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_engine_create": (c_int32, [POINTER(LskConfig), c_void_p, c_size_t, c_void_p, c_size_t, POINTER(c_void_p)]),
    "lsk_engine_destroy": (c_int32, [c_void_p]),
    "lsk_engine_set_layer": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lsk_engine_set_globals": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32]),
    "lsk_engine_weights_checksum": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(ctypes.c_int64), c_int32, POINTER(c_uint64), c_void_p]),
    "lsk_engine_set_block_table": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_void_p]),
    "lsk_engine_reset": (c_int32, [c_void_p, c_void_p]),
    "lsk_engine_set_kv_len": (c_int32, [c_void_p, c_int32, c_void_p]),
    "lsk_engine_get_kv_len": (c_int32, [c_void_p, POINTER(c_int32)]),
    "lsk_spec_step": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, POINTER(c_int32), c_int32,
                                POINTER(LskStepResult), c_void_p]),
    "lsk_spec_generate": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, POINTER(c_int32), c_int32, c_int32,
                                    POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                    POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "lsk_ar_step": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, POINTER(c_int32), c_void_p]),
    "lsk_ar_generate": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, POINTER(c_int32), c_int32, c_int32,
                                  POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "lsk_draft_block": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_engine_set_eos": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_void_p]),
    "lsk_pipeline_pack": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_pipeline_apply": (c_int32, [c_void_p, c_int32, c_void_p]),
    "lsk_pipeline_tail": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p]),
    "lsk_pipeline_result_words": (c_int32, [POINTER(LskConfig), POINTER(c_int32)]),
    "lsk_draft_block_sampled": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32,
                                          c_float, c_uint64, c_uint64, c_void_p, c_size_t, c_void_p]),
    "lsk_pipeline_pack_sampled": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint64, c_void_p, c_size_t, c_void_p]),
    "lsk_pipeline_tail_sampled": (c_int32, [c_void_p, c_int32, c_float, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_size_t, c_void_p,
                                            c_int32, c_void_p]),
    "lsk_pipeline_residual": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_uint64, c_uint64, c_void_p, c_size_t, c_void_p]),
    "lsk_get_row_tokens": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_int32), c_void_p]),
    "lsk_shift_rows": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_rows_offset": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_size_t)]),
    "lsk_embed_rows": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, c_void_p]),
    "lsk_run_layers": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_run_bulk": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "lsk_engine_set_option": (c_int32, [c_void_p, c_int32, c_int32]),
    "lsk_run_head": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, POINTER(c_int32), c_void_p]),
    "lsk_sampling_scratch_bytes": (c_int32, [POINTER(LskConfig), POINTER(c_size_t)]),
    "lsk_sample_rows": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_int32, c_float, c_uint64, c_uint64, c_int32,
                                  c_void_p, c_void_p, c_void_p]),
    "lsk_spec_step_sampled": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, POINTER(c_int32), c_int32, c_float,
                                        c_int32, c_float, c_uint64, c_uint64, c_void_p, c_size_t, POINTER(LskStepResult), c_void_p]),
    "lsk_spec_generate_sampled": (c_int32, [c_void_p, POINTER(c_int32), c_int32, c_int32, c_int32, POINTER(c_int32), c_int32, c_int32,
                                            c_float, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_size_t, POINTER(c_int32),
                                            POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                            POINTER(c_int32), c_void_p]),
    "lsk_read_rows": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "lsk_write_rows": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "lsk_time_gateup": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_float), c_void_p]),
    "lsk_engine_get_host_stats": (c_int32, [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_int64)]),
    "lsk_engine_set_profile": (c_int32, [c_void_p, c_int32]),
    "lsk_engine_get_profile": (c_int32, [c_void_p, POINTER(c_float), POINTER(c_int32)]),
    "lsk_engine_get_profile_table": (c_int32, [c_void_p, c_int32, POINTER(c_float), POINTER(c_int32), POINTER(ctypes.c_double)]),
}

_LIBS: dict = {}


class LskError(RuntimeError):
    """An entry point of liblayerskip_hip.so returned a non-zero status."""


def load(path: str | None = None, dtype: str = "bf16") -> ctypes.CDLL:
    """Load the HIP extension for a model dtype (once) and type every exported symbol.  Raises if anything is missing."""
    if dtype not in DTYPE_CODES:
        raise LskError(f"unsupported model dtype {dtype!r} (bf16 or fp16)")
    if path is None and dtype in _LIBS:
        return _LIBS[dtype]
    explicit = path is not None
    path = path or (LIB_PATH if dtype == "bf16" else LIB_PATH_F16)
    import torch  # noqa: F401  -- loads the ROCm runtime (libamdhip64.so.7) the extension links against
    if not os.path.exists(path):
        raise LskError(
            f"HIP extension not built: {path} is missing. Run `python -m layerskip_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback for the decoding engine.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise LskError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lsk_abi_version() != LSK_ABI_VERSION:
        raise LskError(f"ABI mismatch: library {lib.lsk_abi_version()} vs binding {LSK_ABI_VERSION}")
    if not explicit and lib.lsk_elem_dtype() != DTYPE_CODES[dtype]:
        raise LskError(f"{path} computes in dtype code {lib.lsk_elem_dtype()}, expected {dtype}")
    if not explicit:
        _LIBS[dtype] = lib
    return lib


def check(status: int, lib: ctypes.CDLL | None = None) -> None:
    """Raise LskError with the library's (thread-local) message for a non-zero status."""
    if status != 0:
        msg = (lib or load()).lsk_last_error()
        raise LskError(msg.decode("utf-8", "replace") if msg else f"liblayerskip_hip status {status}")
