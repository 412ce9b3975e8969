"""The C-ABI library loads on a GPU-less box and exports exactly what include/layerskip_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols(header="layerskip_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from layerskip_amd import _lib, build
    build.build()
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no symbols parsed from the header"
    assert sorted(_lib.PROTOTYPES) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lsk_abi_version() == _lib.LSK_ABI_VERSION
    assert lib.lsk_elem_dtype() == 0
    # the fp16 build of the same sources exports the same boundary
    lib16 = _lib.load(dtype="fp16")
    for name in declared:
        assert hasattr(lib16, name), name
    assert lib16.lsk_elem_dtype() == 1 and lib16.lsk_abi_version() == _lib.LSK_ABI_VERSION
    assert lib16 is not lib


def test_test_library_exports_its_header_and_the_product_library_holds_no_test_code():
    """liblayerskip_hip_test.so = include/layerskip_hip_test.h; nothing named lsk_test_* lives in the product library, and the
    package's binding does not know the test library."""
    import lsk_test_lib
    from layerskip_amd import _lib
    declared = _header_symbols("layerskip_hip_test.h")
    assert declared and all(n.startswith("lsk_test_") for n in declared)
    assert sorted(lsk_test_lib.PROTOTYPES) == declared
    for dtype, code in (("bf16", 0), ("fp16", 1)):
        tlib = lsk_test_lib.load(dtype)
        for name in declared:
            assert hasattr(tlib, name), name
        assert tlib.lsk_test_abi_version() == _lib.LSK_ABI_VERSION and tlib.lsk_test_elem_dtype() == code
        assert not hasattr(tlib, "lsk_engine_create")
    lib = _lib.load()
    for name in declared:
        assert not hasattr(lib, name), f"{name} is exported by the product library"
    assert not any(n.startswith("lsk_test_") for n in _lib.PROTOTYPES)
    src = open(os.path.join(ROOT, "layerskip_amd", "_lib.py")).read() + open(os.path.join(ROOT, "layerskip_amd", "engine.py")).read()
    assert "hip_test" not in src


def test_size_queries_need_no_gpu():
    from layerskip_amd import _lib
    lib = _lib.load()
    cfg = _lib.LskConfig(32, 4096, 11008, 32, 32, 128, 32000, 1e-5, 1152, 128, 512, 0)
    ws, kv, pk = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.check(lib.lsk_workspace_bytes(ctypes.byref(cfg), ctypes.byref(ws)))
    _lib.check(lib.lsk_kv_pool_bytes(ctypes.byref(cfg), ctypes.byref(kv)))
    _lib.check(lib.lsk_packed_bytes(11008, 4096, ctypes.byref(pk)))
    assert kv.value == 32 * 2 * 1152 * 32 * 128 * 2
    assert pk.value == 11008 * 4096 * 2
    assert ws.value > 16 * 4096 * 2
    # ragged N pads to a whole 16-row tile
    _lib.check(lib.lsk_packed_bytes(1000, 512, ctypes.byref(pk)))
    assert pk.value == 1008 * 512 * 2


def test_errors_are_status_codes_with_messages():
    from layerskip_amd import _lib
    lib = _lib.load()
    bad = _lib.LskConfig(32, 4096, 11008, 32, 32, 96, 32000, 1e-5, 1152, 128, 512, 0)   # head_dim 96
    out = ctypes.c_size_t(0)
    assert lib.lsk_workspace_bytes(ctypes.byref(bad), ctypes.byref(out)) != 0
    assert b"head_dim" in lib.lsk_last_error()
    assert lib.lsk_packed_bytes(16, 100, ctypes.byref(out)) != 0          # k not a multiple of 32
    try:
        _lib.check(lib.lsk_packed_bytes(16, 100, ctypes.byref(out)))
    except _lib.LskError as exc:
        assert "multiple of 32" in str(exc)
    else:
        raise AssertionError("expected LskError")


def test_engine_refuses_cpu_models():
    """No silent CPU fallback: a model that is not on a HIP device is an error, not a slow path."""
    import pytest
    import torch
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import HipEngine
    model = synthetic.build_model(synthetic.make_config("tiny-mha"), seed=0, exit_layer=2)
    with pytest.raises(_lib.LskError):
        HipEngine(model)
    assert not torch.cuda.is_available() or True


def test_error_messages_come_from_the_library_that_failed():
    """Two libraries (bf16 / fp16) are loaded side by side, each with its own thread-local message."""
    from layerskip_amd import _lib
    a, b = _lib.load(dtype="bf16"), _lib.load(dtype="fp16")
    out = ctypes.c_size_t(0)
    assert a.lsk_packed_bytes(16, 100, ctypes.byref(out)) != 0
    assert b.lsk_packed_bytes(-1, 64, ctypes.byref(out)) != 0
    import pytest
    with pytest.raises(_lib.LskError, match="k=100"):
        _lib.check(1, a)
    with pytest.raises(_lib.LskError, match="n_rows=-1"):
        _lib.check(1, b)


def test_kernel_resources_of_the_product_build():
    """Two properties of the compiled kernels no parity test sees (read from the kernel descriptors in the device assembly of the product
    sources, built with the product flags):
    * every kernel stays in registers (ScratchSize 0).  A spill changes no result, only the speed: the 16-row templates of the
      skinny projection kernel sit at the 256-register limit and an innocent-looking edit of the prologue pushed them into scratch
      (llama2-13B verify passes 30 % slower);
    * the decode kernels get their first 14 argument dwords preloaded into SGPRs (kernel descriptor field
      kernarg_preload_length): without it every launch starts with a ~1 us scalar round trip (profiles/r03_kernel_timeline.md)."""
    import re
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from layerskip_amd import build
    csrc = os.path.join(ROOT, "layerskip_amd", "csrc")

    def device_asm(src):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "x.s")
            proc = subprocess.run(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["--cuda-device-only", "-S", "-o", out, os.path.join(csrc, src)],
                                  capture_output=True, text=True)
            assert proc.returncode == 0, proc.stderr[-2000:]
            return open(out).read()

    with ThreadPoolExecutor(max_workers=len(build.SOURCES)) as pool:
        texts = list(pool.map(device_asm, build.SOURCES))
    kernels = {}
    for text in texts:
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
            scratch = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
            pre = re.search(r"\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", m.group(2))
            kernels[m.group(1)] = {"scratch": int(scratch.group(1)), "preload": int(pre.group(1)) if pre else 0}
    assert len(kernels) >= 40
    spilled = {k: v["scratch"] for k, v in kernels.items() if v["scratch"] != 0}
    assert not spilled, spilled
    decode = {k: v["preload"] for k, v in kernels.items() if "lsk_gemm_kernel" in k or "lsk_attn_split_kernel" in k}
    assert len(decode) >= 20 and all(n == 14 for n in decode.values()), decode


def test_the_timeline_instrumentation_still_builds():
    """tools/kernel_timeline.py needs a -DLSK_TRACE build of the engine (in-kernel time stamps, lsk_common.h); the product build
    compiles none of it, so only this test keeps it from rotting."""
    import subprocess
    import tempfile
    from layerskip_amd import build
    csrc = os.path.join(ROOT, "layerskip_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        proc = subprocess.run(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DLSK_TRACE", "--cuda-device-only", "-S", "-o", os.path.join(td, "x.s"),
                               os.path.join(csrc, "lsk_engine.hip")], capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr[-2000:]
        text = open(os.path.join(td, "x.s")).read()
    assert text.count("s_memrealtime") > 100          # the stamps are there
