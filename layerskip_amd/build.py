"""Builds the HIP extension in-tree: layerskip_amd/csrc/liblayerskip_hip.so (gfx950 only).

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblayerskip_hip.so")              # bf16: the BASELINE configs
LIB_F16 = os.path.join(CSRC, "liblayerskip_hip_f16.so")      # fp16: same sources, -DLSK_ELEM_F16 (generate.py:63's dtype)
# test infrastructure: single kernels on caller-owned buffers (include/layerskip_hip_test.h); never loaded by the package
LIB_TEST = os.path.join(CSRC, "liblayerskip_hip_test.so")
LIB_TEST_F16 = os.path.join(CSRC, "liblayerskip_hip_test_f16.so")
SOURCES = ["lsk_engine.hip", "lsk_generate.hip"]             # the product library: two translation units
TEST_SOURCES = ["lsk_test_exports.hip"]


def _inputs():
    """Everything the translation unit reads: the .hip sources, EVERY kernel header next to them, the public header."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "layerskip_hip.h"))
    return files


# The compile flags of every build of the extension (the libraries, the measurement variants of tools/, the resource checks of
# tests/test_abi.py).  -amdgpu-kernarg-preload-count: the first 14 argument dwords of every kernel are in SGPRs when a wave starts
# (GemmHot in csrc/lsk_gemm.h says what rides there and why); the code object keeps a compatibility prologue for firmware without it.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
               "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(f) > t for f in _inputs())


def _compile(lib: str, defines, verbose: bool, sources=None) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-fPIC", "-shared"] + list(defines) + ["-o", lib] + [os.path.join(CSRC, s) for s in (sources or SOURCES)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError(f"hipcc failed building {os.path.basename(lib)}")
    if verbose and proc.stderr:
        sys.stderr.write(proc.stderr)


def build(force: bool = False, verbose: bool = False) -> str:
    """Builds the product libraries (bf16, fp16) and the test libraries in-tree (when stale); returns the bf16 product one."""
    jobs = ((LIB, [], SOURCES), (LIB_F16, ["-DLSK_ELEM_F16"], SOURCES),
            (LIB_TEST, [], TEST_SOURCES), (LIB_TEST_F16, ["-DLSK_ELEM_F16"], TEST_SOURCES))
    stale = [j for j in jobs if force or _stale(j[0])]
    if len(stale) > 1:
        from concurrent.futures import ThreadPoolExecutor      # hipcc is a subprocess: the four builds run side by side
        with ThreadPoolExecutor(max_workers=len(stale)) as pool:
            list(pool.map(lambda j: _compile(j[0], j[1], verbose, j[2]), stale))
    else:
        for lib, defines, sources in stale:
            _compile(lib, defines, verbose, sources)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
