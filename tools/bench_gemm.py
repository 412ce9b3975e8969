"""GPU micro-benchmark (not a test): skinny projection kernel bandwidth vs shape / M / grid size."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import lsk_test_lib  # noqa: E402
from layerskip_amd import _lib  # noqa: E402

lib = _lib.load()
# LSK_TEST_LIB: a variant build of the TEST library (hipcc -D... layerskip_amd/csrc/lsk_test_exports.hip), for kernel experiments
tlib = lsk_test_lib.load(path=os.environ["LSK_TEST_LIB"]) if os.environ.get("LSK_TEST_LIB") else lsk_test_lib.load()
dev = torch.device("cuda:0")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


NBUF = int(os.environ.get('NBUF', '6'))


def run(k, n, m, wgs, with_norm, nbuf=None, iters=60):
    nbuf = nbuf or NBUF
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.lsk_packed_bytes(n, k, ctypes.byref(nbytes)))
    bufs = [torch.randint(0, 255, (nbytes.value,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    for b in bufs:  # keep bf16 values finite-ish: clear exponent top bits
        b.view(torch.int16).bitwise_and_(0x3FFF)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    nw = torch.ones(k, device=dev, dtype=torch.bfloat16)
    y = torch.empty(m, n, dtype=torch.float32, device=dev)
    def launch(i):
        lsk_test_lib.check(tlib.lsk_test_gemm(x.data_ptr(), m, k, bufs[i % nbuf].data_ptr(), n, nw.data_ptr() if with_norm else None,
                                     1e-5, y.data_ptr(), wgs, st()))
    for i in range(nbuf):
        launch(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        launch(i)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    return us, nbytes.value / us / 1e3   # us, GB/s


shapes = {"qkv": (4096, 12288, True), "o": (4096, 4096, False), "gateup": (4096, 22016, True), "down": (11008, 4096, False),
          "head": (4096, 32000, True)}
import os
WGS_LIST = [int(x) for x in os.environ.get('WGS', '128,192,256,384,512,768,1024').split(',')]
MS = [int(x) for x in os.environ.get('MS', '1,7').split(',')]
which = sys.argv[1:] or list(shapes)
for name in which:
    k, n, norm = shapes[name]
    for m in MS:
        row = []
        for wgs in (WGS_LIST):
            us, gbs = run(k, n, m, wgs, norm)
            row.append(f"{wgs}:{us:6.1f}us/{gbs:5.0f}")
        print(f"{name:7s} M={m}  " + "  ".join(row), flush=True)
