"""GPU micro-benchmark (not a test): prompt prefill time (all layers) of the llama2-7B shape.

    python tools/bench_prefill.py [rows ...]          # default 511 2047; FLASH=0 also times the chunked-attention path
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layerskip_amd import _lib, synthetic  # noqa: E402
from layerskip_amd.engine import BUF_BULK, HipEngine  # noqa: E402

if os.environ.get("LSK_LIB"):          # a variant build of the extension (kernel experiments)
    _lib._LIBS["bf16"] = _lib.load(os.environ["LSK_LIB"])
rows = [int(a) for a in sys.argv[1:]] or [511, 2047]
cfg = synthetic.make_config(os.environ.get("MODEL", "llama2-7B"))
model = synthetic.build_model(cfg, seed=0, exit_layer=8, late_damping=0.03, device="cuda:0", gen_device="cuda:0")
for n in rows:
    eng = HipEngine(model, max_ctx=n + 129, max_prompt=n + 1)
    ids = synthetic.make_prompt(cfg.vocab_size, n, 1)
    for flash in ((0, 1) if os.environ.get("FLASH") == "0" else (1,)):
        eng.set_option(_lib.LSK_OPT_FLASH_PREFILL, flash)
        ts = []
        for it in range(4):
            eng.reset()
            eng.embed_rows(ids, BUF_BULK, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run_bulk(n, 0, eng.num_layers)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        h = eng.rows_view(BUF_BULK, 0, n).view(torch.int16).to(torch.int64)          # bit pattern of the final hidden rows
        print(f"rows {n} flash {flash}: {min(ts) * 1e3:.2f} ms  checksum {int((h * torch.arange(1, h.numel() + 1, device=h.device).view_as(h) % 1000003).sum().item())}", flush=True)
    eng.close()
