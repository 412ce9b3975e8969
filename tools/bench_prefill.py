"""GPU micro-benchmark (not a test): prompt prefill time, flash-shaped vs chunked attention."""
import sys
import time

import torch

sys.path.insert(0, ".")
from layerskip_amd import _lib, synthetic  # noqa: E402
from layerskip_amd.engine import BUF_BULK, HipEngine  # noqa: E402

cfg = synthetic.make_config("llama2-7B")
model = synthetic.build_model(cfg, seed=0, exit_layer=8, late_damping=0.03, device="cuda:0", gen_device="cuda:0")
for n in (511, 2047):
    eng = HipEngine(model, max_ctx=n + 129, max_prompt=n + 1)
    ids = synthetic.make_prompt(cfg.vocab_size, n, 1)
    for flash in (0, 1):
        eng.set_option(_lib.LSK_OPT_FLASH_PREFILL, flash)
        ts = []
        for it in range(3):
            eng.reset()
            eng.embed_rows(ids, BUF_BULK, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run_bulk(n, 0, eng.num_layers)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"rows {n} flash {flash}: {min(ts) * 1e3:.2f} ms", flush=True)
    eng.close()
