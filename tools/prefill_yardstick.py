"""A measuring stick for the prefill projections, never a product path: `torch.mm` (hipBLASLt / rocBLAS behind torch-ROCm) at the
four GEMM shapes of a llama2-7B-shaped prompt pass, timed with HIP events on the same box the engine's MFMA-tiled prefill kernels
(csrc/lsk_gemm_big.h) are timed on.

    python tools/prefill_yardstick.py [--model llama2-7B] [--rows 511,2047] [--out x.json]

Per (rows, projection): the library's microseconds and TFLOP/s, the engine's own prefill time for the same rows (all layers, best of
5 of lsk_run_bulk) and the time 32 layers of library GEMMs alone would take -- the library number excludes the RMSNorm / RoPE / KV
append / SwiGLU / residual epilogues the engine's kernels fuse, and the attention, so it is a LOWER bound for a library-built layer.
("no hipBLASLt yardstick was ever timed beside it".)"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from layerskip_amd import synthetic  # noqa: E402


def time_mm(m, k, n, iters=50):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a, w.t())
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        start.record()
        for _ in range(iters):
            torch.mm(a, w.t())
        stop.record()
        torch.cuda.synchronize()
        best = min(best, start.elapsed_time(stop) / iters)
    return best * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7B")
    ap.add_argument("--rows", default="511,2047")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-engine", action="store_true")
    args = ap.parse_args()
    cfg = synthetic.make_config(args.model)
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    hd = getattr(cfg, "head_dim", None) or H // cfg.num_attention_heads
    qkv_n = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * hd
    shapes = {"qkv": (H, qkv_n), "o_proj": (cfg.num_attention_heads * hd, H), "gate_up": (H, 2 * I), "down": (I, H)}
    rows_list = [int(r) for r in args.rows.split(",")]
    out = {"model": args.model, "device": torch.cuda.get_device_name(0), "torch": torch.__version__, "rows": {}}
    eng = None
    if not args.no_engine:
        from layerskip_amd.engine import BUF_BULK, HipEngine
        E = synthetic.default_exit_layer(args.model)
        model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device="cuda:0", gen_device="cuda:0")
        eng = HipEngine(model, max_ctx=max(rows_list) + 64, max_prompt=max(rows_list) + 1)
    for rows in rows_list:
        rec = {"library": {}}
        layer_us, layer_flops = 0.0, 0.0
        for name, (k, n) in shapes.items():
            us = time_mm(rows, k, n)
            fl = 2.0 * rows * k * n
            rec["library"][name] = {"M": rows, "K": k, "N": n, "us": round(us, 2), "tflops": round(fl / us / 1e6, 1)}
            layer_us += us
            layer_flops += fl
        rec["library_layer_us"] = round(layer_us, 2)
        rec["library_all_layers_ms"] = round(layer_us * L / 1e3, 3)
        rec["library_tflops"] = round(layer_flops / layer_us / 1e6, 1)
        if eng is not None:
            prompt = synthetic.make_prompt(cfg.vocab_size, rows, 0)
            best = 1e9
            for _ in range(5):
                eng.reset()
                eng.embed_rows(prompt, BUF_BULK, 0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run_bulk(rows, 0, eng.num_layers)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            eng.reset()
            rec["engine_prefill_ms"] = round(best * 1e3, 3)
            rec["engine_over_library_gemms"] = round(best * 1e3 / rec["library_all_layers_ms"], 3)
        out["rows"][str(rows)] = rec
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
