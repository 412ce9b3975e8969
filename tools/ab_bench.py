"""A/B of variant builds of the extension inside ONE process (one model build, interleaved rounds, guide rule 24).

    python tools/ab_bench.py [--model llama2-7B] [--rounds 3] [--max-steps 512] [--prompt-len 512] name=path.so[,OPT=VALUE...] ...

(`,OPT=VALUE`: lsk_engine_set_option pairs applied to that variant's engine, e.g. graph=build/variants/new.so,7=1 (LSK_OPT_GRAPH_STEPS).)

Every variant gets its own HipEngine over the SAME model object (its own packed weights, KV pool and workspace), the
variants take turns generating the same prompts, and per variant the script prints tokens/s (median and best of the rounds),
the acceptance rate, a checksum of the produced ids (variants that must be bit-identical are), and the kernels[] table
(per-dispatch begin/end timestamps) of one traced generation.  One JSON object per variant + a summary table.
Measurement tool, not product code."""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from layerskip_amd import _lib, synthetic  # noqa: E402
from layerskip_amd.engine import HipEngine  # noqa: E402


class ToolEngine(HipEngine):
    """No weight-change tracking (variant builds of older sources may lack lsk_engine_weights_checksum)."""

    def _weights_fingerprint(self, m):
        return None


def load_variant(path):
    """_lib.load for a variant build; entry points an OLDER build does not have yet are skipped (the harness only needs the
    engine lifetime, generate and profile calls)."""
    import ctypes
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7B")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--prompts", type=int, default=2)
    ap.add_argument("--max-steps", type=int, default=512)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--late-damping", type=float, default=0.03)
    ap.add_argument("--exit-layer", type=int, default=None)
    ap.add_argument("--num-speculations", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    E = args.exit_layer or synthetic.default_exit_layer(args.model)
    S = args.num_speculations or synthetic.default_num_speculations(args.model)
    cfg = synthetic.make_config(args.model)
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=args.late_damping, dtype=torch.bfloat16, device=dev, gen_device=dev)
    torch.cuda.synchronize()
    eos = [cfg.vocab_size]
    engines = {}
    for v in args.variants:
        name, spec = v.split("=", 1)
        path, *opts = spec.split(",")
        _lib._LIBS["bf16"] = load_variant(os.path.abspath(path))
        engines[name] = ToolEngine(model, max_ctx=args.prompt_len + args.max_steps + S + 16, max_prompt=args.prompt_len)
        for o in opts:
            k, val = o.split("=")
            engines[name].set_option(int(k), int(val))
    prompts = [synthetic.make_prompt(cfg.vocab_size, args.prompt_len, i) for i in range(args.prompts)]
    tps = {n: [] for n in engines}
    info = {}
    for n, eng in engines.items():       # warm-up
        eng.spec_generate(synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 1000), S, E, eos, args.max_steps)
    for r in range(args.rounds):
        for n, eng in engines.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks = 0
            crc = 0
            m = d = 0
            for p in prompts:
                out, mm, dd, steps = eng.spec_generate(p, S, E, eos, args.max_steps)
                toks += len(out)
                crc = zlib.crc32(bytes(str(out), "ascii"), crc)
                m += mm
                d += dd
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tps[n].append(toks / dt)
            info[n] = {"crc": crc, "acceptance": round(m / max(1, d), 4), "steps": len(steps)}
    # the prompt prefill alone (511 rows through every layer, MFMA-tiled kernels): best of 5, and a checksum of the rows
    prefill = {}
    from layerskip_amd.engine import BUF_BULK
    for n, eng in engines.items():
        rows = args.prompt_len - 1
        best = 1e9
        for _ in range(5):
            eng.reset()
            eng.embed_rows(prompts[0][:rows], BUF_BULK, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run_bulk(rows, 0, eng.num_layers)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        crc = zlib.crc32(eng.read_rows(BUF_BULK, 0, rows).view(torch.int16).cpu().numpy().tobytes())
        eng.reset()
        prefill[n] = {"prefill_ms": round(1e3 * best, 3), "prefill_rows_crc": crc}
    results = []
    for n, eng in engines.items():
        eng.set_profile(True)
        eng.spec_generate(prompts[0], S, E, eos, args.max_steps)
        torch.cuda.synchronize()
        table = eng.get_profile_table()
        eng.set_profile(False)
        kern = {f"{row['kernel']}{'M' if row['rows'] != '1' else '1'}": round(1e3 * row["ms"] / row["launches"], 2) for row in table}
        results.append({"variant": n, "tok_s_median": round(statistics.median(tps[n]), 1), "tok_s_best": round(max(tps[n]), 1),
                        "rounds": [round(x, 1) for x in tps[n]], **info[n], **prefill[n], "kernels_us": kern})
    for res in results:
        print(json.dumps(res), flush=True)
    keys = []
    for res in results:
        keys += [k for k in res["kernels_us"] if k not in keys]
    print(f"{'variant':24s} {'tok/s':>8s} {'best':>8s} {'acc':>6s} {'crc':>10s} {'prefill':>8s} " + " ".join(f"{k:>9s}" for k in keys))
    for res in results:
        print(f"{res['variant']:24s} {res['tok_s_median']:8.1f} {res['tok_s_best']:8.1f} {res['acceptance']:6.3f} {res['crc']:10d} {res['prefill_ms']:8.3f} "
              + " ".join(f"{res['kernels_us'].get(k, 0):9.2f}" for k in keys))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
