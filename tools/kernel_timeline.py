"""In-kernel timeline of the decode kernels (measurement tool): where the microseconds of a dependent launch go.

    python tools/kernel_timeline.py [--model llama2-7B] [--lib build/variants/trace.so] [--skip 3000] [--launches 900] [--out x.json]

Needs a -DLSK_TRACE build of the extension (see tools/README.md): every workgroup of the projection and attention kernels stamps
the 100 MHz realtime counter at fixed points of its life (lsk_common.h) and stores the stamps with the XCD / CU it ran on.  The tool
generates with the benchmark workload, skips the first launches (prefill, first steps), records the next ones and prints, per
kernel class: the gap between the previous launch's last store acknowledgement and this launch's first instruction, the spread
of workgroup start times, and the medians over workgroups of every stamp relative to the launch's first instruction."""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ab_bench import ToolEngine, load_variant  # noqa: E402
from layerskip_amd import _lib, synthetic  # noqa: E402

WORDS, MAX_WGS = 12, 2048
TICK_US = 0.01
GEMM_NAMES = {(0, 1): "o_proj/down", (1, 2): "gate_up", (1, 3): "qkv", (1, 4): "lm_head", (0, 0): "plain_f32", (1, 0): "rms_f32"}
GEMM_POINTS = ["ring_req", "rows_staged", "w_first", "w_unit0", "reduced", "stores_issued", "stores_acked"]
CHAIN_POINTS = ["ring_req", "attn_row_staged", "o_reduced", "at_edge1", "edge1_passed", "norm_staged", "gu_reduced", "edge2_passed", "down_reduced"]
ATTN_POINTS = ["loads_req", "qk_arrived", "pv_in_lds", "part_issued", "part_drained", "ticket", "out_issued", "out_acked"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7B")
    ap.add_argument("--lib", default=os.path.join(ROOT, "build", "variants", "trace.so"))
    ap.add_argument("--skip", type=int, default=3000)
    ap.add_argument("--launches", type=int, default=900)
    ap.add_argument("--max-steps", type=int, default=96)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    E, S = synthetic.default_exit_layer(args.model), synthetic.default_num_speculations(args.model)
    cfg = synthetic.make_config(args.model)
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=dev, gen_device=dev)
    lib = load_variant(os.path.abspath(args.lib))
    lib.lsk_trace_begin.restype, lib.lsk_trace_begin.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.lsk_trace_end.restype, lib.lsk_trace_end.argtypes = ctypes.c_int, [ctypes.c_void_p]
    _lib._LIBS["bf16"] = lib
    eng = ToolEngine(model, max_ctx=args.prompt_len + args.max_steps + S + 16, max_prompt=args.prompt_len)
    eos = [cfg.vocab_size]
    eng.spec_generate(synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 1000), S, E, eos, args.max_steps)      # warm-up
    torch.cuda.synchronize()
    buf = torch.zeros((args.launches, MAX_WGS, WORDS), dtype=torch.int64, device=dev)
    lib.lsk_trace_begin(ctypes.c_void_p(buf.data_ptr()), args.launches, args.skip)
    eng.spec_generate(synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 0), S, E, eos, args.max_steps)
    torch.cuda.synchronize()
    tags = (ctypes.c_int * (4 * 8192))()
    n = lib.lsk_trace_end(tags)
    data = buf[:n].cpu().numpy().astype(np.int64)
    tags = np.frombuffer(tags, dtype=np.int32)[: 4 * n].reshape(n, 4)
    print(f"{n} launches recorded", flush=True)

    groups = defaultdict(list)
    prev_end = None
    for i in range(n):
        kind, sub, m, grid = (int(v) for v in tags[i])
        g = min(grid, MAX_WGS)
        d = data[i, :g]
        t0 = d[:, 0]
        if (t0 == 0).any():
            prev_end = None
            continue
        start = int(t0.min())
        stamps = d[:, 1:9].copy()
        stamps[stamps == 0] = start                         # stamps a workgroup never reached
        end = int(max(stamps.max(), t0.max()))
        cu = (d[:, 11] & 0xF) * 4096 + ((d[:, 10] >> 13) & 7) * 256 + ((d[:, 10] >> 12) & 1) * 16 + ((d[:, 10] >> 8) & 0xF)
        _, counts = np.unique(cu, return_counts=True)
        rec = {"gap": None if prev_end is None else (start - prev_end) * TICK_US, "start_spread": (int(t0.max()) - start) * TICK_US,
               "span": (end - start) * TICK_US, "grid": grid, "cus": int(len(counts)), "max_wg_per_cu": int(counts.max()),
               "xcds": int(len(np.unique(d[:, 11] & 0xF)))}
        if kind == 2:
            # the resident one-row grid (lsk_chain.h): compute wave 0's stamps, and service wave 0's in the rows behind them
            name = "chain"
            rel = (d[:, 1:10] - start) * TICK_US
            for k, pn in enumerate(CHAIN_POINTS):
                col = rel[:, k][d[:, 1 + k] != 0]
                if col.size:
                    rec[pn] = float(np.median(col))
                    rec[pn + "_max"] = float(col.max())
                    rec[pn + "_min"] = float(col.min())
            sv = data[i, g:2 * g]
            ok = sv[:, 0] != 0
            if ok.any():
                rec["sv_gather1_begin"] = float(np.median((sv[ok, 0] - start) * TICK_US))
                rec["sv_gather1_end"] = float(np.median((sv[ok, 1] - start) * TICK_US))
                rec["sv_gather1_end_max"] = float(((sv[ok, 1] - start) * TICK_US).max())
                rec["sv_sweeps1"] = float(np.mean(sv[ok, 2]))
                rec["sv_gather2_begin"] = float(np.median((sv[ok, 3] - start) * TICK_US))
                rec["sv_gather2_end"] = float(np.median((sv[ok, 4] - start) * TICK_US))
                rec["sv_gather2_end_max"] = float(((sv[ok, 4] - start) * TICK_US).max())
                rec["sv_sweeps2"] = float(np.mean(sv[ok, 5]))
            end = int(max(d[:, 1:10].max(), t0.max()))
            rec["span"] = (end - start) * TICK_US
        elif kind == 0:
            name = GEMM_NAMES.get(((sub >> 4) & 15, sub & 15), str(sub & 255))
            if name == "o_proj/down":
                name = "o_proj" if (sub >> 8) == cfg.hidden_size else "down"
            rel = (d[:, 1:8] - start) * TICK_US
            for k, pn in enumerate(GEMM_POINTS):
                rec[pn] = float(np.median(rel[:, k]))
                rec[pn + "_max"] = float(rel[:, k].max())
            xcd = d[:, 11] & 0xF
            for x in range(8):                                    # when the workgroups of each XCD finish (mean; NaN: none there)
                rec[f"reduced_xcd{x}"] = float(np.mean(rel[xcd == x, 4])) if (xcd == x).any() else float("nan")
            rec["reduced_p10"] = float(np.percentile(rel[:, 4], 10))
            rec["reduced_p90"] = float(np.percentile(rel[:, 4], 90))
            rec["own_ring_req"] = float(np.median((d[:, 1] - t0) * TICK_US))      # first instruction -> ring requested, per workgroup
            rec["own_addr_done"] = float(np.median((d[:, 9] - t0) * TICK_US))
            rec["rows_wave0"] = float(np.median((d[:, 8] - start) * TICK_US))
            rec["own_w_first"] = float(np.median((d[:, 3] - t0) * TICK_US))
        else:
            name = "attention"
            last = d[:, 9] == 1
            rel = (d[:, 1:9] - start) * TICK_US
            for k, pn in enumerate(ATTN_POINTS[:6]):
                rec[pn] = float(np.median(rel[:, k]))
                rec[pn + "_max"] = float(rel[:, k].max())
            if last.any():
                for k, pn in enumerate(ATTN_POINTS[6:], start=6):
                    rec[pn] = float(np.median(rel[last, k]))
                    rec[pn + "_max"] = float(rel[last, k].max())
        groups[(name, "1" if m == 1 else ">1", grid)].append(rec)
        prev_end = end

    out = []
    for (name, rows, grid), recs in sorted(groups.items()):
        row = {"kernel": name, "rows": rows, "grid": grid, "launches": len(recs)}
        for k in recs[0]:
            vals = [r[k] for r in recs if r.get(k) is not None]
            if vals:
                row[k] = round(float(np.mean(vals)), 2)
        out.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
