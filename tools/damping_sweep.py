"""Acceptance rate and tokens/s of the benchmark workload as a function of the late-layer damping of the synthetic checkpoint
(measurement tool): one model build, the damped projections rescaled in place between points.

    python tools/damping_sweep.py [--model llama2-7B] 0.01 0.02 0.03 0.05 0.08
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from layerskip_amd import GenerationConfig, synthetic  # noqa: E402
from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy  # noqa: E402


def rescale_late_projections(model, exit_layer: int, ratio: float) -> None:
    with torch.no_grad():
        for layer in model.model.layers[exit_layer:]:
            layer.self_attn.o_proj.weight.mul_(ratio)
            layer.mlp.down_proj.weight.mul_(ratio)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7B")
    ap.add_argument("--prompts", type=int, default=2)
    ap.add_argument("dampings", nargs="+", type=float)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    E, S = synthetic.default_exit_layer(args.model), synthetic.default_num_speculations(args.model)
    cfg = synthetic.make_config(args.model)
    cur = args.dampings[0]
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=cur, dtype=torch.bfloat16, device=dev, gen_device=dev)
    strat = HipSelfSpeculativeGenerationStrategy()
    gen = GenerationConfig(max_steps=512, exit_layer=E, num_speculations=S, sample=False, generation_strategy="self_speculative")
    for d in args.dampings:
        if d != cur:
            rescale_late_projections(model, E, d / cur)
            cur = d
        strat.generate_token_ids(model, synthetic.make_prompt(cfg.vocab_size, 512, 1000), [cfg.vocab_size], gen)      # warm-up + re-pack
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = [strat.generate_token_ids(model, synthetic.make_prompt(cfg.vocab_size, 512, i), [cfg.vocab_size], gen) for i in range(args.prompts)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"late_damping": d, "acceptance_rate": round(sum(r.acceptance_rate for r in res) / len(res), 4),
                          "tokens_per_s": round(sum(len(r.predicted_tokens) for r in res) / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
