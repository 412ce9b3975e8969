"""Speculative vs autoregressive equality check, the reference's own parity definition
(reference correctness.py:38-99) at token-id level: for every sample both strategies decode greedily and the
outputs must be identical (`--sample False`, README.md:145-156).  Exit code 1 on any mismatch."""
from __future__ import annotations

import datetime
import json
import os
import sys
from dataclasses import replace

import torch
import transformers

from layerskip_amd import GenerationConfig, TokenGenerator
from layerskip_amd.cli.common import (Arguments, SyntheticArguments, load_model_and_tokenizer, make_strategy, run_on_rank0, run_partition,
                                      setup)
from benchmark import BenchmarkArguments, load_prompts


def compare(model, tokenizer, b, syn, spec, ar, spec_cfg, ar_cfg, seed):
    eos = [model.config.vocab_size] if tokenizer is None else [tokenizer.eos_token_id]
    errors = 0
    prompts = load_prompts(b, model.config.vocab_size, syn.prompt_len, seed, tokenizer)
    for i, ids in enumerate(prompts):
        a = spec.generate_from_ids(ids, eos, spec_cfg).generation_strategy_result.predicted_tokens
        r = ar.generate_from_ids(ids, eos, ar_cfg).generation_strategy_result.predicted_tokens
        if a != r:
            errors += 1
            first = next((k for k, (x, y) in enumerate(zip(a, r)) if x != y), min(len(a), len(r)))
            print(f"sample {i}: mismatch at token {first}")
    return {"errors": errors, "error_pct": errors / max(1, len(prompts)), "num_samples": len(prompts)}


def main(argv=None, backend_factory=None):
    parser = transformers.HfArgumentParser((Arguments, BenchmarkArguments, GenerationConfig, SyntheticArguments))
    args, b, gen, syn = parser.parse_args_into_dataclasses(args=argv, return_remaining_strings=False)
    seed = args.seed
    args.seed = 0                                                 # correctness.py:38-41 seeds with 0
    ctx = setup(args, syn)
    gen = replace(gen, sample=False)
    partition = run_partition(args, syn, gen.exit_layer, ctx)
    model, tokenizer = load_model_and_tokenizer(args, syn, gen.exit_layer, ctx, partition)
    spec_cfg = replace(gen, generation_strategy="self_speculative")
    ar_cfg = replace(gen, generation_strategy="autoregressive", exit_layer=-1)
    spec_strategy = make_strategy(spec_cfg, ctx, partition, backend_factory)
    spec = TokenGenerator(tokenizer, model, spec_strategy)
    ar = TokenGenerator(tokenizer, model, make_strategy(ar_cfg, ctx, partition, backend_factory))
    out = run_on_rank0(ctx, spec_strategy, model, lambda: compare(model, tokenizer, b, syn, spec, ar, spec_cfg, ar_cfg, seed))
    if out is None:
        return 0                                                  # ranks > 0
    print(json.dumps(out))
    os.makedirs(args.output_dir, exist_ok=True)                   # correctness.py:90-99: the result file next to the logs
    with open(os.path.join(args.output_dir, f"correctness_{datetime.datetime.now().strftime('%Y%m%d_%H%M%S')}.json"), "w") as f:
        json.dump(out, f)
    return 1 if out["errors"] else 0


if __name__ == "__main__":
    sys.exit(main())
