"""Grid sweep over exit_layer x num_speculations on the HIP engine (reference sweep.py:27-115: same flags,
CSV output; the contour PDF is left out).  Finds the MI355X-optimal (E, S): the draft/verify cost ratio
differs from the reference's eager path once launch overhead is gone.

    python sweep.py --model synthetic:llama2-7B --num_samples 2 --max_steps 128 \
        --exit_layer_first 4 --exit_layer_last 16 --exit_layer_step 4 \
        --num_speculations_first 2 --num_speculations_last 10 --num_speculations_step 2
"""
from __future__ import annotations

import csv
import datetime
import os
from dataclasses import dataclass, replace

import torch
import transformers

from benchmark import BenchmarkArguments, benchmark
from layerskip_amd import GenerationConfig
from layerskip_amd.cli.common import Arguments, SyntheticArguments, load_model_and_tokenizer


@dataclass
class SweepArguments:                 # sweep.py:27-34
    exit_layer_first: int = 1
    exit_layer_last: int = 15
    exit_layer_step: int = 1
    num_speculations_first: int = 1
    num_speculations_last: int = 6
    num_speculations_step: int = 1


def main():
    parser = transformers.HfArgumentParser((Arguments, BenchmarkArguments, GenerationConfig, SyntheticArguments, SweepArguments))
    args, b, gen, syn, sw = parser.parse_args_into_dataclasses(return_remaining_strings=False)
    torch.manual_seed(args.seed)
    # the synthetic checkpoint's late-layer damping depends on the exit layer: rebuild per exit layer
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, f"sweep_{datetime.datetime.now().strftime('%Y%m%d_%H%M%S')}.csv")      # sweep.py:43-44
    rows = []
    for e in range(sw.exit_layer_first, sw.exit_layer_last + 1, sw.exit_layer_step):
        model, tokenizer = load_model_and_tokenizer(args, syn, e)
        for s in range(sw.num_speculations_first, sw.num_speculations_last + 1, sw.num_speculations_step):
            cfg = replace(gen, exit_layer=e, num_speculations=s, generation_strategy="self_speculative", sample=False)
            m = benchmark(model, tokenizer, b, cfg, syn, args.seed)
            rows.append({"exit_layer": e, "num_speculations": s, "acceptance_rate": m["acceptance_rate"]["mean"],     # sweep.py:54-61
                         "total_time": m["total_time"]["mean"], "time_per_token": m["time_per_token"]["mean"],
                         "tokens_per_second": m["tokens_per_second"]["mean"]})
            print(rows[-1], flush=True)
            with open(path, "w", newline="") as f:             # rewritten after every grid point (sweep.py:62-64)
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
