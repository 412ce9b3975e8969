"""Dataset-style benchmark driver with the reference's flags and metric names
(reference benchmark.py:43-50, :95-117, :155-205).

    python benchmark.py --model synthetic:llama2-7B --num_samples 8 --generation_strategy self_speculative \
        --exit_layer 8 --num_speculations 6 --sample False --output_dir ./logs

Every sample is decoded through the HIP engine; the reported means are the reference's
`acceptance_rate / total_time / time_per_token / tokens_per_second` (torcheval Mean there, plain means
here).  Text-quality metrics (ROUGE/BLEU, benchmark.py:119-147) need real weights and datasets and are out
of scope; prompts are synthetic token ids, or a JSONL file with {"input_ids": [...]} per line
(`--dataset custom_jsonl --data_path file`, the offline hook of reference data.py:175-185).
"""
from __future__ import annotations

import datetime
import json
import os
import random
from dataclasses import asdict, dataclass
from typing import Optional

import torch
import transformers

from layerskip_amd import GenerationConfig, TokenGenerator, synthetic
from layerskip_amd.cli.common import Arguments, SyntheticArguments, dump_json, load_model_and_tokenizer, make_strategy


@dataclass
class BenchmarkArguments:             # benchmark.py:43-50
    dataset: str = "synthetic"
    data_path: Optional[str] = None
    num_samples: int = 8
    random_shuffle: bool = True
    n_shot: int = 0
    template: Optional[str] = None


def load_prompts(b: BenchmarkArguments, vocab: int, prompt_len: int, seed: int):
    if b.dataset == "custom_jsonl":
        rows = [json.loads(line) for line in open(b.data_path)]
        prompts = [r["input_ids"] for r in rows]
        if b.random_shuffle:
            random.Random(seed).shuffle(prompts)
        return prompts[: b.num_samples]
    return [synthetic.make_prompt(vocab, prompt_len, i) for i in range(b.num_samples)]


def benchmark(model, tokenizer, b: BenchmarkArguments, gen: GenerationConfig, syn: SyntheticArguments, seed: int):
    generator = TokenGenerator(tokenizer, model, make_strategy(gen))
    eos = list(gen.stop_token_ids) + ([tokenizer.eos_token_id] if tokenizer is not None else [model.config.vocab_size])
    sums = {"acceptance_rate": 0.0, "total_time": 0.0, "time_per_token": 0.0, "tokens_per_second": 0.0}
    n = 0
    for ids in load_prompts(b, model.config.vocab_size, syn.prompt_len, seed):
        res = generator.generate_from_ids(ids, eos, gen)
        if res.num_tokens_generated == 0:
            continue                                             # benchmark.py:197-199
        acc = res.generation_strategy_result.acceptance_rate
        sums["acceptance_rate"] += acc if acc is not None else 0.0   # benchmark.py:77-84
        sums["total_time"] += res.total_time
        sums["time_per_token"] += res.time_per_token
        sums["tokens_per_second"] += res.tokens_per_second
        n += 1
    return {k: {"mean": v / max(1, n)} for k, v in sums.items()}


def main():
    parser = transformers.HfArgumentParser((Arguments, BenchmarkArguments, GenerationConfig, SyntheticArguments))
    args, b, gen, syn = parser.parse_args_into_dataclasses(return_remaining_strings=False)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    model, tokenizer = load_model_and_tokenizer(args, syn, gen.exit_layer)
    metrics = benchmark(model, tokenizer, b, gen, syn, args.seed)
    print(json.dumps(metrics))
    os.makedirs(args.output_dir, exist_ok=True)
    stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
    dump_json({"args": asdict(args), "benchmark_arguments": asdict(b), "generation_config": asdict(gen), "metrics": metrics},
              os.path.join(args.output_dir, f"benchmark_{stamp}.json"))


if __name__ == "__main__":
    main()
