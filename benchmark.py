"""Dataset-style benchmark driver with the reference's flags and metric names
(reference benchmark.py:43-50, :95-117, :155-205).

    python benchmark.py --model synthetic:llama2-7B --num_samples 8 --generation_strategy self_speculative \
        --exit_layer 8 --num_speculations 6 --sample False --output_dir ./logs
    torchrun --nproc-per-node 8 benchmark.py --model /ckpt/layerskip-llama2-70B --dataset custom_jsonl --data_path p.jsonl \
        --generation_strategy self_speculative --exit_layer 12 --num_speculations 12 --sample False      # one rank per GPU

Every sample is decoded through the HIP engine; the reported means are the reference's
`acceptance_rate / total_time / time_per_token / tokens_per_second` (torcheval Mean there, plain means
here).  Text-quality metrics (ROUGE/BLEU, benchmark.py:119-147) need real weights and datasets and are out
of scope.  Prompts: synthetic token ids; or `--dataset custom_jsonl --data_path file`, the offline hook of reference
data.py:175-185 -- one JSON object per line with the reference's `prompt` / `response` text fields (`--template "...{message}..."`
is applied to the prompt as data.py:40-53 does; needs the checkpoint's tokenizer), or with an `input_ids` list (no tokenizer
exists offline for the synthetic checkpoints).

Under torchrun (WORLD_SIZE > 1) the model is NOT spread by `device_map="auto"` (reference generate.py:59-64): every rank
materialises its own layer range on its own GPU and the generations run through the layer pipeline
(layerskip_amd/pipeline_strategy.py); rank 0 prints the metrics.
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
from layerskip_amd.cli.common import (Arguments, SyntheticArguments, apply_template, dump_json, load_model_and_tokenizer, make_strategy,
                                      run_on_rank0, run_partition, setup)


@dataclass
class BenchmarkArguments:             # benchmark.py:43-50
    dataset: str = "synthetic"
    data_path: Optional[str] = None
    num_samples: int = 8
    random_shuffle: bool = True
    n_shot: int = 0
    template: Optional[str] = None
    prompt_field: str = "prompt"      # data.py:175 (prepare_custom's keyword arguments)
    response_field: str = "response"


def load_prompts(b: BenchmarkArguments, vocab: int, prompt_len: int, seed: int, tokenizer=None):
    """Token-id prompts.  custom_jsonl rows are either the reference's text rows ({prompt_field, response_field}: templated, then
    tokenised exactly as HuggingfaceLlamaGenerator.generate does, generator_base.py:104) or {"input_ids": [...]} rows."""
    if b.dataset == "custom_jsonl":
        rows = [json.loads(line) for line in open(b.data_path) if line.strip()]
        prompts = []
        for r in rows:
            if "input_ids" in r:
                prompts.append([int(t) for t in r["input_ids"]])
            elif b.prompt_field in r:
                if tokenizer is None:
                    raise ValueError(f"custom_jsonl rows with a {b.prompt_field!r} text field need a tokenizer (a real checkpoint path); "
                                     f"synthetic checkpoints take {{\"input_ids\": [...]}} rows")
                text = apply_template(r[b.prompt_field], b.template)            # data.py:178
                prompts.append(tokenizer(text, return_tensors="pt", add_special_tokens=True)["input_ids"].tolist()[0])
            else:
                raise KeyError(f"custom_jsonl row without {b.prompt_field!r} or 'input_ids': {sorted(r)}")
        if b.random_shuffle:
            random.Random(seed).shuffle(prompts)                                # data.py:228-229
        return prompts[: b.num_samples] if b.num_samples else prompts
    return [synthetic.make_prompt(vocab, prompt_len, i) for i in range(b.num_samples)]


def benchmark(model, tokenizer, b: BenchmarkArguments, gen: GenerationConfig, syn: SyntheticArguments, seed: int, strategy=None):
    generator = TokenGenerator(tokenizer, model, strategy if strategy is not None else make_strategy(gen))
    eos = list(gen.stop_token_ids) + ([tokenizer.eos_token_id] if tokenizer is not None else [model.config.vocab_size])
    sums = {"acceptance_rate": 0.0, "total_time": 0.0, "time_per_token": 0.0, "tokens_per_second": 0.0}
    n = 0
    outputs = []
    for ids in load_prompts(b, model.config.vocab_size, syn.prompt_len, seed, tokenizer):
        res = generator.generate_from_ids(ids, eos, gen, decode=tokenizer is not None)
        outputs.append(res.generation_strategy_result.predicted_tokens)
        if res.num_tokens_generated == 0:
            continue                                             # benchmark.py:197-199
        acc = res.generation_strategy_result.acceptance_rate
        sums["acceptance_rate"] += acc if acc is not None else 0.0   # benchmark.py:77-84
        sums["total_time"] += res.total_time
        sums["time_per_token"] += res.time_per_token
        sums["tokens_per_second"] += res.tokens_per_second
        n += 1
    metrics = {k: {"mean": v / max(1, n)} for k, v in sums.items()}
    benchmark.last_outputs = outputs          # token ids of every sample (tests compare runs; not part of the metric file)
    return metrics


def main(argv=None, backend_factory=None):
    """argv: command line (default sys.argv); backend_factory: stage backend of the multi-process path (tests run the protocol
    over gloo with a CPU backend; the default is the HIP engine)."""
    parser = transformers.HfArgumentParser((Arguments, BenchmarkArguments, GenerationConfig, SyntheticArguments))
    args, b, gen, syn = parser.parse_args_into_dataclasses(args=argv, return_remaining_strings=False)
    random.seed(args.seed)
    ctx = setup(args, syn)                                        # benchmark.py:218 (seeds, process group under torchrun)
    partition = run_partition(args, syn, gen.exit_layer if gen.generation_strategy.startswith("self_speculative") else -1, ctx)
    model, tokenizer = load_model_and_tokenizer(args, syn, gen.exit_layer, ctx, partition)
    strategy = make_strategy(gen, ctx, partition, backend_factory)
    metrics = run_on_rank0(ctx, strategy, model, lambda: benchmark(model, tokenizer, b, gen, syn, args.seed, strategy))
    if metrics is None:
        return None                                               # ranks > 0: served, nothing to report
    print(json.dumps(metrics))
    os.makedirs(args.output_dir, exist_ok=True)
    stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
    dump_json({"args": asdict(args), "benchmark_arguments": asdict(b), "generation_config": asdict(gen), "metrics": metrics,
               **({"world_size": ctx.world, "layer_ranges": partition} if ctx is not None else {})},
              os.path.join(args.output_dir, f"benchmark_{stamp}.json"))
    return metrics


if __name__ == "__main__":
    main()
