"""Interactive generation driver with the reference's flags (reference generate.py:32-161): a REPL that
reads a prompt, decodes it through the chosen HIP strategy and prints the continuation with
tokens/s and acceptance rate.  With a real checkpoint path the prompt is text (tokenizer from the same
path); with `--model synthetic:<shape>` (no tokenizer exists offline) the prompt is a list of token ids.
Under torchrun (WORLD_SIZE > 1) rank 0 runs the REPL and the other ranks serve their layer ranges (layerskip_amd/pipeline_strategy.py)
instead of exiting (reference generate.py:49-51)."""
from __future__ import annotations

import sys
import traceback

import torch
import transformers

from layerskip_amd import GenerationConfig, TokenGenerator
from layerskip_amd.cli.common import (Arguments, SyntheticArguments, load_model_and_tokenizer, make_strategy, parse_ids, run_on_rank0,
                                      run_partition, setup)


def repl(generator, tokenizer, model, gen, streamer, lines=None):
    """The prompt loop of reference generate.py:103-160.  `lines`: an iterable of prompts instead of stdin (tests)."""
    eos = list(gen.stop_token_ids) + ([tokenizer.eos_token_id] if tokenizer is not None else [model.config.vocab_size])
    results = []
    it = iter(lines) if lines is not None else None
    while True:
        try:
            line = next(it) if it is not None else input("prompt> " if tokenizer is not None else "token ids> ")
        except (EOFError, StopIteration):
            break
        if line.strip() in ("", "exit", "quit"):
            break
        try:
            if tokenizer is not None:
                res = generator.generate(line, gen, streamer=streamer)
                print(res.decoded_prediction)
            else:
                res = generator.generate_from_ids(parse_ids(line), eos, gen)
                print(res.generation_strategy_result.predicted_tokens)
        except Exception:
            traceback.print_exc()
            raise
        results.append(res)
        acc = res.generation_strategy_result.acceptance_rate
        print(f"\n\t=========================\n\tTime per token: {res.time_per_token * 1000:.2f} ms"
              f"\n\tTokens per second: {res.tokens_per_second:.2f}"
              + (f"\n\tAcceptance rate: {acc:.2%}" if acc is not None else ""))
    return results


def main(argv=None, lines=None, backend_factory=None):
    parser = transformers.HfArgumentParser((Arguments, GenerationConfig, SyntheticArguments))
    args, gen, syn = parser.parse_args_into_dataclasses(args=argv, return_remaining_strings=False)
    ctx = setup(args, syn)               # generate.py:41-52 -- under torchrun every rank stays and owns a layer range
    partition = run_partition(args, syn, gen.exit_layer if gen.generation_strategy.startswith("self_speculative") else -1, ctx)
    model, tokenizer = load_model_and_tokenizer(args, syn, gen.exit_layer, ctx, partition)
    strategy = make_strategy(gen, ctx, partition, backend_factory)
    generator = TokenGenerator(tokenizer, model, strategy)
    streamer = transformers.TextStreamer(tokenizer) if tokenizer is not None else None
    return run_on_rank0(ctx, strategy, model, lambda: repl(generator, tokenizer, model, gen, streamer, lines))


if __name__ == "__main__":
    main()
    sys.exit(0)
