"""Interactive generation driver with the reference's flags (reference generate.py:32-161): a REPL that
reads a prompt, decodes it through the chosen HIP strategy and prints the continuation with
tokens/s and acceptance rate.  With a real checkpoint path the prompt is text (tokenizer from the same
path); with `--model synthetic:<shape>` (no tokenizer exists offline) the prompt is a list of token ids."""
from __future__ import annotations

import sys
import traceback

import torch
import transformers

from layerskip_amd import GenerationConfig, TokenGenerator
from layerskip_amd.cli.common import Arguments, SyntheticArguments, load_model_and_tokenizer, make_strategy, parse_ids


def main():
    parser = transformers.HfArgumentParser((Arguments, GenerationConfig, SyntheticArguments))
    args, gen, syn = parser.parse_args_into_dataclasses(return_remaining_strings=False)
    torch.manual_seed(args.seed)
    model, tokenizer = load_model_and_tokenizer(args, syn, gen.exit_layer)
    generator = TokenGenerator(tokenizer, model, make_strategy(gen))
    streamer = transformers.TextStreamer(tokenizer) if tokenizer is not None else None
    eos = list(gen.stop_token_ids) + ([tokenizer.eos_token_id] if tokenizer is not None else [model.config.vocab_size])
    while True:
        try:
            line = input("prompt> " if tokenizer is not None else "token ids> ")
        except EOFError:
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
        acc = res.generation_strategy_result.acceptance_rate
        print(f"\n\t=========================\n\tTime per token: {res.time_per_token * 1000:.2f} ms"
              f"\n\tTokens per second: {res.tokens_per_second:.2f}"
              + (f"\n\tAcceptance rate: {acc:.2%}" if acc is not None else ""))


if __name__ == "__main__":
    sys.exit(main())
