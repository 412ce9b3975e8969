"""Shared pieces of the command-line drivers (the reference's arguments.py / generate.py:41-67 surface).

Flag names follow the reference (`--model --seed --output_dir`, every `GenerationConfig` field as a flag),
parsed with ``transformers.HfArgumentParser`` like the reference does (arguments.py:19-24).  Because no
checkpoint or tokenizer can be downloaded here, ``--model`` also accepts ``synthetic:<shape>`` (for example
``synthetic:llama2-7B``), which builds the deterministic random-init checkpoint of that architecture, and
prompts can be given as token ids.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .. import synthetic
from ..strategy_api import GenerationConfig


@dataclass
class Arguments:                      # arguments.py:19-24
    model: str = "synthetic:llama2-7B"
    model_args: Optional[str] = None
    seed: int = 42
    output_dir: str = "./logs"


@dataclass
class SyntheticArguments:
    late_damping: float = 0.03
    prompt_len: int = 512
    device: str = "cuda:0"


def load_model_and_tokenizer(args: Arguments, syn: SyntheticArguments, exit_layer: int):
    """(model, tokenizer or None).  A real checkpoint path goes through transformers in bf16."""
    if args.model.startswith("synthetic:"):
        shape = args.model.split(":", 1)[1]
        cfg = synthetic.make_config(shape)
        e = exit_layer if exit_layer > 0 else synthetic.default_exit_layer(shape)
        model = synthetic.build_model(cfg, seed=0, exit_layer=e, late_damping=syn.late_damping, dtype=torch.bfloat16,
                                      device=syn.device, gen_device=syn.device)
        return model, None
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(args.model, use_fast=False)
    model = transformers.AutoModelForCausalLM.from_pretrained(args.model, use_safetensors=True, torch_dtype=torch.bfloat16)
    model.to(syn.device).eval()
    return model, tokenizer


def make_strategy(cfg: GenerationConfig):
    from ..hip_strategies import STRATEGIES
    name = cfg.generation_strategy.replace("_hip", "")
    if name not in STRATEGIES:
        raise ValueError(f"Unsupported generation strategy: {cfg.generation_strategy}")
    return STRATEGIES[name]()


def parse_ids(text: str) -> List[int]:
    return [int(t) for t in text.replace(",", " ").split()]


def dump_json(obj, path: str) -> None:
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)
