"""Shared pieces of the command-line drivers (the reference's arguments.py / generate.py:41-67 surface).

Flag names follow the reference (`--model --seed --output_dir`, every `GenerationConfig` field as a flag),
parsed with ``transformers.HfArgumentParser`` like the reference does (arguments.py:19-24).  Because no
checkpoint or tokenizer can be downloaded here, ``--model`` also accepts ``synthetic:<shape>`` (for example
``synthetic:llama2-7B``), which builds the deterministic random-init checkpoint of that architecture, and
prompts can be given as token ids.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
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
    pp_balance: str = "draft"          # multi-GPU (torchrun) layer split: "draft" | "memory" (layerskip_amd.pipeline.plan_partition)


def setup(args: Arguments, syn: SyntheticArguments):
    """The reference's `setup` (generate.py:41-52): process group from torchrun's environment, seeds.  Where the reference makes
    every rank but 0 `exit()` ("we don't support parallel inference yet"), every rank stays here: it owns a layer range of the
    model on its own GPU.  Returns the DistContext, or None for a single-process run."""
    import random
    from ..pipeline_strategy import init_distributed
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    ctx = init_distributed(device=syn.device)
    if ctx is not None:
        syn.device = str(ctx.device)
    return ctx


def run_partition(args: Arguments, syn: SyntheticArguments, exit_layer: int, ctx):
    """The layer ranges of a multi-process run (None for one process).  Needs only the checkpoint's config."""
    if ctx is None:
        return None
    from ..pipeline_strategy import partition_for
    if args.model.startswith("synthetic:"):
        shape = args.model.split(":", 1)[1]
        n_layers = synthetic.make_config(shape).num_hidden_layers
        e = exit_layer if exit_layer > 0 else -1
    else:
        import transformers
        from ..checkpoint import resolve_checkpoint_dir
        n_layers = transformers.AutoConfig.from_pretrained(resolve_checkpoint_dir(args.model)).num_hidden_layers
        e = exit_layer
    return partition_for(n_layers, e, ctx.world, syn.pp_balance)


def _load_tokenizer(path: str):
    import transformers
    try:
        return transformers.AutoTokenizer.from_pretrained(path, use_fast=False)
    except Exception:       # noqa: BLE001 -- a checkpoint directory that only carries tokenizer.json (fast tokenizers)
        return transformers.AutoTokenizer.from_pretrained(path)


def load_model_and_tokenizer(args: Arguments, syn: SyntheticArguments, exit_layer: int, ctx=None, partition=None):
    """(model, tokenizer or None).  A real checkpoint path is read in bf16 (BASELINE.json's dtype; the reference hard-codes fp16,
    generate.py:63) straight onto the device.  Multi-process (`ctx`, `partition`): only this rank's decoder layers are
    materialised -- `device_map="auto"` of generate.py:59-64 becomes one process per GPU with its own layer range."""
    layer_range = partition[ctx.rank] if (ctx is not None and partition is not None) else None
    if args.model.startswith("synthetic:"):
        shape = args.model.split(":", 1)[1]
        cfg = synthetic.make_config(shape)
        e = exit_layer if exit_layer > 0 else synthetic.default_exit_layer(shape)
        gen_device = syn.device if str(syn.device).startswith("cuda") else "cpu"
        model = synthetic.build_model(cfg, seed=0, exit_layer=e, late_damping=syn.late_damping, dtype=torch.bfloat16,
                                      device=syn.device, gen_device=gen_device, layer_range=layer_range)
        return model, None
    from ..checkpoint import load_layer_range, resolve_checkpoint_dir
    path = resolve_checkpoint_dir(args.model)          # a directory, or a hub id as the reference passes it (generate.py:59-64)
    tokenizer = _load_tokenizer(path)
    # rank 0 embeds and drafts with its own head copy, the last rank runs the verify head; a middle rank needs neither tensor
    first = ctx is None or ctx.rank == 0
    last = ctx is None or ctx.rank == ctx.world - 1
    model = load_layer_range(path, layer_range, device=syn.device, dtype=torch.bfloat16, embed=first, head=first or last)
    return model, tokenizer


def make_strategy(cfg: GenerationConfig, ctx=None, partition=None, backend_factory=None):
    name = cfg.generation_strategy.replace("_hip", "")
    if ctx is not None:
        from ..pipeline_strategy import PIPELINE_STRATEGIES
        if name not in PIPELINE_STRATEGIES:
            raise ValueError(f"Unsupported generation strategy: {cfg.generation_strategy}")
        return PIPELINE_STRATEGIES[name](ctx, partition, backend_factory=backend_factory)
    from ..hip_strategies import STRATEGIES
    if name not in STRATEGIES:
        raise ValueError(f"Unsupported generation strategy: {cfg.generation_strategy}")
    return STRATEGIES[name]()


def run_on_rank0(ctx, strategy, model, fn):
    """Single process: `fn()`.  Multi-process: rank 0 runs `fn()` and then releases the others; ranks > 0 serve rank 0's
    generations (ONE loop per model, whichever strategy object rank 0 decodes with: layerskip_amd/pipeline_strategy.py).
    Returns fn's result on rank 0, None elsewhere."""
    if ctx is None:
        return fn()
    if ctx.rank != 0:
        strategy.serve(model)
        return None
    try:
        return fn()
    finally:
        try:
            strategy._decoder(model)        # even if rank 0 failed before its first generation, its peers are owed a shutdown
            strategy.shutdown()
        except Exception:                   # noqa: BLE001 -- the first error is the one worth reporting
            pass


def apply_template(message: str, template: Optional[str]) -> str:
    """reference data.py:40-53"""
    return message if template is None else template.format(message=message)


def parse_ids(text: str) -> List[int]:
    return [int(t) for t in text.replace(",", " ").split()]


def dump_json(obj, path: str) -> None:
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)
