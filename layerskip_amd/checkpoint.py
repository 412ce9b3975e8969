"""Materialise a LAYER RANGE of a Llama checkpoint on one device (one pipeline rank = one process = one GPU).

The reference loads every checkpoint whole and lets accelerate spread it (`device_map="auto"`, reference generate.py:54-67):
one process drives all GPUs through forward hooks that copy activations from device to device.  The MI355X-native partition
is one process per GPU with point-to-point hand-offs (layerskip_amd/pipeline.py), so a rank needs exactly its decoder layers
`[a, b)` plus the small shared tensors (embedding, final norm, lm_head: rank 0 drafts with its own head copy, the last rank runs
the verify head) -- and must never pull the other 7/8 of a 140 GB checkpoint through its HBM.

`load_layer_range` reads a `save_pretrained` / hub-layout directory (`config.json` + `model.safetensors` or the sharded
`model-0000x-of-0000y.safetensors` with `model.safetensors.index.json`), opens only the shard files that hold tensors this rank
owns and copies those tensors straight to the rank's device.  Decoder layers outside the range stay on the meta device (no
storage at all); `HipEngine(model, layer_range=(a, b))` only touches the owned ones.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch
import transformers

INDEX_NAME = "model.safetensors.index.json"
SINGLE_NAME = "model.safetensors"


def _weight_map(path: str) -> Dict[str, str]:
    """tensor name -> file name (relative to `path`)."""
    index = os.path.join(path, INDEX_NAME)
    if os.path.exists(index):
        with open(index) as f:
            return dict(json.load(f)["weight_map"])
    single = os.path.join(path, SINGLE_NAME)
    if os.path.exists(single):
        from safetensors import safe_open
        with safe_open(single, framework="pt", device="cpu") as f:
            return {k: SINGLE_NAME for k in f.keys()}
    raise FileNotFoundError(f"{path}: neither {INDEX_NAME} nor {SINGLE_NAME} (only safetensors checkpoints are read; the reference "
                            f"passes use_safetensors=True as well, generate.py:61)")


def owned_parameter_names(names: Iterable[str], layer_range: Optional[Sequence[int]]) -> list:
    """The parameters a rank materialises: its decoder layers and everything that is not a decoder layer."""
    out = []
    for name in names:
        if layer_range is not None and name.startswith("model.layers."):
            if not (layer_range[0] <= int(name.split(".")[2]) < layer_range[1]):
                continue
        out.append(name)
    return out


def _assign(model: torch.nn.Module, dotted: str, value: torch.Tensor) -> None:
    mod = model
    parts = dotted.split(".")
    for p in parts[:-1]:
        mod = getattr(mod, p)
    setattr(mod, parts[-1], torch.nn.Parameter(value, requires_grad=False))


@torch.no_grad()
def load_layer_range(path: str, layer_range: Optional[Sequence[int]] = None, device: str | torch.device = "cuda:0",
                     dtype: torch.dtype = torch.bfloat16) -> transformers.LlamaForCausalLM:
    """`LlamaForCausalLM` with decoder layers `[a, b)` (all of them for `layer_range=None`), the embedding, the final norm and
    the lm_head on `device` in `dtype`; the other decoder layers on the meta device.  `model.loaded_layer_range` = (a, b)."""
    device = torch.device(device)
    config = transformers.AutoConfig.from_pretrained(path)
    if getattr(config, "model_type", "llama") != "llama":
        raise ValueError(f"{path}: model_type {config.model_type!r}; the engine implements the Llama decoder (SURVEY.md 8a)")
    n_layers = config.num_hidden_layers
    rng: Tuple[int, int] = (0, n_layers) if layer_range is None else (int(layer_range[0]), int(layer_range[1]))
    if not (0 <= rng[0] < rng[1] <= n_layers):
        raise ValueError(f"layer_range {rng} outside [0, {n_layers}]")
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(config)
    model.eval()
    wmap = _weight_map(path)
    tied = bool(getattr(config, "tie_word_embeddings", False))
    wanted = owned_parameter_names([n for n, _ in model.named_parameters()], rng)
    by_file: Dict[str, list] = {}
    for name in wanted:
        if name == "lm_head.weight" and name not in wmap:
            if tied:
                continue            # tied checkpoints store the embedding only
            raise KeyError(f"{path}: lm_head.weight is missing and the config does not tie it to the embedding")
        if name not in wmap:
            raise KeyError(f"{path}: tensor {name} not in the checkpoint")
        by_file.setdefault(wmap[name], []).append(name)
    from safetensors import safe_open
    for fname, names in sorted(by_file.items()):
        # only the shard files that hold owned tensors are opened; tensors go file -> device without a host-side model copy
        with safe_open(os.path.join(path, fname), framework="pt", device=str(device)) as f:
            for name in names:
                _assign(model, name, f.get_tensor(name).to(dtype).contiguous())
    if tied or model.lm_head.weight.device.type == "meta":
        model.lm_head.weight = model.model.embed_tokens.weight
    # non-persistent buffers (rotary inv_freq) were created on the meta device: rebuild them
    model.model.rotary_emb = type(model.model.rotary_emb)(config).to(device)
    model.requires_grad_(False)
    model.loaded_layer_range = rng
    return model
