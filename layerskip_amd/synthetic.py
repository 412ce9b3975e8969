"""Deterministic synthetic LayerSkip-shaped Llama checkpoints and prompts.

There is no network, tokenizer or checkpoint on the build / GPU boxes, so every
measurement and parity test runs on random-init weights of the real
architectures.  Late layers (>= exit_layer) have their
``o_proj`` / ``down_proj`` damped so that the early-exit head and the full model
agree on a non-trivial fraction of tokens (acceptance rate 0.3-0.8 instead of
~0 for an undamped random network) -- the draft/verify/rollback machinery is
then exercised the way a trained LayerSkip checkpoint exercises it.

The model object that comes out is a plain ``transformers.LlamaForCausalLM``:
exactly what the reference's ``GenerationStrategy.generate_token_ids`` receives
(reference self_speculation/generator_base.py:52-62).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import torch
import transformers

# Public model-card shapes of the checkpoints BASELINE.json names.
SHAPES: Dict[str, dict] = {
    "llama3.2-1B": dict(num_hidden_layers=16, hidden_size=2048, intermediate_size=8192,
                        num_attention_heads=32, num_key_value_heads=8, head_dim=64,
                        vocab_size=128256, rope_theta=500000.0, tie_word_embeddings=True,
                        rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                                          high_freq_factor=4.0,
                                          original_max_position_embeddings=8192),
                        max_position_embeddings=131072, exit_layer=4, num_speculations=4),
    "llama2-7B": dict(num_hidden_layers=32, hidden_size=4096, intermediate_size=11008,
                      num_attention_heads=32, num_key_value_heads=32, head_dim=128,
                      vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                      exit_layer=8, num_speculations=6),
    "llama3-8B": dict(num_hidden_layers=32, hidden_size=4096, intermediate_size=14336,
                      num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                      vocab_size=128256, rope_theta=500000.0, max_position_embeddings=8192,
                      exit_layer=8, num_speculations=6),
    "llama2-13B": dict(num_hidden_layers=40, hidden_size=5120, intermediate_size=13824,
                       num_attention_heads=40, num_key_value_heads=40, head_dim=128,
                       vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                       exit_layer=10, num_speculations=8),
    "llama2-70B": dict(num_hidden_layers=80, hidden_size=8192, intermediate_size=28672,
                       num_attention_heads=64, num_key_value_heads=8, head_dim=128,
                       vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                       exit_layer=12, num_speculations=12),
    # Small shapes for parity tests: every kernel code path of the big ones
    # (MHA / GQA, d=128 / d=64, multi-chunk K, ragged tiles) at CPU-oracle cost of seconds.
    "tiny-mha": dict(num_hidden_layers=6, hidden_size=256, intermediate_size=704,
                     num_attention_heads=2, num_key_value_heads=2, head_dim=128,
                     vocab_size=512, rope_theta=10000.0, max_position_embeddings=2048,
                     exit_layer=2, num_speculations=4),
    "tiny-gqa": dict(num_hidden_layers=6, hidden_size=512, intermediate_size=1408,
                     num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                     vocab_size=1000, rope_theta=500000.0, max_position_embeddings=2048,
                     exit_layer=3, num_speculations=6),
    "tiny-d64": dict(num_hidden_layers=4, hidden_size=256, intermediate_size=1024,
                     num_attention_heads=4, num_key_value_heads=2, head_dim=64,
                     vocab_size=768, rope_theta=500000.0, tie_word_embeddings=True,
                     rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                                       high_freq_factor=4.0,
                                       original_max_position_embeddings=8192),
                     max_position_embeddings=131072, exit_layer=2, num_speculations=4),
    # The exact llama2-7B projection / vocabulary shapes (K = 4096 and 11008, V = 32000) on 4 layers.
    "slice-7B": dict(num_hidden_layers=4, hidden_size=4096, intermediate_size=11008,
                     num_attention_heads=32, num_key_value_heads=32, head_dim=128,
                     vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                     exit_layer=2, num_speculations=6),
    # Wide enough that K > 4096 (two K-chunks in every projection) like 70B / 13B.
    "small-wide": dict(num_hidden_layers=4, hidden_size=5120, intermediate_size=6144,
                       num_attention_heads=40, num_key_value_heads=8, head_dim=128,
                       vocab_size=2048, rope_theta=10000.0, max_position_embeddings=2048,
                       exit_layer=2, num_speculations=5),
}


def make_config(name: str, **overrides) -> transformers.LlamaConfig:
    """LlamaConfig for one of ``SHAPES`` (keys that are not config fields are dropped)."""
    shape = dict(SHAPES[name])
    shape.update(overrides)
    shape.pop("exit_layer", None)
    shape.pop("num_speculations", None)
    rope_theta = shape.pop("rope_theta", 10000.0)
    rope_scaling = shape.pop("rope_scaling", None)
    rope_parameters = {"rope_type": "default", "rope_theta": rope_theta}
    if rope_scaling is not None:
        rope_parameters = dict(rope_scaling)
        rope_parameters["rope_theta"] = rope_theta
    cfg = transformers.LlamaConfig(
        rms_norm_eps=1e-5,
        attention_bias=False,
        mlp_bias=False,
        hidden_act="silu",
        rope_parameters=rope_parameters,
        bos_token_id=1,
        eos_token_id=2,
        pad_token_id=None,
        **shape,
    )
    return cfg


def default_exit_layer(name: str) -> int:
    return SHAPES[name]["exit_layer"]


def default_num_speculations(name: str) -> int:
    return SHAPES[name]["num_speculations"]


def _param_generator(seed: int, index: int, device: torch.device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed((seed * 1000003 + index * 7919 + 12345) % (2 ** 63 - 1))
    return g


@torch.no_grad()
def build_model(
    config: transformers.LlamaConfig,
    seed: int = 0,
    exit_layer: int = -1,
    late_damping: float = 0.1,
    dtype: torch.dtype = torch.bfloat16,
    device: str | torch.device = "cpu",
    gen_device: Optional[str | torch.device] = None,
    init_std: float = 0.02,
    layer_range: Optional[tuple] = None,
) -> transformers.LlamaForCausalLM:
    """Random-init ``LlamaForCausalLM`` with reproducible weights.

    Every parameter is drawn from its own generator (seed, parameter index), in fp32,
    ``normal(0, init_std)`` for matrices and ``1 + 0.1*normal`` for RMSNorm gains, then
    ``o_proj`` / ``down_proj`` of layers ``>= exit_layer`` are multiplied by
    ``late_damping`` and everything is cast to ``dtype``.  ``layer_range`` materialises only
    decoder layers [a, b) (pipeline ranks).  ``gen_device`` chooses where the random numbers are drawn: "cpu" (default; bit-reproducible everywhere, used by
    the golden fixtures) or the GPU (fast path for 7B+ shapes in bench.py).
    """
    device = torch.device(device)
    gen_device = torch.device(gen_device) if gen_device is not None else torch.device("cpu")
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(config)
    model.eval()
    tied = bool(getattr(config, "tie_word_embeddings", False))
    named = list(model.named_parameters())
    for index, (name, param) in enumerate(named):
        if tied and name == "lm_head.weight":
            continue
        if layer_range is not None and name.startswith("model.layers."):
            if not (layer_range[0] <= int(name.split(".")[2]) < layer_range[1]):
                continue        # stays on the meta device: a pipeline rank only materialises its own layers
        g = _param_generator(seed, index, gen_device)
        shape = tuple(param.shape)
        if param.dim() == 1:
            value = 1.0 + 0.1 * torch.randn(shape, generator=g, device=gen_device, dtype=torch.float32)
        else:
            value = init_std * torch.randn(shape, generator=g, device=gen_device, dtype=torch.float32)
            if exit_layer > 0 and (name.endswith("self_attn.o_proj.weight") or name.endswith("mlp.down_proj.weight")):
                layer_idx = int(name.split(".")[2])
                if layer_idx >= exit_layer:
                    value = value * late_damping
        value = value.to(dtype).to(device)
        _assign(model, name, value)
    if tied:
        model.lm_head.weight = model.model.embed_tokens.weight
    # Non-persistent buffers (rotary inv_freq) were created on the meta device: rebuild them.
    rotary = type(model.model.rotary_emb)(config).to(device)
    model.model.rotary_emb = rotary
    model.requires_grad_(False)
    return model


def _assign(model: torch.nn.Module, dotted: str, value: torch.Tensor) -> None:
    parts = dotted.split(".")
    mod = model
    for p in parts[:-1]:
        mod = getattr(mod, p)
    setattr(mod, parts[-1], torch.nn.Parameter(value, requires_grad=False))


def make_prompt(vocab_size: int, length: int, seed: int) -> List[int]:
    """``length`` token ids uniform in [3, vocab) from a CPU generator."""
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randint(3, vocab_size, (length,), generator=g).tolist()


@dataclasses.dataclass
class SyntheticCase:
    """One parity / bench case: model shape + weights seed + generation settings."""
    shape: str
    seed: int = 0
    late_damping: float = 0.1
    exit_layer: Optional[int] = None
    num_speculations: Optional[int] = None
    prompt_len: int = 24
    prompt_seed: int = 0
    max_steps: int = 32

    def resolved(self) -> "SyntheticCase":
        c = dataclasses.replace(self)
        if c.exit_layer is None:
            c.exit_layer = default_exit_layer(self.shape)
        if c.num_speculations is None:
            c.num_speculations = default_num_speculations(self.shape)
        return c
