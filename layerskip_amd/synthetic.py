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
    # a checkpoint with ONE ADDED TOKEN (V = 32 001: not a multiple of 16, nor of 4 -- the last lm_head tile holds one real column,
    # sampling rows are padded to 32 004): what `resize_token_embeddings(len(tokenizer) + 1)` produces
    "tiny-gqa-v32001": dict(num_hidden_layers=6, hidden_size=512, intermediate_size=1408,
                            num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                            vocab_size=32001, rope_theta=500000.0, max_position_embeddings=2048,
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
    # 4-layer slices with the exact projection / vocabulary / RoPE geometry of the other BASELINE configs
    "slice-8B": dict(num_hidden_layers=4, hidden_size=4096, intermediate_size=14336,
                     num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                     vocab_size=128256, rope_theta=500000.0, max_position_embeddings=8192,
                     exit_layer=2, num_speculations=6),
    "slice-13B": dict(num_hidden_layers=4, hidden_size=5120, intermediate_size=13824,
                      num_attention_heads=40, num_key_value_heads=40, head_dim=128,
                      vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                      exit_layer=2, num_speculations=8),
    "slice-1B": dict(num_hidden_layers=4, hidden_size=2048, intermediate_size=8192,
                     num_attention_heads=32, num_key_value_heads=8, head_dim=64,
                     vocab_size=128256, rope_theta=500000.0, tie_word_embeddings=True,
                     rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                                       high_freq_factor=4.0,
                                       original_max_position_embeddings=8192),
                     max_position_embeddings=131072, exit_layer=2, num_speculations=4),
    # llama2-70B geometry on 4 layers: H = 8192 (two K-chunks), I = 28672 (seven K-chunks in down_proj), 64 : 8 GQA, and the
    # config's 12 speculations (a 13-row verify block: the 16-row template, one query head per attention workgroup)
    "slice-70B": dict(num_hidden_layers=4, hidden_size=8192, intermediate_size=28672,
                      num_attention_heads=64, num_key_value_heads=8, head_dim=128,
                      vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                      exit_layer=2, num_speculations=12),
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


# --------------------------------------------------------------------------------------------------
# Structured checkpoints: greedy decoding with HEALTHY top-2 margins on every decision.
# --------------------------------------------------------------------------------------------------
# Random-init logits are Gaussian, so a few percent of greedy decisions are ties at bf16 resolution and no
# fixture built from them can show token-exact parity (the reference is not even reproducible against itself
# there).  A structured checkpoint makes every decision of the draft head AND of the full model a wide-margin
# one while every kernel still does real arithmetic:
#   * token embeddings are random UNIT vectors u_t; lm_head rows are the same directions (tied, or an untied
#     noisy copy), the final RMSNorm gain scales the logits to <= ~16;
#   * an "active" vocabulary A is ordered into one cycle pi; a lookup MLP in an EARLY layer (gate/up rows keyed
#     on u_t, down column 4*u_pi(t)) makes the early-exit head predict pi(t);
#   * a lookup MLP in a LATE layer, keyed on u_pi(t) for t in a subset B of A, adds 16*u_sigma(t) with
#     sigma = pi^jump: the full model predicts sigma(t) for t in B and pi(t) otherwise.  Drafts are therefore
#     rejected exactly behind tokens of B: |B|/|A| sets the acceptance rate (the role damping played for the
#     random checkpoints);
#   * all other weights are small random matrices; q/k give O(1) score spreads and v/o carry an identity
#     component, so the whole context (every KV entry, RoPE, the masks) moves the logits by many bf16 ulp --
#     visible to the teacher-forced logits tests -- without ever approaching the decision margins.
STRUCT_DEFAULTS = dict(active=96, override_frac=0.3, jump=3, early_gain=4.0, late_gain=16.0, key_gain=6.0,
                       logit_scale=16.0, noise=0.5, down_noise=2.0, mix=2.0, qk_gain=1.6, head_noise=0.02)


def _unit_rows(n: int, h: int, g: torch.Generator) -> torch.Tensor:
    v = torch.randn(n, h, generator=g, dtype=torch.float32)
    return v / v.norm(dim=1, keepdim=True)


@torch.no_grad()
def build_structured_model(config: transformers.LlamaConfig, seed: int = 0, exit_layer: int = 2,
                           dtype: torch.dtype = torch.bfloat16, device: str | torch.device = "cpu",
                           layer_range: Optional[tuple] = None, **knobs) -> transformers.LlamaForCausalLM:
    """Deterministic (CPU generator) structured ``LlamaForCausalLM``; see the block comment above.
    The token program is attached as ``model.struct_program`` (dict: active, pi, sigma, override).

    Scales (x = RMS-normed residual row, |x| = sqrt(H)): q/k rows ~ N(0, 1/H) (scores of std ~1); v = x on the kv
    dims / sqrt(H) plus noise, o scatters each head back with weight ``mix`` plus noise: an attention block moves
    the residual by <~ 0.3 at |h| >= 1; random gate/up pre-activations ~ N(0, noise^2), random down columns
    sized so that an MLP block moves it by ~0.3 as well; the lookup rows give 4 (early) and 16 (late)."""
    k = dict(STRUCT_DEFAULTS)
    k.update(knobs)
    device = torch.device(device)
    H, I, L, V = config.hidden_size, config.intermediate_size, config.num_hidden_layers, config.vocab_size
    hd = getattr(config, "head_dim", None) or H // config.num_attention_heads
    nh, nkv = config.num_attention_heads, config.num_key_value_heads
    n_act = int(min(k["active"], I // 2, V - 3))
    if not (1 <= exit_layer < L):
        raise ValueError("exit_layer must be in [1, num_layers)")
    g = torch.Generator().manual_seed((seed * 2654435761 + 97) % (2 ** 63 - 1))
    with torch.device("meta"):
        model = transformers.LlamaForCausalLM(config)
    model.eval()
    tied = bool(getattr(config, "tie_word_embeddings", False))
    rs = 1.0 / (H ** 0.5)

    def put(name, value):
        _assign(model, name, value.to(dtype).to(device))

    embed = _unit_rows(V, H, g)
    put("model.embed_tokens.weight", embed)
    if tied:
        model.lm_head.weight = model.model.embed_tokens.weight
    else:
        put("lm_head.weight", embed + k["head_noise"] * rs * torch.randn(V, H, generator=g))
    put("model.norm.weight", torch.full((H,), k["logit_scale"] * rs) * (1.0 + 0.05 * torch.randn(H, generator=g)))
    # ---- token program -------------------------------------------------------------------------
    perm = torch.randperm(V - 3, generator=g)[:n_act] + 3
    cyc = perm.tolist()
    pi = {cyc[i]: cyc[(i + 1) % n_act] for i in range(n_act)}
    jump = max(2, int(k["jump"]))
    pos_of = {t: i for i, t in enumerate(cyc)}
    sigma = {t: cyc[(pos_of[t] + jump) % n_act] for t in cyc}
    n_over = max(1, int(round(k["override_frac"] * n_act))) if k["override_frac"] > 0 else 0     # 0: the full model agrees with its early exit
    override = [cyc[i] for i in torch.randperm(n_act, generator=g)[:n_over].tolist()]
    early_layer, late_layer = 0, exit_layer
    kg = k["key_gain"] * rs          # gate/up rows: kg * u  ->  pre-activation key_gain * cos(x, u)
    act = k["key_gain"] * k["key_gain"] * (1.0 / (1.0 + 2.718281828 ** (-k["key_gain"])))   # silu(g) * u at cos = 1
    kvd = min(nkv * hd, H)
    group = nh // nkv
    for idx in range(L):
        p = f"model.layers.{idx}."
        # every layer draws the same amount of randomness whether or not this rank materialises it
        n1 = 1.0 + 0.1 * torch.randn(H, generator=g)
        n2 = 1.0 + 0.1 * torch.randn(H, generator=g)
        wq = k["qk_gain"] * rs * torch.randn(nh * hd, H, generator=g)
        wk = k["qk_gain"] * rs * torch.randn(nkv * hd, H, generator=g)
        wv = k["noise"] * rs * rs * torch.randn(nkv * hd, H, generator=g)
        wo = k["noise"] * torch.randn(H, nh * hd, generator=g) / ((nh * hd) ** 0.5)
        wg = k["noise"] * rs * torch.randn(I, H, generator=g)
        wu = k["noise"] * rs * torch.randn(I, H, generator=g)
        wd = k["down_noise"] * torch.randn(H, I, generator=g) / ((I * H) ** 0.5)
        if layer_range is not None and not (layer_range[0] <= idx < layer_range[1]):
            continue
        # identity component: v = x[kv dims] / sqrt(H); o scatters head h's output back onto those dims
        wv[torch.arange(kvd), torch.arange(kvd)] += rs
        for h in range(nh):
            kvh = h // group
            rows = torch.arange(kvh * hd, min((kvh + 1) * hd, H))
            if rows.numel():
                wo[rows, h * hd + (rows - kvh * hd)] += k["mix"] / (group * L ** 0.5)
        if idx == early_layer:
            keys = embed[cyc]                                   # [n_act, H] key on u_t
            vals = embed[[pi[t] for t in cyc]]                  # value u_pi(t)
            wg[:n_act] = kg * keys
            wu[:n_act] = kg * keys
            wd[:, :n_act] = (k["early_gain"] / act) * vals.t()
        if idx == late_layer and n_over:
            keys = embed[[pi[t] for t in override]]             # key on the early path's prediction u_pi(t)
            vals = embed[[sigma[t] for t in override]]
            wg[:n_over] = kg * keys
            wu[:n_over] = kg * keys
            wd[:, :n_over] = (k["late_gain"] / act) * vals.t()
        put(p + "input_layernorm.weight", n1)
        put(p + "post_attention_layernorm.weight", n2)
        put(p + "self_attn.q_proj.weight", wq)
        put(p + "self_attn.k_proj.weight", wk)
        put(p + "self_attn.v_proj.weight", wv)
        put(p + "self_attn.o_proj.weight", wo)
        put(p + "mlp.gate_proj.weight", wg)
        put(p + "mlp.up_proj.weight", wu)
        put(p + "mlp.down_proj.weight", wd)
    rotary = type(model.model.rotary_emb)(config).to(device)
    model.model.rotary_emb = rotary
    model.requires_grad_(False)
    model.struct_program = {"active": cyc, "pi": pi, "sigma": sigma, "override": sorted(override),
                            "early_layer": early_layer, "late_layer": late_layer}
    return model


def struct_next_token(program: dict, tok: int, full: bool) -> int:
    """What the structured checkpoint is built to emit after ``tok``: the early-exit head's pi(t), or the full
    model's sigma(t) behind an override token."""
    if full and tok in set(program["override"]):
        return program["sigma"][tok]
    return program["pi"][tok]


def make_struct_prompt(program: dict, length: int, seed: int) -> List[int]:
    """``length`` tokens drawn from the checkpoint's active vocabulary (CPU generator)."""
    g = torch.Generator().manual_seed(5000 + seed)
    act = program["active"]
    return [act[i] for i in torch.randint(0, len(act), (length,), generator=g).tolist()]
