"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  CPU restatement of the reference hot path.

This file restates, with plain torch CPU ops on explicit weight tensors, the
algorithm of the reference's self-speculative decoding path:

  * control flow: reference self_speculation/llama_model_utils.py (``LMU``) and
    self_speculation/self_speculation_generator.py (``SSG``),
    self_speculation/autoregressive_generator.py (``ARG``);
  * arithmetic: the third-party HF ``transformers`` Llama blocks the reference
    calls into (pinned ``transformers==4.50.0`` in the reference's
    requirements.txt:4; validated here against the installed 5.15.0,
    models/llama/modeling_llama.py = ``TF5``).  The reference tree holds no
    arithmetic of its own and no golden vectors.

Parity pinning: ``oracle/make_golden.py`` runs the UNMODIFIED reference modules
(through ``oracle/ref_shim.py``) on deterministic synthetic checkpoints and checks
that this restatement reproduces them bit-for-bit on CPU (token ids, acceptance
counters, logits) in fp32 and bf16; the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` on every box.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs (``cpu_baseline``, and
the optional ``--gpu-reference`` leg that runs this same restatement with its tensors on the GPU, i.e. the
reference's torch-ROCm eager path) may import this module -- always as the checker / the thing timed BESIDE the
engine, never as part of it.  The product (``layerskip_amd``) never imports it.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
@dataclass
class LayerWeights:
    input_norm: torch.Tensor
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    o: torch.Tensor
    post_norm: torch.Tensor
    gate: torch.Tensor
    up: torch.Tensor
    down: torch.Tensor


@dataclass
class OracleModel:
    embed: torch.Tensor
    layers: List[LayerWeights]
    final_norm: torch.Tensor
    lm_head: torch.Tensor
    inv_freq: torch.Tensor
    attention_scaling: float
    n_heads: int
    n_kv_heads: int
    head_dim: int
    eps: float
    attn_impl: str = "sdpa"  # what LlamaForCausalLM(config) selects by default in TF5
    dtype: torch.dtype = torch.float32
    device: torch.device = torch.device("cpu")   # "cuda" = the reference's torch-ROCm eager path (baseline only)

    @property
    def num_layers(self) -> int:
        return len(self.layers)

    @classmethod
    def from_hf(cls, model, dtype: Optional[torch.dtype] = None, attn_impl: Optional[str] = None,
                device: str = "cpu") -> "OracleModel":
        """Borrow (or convert) the weights of a ``transformers.LlamaForCausalLM``."""
        cfg = model.config
        dev = torch.device(device)

        def get(t):
            t = t.detach().to(dev)
            return t if dtype is None else t.to(dtype)

        layers = []
        for layer in model.model.layers:
            a, m = layer.self_attn, layer.mlp
            layers.append(LayerWeights(
                input_norm=get(layer.input_layernorm.weight),
                q=get(a.q_proj.weight), k=get(a.k_proj.weight), v=get(a.v_proj.weight), o=get(a.o_proj.weight),
                post_norm=get(layer.post_attention_layernorm.weight),
                gate=get(m.gate_proj.weight), up=get(m.up_proj.weight), down=get(m.down_proj.weight)))
        head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        rot = model.model.rotary_emb
        embed = get(model.model.embed_tokens.weight)
        return cls(
            embed=embed, layers=layers, final_norm=get(model.model.norm.weight),
            lm_head=get(model.lm_head.weight),
            inv_freq=rot.inv_freq.detach().to(dev, torch.float32).clone(),
            attention_scaling=float(rot.attention_scaling),
            n_heads=cfg.num_attention_heads, n_kv_heads=cfg.num_key_value_heads, head_dim=head_dim,
            eps=float(cfg.rms_norm_eps),
            attn_impl=attn_impl or getattr(cfg, "_attn_implementation", "sdpa") or "sdpa",
            dtype=embed.dtype, device=dev)


# --------------------------------------------------------------------------------------
# arithmetic (TF5 = transformers/models/llama/modeling_llama.py)
# --------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """TF5:62-67 -- fp32 statistics, cast to input dtype, THEN multiply by the gain."""
    in_dtype = x.dtype
    x32 = x.to(torch.float32)
    variance = x32.pow(2).mean(-1, keepdim=True)
    x32 = x32 * torch.rsqrt(variance + eps)
    return weight * x32.to(in_dtype)


def rope_cos_sin(inv_freq: torch.Tensor, attention_scaling: float, position_ids: torch.Tensor,
                 dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """TF5:113-127 -- fp32 outer product, cat(freqs, freqs), cos/sin in fp32, cast to model dtype."""
    inv = inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
    pos = position_ids[:, None, :].float()
    freqs = (inv @ pos).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos = emb.cos() * attention_scaling
    sin = emb.sin() * attention_scaling
    return cos.to(dtype), sin.to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """TF5:130-134."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """TF5:138-160 (unsqueeze_dim=1) -- three roundings per element in the model dtype."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """TF5:179-188."""
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def attention_core(om: OracleModel, q, k, v, mask):
    """TF5:191-213 (eager) or transformers/integrations/sdpa_attention.py (sdpa, the default).

    The reference always passes an additive float mask (LMU:21-73), so the sdpa branch
    runs with ``attn_mask=mask, is_causal=False`` and explicit ``repeat_kv`` for GQA.
    """
    n_rep = om.n_heads // om.n_kv_heads
    scaling = om.head_dim ** -0.5
    k = repeat_kv(k, n_rep)
    v = repeat_kv(v, n_rep)
    if om.attn_impl == "eager":
        w = torch.matmul(q, k.transpose(2, 3)) * scaling
        if mask is not None:
            w = w + mask
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(w, v)
    else:
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=scaling, is_causal=False)
    return out.transpose(1, 2).contiguous()


def decoder_layer(om: OracleModel, lw: LayerWeights, h, mask, position_ids, kv: Optional[Tuple[torch.Tensor, torch.Tensor]]):
    """TF5:295-324 + TF5:243-281.  ``kv`` is this layer's (K, V) so far or None; returns (h, (K, V))."""
    b, m, _ = h.shape
    resid = h
    x = rms_norm(h, lw.input_norm, om.eps)
    q = F.linear(x, lw.q).view(b, m, -1, om.head_dim).transpose(1, 2)
    k = F.linear(x, lw.k).view(b, m, -1, om.head_dim).transpose(1, 2)
    v = F.linear(x, lw.v).view(b, m, -1, om.head_dim).transpose(1, 2)
    cos, sin = rope_cos_sin(om.inv_freq, om.attention_scaling, position_ids, h.dtype)
    q, k = apply_rope(q, k, cos, sin)
    if kv is not None and kv[0] is not None and kv[0].numel() > 0:
        k = torch.cat([kv[0], k], dim=-2)  # DynamicLayer.update, cache_utils.py:144-145
        v = torch.cat([kv[1], v], dim=-2)
    a = attention_core(om, q, k, v, mask)
    a = F.linear(a.reshape(b, m, -1).contiguous(), lw.o)
    h = resid + a
    resid = h
    x = rms_norm(h, lw.post_norm, om.eps)
    x = F.linear(F.silu(F.linear(x, lw.gate)) * F.linear(x, lw.up), lw.down)  # TF5:174-176
    return resid + x, (k, v)


def head(om: OracleModel, h):
    """LMU:204-205 / :271-273 / :386-387 -- final RMSNorm then lm_head on ALL rows."""
    return F.linear(rms_norm(h, om.final_norm, om.eps), om.lm_head)


# --------------------------------------------------------------------------------------
# masks (LMU:21-73)
# --------------------------------------------------------------------------------------
def make_causal_mask(tgt_len: int, dtype: torch.dtype, past: int, device="cpu") -> torch.Tensor:
    """LMU:45-59."""
    mask = torch.full((tgt_len, tgt_len), torch.finfo(dtype).min, device=device)
    cond = torch.arange(tgt_len, device=device)
    mask.masked_fill_(cond < (cond + 1).view(tgt_len, 1), 0)
    mask = mask.to(dtype)
    if past > 0:
        mask = torch.cat([torch.zeros(tgt_len, past, dtype=dtype, device=device), mask], dim=-1)
    return mask[None, None, :, :]


def decoder_mask(tgt_len: int, src_len: int, dtype: torch.dtype, past: int, device="cpu") -> torch.Tensor:
    """LMU:21-42 with an all-ones boolean attention_mask of width ``src_len`` (LMU:62-73)."""
    ones = torch.ones(1, src_len, dtype=torch.bool, device=device)
    expanded = ones[:, None, None, :].expand(1, 1, tgt_len, src_len).to(dtype)
    inverted = 1.0 - expanded
    expanded_mask = inverted.masked_fill(inverted.to(torch.bool), torch.finfo(dtype).min)
    if tgt_len > 1:
        return expanded_mask + make_causal_mask(tgt_len, dtype, past, device)
    return expanded_mask


# --------------------------------------------------------------------------------------
# LMU forward functions.  past = list (len = #layers that have a cache) of (K, V).
# --------------------------------------------------------------------------------------
@dataclass
class ForwardResult:
    logits: torch.Tensor
    past: List[Tuple[torch.Tensor, torch.Tensor]]
    exit_query_cache: Optional[torch.Tensor] = None


def _kv(past, idx):
    return past[idx] if (past is not None and idx < len(past)) else None


def forward(om: OracleModel, input_ids: torch.Tensor, past) -> ForwardResult:
    """LMU:155-209."""
    _, m = input_ids.shape
    past_len = past[0][0].shape[2] if past else 0
    input_ids = input_ids.to(om.device)
    pos = torch.arange(past_len, past_len + m, dtype=torch.long, device=om.device).unsqueeze(0)
    h = F.embedding(input_ids, om.embed)
    mask = decoder_mask(m, past_len + m, h.dtype, past_len, om.device)
    new_past = []
    for idx, lw in enumerate(om.layers):
        h, kv = decoder_layer(om, lw, h, mask, pos, _kv(past, idx))
        new_past.append(kv)
    return ForwardResult(head(om, h), new_past)


def forward_early(om: OracleModel, input_ids, past, exit_layer: int, exit_query_cache) -> ForwardResult:
    """LMU:213-276.  Layers >= exit_layer keep whatever cache they already had."""
    _, m = input_ids.shape
    past_len = past[0][0].shape[2] if past else 0
    input_ids = input_ids.to(om.device)
    pos = torch.arange(past_len, past_len + m, dtype=torch.long, device=om.device).unsqueeze(0)
    h = F.embedding(input_ids, om.embed)
    mask = decoder_mask(m, past_len + m, h.dtype, past_len, om.device)
    new_past = list(past) if past else []
    for idx, lw in enumerate(om.layers[:exit_layer]):
        h, kv = decoder_layer(om, lw, h, mask, pos, _kv(past, idx))
        if idx < len(new_past):
            new_past[idx] = kv
        else:
            new_past.append(kv)
    eqc = h if exit_query_cache is None else torch.cat([exit_query_cache, h], dim=1)  # LMU:266-269
    return ForwardResult(head(om, h), new_past, eqc)


def forward_remainder(om: OracleModel, input_ids, past, exit_layer: int, exit_query_cache) -> ForwardResult:
    """LMU:280-391."""
    _, seq = input_ids.shape
    n_gen = 1
    draft_past = 0
    full_past = 0
    seq_with_past = seq
    if past:
        draft_past = past[0][0].shape[2]                                   # LMU:296
        full_past = past[-1][0].shape[2] if len(past) == om.num_layers else 0  # LMU:301-305
        seq_with_past = n_gen + draft_past                                 # LMU:307
    input_ids = input_ids.to(om.device)
    h = F.embedding(input_ids, om.embed)
    pos = torch.arange(full_past, seq_with_past, dtype=torch.long, device=om.device).unsqueeze(0).view(-1, seq)
    early_mask = decoder_mask(n_gen, seq_with_past, h.dtype, draft_past, om.device)
    full_mask = decoder_mask(seq, seq_with_past, h.dtype, full_past, om.device)
    new_past = list(past) if past else []
    full_h = None
    for idx, lw in enumerate(om.layers):
        if idx < exit_layer:
            h, kv = decoder_layer(om, lw, h[:, -n_gen:], early_mask, pos[:, -n_gen:], _kv(past, idx))
        else:
            if full_h is None and exit_query_cache is not None:
                full_h = torch.cat([exit_query_cache, h[:, -n_gen:]], dim=1)   # LMU:364-371
            else:
                full_h = h
            h, kv = decoder_layer(om, lw, full_h, full_mask, pos, _kv(past, idx))
        if idx < len(new_past):
            new_past[idx] = kv
        else:
            new_past.append(kv)
    return ForwardResult(head(om, h), new_past, exit_query_cache)


def crop_past(past, maximum_length: int):
    """LMU:134-149 (slice views)."""
    return [(k[:, :, :maximum_length, :], v[:, :, :maximum_length, :]) for (k, v) in past]


def decode_next_token_greedy(logits: torch.Tensor, token_idx=None) -> torch.Tensor:
    """LMU:109-122, sample=False branch.  ``token_idx`` is used for truthiness only (LMU:117)."""
    if token_idx:
        logits = logits[:, -1, :]
    return logits.argmax(dim=-1)


# --------------------------------------------------------------------------------------
# strategies (greedy).  Traces record what the HIP engine must reproduce.
# --------------------------------------------------------------------------------------
@dataclass
class StepTrace:
    num_drafts: int
    num_matches: int
    draft_tokens: List[int]
    verified_tokens: List[int]


@dataclass
class GenerationTrace:
    predicted_tokens: List[int]
    acceptance_rate: Optional[float]
    steps: List[StepTrace] = field(default_factory=list)
    # top-1 minus top-2 of the (model-dtype) verify logits row that produced each emitted token
    margins: List[float] = field(default_factory=list)
    # the same for every draft-head decision of the run (SSG:140-141), in units of the bf16 ulp of the top logit;
    # and the emitted-token margins in those units: what "a healthy decision" is measured in (scale invariant)
    draft_margins_ulp: List[float] = field(default_factory=list)
    margins_ulp: List[float] = field(default_factory=list)


def _margin(logits_row: torch.Tensor) -> float:
    top2 = torch.topk(logits_row.float(), 2).values
    return float(top2[0] - top2[1])


def bf16_ulp(value: float) -> float:
    """Spacing of bf16 numbers at |value| (8 significand bits)."""
    import math
    a = abs(float(value))
    if a == 0.0 or not math.isfinite(a):
        return 2.0 ** -133
    return 2.0 ** (math.floor(math.log2(a)) - 7)


def _margin_ulp(logits_row: torch.Tensor) -> float:
    top2 = torch.topk(logits_row.float(), 2).values
    return float(top2[0] - top2[1]) / bf16_ulp(float(top2[0]))


def single_step_speculation(om, input_ids, input_ids_list, output_ids, num_speculations, past, eos_token_ids,
                            exit_layer, margins: Optional[List[float]] = None, draft_margins_ulp: Optional[List[float]] = None,
                            margins_ulp: Optional[List[float]] = None):
    """SSG:102-229, sample=False, no processors / criteria / streamer."""
    prompt_length = input_ids.size(1)
    draft_input = input_ids.clone()
    drafts: List[int] = []
    eqc = None
    for _ in range(num_speculations):
        r = forward_early(om, draft_input, past, exit_layer, eqc)
        past, eqc = r.past, r.exit_query_cache
        tok = int(decode_next_token_greedy(r.logits, token_idx=-1).item())
        drafts.append(tok)
        if draft_margins_ulp is not None:
            draft_margins_ulp.append(_margin_ulp(r.logits[0, -1]))
        draft_input = torch.tensor([[tok]])
        if tok in eos_token_ids:
            break
    input_ids = input_ids.to(om.device)
    draft_t = torch.tensor(drafts, dtype=input_ids.dtype, device=om.device).unsqueeze(0)
    if len(drafts) == 0:
        draft_t = draft_t.reshape(1, 0)
    prefill = torch.cat([input_ids, draft_t], dim=-1)
    vr = forward_remainder(om, prefill.int(), past, exit_layer, eqc)
    past = vr.past
    vlogits = vr.logits[:, prompt_length - 1:, :]
    verified = decode_next_token_greedy(vlogits).to(prefill)
    ok = draft_t == verified[:, :-1]
    n = int(((~ok).cumsum(dim=-1) < 1).sum().item())   # SSG:190
    new_input = verified[:, n:n + 1]
    output_ids = output_ids + drafts[:n] + verified[0, n:n + 1].tolist()
    if margins is not None:
        for i in range(n + 1):
            margins.append(_margin(vlogits[0, i]))
    if margins_ulp is not None:
        for i in range(n + 1):
            margins_ulp.append(_margin_ulp(vlogits[0, i]))
    past = crop_past(past, len(input_ids_list) + len(output_ids) - 1)  # SSG:219-221
    return new_input, output_ids, past, n, len(drafts), StepTrace(len(drafts), n, list(drafts), verified[0].tolist())


def self_speculative_generate(om: OracleModel, input_ids: List[int], eos_token_ids: List[int], max_steps: int,
                              exit_layer: int, num_speculations: int) -> GenerationTrace:
    """SSG:32-99 (greedy)."""
    past = None
    ids = torch.tensor([input_ids])
    out: List[int] = []
    matches = 0
    gens = 0
    steps: List[StepTrace] = []
    margins: List[float] = []
    dm_ulp: List[float] = []
    m_ulp: List[float] = []
    while len(out) < max_steps:
        ids, out, past, n, td, tr = single_step_speculation(
            om, ids, input_ids, out, min(num_speculations, max_steps - len(out) - 1), past, eos_token_ids,
            exit_layer, margins, dm_ulp, m_ulp)
        steps.append(tr)
        matches += n
        gens += td
        eos_found = False
        for e in eos_token_ids:
            if e in out:
                out = out[: out.index(e)]
                eos_found = True
                break
        if eos_found:
            break
    rate = matches / gens  # ZeroDivisionError when no draft was ever made, like SSG:98
    return GenerationTrace(out, rate, steps, margins[: len(out)], dm_ulp, m_ulp[: len(out)])


def autoregressive_generate(om: OracleModel, input_ids: List[int], eos_token_ids: List[int], max_steps: int,
                            exit_layer: int = -1) -> GenerationTrace:
    """ARG:26-80 (greedy); ``exit_layer > 0`` = early-exit-only decoding (ARG:44-51)."""
    past = None
    ids = torch.tensor([input_ids])
    out: List[int] = []
    margins: List[float] = []
    eqc = None
    for _ in range(max_steps):
        if exit_layer > 0:
            r = forward_early(om, ids, past, exit_layer, eqc)
        else:
            r = forward(om, ids, past)
        past = r.past
        tok = int(decode_next_token_greedy(r.logits, token_idx=-1).item())
        if tok in eos_token_ids:
            break
        margins.append(_margin(r.logits[0, -1]))
        out.append(tok)
        ids = torch.tensor([[tok]])
    return GenerationTrace(out, None, [], margins)


def teacher_forced_logits(om: OracleModel, token_ids: List[int]) -> torch.Tensor:
    """Full-depth logits of every position of ``token_ids`` in one pass (LMU.forward, no past)."""
    return forward(om, torch.tensor([token_ids]), None).logits[0]
