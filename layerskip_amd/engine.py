"""Host side of the MI355X self-speculative decoding engine.

``HipEngine`` borrows a ``transformers.LlamaForCausalLM`` (the object the reference's strategies are
handed, reference self_speculation/generator_base.py:52-62), re-lays its projection weights once into
the MFMA-fragment layout the HIP kernels stream, owns the paged KV pool and the workspace, and exposes
the C-ABI entry points of ``include/layerskip_hip.h`` as methods.

PyTorch is used for storage and stream handles only: every tensor here is a buffer whose
``data_ptr()`` is passed to liblayerskip_hip.so.  No torch op computes anything on the decoding path.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import LskConfig, LskStepResult

BUF_STEP = 0   # 16-row step buffer (draft rows / verify block)
BUF_BULK = 1   # prompt rows (prefill), doubles as the exit_query_cache of the first step
BUF_MSG = 2    # the layer pipeline's message: row 0 = header words, rows 1.. = the verify block (lsk_pipeline_*)


@dataclass
class StepResult:
    """What ``single_step_speculation`` returns in the reference (SSG:223-229) plus the tokens."""
    num_matches: int
    num_drafts: int
    next_token: int
    kv_len: int
    emitted: List[int]
    draft_tokens: List[int]
    verified_tokens: List[int]


def _round_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def _i32_array(values: Sequence[int]):
    arr = (ctypes.c_int32 * max(1, len(values)))()
    for i, v in enumerate(values):
        arr[i] = int(v)
    return arr


class HipEngine:
    """One engine per (model, device).  Not re-entrant; one caller thread (like the reference)."""

    def _ck(self, status: int) -> None:
        _lib.check(status, self.lib)

    def _eos_array(self, eos_token_ids: Sequence[int]):
        """The eos list as the device sees it: ids outside the vocabulary can never be produced and are dropped, repeats are dropped
        (first occurrence kept: the list ORDER decides which id truncates the output, SSG:82-91).  The reference folds any number of
        `stop_token_ids` into this list (generator_base.py:106); the device list holds LSK_MAX_EOS = 1024 -- more than that is an error,
        never a silent truncation (the device-side draft cut and the host-side output cut must agree)."""
        eos = list(dict.fromkeys(int(t) for t in eos_token_ids if t is not None and 0 <= int(t) < self.vocab))     # None: a tokenizer without eos
        if len(eos) > _lib.LSK_MAX_EOS:
            raise _lib.LskError(f"{len(eos)} distinct eos / stop token ids; the engine supports at most {_lib.LSK_MAX_EOS}")
        return eos, _i32_array(eos)

    def __init__(self, model, max_ctx: int = 2048, max_prompt: int = 1024, page_size: int = 128,
                 target_wgs: int = 0, layer_range: Optional[Sequence[int]] = None, release_weights: bool = False):
        cfg = model.config
        # (the final norm is the one tensor every rank of a pipeline materialises: a middle rank holds neither the embedding nor the head)
        weight = model.model.norm.weight if model.model.embed_tokens.weight.is_meta else model.model.embed_tokens.weight
        dmap = getattr(model, "hf_device_map", None)
        if dmap and len(set(dmap.values())) > 1:
            # the reference's multi-GPU form (generate.py:59-64): ONE process, layers spread by accelerate hooks
            raise _lib.LskError("this model was spread over several devices by device_map='auto'; the engine wants one process per GPU: "
                                "launch the driver under torchrun (every rank loads its own layer range, layerskip_amd/checkpoint.py) "
                                "or load the model on one device")
        if weight.device.type != "cuda":
            raise _lib.LskError("HipEngine needs the model on a HIP device (model.to('cuda')); there is no CPU path")
        if weight.dtype == torch.bfloat16:
            self.dtype, dtype_name = torch.bfloat16, "bf16"
        elif weight.dtype == torch.float16:
            # the fp16 library: the same sources built with -DLSK_ELEM_F16 (the dtype generate.py:63 hard-codes)
            self.dtype, dtype_name = torch.float16, "fp16"
        else:
            raise _lib.LskError(f"HipEngine computes in bf16 (or fp16); model dtype is {weight.dtype}")
        self.lib = _lib.load(dtype=dtype_name)
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            raise _lib.LskError("biased projections are not supported")
        # The engine keeps its own references to every tensor it borrows (embedding, norm gains) and copies of the
        # scalars it needs later, but NOT the model: `get_engine` maps model -> engine weakly, and a strong
        # reference back to the key would keep both (and ~2x the weights in HBM) alive forever.
        import weakref
        self._model_ref = weakref.ref(model)
        self.rms_eps = float(cfg.rms_norm_eps)
        rot = model.model.rotary_emb
        self._inv_freq = rot.inv_freq.detach().to("cpu", torch.float32).clone()
        self._rope_scaling = float(rot.attention_scaling)
        self.device = weight.device
        self.num_layers = cfg.num_hidden_layers
        self.hidden = cfg.hidden_size
        self.intermediate = cfg.intermediate_size
        self.n_heads = cfg.num_attention_heads
        self.n_kv_heads = cfg.num_key_value_heads
        self.head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        self.vocab = cfg.vocab_size
        self.page_size = page_size
        self.target_wgs = target_wgs
        self.layer_range = tuple(layer_range) if layer_range is not None else (0, self.num_layers)
        # release_weights: after a projection is packed its HF original is dropped (the model object then
        # only serves this engine) -- what lets a 70B checkpoint (140 GB bf16) live on ONE 288 GB MI355X
        self.release_weights = release_weights
        self._handle = ctypes.c_void_p(None)
        self._packed = []           # per-layer packed buffers (kept alive); None outside layer_range
        self._globals = {}
        self._buffers = {}
        self._options = {}          # lsk_engine_set_option values, re-applied whenever the C engine is recreated
        self._block_table = None
        self._profile = False
        self._fingerprint = None
        with torch.cuda.device(self.device):
            self._pack_weights(model)
            self._allocate(max_ctx, max_prompt)
        self._fingerprint = self._weights_fingerprint(model)

    @property
    def model(self):
        """The borrowed model, or None once the caller dropped it (the engine keeps working: it owns packed
        copies of the projections and references to the tensors it shares)."""
        return self._model_ref()

    # ------------------------------------------------------------------ weights
    def _weight_tensors(self, m):
        out = []
        for idx, layer in enumerate(m.model.layers):
            if not (self.layer_range[0] <= idx < self.layer_range[1]):
                continue
            a, mlp = layer.self_attn, layer.mlp
            out += [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight,
                    mlp.down_proj.weight, layer.input_layernorm.weight, layer.post_attention_layernorm.weight]
        out += [t for t in (m.lm_head.weight, m.model.embed_tokens.weight, m.model.norm.weight) if not t.is_meta]
        return out

    def _weights_fingerprint(self, m):
        """What the engine copied or borrowed, in two parts: (storage address, in-place version) of every tensor --
        load_state_dict(), an in-place op on the Parameter or assigning a new Parameter shows there -- and a sampled CONTENT
        checksum per tensor (lsk_engine_weights_checksum: up to 4096 evenly strided elements each, one launch), because an
        edit through `.data` (`p.data += delta`: PEFT's default LoRA merge) moves neither the address nor the version.  An edit
        confined to elements the sample misses is not detected: call refresh_weights() after such surgery."""
        if self.release_weights:
            return None                      # the originals were dropped: the model object only serves this engine
        tensors = self._weight_tensors(m)
        meta = tuple((t.data_ptr(), t._version) for t in tensors)
        if not self._handle or any(t.device != self.device or t.element_size() != 2 or not t.is_contiguous() for t in tensors):
            return (meta, None)
        n = len(tensors)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
        cnts = (ctypes.c_int64 * n)(*[t.numel() for t in tensors])
        sums = (ctypes.c_uint64 * n)()
        self._ck(self.lib.lsk_engine_weights_checksum(self._handle, ptrs, cnts, n, sums, self._stream))
        return (meta, tuple(sums))

    def weights_changed(self, model=None) -> bool:
        model = model if model is not None else self.model
        if model is None or self._fingerprint is None:
            return False
        return self._weights_fingerprint(model) != self._fingerprint

    def refresh_weights(self, model=None) -> None:
        """Re-pack every projection from the model's CURRENT weights and re-bind the borrowed tensors (after
        load_state_dict / in-place edits / a LoRA merge on the same model object).  The cached context is dropped."""
        model = model if model is not None else self.model
        if model is None:
            raise _lib.LskError("refresh_weights: the model this engine was built from is gone")
        if self.release_weights:
            raise _lib.LskError("refresh_weights: this engine released the original weights (release_weights=True)")
        with torch.cuda.device(self.device):
            self._packed = []
            self._globals = {}
            self._pack_weights(model)
            self._bind_weights()
        self._fingerprint = self._weights_fingerprint(model)
        self.reset()

    @property
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _packed_buffer(self, n_rows: int, k: int) -> torch.Tensor:
        nbytes = ctypes.c_size_t(0)
        self._ck(self.lib.lsk_packed_bytes(n_rows, k, ctypes.byref(nbytes)))
        return torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)

    def _pack_into(self, dst: torch.Tensor, w: torch.Tensor, tile_offset: int, tile_stride: int, rope_hd: int) -> None:
        w = w.detach()
        if w.dtype != self.dtype or w.device != self.device:
            raise _lib.LskError(f"all projection weights must be {self.dtype} on the engine's device")
        if w.stride(1) != 1:
            w = w.contiguous()
        self._ck(self.lib.lsk_pack_linear(w.data_ptr(), w.shape[0], w.shape[1], w.stride(0), dst.data_ptr(),
                                       tile_offset, tile_stride, rope_hd, self._stream))

    def _pack_weights(self, m) -> None:
        hd, H, I = self.head_dim, self.hidden, self.intermediate
        qdim, kvdim = self.n_heads * hd, self.n_kv_heads * hd
        for idx, layer in enumerate(m.model.layers):
            if not (self.layer_range[0] <= idx < self.layer_range[1]):
                self._packed.append(None)       # another pipeline rank owns this layer
                continue
            a, mlp = layer.self_attn, layer.mlp
            wqkv = self._packed_buffer(qdim + 2 * kvdim, H)
            self._pack_into(wqkv, a.q_proj.weight, 0, 1, hd)
            self._pack_into(wqkv, a.k_proj.weight, qdim // 16, 1, hd)
            self._pack_into(wqkv, a.v_proj.weight, (qdim + kvdim) // 16, 1, 0)
            wo = self._packed_buffer(H, qdim)
            self._pack_into(wo, a.o_proj.weight, 0, 1, 0)
            wgu = self._packed_buffer(2 * I, H)
            self._pack_into(wgu, mlp.gate_proj.weight, 0, 2, 0)
            self._pack_into(wgu, mlp.up_proj.weight, 1, 2, 0)
            wdown = self._packed_buffer(H, I)
            self._pack_into(wdown, mlp.down_proj.weight, 0, 1, 0)
            n1 = layer.input_layernorm.weight.detach().contiguous()
            n2 = layer.post_attention_layernorm.weight.detach().contiguous()
            self._packed.append((wqkv, wo, wgu, wdown, n1, n2))
            if self.release_weights:
                torch.cuda.synchronize(self.device)
                for lin in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj):
                    lin.weight = torch.nn.Parameter(torch.empty(0, dtype=self.dtype, device=self.device),
                                                    requires_grad=False)
        # a pipeline rank that runs no head / embeds no token (checkpoint.load_layer_range(embed=False, head=False)) leaves them on meta
        have_head, have_embed = not m.lm_head.weight.is_meta, not m.model.embed_tokens.weight.is_meta
        self._globals["lm_head"] = None
        if have_head:
            head = self._packed_buffer(self.vocab, H)
            self._pack_into(head, m.lm_head.weight, 0, 1, 0)
            self._globals["lm_head"] = head
            tied = have_embed and m.lm_head.weight.data_ptr() == m.model.embed_tokens.weight.data_ptr()
            if self.release_weights and not tied:
                torch.cuda.synchronize(self.device)
                m.lm_head.weight = torch.nn.Parameter(torch.empty(0, dtype=self.dtype, device=self.device),
                                                      requires_grad=False)
        self._globals["embed"] = m.model.embed_tokens.weight.detach().contiguous() if have_embed else None
        self._globals["final_norm"] = m.model.norm.weight.detach().contiguous() if have_head else None
        torch.cuda.synchronize(self.device)

    def _rope_tables(self, length: int):
        """cos/sin exactly as LlamaRotaryEmbedding.forward computes them on CPU (fp32 -> model dtype)."""
        inv_freq, scaling = self._inv_freq, self._rope_scaling
        pos = torch.arange(length, dtype=torch.float32)
        freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]   # [length, d/2]
        cos = (freqs.cos() * scaling).to(self.dtype)
        sin = (freqs.sin() * scaling).to(self.dtype)
        return cos.contiguous().to(self.device), sin.contiguous().to(self.device)

    # ------------------------------------------------------------------ buffers / C engine
    def _allocate(self, max_ctx: int, max_prompt: int) -> None:
        if self._handle:
            self._ck(self.lib.lsk_engine_destroy(self._handle))
            self._handle = ctypes.c_void_p(None)
        self.max_ctx = _round_up(max(max_ctx, self.page_size), self.page_size)
        self.max_prompt = max(16, int(max_prompt))
        self.cfg = LskConfig(self.num_layers, self.hidden, self.intermediate, self.n_heads, self.n_kv_heads,
                             self.head_dim, self.vocab, self.rms_eps, self.max_ctx, self.page_size,
                             self.max_prompt, self.target_wgs)
        ws, kv = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._ck(self.lib.lsk_workspace_bytes(ctypes.byref(self.cfg), ctypes.byref(ws)))
        self._ck(self.lib.lsk_kv_pool_bytes(ctypes.byref(self.cfg), ctypes.byref(kv)))
        self._buffers.pop("sampling", None)
        self._buffers["ws"] = torch.zeros(ws.value, dtype=torch.uint8, device=self.device)
        self._buffers["kv"] = torch.zeros(kv.value, dtype=torch.uint8, device=self.device)
        cos, sin = self._rope_tables(self.max_ctx)
        self._buffers["cos"], self._buffers["sin"] = cos, sin
        torch.cuda.synchronize(self.device)
        handle = ctypes.c_void_p(None)
        self._ck(self.lib.lsk_engine_create(ctypes.byref(self.cfg), self._buffers["ws"].data_ptr(), ws.value,
                                         self._buffers["kv"].data_ptr(), kv.value, ctypes.byref(handle)))
        self._handle = handle
        self._bind_weights()
        # a recreated engine keeps what the caller configured on the old one
        for option, value in self._options.items():
            self._ck(self.lib.lsk_engine_set_option(handle, option, value))
        if self._block_table is not None and len(self._block_table) == self.max_ctx // self.page_size:
            arr = _i32_array(self._block_table)
            self._ck(self.lib.lsk_engine_set_block_table(handle, arr, len(self._block_table), self._stream))
        else:
            self._block_table = None         # a grown pool has more pages: back to the identity mapping
        if self._profile:
            self._ck(self.lib.lsk_engine_set_profile(handle, 1))

    def _bind_weights(self) -> None:
        handle = self._handle
        for i, packed in enumerate(self._packed):
            if packed is None:
                continue
            wqkv, wo, wgu, wdown, n1, n2 = packed
            self._ck(self.lib.lsk_engine_set_layer(handle, i, wqkv.data_ptr(), wo.data_ptr(), wgu.data_ptr(),
                                                wdown.data_ptr(), n1.data_ptr(), n2.data_ptr()))
        g = self._globals
        ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731 -- NULL: not bound on this rank
        self._ck(self.lib.lsk_engine_set_globals(handle, ptr(g["embed"]), ptr(g["final_norm"]), ptr(g["lm_head"]),
                                              self._buffers["cos"].data_ptr(), self._buffers["sin"].data_ptr(), self.max_ctx))

    def ensure_capacity(self, total_tokens: int, prompt_len: int) -> None:
        """Grow the KV pool / prompt buffer if a request needs more (context is lost)."""
        if total_tokens > self.max_ctx or prompt_len > self.max_prompt:
            with torch.cuda.device(self.device):
                self._allocate(max(total_tokens, self.max_ctx), max(prompt_len, self.max_prompt))

    def close(self) -> None:
        if self._handle:
            self.lib.lsk_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def reset(self) -> None:
        self._ck(self.lib.lsk_engine_reset(self._handle, self._stream))

    @property
    def kv_len(self) -> int:
        v = ctypes.c_int32(0)
        self._ck(self.lib.lsk_engine_get_kv_len(self._handle, ctypes.byref(v)))
        return v.value

    def set_kv_len(self, kv_len: int) -> None:
        self._ck(self.lib.lsk_engine_set_kv_len(self._handle, int(kv_len), self._stream))

    def set_block_table(self, table: Sequence[int]) -> None:
        arr = _i32_array(table)
        self._ck(self.lib.lsk_engine_set_block_table(self._handle, arr, len(table), self._stream))
        self._block_table = [int(t) for t in table]

    # ------------------------------------------------------------------ fused fast paths
    def spec_step(self, input_ids: Sequence[int], num_speculations: int, exit_layer: int,
                  eos_token_ids: Sequence[int]) -> StepResult:
        ids = _i32_array(input_ids)
        eos, eos_arr = self._eos_array(eos_token_ids)
        res = LskStepResult()
        self._ck(self.lib.lsk_spec_step(self._handle, ids, len(input_ids), int(num_speculations), int(exit_layer),
                                     eos_arr, len(eos), ctypes.byref(res), self._stream))
        n, s = res.num_matches, int(num_speculations)
        return StepResult(n, res.num_drafts, res.next_token, res.kv_len, list(res.emitted[: n + 1]),
                          list(res.draft_tokens[:s]), list(res.verified_tokens[: s + 1]))

    def spec_generate(self, prompt_ids: Sequence[int], num_speculations: int, exit_layer: int,
                      eos_token_ids: Sequence[int], max_steps: int):
        """Whole greedy generation in one C-ABI call.  Returns (tokens, matches, drafts, [(T_d, n) per step])."""
        ids = _i32_array(prompt_ids)
        eos, eos_arr = self._eos_array(eos_token_ids)
        out = (ctypes.c_int32 * max_steps)()
        sd = (ctypes.c_int32 * max_steps)()
        sm = (ctypes.c_int32 * max_steps)()
        n_out, tm, td, ns = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        self._ck(self.lib.lsk_spec_generate(self._handle, ids, len(prompt_ids), int(num_speculations), int(exit_layer), eos_arr,
                                         len(eos), int(max_steps), out, ctypes.byref(n_out), ctypes.byref(tm), ctypes.byref(td),
                                         sd, sm, ctypes.byref(ns), self._stream))
        steps = [(sd[i], sm[i]) for i in range(ns.value)]
        return list(out[: n_out.value]), tm.value, td.value, steps

    # ------------------------------------------------------------------ sampling on the device (opt-in)
    def _sampling_scratch(self) -> torch.Tensor:
        buf = self._buffers.get("sampling")
        if buf is None:
            n = ctypes.c_size_t(0)
            self._ck(self.lib.lsk_sampling_scratch_bytes(ctypes.byref(self.cfg), ctypes.byref(n)))
            buf = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._buffers["sampling"] = buf
        return buf

    def sample_rows(self, logits: torch.Tensor, temperature: float, top_k: int, top_p: float, seed: int, offset: int,
                    tag0: int = 0):
        """decode_next_token(sample=True) over rows of device logits [m, ld] (fp32).  Returns (tokens int32[m] on
        the device, probabilities fp32 [m, ld])."""
        if logits.dtype != torch.float32 or logits.device != self.device or logits.dim() != 2 or logits.stride(1) != 1:
            raise _lib.LskError("logits must be a [m, ld] fp32 tensor on the engine device")
        m, ld = logits.shape[0], logits.stride(0)
        toks = torch.empty(m, dtype=torch.int32, device=self.device)
        probs = torch.empty(m, ld, dtype=torch.float32, device=self.device)
        self._ck(self.lib.lsk_sample_rows(self._handle, logits.data_ptr(), ld, m, float(temperature), int(top_k), float(top_p),
                                       int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(tag0), toks.data_ptr(),
                                       probs.data_ptr(), self._stream))
        return toks, probs

    def spec_step_sampled(self, input_ids: Sequence[int], num_speculations: int, exit_layer: int,
                          eos_token_ids: Sequence[int], temperature: float, top_k: int, top_p: float, seed: int,
                          offset: int) -> StepResult:
        """single_step_speculation with sample=True, entirely on the device (lsk_spec_step_sampled)."""
        ids = _i32_array(input_ids)
        eos, eos_arr = self._eos_array(eos_token_ids)
        res = LskStepResult()
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_spec_step_sampled(self._handle, ids, len(input_ids), int(num_speculations), int(exit_layer), eos_arr,
                                             len(eos), float(temperature), int(top_k), float(top_p), int(seed) & (2 ** 64 - 1),
                                             int(offset) & (2 ** 64 - 1), scratch.data_ptr(), scratch.numel(), ctypes.byref(res),
                                             self._stream))
        n, s = res.num_matches, int(num_speculations)
        return StepResult(n, res.num_drafts, res.next_token, res.kv_len, list(res.emitted[: n + 1]),
                          list(res.draft_tokens[:s]), list(res.verified_tokens[: s + 1]))

    def spec_generate_sampled(self, prompt_ids: Sequence[int], num_speculations: int, exit_layer: int,
                              eos_token_ids: Sequence[int], max_steps: int, temperature: float, top_k: int, top_p: float,
                              seed: int, offset: int):
        """Whole sample=True generation in one C-ABI call (lsk_spec_generate_sampled).  Same return as spec_generate."""
        ids = _i32_array(prompt_ids)
        eos, eos_arr = self._eos_array(eos_token_ids)
        out = (ctypes.c_int32 * max_steps)()
        sd = (ctypes.c_int32 * max_steps)()
        sm = (ctypes.c_int32 * max_steps)()
        n_out, tm, td, ns = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_spec_generate_sampled(
            self._handle, ids, len(prompt_ids), int(num_speculations), int(exit_layer), eos_arr, len(eos), int(max_steps),
            float(temperature), int(top_k), float(top_p), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
            scratch.data_ptr(), scratch.numel(), out, ctypes.byref(n_out), ctypes.byref(tm), ctypes.byref(td), sd, sm,
            ctypes.byref(ns), self._stream))
        steps = [(sd[i], sm[i]) for i in range(ns.value)]
        return list(out[: n_out.value]), tm.value, td.value, steps

    def ar_generate(self, input_ids: Sequence[int], layer_end: Optional[int], eos_token_ids: Sequence[int],
                    max_steps: int) -> List[int]:
        """Whole greedy autoregressive generation in one C-ABI call."""
        ids = _i32_array(input_ids)
        eos, eos_arr = self._eos_array(eos_token_ids)
        out = (ctypes.c_int32 * max_steps)()
        n_out = ctypes.c_int32(0)
        self._ck(self.lib.lsk_ar_generate(self._handle, ids, len(input_ids), int(layer_end or self.num_layers), eos_arr,
                                       len(eos), int(max_steps), out, ctypes.byref(n_out), self._stream))
        return list(out[: n_out.value])

    def ar_step(self, input_ids: Sequence[int], layer_end: Optional[int] = None) -> int:
        ids = _i32_array(input_ids)
        tok = ctypes.c_int32(0)
        self._ck(self.lib.lsk_ar_step(self._handle, ids, len(input_ids), int(layer_end or self.num_layers),
                                   ctypes.byref(tok), self._stream))
        return tok.value

    # ------------------------------------------------------------------ layer-range pipeline (rank-0 half of a step)
    def draft_block(self, input_ids: Optional[Sequence[int]], row0: int, n_rows: int, pos_off0: int, exit_layer: int,
                    head_last: bool = False) -> None:
        """Asynchronous device-resident draft loop over step rows [row0, row0 + n_rows) (lsk_draft_block)."""
        if input_ids is None:
            ids, n = None, 1
        else:
            ids, n = _i32_array(input_ids), len(input_ids)
        self._ck(self.lib.lsk_draft_block(self._handle, ids, n, int(row0), int(n_rows), int(pos_off0), int(exit_layer),
                                       1 if head_last else 0, self._stream))

    def set_eos(self, eos_token_ids: Sequence[int]) -> None:
        """The eos list of the acceptance kernel (pipeline ranks: lsk_engine_set_eos)."""
        eos, arr = self._eos_array(eos_token_ids)
        self._ck(self.lib.lsk_engine_set_eos(self._handle, arr, len(eos), self._stream))

    def pipeline_pack(self, go: int, prompt_len: int, src_row: int, m: int, kv: int) -> None:
        """Rank 0: step rows [src_row, src_row + m) + header -> the message buffer (lsk_pipeline_pack), asynchronous."""
        self._ck(self.lib.lsk_pipeline_pack(self._handle, int(go), int(prompt_len), int(src_row), int(m), int(kv), self._stream))

    def pipeline_apply(self, kv_bound: int) -> None:
        """Ranks > 0: the received header's rollback applied on the device; the host keeps `kv_bound` (lsk_pipeline_apply)."""
        self._ck(self.lib.lsk_pipeline_apply(self._handle, int(min(kv_bound, self.max_ctx)), self._stream))

    def pipeline_tail(self, m: int) -> torch.Tensor:
        """Last rank: head + argmax + acceptance kernel over message rows [1, 1 + m) -> device int32[64] result block."""
        res = self._buffers.get("pp_result")
        if res is None:
            res = torch.zeros(64, dtype=torch.int32, device=self.device)
            self._buffers["pp_result"] = res
        self._ck(self.lib.lsk_pipeline_tail(self._handle, int(m), res.data_ptr(), self._stream))
        return res

    # ---- sample=True on the pipeline (lsk_draft_block_sampled / lsk_pipeline_pack_sampled / _tail_sampled / _residual) ----
    def pipeline_result_words(self) -> int:
        """int32 words of a sampled step's result block: 64 result words + one fp32 probability row (q_n)."""
        n = ctypes.c_int32(0)
        self._ck(self.lib.lsk_pipeline_result_words(ctypes.byref(self.cfg), ctypes.byref(n)))
        return n.value

    def draft_block_sampled(self, input_ids: Optional[Sequence[int]], row0: int, n_rows: int, pos_off0: int, exit_layer: int,
                            head_last: bool, temperature: float, top_k: int, top_p: float, seed: int, offset: int) -> None:
        """`draft_block` with every argmax replaced by a draw (draft j of the block: Philox tag j at `offset`); the warped draft
        distributions stay in the sampling scratch (rows row0 + j) for `pipeline_pack_sampled` / `pipeline_residual`."""
        if input_ids is None:
            ids, n = None, 1
        else:
            ids, n = _i32_array(input_ids), len(input_ids)
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_draft_block_sampled(self._handle, ids, n, int(row0), int(n_rows), int(pos_off0), int(exit_layer),
                                               1 if head_last else 0, float(temperature), int(top_k), float(top_p),
                                               int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), scratch.data_ptr(), scratch.numel(),
                                               self._stream))

    def pipeline_pack_sampled(self, go: int, prompt_len: int, src_row: int, m: int, kv: int, offset: int) -> None:
        """`pipeline_pack` + the sampled step's header words: mode, Philox offset, p_i(x_i) of every draft."""
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_pipeline_pack_sampled(self._handle, int(go), int(prompt_len), int(src_row), int(m), int(kv),
                                                 int(offset) & (2 ** 64 - 1), scratch.data_ptr(), scratch.numel(), self._stream))

    def pipeline_tail_sampled(self, m: int, temperature: float, top_k: int, top_p: float, seed: int, offset: int) -> torch.Tensor:
        """Last rank, sample=True: head + verify draws + the acceptance test against the header's p_i(x_i) -> device int32
        [pipeline_result_words()] = result words + the verify row q_n at the first rejection."""
        words = self.pipeline_result_words()
        res = self._buffers.get("pp_result_sampled")
        if res is None or res.numel() != words:
            res = torch.zeros(words, dtype=torch.int32, device=self.device)
            self._buffers["pp_result_sampled"] = res
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_pipeline_tail_sampled(self._handle, int(m), float(temperature), int(top_k), float(top_p),
                                                 int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), scratch.data_ptr(), scratch.numel(),
                                                 res.data_ptr(), words, self._stream))
        return res

    def pipeline_residual(self, block: torch.Tensor, src_row: int, seed: int, offset: int) -> None:
        """Rank 0: finish a sampled result block in place -- if its residual draw is pending, the token from max(q_n - p_n, 0) with
        this rank's p_n (the draft rows start at step row `src_row`)."""
        if block.dtype != torch.int32 or block.device != self.device or not block.is_contiguous():
            raise _lib.LskError("the result block must be a contiguous int32 tensor on the engine device")
        scratch = self._sampling_scratch()
        self._ck(self.lib.lsk_pipeline_residual(self._handle, block.data_ptr(), block.numel(), int(src_row), int(seed) & (2 ** 64 - 1),
                                             int(offset) & (2 ** 64 - 1), scratch.data_ptr(), scratch.numel(), self._stream))

    def logits_rows(self, blocks: Sequence[Sequence[int]]) -> torch.Tensor:
        """Final norm + lm_head logits of the listed (buffer, row_base, count) blocks -> [rows, vocab] in the model dtype (the values
        the reference's `model.lm_head` returns, LMU:273, :387): what logits processors are shown."""
        total = sum(c for _, _, c in blocks)
        # (16 rows at a time through ONE fp32 staging block straight into the model dtype: a 2 048-row prompt at V = 128 256 is 0.5 GB in
        # bf16; the fp32 image of all rows plus a converted copy was three times that)
        out = torch.empty(total, self.vocab, dtype=self.dtype, device=self.device)
        stage = torch.empty(_lib.LSK_MAX_ROWS, self.vocab, dtype=torch.float32, device=self.device)
        at = 0
        for buf, base, count in blocks:
            for r0 in range(0, count, _lib.LSK_MAX_ROWS):
                m = min(_lib.LSK_MAX_ROWS, count - r0)
                self.run_head(buf, base + r0, m, logits=stage[:m], want_tokens=False)
                out[at:at + m].copy_(stage[:m])
                at += m
        return out

    def header(self) -> List[int]:
        """The int32 words of the message header (synchronises: a device -> host read)."""
        return [int(v) for v in self.rows_view(BUF_MSG, 0, 1).view(torch.int32)[0, :24].tolist()]

    def row_tokens(self, row0: int, n: int) -> List[int]:
        out = (ctypes.c_int32 * n)()
        self._ck(self.lib.lsk_get_row_tokens(self._handle, int(row0), int(n), out, self._stream))
        return list(out)

    def shift_rows(self, src: int, dst: int, n: int) -> None:
        self._ck(self.lib.lsk_shift_rows(self._handle, int(src), int(dst), int(n), self._stream))

    def rows_view(self, buffer: int, row_base: int, m: int) -> torch.Tensor:
        """Rows of a hidden-state buffer as a ZERO-COPY tensor over the engine's workspace ([m, hidden], model dtype):
        point-to-point send / recv go straight from / into the engine's buffers on the current stream."""
        off = ctypes.c_size_t(0)
        self._ck(self.lib.lsk_rows_offset(self._handle, buffer, row_base, ctypes.byref(off)))
        nbytes = m * self.hidden * 2
        return self._buffers["ws"][off.value: off.value + nbytes].view(self.dtype).view(m, self.hidden)

    # ------------------------------------------------------------------ building blocks
    def embed_rows(self, ids: Sequence[int], buffer: int, row_base: int) -> None:
        arr = _i32_array(ids)
        self._ck(self.lib.lsk_embed_rows(self._handle, arr, len(ids), buffer, row_base, self._stream))

    def run_layers(self, buffer: int, row_base: int, m: int, pos_offset: int, layer_begin: int, layer_end: int) -> None:
        self._ck(self.lib.lsk_run_layers(self._handle, buffer, row_base, m, pos_offset, layer_begin, layer_end, self._stream))

    def run_layers_chunked(self, buffer: int, row_base: int, n: int, pos_offset: int, layer_begin: int, layer_end: int) -> None:
        for r0 in range(0, n, _lib.LSK_MAX_ROWS):
            m = min(_lib.LSK_MAX_ROWS, n - r0)
            self.run_layers(buffer, row_base + r0, m, pos_offset + r0, layer_begin, layer_end)

    def run_bulk(self, n: int, layer_begin: int, layer_end: int) -> None:
        self._ck(self.lib.lsk_run_bulk(self._handle, n, layer_begin, layer_end, self._stream))

    def set_option(self, option: int, value: int) -> None:
        self._ck(self.lib.lsk_engine_set_option(self._handle, option, value))
        self._options[int(option)] = int(value)

    def run_head(self, buffer: int, row_base: int, m: int, logits: Optional[torch.Tensor] = None,
                 want_tokens: bool = True) -> Optional[List[int]]:
        """logits: optional fp32 CUDA tensor [m, >=vocab] that receives the (bf16-rounded) logits."""
        ptr, ld = None, 0
        if logits is not None:
            if logits.dtype != torch.float32 or logits.device != self.device or logits.stride(-1) != 1:
                raise _lib.LskError("logits buffer must be a contiguous fp32 tensor on the engine device")
            ptr, ld = logits.data_ptr(), logits.stride(0)
        toks = (ctypes.c_int32 * m)() if want_tokens else None
        self._ck(self.lib.lsk_run_head(self._handle, buffer, row_base, m, ptr, ld, toks, self._stream))
        return list(toks) if want_tokens else None

    def read_rows(self, buffer: int, row_base: int, m: int) -> torch.Tensor:
        out = torch.empty(m, self.hidden, dtype=self.dtype, device=self.device)
        self._ck(self.lib.lsk_read_rows(self._handle, buffer, row_base, m, out.data_ptr(), self._stream))
        return out

    def write_rows(self, buffer: int, row_base: int, rows: torch.Tensor) -> None:
        rows = rows.to(self.device, self.dtype).contiguous()
        self._ck(self.lib.lsk_write_rows(self._handle, buffer, row_base, rows.shape[0], rows.data_ptr(), self._stream))

    # ------------------------------------------------------------------ measurement hooks
    def time_gateup(self, layer: int, m: int, iters: int) -> float:
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.lsk_time_gateup(self._handle, layer, m, iters, ctypes.byref(ms), self._stream))
        return ms.value

    def set_profile(self, enable: bool) -> None:
        self._ck(self.lib.lsk_engine_set_profile(self._handle, 1 if enable else 0))
        self._profile = bool(enable)

    def get_profile(self):
        """(sum of the gate/up dispatch durations in ms, number of launches)."""
        ms, n = ctypes.c_float(0.0), ctypes.c_int32(0)
        self._ck(self.lib.lsk_engine_get_profile(self._handle, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def host_stats(self) -> dict:
        """Host cost of the fused generate calls since the last query (lsk_engine_get_host_stats)."""
        a, b, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        self._ck(self.lib.lsk_engine_get_host_stats(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return {"enqueue_s": a.value, "wall_s": b.value, "steps": n.value}

    PROFILE_CLASSES = ("qkv", "attention", "o_proj", "gate_up", "down", "lm_head")

    def get_profile_table(self):
        """[{kernel, rows ("1" | ">1"), launches, ms, bytes}] for every decode-path kernel class since set_profile(True)."""
        n = 2 * len(self.PROFILE_CLASSES)
        ms, cnt, by = (ctypes.c_float * n)(), (ctypes.c_int32 * n)(), (ctypes.c_double * n)()
        self._ck(self.lib.lsk_engine_get_profile_table(self._handle, n, ms, cnt, by))
        out = []
        for c, name in enumerate(self.PROFILE_CLASSES):
            for multi in (0, 1):
                i = 2 * c + multi
                if cnt[i]:
                    out.append({"kernel": name, "rows": ">1" if multi else "1", "launches": cnt[i], "ms": ms[i], "bytes": by[i]})
        return out

    # bytes one launch of each projection streams from HBM (algorithmic: the packed weights once)
    def projection_bytes(self) -> dict:
        hd = self.head_dim
        qdim, kvdim = self.n_heads * hd, self.n_kv_heads * hd
        return {
            "qkv": 2 * (qdim + 2 * kvdim) * self.hidden,
            "o": 2 * self.hidden * qdim,
            "gate_up": 2 * 2 * self.intermediate * self.hidden,
            "down": 2 * self.hidden * self.intermediate,
            "lm_head": 2 * self.vocab * self.hidden,
        }


_ENGINES = None


def get_engine(model, check_weights: bool = True, **kwargs) -> HipEngine:
    """The engine bound to ``model`` (built lazily on first use, reused across calls).  Kept in a weak
    map, not on the module, so ``copy.deepcopy(model)`` and ``state_dict()`` never see it.

    The engine streams PACKED COPIES of the projections, so it must notice when the model's weights change under it
    (load_state_dict, an in-place edit, a LoRA merge on the same object -- the reference always reads live weights):
    with ``check_weights`` the fingerprint taken at pack time -- (address, version) per tensor plus a sampled content
    checksum, which is what catches edits through ``.data`` -- is compared and the weights are re-packed on a mismatch.  Keyword arguments that differ from the cached engine's are applied where that is possible
    (capacity grows, target_wgs is an option) and reported otherwise."""
    global _ENGINES
    if _ENGINES is None:
        import weakref
        _ENGINES = weakref.WeakKeyDictionary()
    eng = _ENGINES.get(model)
    if eng is None:
        eng = HipEngine(model, **kwargs)
        _ENGINES[model] = eng
        return eng
    if check_weights and eng.weights_changed(model):
        eng.refresh_weights(model)
    if kwargs:
        import warnings
        if kwargs.get("max_ctx", 0) > eng.max_ctx or kwargs.get("max_prompt", 0) > eng.max_prompt:
            eng.ensure_capacity(kwargs.get("max_ctx", eng.max_ctx), kwargs.get("max_prompt", eng.max_prompt))
        tw = kwargs.get("target_wgs")
        if tw is not None and tw != eng.target_wgs:
            eng.target_wgs = tw
            eng.set_option(_lib.LSK_OPT_TARGET_WGS, tw)
        for key in ("release_weights", "layer_range", "page_size"):
            if key in kwargs and kwargs[key] is not None:
                have = getattr(eng, key)
                want = tuple(kwargs[key]) if key == "layer_range" else kwargs[key]
                if have != want:
                    warnings.warn(f"get_engine: {key}={kwargs[key]!r} ignored, the engine cached for this model was built "
                                  f"with {key}={have!r}", RuntimeWarning, stacklevel=2)
    return eng
