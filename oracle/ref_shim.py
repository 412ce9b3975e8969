"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Loads the UNMODIFIED reference modules from /root/reference (read-only mount,
present only in the build container) behind a small compatibility shim so that
they run against the transformers release installed here (5.15) instead of the
4.45-era API they were written for.  Nothing from the reference is copied: the
modules are imported from where they lie.

The shim does three things:
  1. registers a stub ``colorama`` module (imported by
     self_speculation/self_speculation_generator.py:10, never needed without a
     SpeculativeTextStreamer);
  2. re-adds ``DynamicCache.from_legacy_cache / to_legacy_cache / __getitem__``
     which llama_model_utils.py:169,203,229,263,308,385 call and transformers
     5.x removed;
  3. wraps every ``LlamaDecoderLayer.forward`` so it accepts the old keyword
     names (``past_key_value=``, ``padding_mask=``), computes
     ``position_embeddings`` from ``position_ids`` itself and returns the
     ``(hidden, cache)`` 2-tuple llama_model_utils.py:193,253,354,375 unpack.

Only ``oracle/make_golden.py`` and ``bench.py``'s cpu_baseline leg (when
/root/reference exists) use this file.  It cannot travel to the GPU box.
"""
from __future__ import annotations

import os
import sys
import types

import torch
import transformers
from transformers.cache_utils import DynamicCache

REFERENCE_ROOT = os.environ.get("LAYERSKIP_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "self_speculation"))


def _install_colorama_stub() -> None:
    if "colorama" in sys.modules:
        return
    try:
        import colorama  # noqa: F401
        return
    except ImportError:
        pass

    class _Blank:
        def __getattr__(self, _name):
            return ""

    stub = types.ModuleType("colorama")
    stub.Fore = _Blank()
    stub.Style = _Blank()
    stub.Back = _Blank()
    sys.modules["colorama"] = stub


def _install_legacy_cache_api() -> None:
    if hasattr(DynamicCache, "from_legacy_cache"):
        return

    def from_legacy_cache(cls, past_key_values=None):
        cache = cls()
        if past_key_values is not None:
            for idx, kv in enumerate(past_key_values):
                cache.update(kv[0], kv[1], idx)
        return cache

    def to_legacy_cache(self):
        out = []
        for layer in self.layers:
            if not getattr(layer, "is_initialized", False):
                break
            out.append((layer.keys, layer.values))
        return tuple(out)

    def getitem(self, idx):
        layer = self.layers[idx]
        return (layer.keys, layer.values)

    DynamicCache.from_legacy_cache = classmethod(from_legacy_cache)
    DynamicCache.to_legacy_cache = to_legacy_cache
    DynamicCache.__getitem__ = getitem
    if not hasattr(DynamicCache, "__len__"):
        DynamicCache.__len__ = lambda self: len(self.layers)


def patch_model(model: "transformers.LlamaForCausalLM") -> "transformers.LlamaForCausalLM":
    """Wrap each decoder layer of ``model`` for the old call signature (idempotent)."""
    rotary = model.model.rotary_emb
    for layer in model.model.layers:
        if getattr(layer, "_layerskip_shimmed", False):
            continue
        inner = layer.forward

        def fwd(hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=True, padding_mask=None, _inner=inner, **kw):
            pos_emb = rotary(hidden_states, position_ids)
            out = _inner(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                         past_key_values=past_key_value, use_cache=use_cache,
                         position_embeddings=pos_emb)
            if isinstance(out, tuple):
                out = out[0]
            return out, past_key_value

        layer.forward = fwd
        layer._layerskip_shimmed = True
    return model


_LOADED = None


def load_reference():
    """Import the reference's self_speculation package; returns a namespace of its modules."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_colorama_stub()
    _install_legacy_cache_api()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    ns.generator_base = importlib.import_module("self_speculation.generator_base")
    ns.llama_model_utils = importlib.import_module("self_speculation.llama_model_utils")
    ns.autoregressive_generator = importlib.import_module("self_speculation.autoregressive_generator")
    ns.self_speculation_generator = importlib.import_module("self_speculation.self_speculation_generator")
    _LOADED = ns
    return ns


def cast_parameters(model, dtype):
    """What the reference's loader leaves behind (generate.py:59-64: `from_pretrained(..., torch_dtype=dtype)`): every PARAMETER in
    `dtype`, the buffers -- the rotary `inv_freq` -- as the modules' `__init__` made them (fp32; checked against
    `AutoModelForCausalLM.from_pretrained(dir, dtype=...)` of transformers 5.15).  `model.to(dtype)` is NOT that: `nn.Module.to` casts
    floating-point buffers too, and an `inv_freq` rounded to bf16 is off by 2e-3 relative -- 8 rad of rotary angle at position 4 000.
    (Rounds 2-5 generated the fixtures of every checkpoint below 2e9 parameters that way; round 6 regenerated them.)  In place."""
    for prm in model.parameters():
        prm.data = prm.data.to(dtype)
    return model

