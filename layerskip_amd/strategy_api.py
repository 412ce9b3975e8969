"""The plugin surface of the reference, restated for stand-alone use.

The HIP strategies in ``hip_strategies.py`` are duck-type compatible with the reference's
``GenerationStrategy`` (reference self_speculation/generator_base.py:51-62): they can be handed to
the reference's own ``HuggingfaceLlamaGenerator`` unchanged, or used with the equivalents below when
the reference tree is not installed.  Field names, defaults and result shapes follow
generator_base.py:17-49 so that configs and result consumers are interchangeable.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, List, Optional


@dataclass
class GenerationStrategyResult:            # generator_base.py:17-20
    predicted_tokens: List[int]
    acceptance_rate: Optional[float] = None


@dataclass
class GenerationResult:                    # generator_base.py:23-30
    generation_strategy_result: GenerationStrategyResult
    decoded_prediction: str
    num_tokens_generated: int
    total_time: float
    time_per_token: Optional[float]
    tokens_per_second: float


@dataclass
class GenerationConfig:                    # generator_base.py:33-49 (same names, same defaults)
    max_steps: int = 512
    exit_layer: int = -1
    num_speculations: int = -1
    generation_strategy: str = "autoregressive"
    sample: bool = True
    temperature: float = 0.6
    top_k: int = 0
    top_p: float = 0.9
    no_repeat_ngram_size: Optional[int] = None
    stop_words: Optional[List[str]] = None
    stop_token_ids: List[int] = field(default_factory=list)

    def __post_init__(self):
        if self.stop_token_ids is None:
            self.stop_token_ids = []


class GenerationStrategy:                  # generator_base.py:51-62
    def generate_token_ids(self, model, input_ids: List[int], eos_token_ids: List[int],
                           generation_config: GenerationConfig, logits_processors=None,
                           stopping_criteria=None, streamer=None) -> GenerationStrategyResult:
        raise NotImplementedError()


class TokenGenerator:
    """Facade with the behaviour of ``HuggingfaceLlamaGenerator`` (generator_base.py:65-130).

    ``tokenizer`` needs ``__call__(prompt, return_tensors="pt", add_special_tokens=True)``,
    ``decode(ids)`` and ``eos_token_id``; no tokenizer exists offline, so tests and bench.py drive
    ``generate_from_ids`` directly (same timing bracket, token ids instead of text).
    """

    def __init__(self, tokenizer: Any, model, generation_strategy: GenerationStrategy) -> None:
        self.tokenizer = tokenizer
        self.model = model
        self.generation_strategy = generation_strategy

    def _processors(self, cfg: GenerationConfig):
        import transformers
        procs = transformers.generation.logits_process.LogitsProcessorList()
        if cfg.no_repeat_ngram_size:
            procs.append(transformers.generation.logits_process.NoRepeatNGramLogitsProcessor(cfg.no_repeat_ngram_size))
        return procs

    def _criteria(self, cfg: GenerationConfig):
        import transformers
        crit = transformers.StoppingCriteriaList()
        if cfg.stop_words:
            crit.append(transformers.StopStringCriteria(self.tokenizer, cfg.stop_words))
        return crit

    def generate_from_ids(self, input_ids: List[int], eos_token_ids: List[int], generation_config: GenerationConfig,
                          streamer=None, decode: bool = False) -> GenerationResult:
        import torch
        procs = self._processors(generation_config)
        crit = self._criteria(generation_config)
        with torch.inference_mode():
            start = time.time()                                   # generator_base.py:107-118
            res = self.generation_strategy.generate_token_ids(
                model=self.model, input_ids=list(input_ids), eos_token_ids=list(eos_token_ids),
                generation_config=generation_config, logits_processors=procs, stopping_criteria=crit,
                streamer=streamer)
            total = time.time() - start
        n = len(res.predicted_tokens)
        text = self.tokenizer.decode(res.predicted_tokens) if (decode and self.tokenizer is not None) else ""
        return GenerationResult(res, text, n, total, total / n if n > 0 else None, n / total)

    def generate(self, prompt: str, generation_config: GenerationConfig, streamer=None) -> GenerationResult:
        enc = self.tokenizer(prompt, return_tensors="pt", add_special_tokens=True)
        eos = list(generation_config.stop_token_ids) + [self.tokenizer.eos_token_id]
        return self.generate_from_ids(enc["input_ids"].tolist()[0], eos, generation_config, streamer, decode=True)
