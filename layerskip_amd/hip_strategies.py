"""GenerationStrategy plugins backed by the HIP engine.

Drop-in replacements for the reference's strategies:

* ``HipSelfSpeculativeGenerationStrategy``  <->  SelfSpeculativeGenerationStrategy
  (reference self_speculation/self_speculation_generator.py:31-229)
* ``HipAutoRegressiveGenerationStrategy``   <->  AutoRegressiveGenerationStrategy
  (reference self_speculation/autoregressive_generator.py:25-80)

Same constructor (none), same ``generate_token_ids`` / ``single_step_speculation`` signatures, same
result objects, same corner-case behaviour (EOS dropped and truncating, ``max_steps`` clamp of the
speculation count, ``ZeroDivisionError`` when no draft was ever made, falsy-when-empty processors).

Execution paths:
  fused greedy, no logits processors / stopping criteria / streamer: the whole generation is ONE C-ABI call
        (`lsk_spec_generate` / `lsk_ar_generate`), speculation steps pipelined on the stream;
  step  greedy with a streamer or stopping criteria: ONE C-ABI call per speculation step (`lsk_spec_step`):
        draft loop, verify, ballot acceptance and KV rollback all stay on the device, one sync per step;
  sampled  `sample=True` without logits processors: the same two paths with every argmax replaced by a draw and the
        prefix match by modified rejection sampling ON THE DEVICE (`lsk_spec_generate_sampled` / `lsk_spec_step_sampled`):
        no logits row leaves HBM.  Parity with the reference is in distribution; the Philox stream is seeded from
        torch's generator (`torch.manual_seed(s)` reproduces a generation).
  slow  logits processors or more than 15 speculations: the same kernels driven row-block by row-block with the
        logits materialised as a tensor so the user's callables see what the reference shows them (sampling, if
        any, then happens on those logits with torch).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .engine import BUF_BULK, BUF_STEP, HipEngine, get_engine
from .strategy_api import GenerationConfig, GenerationStrategy, GenerationStrategyResult


class EngineCache:
    """Opaque stand-in for the reference's ``past_key_values`` tuple: the KV lives in the engine."""

    def __init__(self, engine: HipEngine):
        self.engine = engine

    @property
    def length(self) -> int:
        return self.engine.kv_len

    def __repr__(self) -> str:
        return f"EngineCache(kv_len={self.length})"


# --------------------------------------------------------------------------------------------------
# decoding helpers (greedy / sampled), semantics of llama_model_utils.py:75-131
# --------------------------------------------------------------------------------------------------
def _filter_top_k_top_p(logits: torch.Tensor, top_k: int, top_p: float) -> torch.Tensor:
    import transformers
    warp = transformers.generation.logits_process
    if top_k > 0:
        logits = warp.TopKLogitsWarper(top_k=top_k, filter_value=-float("inf"), min_tokens_to_keep=1)(None, logits)
    if 0 <= top_p <= 1.0:
        logits = warp.TopPLogitsWarper(top_p=top_p, filter_value=-float("inf"), min_tokens_to_keep=1)(None, logits)
    return logits


def decode_rows(logits: torch.Tensor, last_only: bool, sample: bool, temperature: float, top_k: int, top_p: float):
    """logits [1, M, V] -> (tokens, probabilities).  ``last_only`` = the reference's truthy token_idx."""
    if last_only:
        logits = logits[:, -1, :]
    if not sample:
        return logits.argmax(dim=-1), None
    rows = logits if last_only else logits[0]
    probs = torch.nn.functional.softmax(_filter_top_k_top_p(rows / temperature, top_k, top_p), dim=-1)
    tok = torch.multinomial(probs, num_samples=1)
    if not last_only:
        tok = tok.transpose(1, 0)
    return tok, probs


def _residual_distribution(p_verify: torch.Tensor, p_draft: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    d = torch.clamp(p_verify - p_draft, min=0)
    return d / (d.sum() + eps)


class HipSelfSpeculativeGenerationStrategy(GenerationStrategy):
    def __init__(self, engine_kwargs: Optional[dict] = None, fused_generate: bool = True, device_sampling: bool = True) -> None:
        self.engine_kwargs = engine_kwargs or {}
        # device_sampling (default): `sample=True` steps without logits processors run on the device (top-k / top-p
        # thresholds, Gumbel-max draws, rejection sampling) instead of materialising the logits for torch.
        # Seeding contract: key = torch.initial_seed(); the counter base of a generation is ONE 62-bit draw from
        # torch's global generator at its start, step i adds i -- so torch.manual_seed(s) before generate_token_ids
        # reproduces it, and two strategy instances (or two generations) never share a stream.
        self.device_sampling = device_sampling
        self._sample_base = None
        self._sample_step = 0
        self._stream_fresh = False        # generate_token_ids has just started a stream: its first step must not start another
        # fused_generate: greedy generations without processors / criteria / streamer run as ONE C-ABI call
        # (lsk_spec_generate: the loop of SSG:51-95 with the steps pipelined on the stream)
        self.fused_generate = fused_generate
        self.last_steps = []              # [(num_drafts, num_matches)] of the last fused generation

    # ------------------------------------------------------------------------------ where the engine and the late layers live
    # (one GPU: here.  The layer pipeline's slow path overrides the three: rank 0's engine owns the early layers and a head, the late
    #  layers and the verify head are other ranks', layerskip_amd/pipeline_strategy.py)
    def _get_engine(self, model, check_weights: bool = True) -> HipEngine:
        return get_engine(model, check_weights=check_weights, **self.engine_kwargs)

    def _verify_logits(self, engine: HipEngine, P: int, td: int, E: int, sbuf: int, sbase: int, prompt_rows: bool) -> torch.Tensor:
        """forward_remainder's late layers + final norm + lm_head (LMU:364-387) over the td + 1 step rows (and the P - 1 prompt rows
        in front): logits [1, (P - 1 if prompt_rows) + td + 1, V]."""
        L = engine.num_layers
        if P > 1:
            engine.run_bulk(P - 1, E, L)
        engine.run_layers_chunked(sbuf, sbase, td + 1, P - 1, E, L)
        blocks = ([(BUF_BULK, 0, P - 1)] if (P > 1 and prompt_rows) else []) + [(sbuf, sbase, td + 1)]
        return self._logits_rows(engine, blocks)

    def _commit(self, engine: HipEngine, kv_len: int) -> None:
        """crop_past_key_values (SSG:219-221): the verified context length after a step."""
        engine.set_kv_len(kv_len)

    # ------------------------------------------------------------------------------ outer loop
    def generate_token_ids(self, model, input_ids: List[int], eos_token_ids: List[int],
                           generation_config: GenerationConfig, logits_processors=None, stopping_criteria=None,
                           streamer=None) -> GenerationStrategyResult:
        engine = self._get_engine(model)
        spec = max(0, int(generation_config.num_speculations))
        extra_rows = spec if spec > _lib.LSK_MAX_SPEC else 0      # long draft blocks live behind the prompt rows
        engine.ensure_capacity(len(input_ids) + generation_config.max_steps + spec + 2, len(input_ids) + extra_rows)
        engine.reset()                                            # past_key_values = None
        eos_token_ids = [t for t in eos_token_ids if t is not None]       # a tokenizer without an eos token (reference: `None in list` is just False)
        device_sampled = bool(generation_config.sample and self.device_sampling and not logits_processors
                              and spec <= _lib.LSK_MAX_SPEC and hasattr(engine, "spec_step_sampled"))
        if device_sampled:
            self._new_sample_stream()
        if (self.fused_generate and not logits_processors and not stopping_criteria and streamer is None
                and spec <= _lib.LSK_MAX_SPEC and "single_step_speculation" not in self.__dict__
                and ((not generation_config.sample and hasattr(engine, "spec_generate"))
                     or (device_sampled and hasattr(engine, "spec_generate_sampled")))):
            if not (1 <= generation_config.exit_layer < engine.num_layers):
                raise ValueError(f"exit_layer={generation_config.exit_layer} must be in [1, {engine.num_layers})")
            if generation_config.sample:
                tokens, matches, drafts, self.last_steps = engine.spec_generate_sampled(
                    list(input_ids), spec, generation_config.exit_layer, eos_token_ids, generation_config.max_steps,
                    generation_config.temperature, generation_config.top_k, generation_config.top_p,
                    torch.initial_seed(), self._sample_base)
                self._sample_step += len(self.last_steps) + 1
                self._stream_fresh = False        # the fused call consumed this generation's stream: a later stand-alone step starts its own
            else:
                tokens, matches, drafts, self.last_steps = engine.spec_generate(
                    list(input_ids), spec, generation_config.exit_layer, eos_token_ids, generation_config.max_steps)
            return GenerationStrategyResult(predicted_tokens=tokens, acceptance_rate=matches / drafts)
        try:
            return self._step_loop(model, input_ids, eos_token_ids, generation_config, logits_processors, stopping_criteria, streamer)
        finally:
            self._stream_fresh = False            # (a generation that ran no sampled step -- max_steps = 0, an error -- must not leave the mark behind)

    def _step_loop(self, model, input_ids, eos_token_ids, generation_config, logits_processors, stopping_criteria, streamer):
        """SSG:51-99, one `single_step_speculation` call per step (streamers, stopping criteria, logits processors)."""
        past = None
        input_ids_list = list(input_ids)
        cur = torch.tensor([input_ids_list])
        output_ids: List[int] = []
        calls = 0
        total_draft_matches = 0
        total_generations = 0
        while len(output_ids) < generation_config.max_steps:
            cur, output_ids, past, n_match, n_draft = self.single_step_speculation(
                model=model, input_ids=cur, input_ids_list=input_ids_list, output_ids=output_ids,
                num_speculations=min(generation_config.num_speculations,
                                     generation_config.max_steps - len(output_ids) - 1),
                past_key_values=past, eos_token_ids=eos_token_ids, calls=calls,
                exit_layer=generation_config.exit_layer, sample=generation_config.sample,
                temperature=generation_config.temperature, top_k=generation_config.top_k,
                top_p=generation_config.top_p, logits_processors=logits_processors,
                stopping_criteria=stopping_criteria, streamer=streamer)
            calls += 1
            total_draft_matches += n_match
            total_generations += n_draft
            hit = [output_ids.index(e) for e in eos_token_ids if e in output_ids]
            if hit:
                # first eos id (in list order) found in the output truncates it; the eos is dropped
                output_ids = output_ids[: hit[0]]
                break
            if stopping_criteria:
                if torch.all(stopping_criteria(cur, scores=None)):
                    break
        return GenerationStrategyResult(predicted_tokens=output_ids,
                                        acceptance_rate=total_draft_matches / total_generations)

    def _new_sample_stream(self) -> None:
        """Counter base of the Philox stream of one generation: one draw from torch's global generator."""
        self._sample_base = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._sample_step = 0
        self._stream_fresh = True

    # ------------------------------------------------------------------------------ one step
    def single_step_speculation(self, model, input_ids: torch.Tensor, input_ids_list: List[int],
                                output_ids: List[int], num_speculations: int, past_key_values,
                                eos_token_ids: List[int], calls: int, exit_layer: int,
                                sample: Optional[bool] = False, temperature: Optional[float] = 0.7,
                                top_k: Optional[int] = 50, top_p: Optional[float] = 0.95,
                                logits_processors=None, stopping_criteria=None, streamer=None):
        # the packed weights are checked against the live model at the START of a generation, not on every step
        engine = self._get_engine(model, check_weights=past_key_values is None)
        if past_key_values is None:
            sp = max(0, num_speculations)
            engine.ensure_capacity(len(input_ids_list) + sp + 2, input_ids.shape[1] + (sp if sp > _lib.LSK_MAX_SPEC else 0))
            engine.reset()
        spec = max(0, int(num_speculations))
        if not (1 <= exit_layer < engine.num_layers):
            raise ValueError(f"exit_layer={exit_layer} must be in [1, {engine.num_layers})")
        new_ids = [int(t) for t in input_ids[0].tolist()]
        eos_token_ids = [t for t in eos_token_ids if t is not None]
        if (sample and self.device_sampling and not logits_processors and spec <= _lib.LSK_MAX_SPEC
                and hasattr(engine, "spec_step_sampled")):
            # a step called on its own (the reference's tests do) starts a stream; the first step of generate_token_ids continues the one
            # that call drew, so the fused call, the step loop and the layer pipeline consume torch's generator identically
            if self._sample_base is None or (past_key_values is None and not self._stream_fresh):
                self._new_sample_stream()
            self._stream_fresh = False
            step = engine.spec_step_sampled(new_ids, spec, exit_layer, eos_token_ids, temperature, top_k, top_p,
                                            torch.initial_seed(), self._sample_base + self._sample_step)
            self._sample_step += 1
        elif sample or logits_processors or spec > _lib.LSK_MAX_SPEC:
            # (more than 15 speculations do not fit the 16-row fused verify block: same kernels, rows walked in
            #  16-row passes from the host)
            step = self._slow_step(engine, new_ids, spec, exit_layer, eos_token_ids, sample, temperature, top_k, top_p,
                                   logits_processors)
        else:
            step = engine.spec_step(new_ids, spec, exit_layer, eos_token_ids)
        n = step.num_matches
        output_ids.extend(step.emitted)              # the CALLER's list grows in place, as in the reference (SSG:204-205), and is returned
        next_input = torch.tensor([[step.next_token]], dtype=input_ids.dtype, device=input_ids.device)
        if streamer:
            drafts = torch.tensor([step.draft_tokens[: step.num_drafts]])
            if hasattr(streamer, "delete"):          # SpeculativeTextStreamer protocol (SSG:158-161, :207-213)
                streamer.put(drafts, is_draft=True)
                streamer.delete(drafts.shape[1])
                streamer.put(drafts[0, :n])
                streamer.put(torch.tensor(step.emitted[n:n + 1]))
            else:
                streamer.put(torch.LongTensor(output_ids[len(output_ids) - n - 1:]))
        # crop_past_key_values(past, len(input_ids_list) + len(output_ids) - 1): a counter write
        target = len(input_ids_list) + len(output_ids) - 1
        if target != engine.kv_len:
            self._commit(engine, target)
        return next_input, output_ids, EngineCache(engine), n, step.num_drafts

    # ------------------------------------------------------------------------------ slow path
    def _logits_rows(self, engine: HipEngine, rows: Sequence[Tuple[int, int, int]], dtype=None) -> torch.Tensor:
        """Final-norm + lm_head logits of the listed (buffer, row_base, count) blocks -> [1, M, V]."""
        total = sum(c for _, _, c in rows)
        out = torch.empty(total, engine.vocab, dtype=torch.float32, device=engine.device)
        at = 0
        for buf, base, count in rows:
            for r0 in range(0, count, _lib.LSK_MAX_ROWS):
                m = min(_lib.LSK_MAX_ROWS, count - r0)
                engine.run_head(buf, base + r0, m, logits=out[at:at + m], want_tokens=False)
                at += m
        return out.to(dtype or getattr(engine, "dtype", torch.bfloat16)).unsqueeze(0)

    def _slow_step(self, engine: HipEngine, ids: List[int], spec: int, exit_layer: int, eos: List[int], sample: bool,
                   temperature: float, top_k: int, top_p: float, processors):
        from .engine import StepResult
        P, E = len(ids), exit_layer
        C = engine.kv_len
        dev = engine.device
        # the step rows (input token + drafts) live in the 16-row step buffer, or -- for more than 15
        # speculations -- right behind the prompt rows of the bulk buffer
        if spec <= _lib.LSK_MAX_SPEC:
            SBUF, SBASE = BUF_STEP, 0
        else:
            if P + spec > engine.max_prompt + 16:
                raise ValueError(f"num_speculations={spec} needs max_prompt >= {P + spec - 16} (engine has {engine.max_prompt})")
            SBUF, SBASE = BUF_BULK, P - 1
        if P > 1:
            engine.embed_rows(ids[:-1], BUF_BULK, 0)
            engine.run_bulk(P - 1, 0, E)            # same prefill kernels as the fused path
        drafts: List[int] = []
        draft_probs = []
        tok = ids[-1]
        draft_input = torch.tensor([ids], device=dev)
        j = 0
        while True:
            engine.embed_rows([tok], SBUF, SBASE + j)
            engine.run_layers(SBUF, SBASE + j, 1, P - 1 + j, 0, E)
            if j >= spec:
                break
            # forward_early returns logits for every input row (LMU:271-273): the prompt rows on call 0
            blocks = ([(BUF_BULK, 0, P - 1)] if (j == 0 and P > 1 and processors) else []) + [(SBUF, SBASE + j, 1)]
            logits = self._logits_rows(engine, blocks)
            if processors:
                logits = processors(draft_input, logits)
            t, prob = decode_rows(logits, True, sample, temperature, top_k, top_p)
            tok = int(t.item())
            drafts.append(tok)
            if sample:
                draft_probs.append(prob)
            draft_input = torch.tensor([[tok]], device=dev)
            j += 1
            if tok in eos:
                engine.embed_rows([tok], SBUF, SBASE + j)
                engine.run_layers(SBUF, SBASE + j, 1, P - 1 + j, 0, E)
                break
        td = len(drafts)
        prefill = torch.tensor([ids + drafts], device=dev)
        logits = self._verify_logits(engine, P, td, E, SBUF, SBASE, bool(processors))
        if processors:
            logits = processors(prefill, logits)
        vlogits = logits[:, -(td + 1):, :]
        vt, vprobs = decode_rows(vlogits, False, sample, temperature, top_k, top_p)
        verified = [int(x) for x in vt.reshape(-1).tolist()]
        n = 0
        if not sample:
            while n < td and drafts[n] == verified[n]:
                n += 1
        else:
            rand = torch.rand(td, device=dev)
            for i in range(td):
                ratio = vprobs[i, drafts[i]].item() / draft_probs[i][0, drafts[i]].item()
                if rand[i].item() < min(1.0, ratio):
                    n += 1
                else:
                    resid = _residual_distribution(vprobs[i, :], draft_probs[i][0])
                    verified[n] = int(torch.multinomial(resid, num_samples=1).item())
                    break
        self._commit(engine, C + P + n)
        return StepResult(n, td, verified[n], C + P + n, drafts[:n] + [verified[n]], drafts, verified)


class HipAutoRegressiveGenerationStrategy(GenerationStrategy):
    def __init__(self, engine_kwargs: Optional[dict] = None, fused_generate: bool = True) -> None:
        self.engine_kwargs = engine_kwargs or {}
        self.fused_generate = fused_generate

    # where the engine and the layers live (overridden by the layer pipeline's rank-0 slow path, layerskip_amd/pipeline_strategy.py)
    def _get_engine(self, model) -> HipEngine:
        return get_engine(model, **self.engine_kwargs)

    def _forward_logits(self, engine: HipEngine, ids: List[int], layer_end: int, prompt_rows: bool) -> torch.Tensor:
        """`forward` / `forward_early` (LMU:155-276) over the new ids: logits [1, (P - 1 if prompt_rows) + 1, V]."""
        P = len(ids)
        if P > 1:
            engine.embed_rows(ids[:-1], BUF_BULK, 0)
            engine.run_bulk(P - 1, 0, layer_end)
        engine.embed_rows(ids[-1:], BUF_STEP, 0)
        engine.run_layers(BUF_STEP, 0, 1, P - 1, 0, layer_end)
        blocks = ([(BUF_BULK, 0, P - 1)] if (P > 1 and prompt_rows) else []) + [(BUF_STEP, 0, 1)]
        return HipSelfSpeculativeGenerationStrategy._logits_rows(None, engine, blocks)

    def _commit(self, engine: HipEngine, kv_len: int) -> None:
        engine.set_kv_len(kv_len)

    def generate_token_ids(self, model, input_ids: List[int], eos_token_ids: List[int],
                           generation_config: GenerationConfig, logits_processors=None, stopping_criteria=None,
                           streamer=None) -> GenerationStrategyResult:
        engine = self._get_engine(model)
        engine.ensure_capacity(len(input_ids) + generation_config.max_steps + 10, len(input_ids))
        engine.reset()
        eos_token_ids = [t for t in eos_token_ids if t is not None]
        layer_end = generation_config.exit_layer if generation_config.exit_layer > 0 else engine.num_layers
        if layer_end > engine.num_layers:
            raise ValueError(f"exit_layer={layer_end} > num_layers={engine.num_layers}")
        if (self.fused_generate and not generation_config.sample and not logits_processors and not stopping_criteria
                and streamer is None and hasattr(engine, "ar_generate")):
            tokens = engine.ar_generate([int(t) for t in input_ids], layer_end, eos_token_ids, generation_config.max_steps)
            return GenerationStrategyResult(predicted_tokens=tokens, acceptance_rate=None)
        cur = [int(t) for t in input_ids]
        cur_t = torch.tensor([cur])
        output_ids: List[int] = []
        slow = bool(logits_processors) or generation_config.sample
        for _ in range(generation_config.max_steps):
            if slow:
                tok_t = self._slow_next(engine, cur, cur_t, layer_end, generation_config, logits_processors)
                tok = int(tok_t.item())
            else:
                tok = engine.ar_step(cur, layer_end)
                tok_t = torch.tensor([tok])
            if streamer:
                streamer.put(tok_t)
            if tok in eos_token_ids:
                break
            if stopping_criteria:
                if torch.all(stopping_criteria(cur_t, scores=None)):
                    break
            output_ids.append(tok)
            cur = [tok]
            cur_t = torch.tensor([[tok]])
        return GenerationStrategyResult(predicted_tokens=output_ids, acceptance_rate=None)

    def _slow_next(self, engine: HipEngine, ids: List[int], ids_t: torch.Tensor, layer_end: int,
                   cfg: GenerationConfig, processors) -> torch.Tensor:
        P = len(ids)
        C = engine.kv_len
        logits = self._forward_logits(engine, ids, layer_end, bool(processors))
        if processors:
            logits = processors(ids_t.to(engine.device), logits)
        tok, _ = decode_rows(logits, True, cfg.sample, cfg.temperature, cfg.top_k, cfg.top_p)
        self._commit(engine, C + P)
        return tok.reshape(-1)[:1].cpu()


STRATEGIES = {
    "autoregressive": HipAutoRegressiveGenerationStrategy,
    "self_speculative": HipSelfSpeculativeGenerationStrategy,
}
