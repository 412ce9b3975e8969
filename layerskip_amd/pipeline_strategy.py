"""The layer-range pipeline behind the reference's plugin surface (multi-GPU: one process per GPU under torchrun).

The reference's multi-GPU story is "launch generate.py / benchmark.py under torchrun; every rank but 0 exits; the model is
spread over the GPUs by `device_map="auto"`" (reference generate.py:41-52, :54-67).  Here the same launch line gives the
MI355X-native partition instead: every rank stays, owns a contiguous layer range on ITS GPU
(`layerskip_amd.checkpoint.load_layer_range` / `synthetic.build_model(layer_range=...)`), and

* rank 0 is the caller of `GenerationStrategy.generate_token_ids` (reference generator_base.py:51-62) -- the facade, the CLI
  drivers and their metrics are unchanged;
* ranks > 0 run `strategy.serve(model)` instead of the reference's `exit()`: they take part in one generation after the
  other (verify blocks over RCCL point-to-point, layerskip_amd/pipeline.py) until rank 0 calls `strategy.shutdown()`.

`HipPipelineSelfSpeculativeGenerationStrategy` covers what the one-GPU strategy covers:
  greedy   SSG:186-190 on the last rank's acceptance kernel;
  sampled  `sample=True` (the reference's default, generator_base.py:39): draws and modified rejection sampling (SSG:191-199) on the
           devices -- rank 0's p_i(x_i) travel in the header, the last rank's q_n comes back with the result
           (layerskip_amd/pipeline.py); draw for draw the one-GPU engine's generation under the same torch seed;
  slow     logits processors (`no_repeat_ngram_size`, generator_base.py:77-85): the one-GPU slow path on rank 0 with the verify's
           logits rows fetched from the last rank.
`HipPipelineAutoRegressiveGenerationStrategy` is the same pipeline with zero speculations (one token per round trip: ARG:26-80 over
a model that does not fit one GPU); early-exit-only decoding (ARG:44-51) runs rank-locally on rank 0, which owns those layers.
Streamers (both protocols) and stopping criteria are served per step on rank 0.
"""
from __future__ import annotations

import datetime
import os
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
from .pipeline import PipelineSpeculativeDecoder, Sampling, plan_partition
from .strategy_api import GenerationConfig, GenerationStrategy, GenerationStrategyResult


@dataclass
class DistContext:
    """What one rank knows about the job (torchrun's environment, reference generate.py:41-52)."""
    rank: int
    world: int
    local_rank: int
    device: torch.device          # this rank's compute device
    backend: str                  # "nccl" (= RCCL over xGMI: one rank per GPU) or "gloo" (CPU tests, ranks sharing one GPU)
    comm_device: torch.device     # where point-to-point tensors live: the GPU for nccl, the host for gloo
    group: object = None


def init_distributed(device: Optional[str] = None, backend: Optional[str] = None, timeout_minutes: int = 30) -> Optional[DistContext]:
    """Process-group set-up from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  None outside a
    multi-process launch.  One rank per GPU over RCCL when the node has a device per local rank; otherwise (a development box
    with fewer GPUs than ranks, or no GPU at all) every rank shares device 0 / the host and the collectives go through gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    have_gpu = torch.cuda.is_available() and (device is None or str(device).startswith("cuda"))
    if have_gpu and torch.cuda.device_count() >= local_world:
        dev = torch.device("cuda", local_rank)
        be = backend or "nccl"
    elif have_gpu:
        dev = torch.device("cuda", 0)
        be = backend or "gloo"
    else:
        dev = torch.device("cpu")
        be = backend or "gloo"
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if not dist.is_initialized():
        limit = datetime.timedelta(minutes=timeout_minutes)        # a lost rank must surface as an error, not as a hang
        if be == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend=be, timeout=limit)
    comm = dev if be == "nccl" else torch.device("cpu")
    return DistContext(rank, world, local_rank, dev, be, comm)


def _hip_backend(model, layer_range, **engine_kwargs):
    from .engine import HipEngine
    return HipEngine(model, layer_range=layer_range, **engine_kwargs)


# ONE decoder (one engine: packed weights, KV pool) per model object and process, shared by every strategy object that decodes
# with that model -- correctness.py alternates a speculative and an autoregressive strategy per sample, and the ranks > 0 serve both
# from one loop (the number of speculations travels with every generation's set-up broadcast).
_DECODERS = weakref.WeakKeyDictionary()


class _Rank0SlowStrategy(HipSelfSpeculativeGenerationStrategy):
    """The one-GPU slow path (logits processors; hip_strategies._slow_step) with the engine and the late layers where the pipeline
    keeps them: the draft loop runs row by row on rank 0's engine (early layers + its head copy), the verify's logits rows come from
    the last rank (`PipelineSpeculativeDecoder.remote_verify`), the rollback is rank 0's counter write + the next header."""

    def __init__(self, dec: PipelineSpeculativeDecoder) -> None:
        super().__init__(fused_generate=False, device_sampling=False)
        self._dec = dec

    def _get_engine(self, model, check_weights: bool = True):
        return self._dec.be

    def _verify_logits(self, engine, P: int, td: int, E: int, sbuf: int, sbase: int, prompt_rows: bool) -> torch.Tensor:
        if sbuf != 0 or sbase != 0:
            raise ValueError("the layer pipeline verifies at most 16 rows per step (num_speculations <= 15)")
        rows = self._dec.remote_verify(P, td + 1)                       # [(P - 1 if the generation asked for prompt rows) + td + 1, V]
        return (rows if prompt_rows else rows[rows.shape[0] - (td + 1):]).unsqueeze(0)

    def _commit(self, engine, kv_len: int) -> None:
        self._dec.commit(kv_len)


class _Rank0SlowARStrategy(HipAutoRegressiveGenerationStrategy):
    """The one-GPU autoregressive slow path (logits processors / sample=True: logits rows + torch on the host) over the pipeline:
    rank 0's layers locally, the rest and the head through `remote_verify` (one row per round trip)."""

    def __init__(self, dec: PipelineSpeculativeDecoder) -> None:
        super().__init__(fused_generate=False)
        self._dec = dec

    def _get_engine(self, model):
        return self._dec.be

    def _forward_logits(self, engine, ids: List[int], layer_end: int, prompt_rows: bool) -> torch.Tensor:
        from .engine import BUF_BULK, BUF_STEP
        P, first = len(ids), self._dec.E            # rank 0 owns [0, first); the pipeline's `exit layer` of an AR run is that boundary
        if P > 1:
            engine.embed_rows(ids[:-1], BUF_BULK, 0)
            engine.run_bulk(P - 1, 0, first)
        engine.embed_rows(ids[-1:], BUF_STEP, 0)
        engine.run_layers(BUF_STEP, 0, 1, P - 1, 0, first)
        rows = self._dec.remote_verify(P, 1)
        return (rows if prompt_rows else rows[rows.shape[0] - 1:]).unsqueeze(0)

    def _commit(self, engine, kv_len: int) -> None:
        self._dec.commit(kv_len)


class _Rank0LocalARStrategy(HipAutoRegressiveGenerationStrategy):
    """Early-exit-only autoregressive decoding (ARG:44-51) on a pipeline: layers [0, exit_layer) and a head copy are rank 0's, so the
    whole generation is rank-local -- the one-GPU strategy on rank 0's engine, no message moves."""

    def __init__(self, dec: PipelineSpeculativeDecoder) -> None:
        super().__init__()
        self._dec = dec

    def _get_engine(self, model):
        return self._dec.be


class HipPipelineSelfSpeculativeGenerationStrategy(GenerationStrategy):
    """`SelfSpeculativeGenerationStrategy` (SSG:31-99) over N GPUs.  All ranks construct it with the same `partition`;
    rank 0 calls `generate_token_ids`, the others `serve(model)`."""

    speculative = True

    def __init__(self, ctx: DistContext, partition: Sequence[Tuple[int, int]], backend_factory: Optional[Callable] = None,
                 optimistic: bool = True, engine_kwargs: Optional[dict] = None) -> None:
        self.ctx = ctx
        self.partition = [tuple(p) for p in partition]
        if len(self.partition) != ctx.world:
            raise ValueError(f"{len(self.partition)} layer ranges for {ctx.world} ranks")
        self.backend_factory = backend_factory or _hip_backend
        self.optimistic = optimistic
        self.engine_kwargs = engine_kwargs or {}
        self._last = None
        self.last_steps: List[Tuple[int, int]] = []

    # ------------------------------------------------------------------ plumbing
    def _decoder(self, model) -> PipelineSpeculativeDecoder:
        dec = _DECODERS.get(model)
        if dec is not None and [tuple(p) for p in dec.partition] != self.partition:
            raise RuntimeError(f"this model is already decoded with the layer ranges {dec.partition}, not {self.partition}")
        if dec is None:
            if getattr(model, "hf_device_map", None) and len(set(model.hf_device_map.values())) > 1:
                raise RuntimeError("this model was spread by device_map='auto'; the pipeline wants one process per GPU with the rank's "
                                   "own layer range on its device (layerskip_amd.checkpoint.load_layer_range)")
            backend = self.backend_factory(model, self.partition[self.ctx.rank], **self.engine_kwargs)
            dec = PipelineSpeculativeDecoder(backend, self.ctx.rank, self.ctx.world, self.partition, self.partition[0][1],
                                             group=self.ctx.group, comm_device=self.ctx.comm_device, optimistic=self.optimistic)
            # every rank builds its decoder exactly once per model (rank 0 at its first generation, the others in serve()): the one
            # collective moment to open the protocol's peer channels, so that no generation -- and no timing of one -- pays for RCCL's
            # lazy communicator set-up
            dec.warm_transport()
            _DECODERS[model] = dec
        self._last = dec
        return dec

    def serve(self, model) -> int:
        """Ranks > 0: serve generations until rank 0 shuts the pipeline down.  Returns how many were served."""
        return self._decoder(model).serve_forever()

    def shutdown(self) -> None:
        """Rank 0, once, when no more generations follow (the drivers call it in a `finally`)."""
        if self.ctx.rank == 0 and self._last is not None:
            self._last.shutdown()

    def stats(self) -> dict:
        return self._last.stats() if self._last is not None else {}

    # ------------------------------------------------------------------ the plugin call (rank 0)
    def _speculations(self, cfg: GenerationConfig) -> int:
        return max(0, int(cfg.num_speculations))

    def _slow_strategy(self, dec: PipelineSpeculativeDecoder):
        return _Rank0SlowStrategy(dec)

    def _exit_layer(self, cfg: GenerationConfig, num_layers: int) -> int:
        e = int(cfg.exit_layer)
        first = self.partition[0][1]
        if not (1 <= e <= first and e < num_layers):
            raise ValueError(f"exit_layer={e}: rank 0 owns layers [0, {first}) -- the draft loop must be rank-local (plan_partition)")
        return e

    def generate_token_ids(self, model, input_ids: List[int], eos_token_ids: List[int], generation_config: GenerationConfig,
                           logits_processors=None, stopping_criteria=None, streamer=None) -> GenerationStrategyResult:
        if self.ctx.rank != 0:
            raise RuntimeError("generate_token_ids is rank 0's call; ranks > 0 run strategy.serve(model)")
        dec = self._decoder(model)
        dec.E = self._exit_layer(generation_config, model.config.num_hidden_layers)
        eos = [t for t in eos_token_ids if t is not None]
        S = self._speculations(generation_config)
        if logits_processors:
            # the slow path: every decision on rank 0, on logits rows the user's callables have seen (SSG:138-139, :172-173)
            inner = self._slow_strategy(dec)
            cfg = generation_config

            def driver(_dec):
                return inner.generate_token_ids(model, input_ids, eos, cfg, logits_processors, stopping_criteria, streamer)

            return dec.generate([int(t) for t in input_ids], eos, int(cfg.max_steps), S, driver=driver, prompt_rows=True)
        sampling = None
        if generation_config.sample and not self.speculative:
            # sampled autoregressive decoding: one logits row per round trip, drawn with torch on rank 0 (the one-GPU strategy's path)
            inner, cfg = self._slow_strategy(dec), generation_config
            return dec.generate([int(t) for t in input_ids], eos, int(cfg.max_steps), S,
                                driver=lambda _dec: inner.generate_token_ids(model, input_ids, eos, cfg, logits_processors,
                                                                             stopping_criteria, streamer))
        if generation_config.sample:
            # the seeding contract of the one-GPU strategy (hip_strategies): key = torch.initial_seed(), counter base = one 62-bit draw
            # from torch's global generator per generation -- torch.manual_seed(s) reproduces a generation, on one GPU or on N
            sampling = Sampling(float(generation_config.temperature), int(generation_config.top_k), float(generation_config.top_p),
                                int(torch.initial_seed()), int(torch.randint(0, 2 ** 62, (1,)).item()))

        def on_step(drafts, n, emitted, nxt):
            if streamer is not None:
                d = torch.tensor([list(drafts)], dtype=torch.long)
                if hasattr(streamer, "delete"):          # SpeculativeTextStreamer protocol (SSG:158-161, :207-213)
                    streamer.put(d, is_draft=True)
                    streamer.delete(d.shape[1])
                    streamer.put(d[0, :n])
                    streamer.put(torch.tensor(emitted[n:n + 1]))
                else:
                    streamer.put(torch.LongTensor(list(emitted)))
            if stopping_criteria:
                return bool(torch.all(stopping_criteria(torch.tensor([[nxt]]), scores=None)))      # SSG:92-95: on the next input
            return False

        res = dec.generate([int(t) for t in input_ids], eos, int(generation_config.max_steps), S,
                           on_step=on_step if (streamer is not None or stopping_criteria) else None, sampling=sampling)
        self.last_steps = list(res.steps)
        return self._result(res)

    def _result(self, res) -> GenerationStrategyResult:
        matches = sum(n for _, n in res.steps)
        drafts = sum(td for td, _ in res.steps)
        return GenerationStrategyResult(predicted_tokens=res.predicted_tokens, acceptance_rate=matches / drafts)   # SSG:98 (ZeroDivisionError too)


class HipPipelineAutoRegressiveGenerationStrategy(HipPipelineSelfSpeculativeGenerationStrategy):
    """`AutoRegressiveGenerationStrategy` (ARG:25-80) over N GPUs: the same pipeline with zero speculations -- every round trip
    carries one row and yields one token.  Early-exit-only decoding (`exit_layer > 0`, ARG:44-51) never leaves rank 0."""

    speculative = False

    def _speculations(self, cfg: GenerationConfig) -> int:
        return 0

    def _slow_strategy(self, dec: PipelineSpeculativeDecoder):
        return _Rank0SlowARStrategy(dec)

    def _exit_layer(self, cfg: GenerationConfig, num_layers: int) -> int:
        return self.partition[0][1]

    def generate_token_ids(self, model, input_ids: List[int], eos_token_ids: List[int], generation_config: GenerationConfig,
                           logits_processors=None, stopping_criteria=None, streamer=None) -> GenerationStrategyResult:
        e, L = int(generation_config.exit_layer), model.config.num_hidden_layers
        if 0 < e < L:
            if self.ctx.rank != 0:
                raise RuntimeError("generate_token_ids is rank 0's call; ranks > 0 run strategy.serve(model)")
            if e > self.partition[0][1]:
                raise ValueError(f"exit_layer={e}: early-exit-only decoding runs on rank 0, which owns layers [0, {self.partition[0][1]})")
            return _Rank0LocalARStrategy(self._decoder(model)).generate_token_ids(model, input_ids, eos_token_ids, generation_config,
                                                                                 logits_processors, stopping_criteria, streamer)
        return super().generate_token_ids(model, input_ids, eos_token_ids, generation_config, logits_processors, stopping_criteria, streamer)

    def _result(self, res) -> GenerationStrategyResult:
        return GenerationStrategyResult(predicted_tokens=res.predicted_tokens, acceptance_rate=None)


PIPELINE_STRATEGIES = {
    "autoregressive": HipPipelineAutoRegressiveGenerationStrategy,
    "self_speculative": HipPipelineSelfSpeculativeGenerationStrategy,
}


def partition_for(num_layers: int, exit_layer: int, world: int, balance: str = "draft") -> List[Tuple[int, int]]:
    """The layer ranges the CLI drivers use: `plan_partition` for the run's exit layer (autoregressive runs, which have none:
    an even split by count)."""
    if exit_layer is None or exit_layer <= 0:
        return plan_partition(num_layers, max(1, num_layers // world), world, balance="memory")
    return plan_partition(num_layers, exit_layer, world, balance=balance)
