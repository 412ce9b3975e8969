"""The layer-range pipeline behind the reference's plugin surface (multi-GPU: one process per GPU under torchrun).

The reference's multi-GPU story is "launch generate.py / benchmark.py under torchrun; every rank but 0 exits; the model is
spread over the GPUs by `device_map="auto"`" (reference generate.py:41-52, :54-67).  Here the same launch line gives the
MI355X-native partition instead: every rank stays, owns a contiguous layer range on ITS GPU
(`layerskip_amd.checkpoint.load_layer_range` / `synthetic.build_model(layer_range=...)`), and

* rank 0 is the caller of `GenerationStrategy.generate_token_ids` (reference generator_base.py:51-62) -- the facade, the CLI
  drivers and their metrics are unchanged;
* ranks > 0 run `strategy.serve(model)` instead of the reference's `exit()`: they take part in one generation after the
  other (verify blocks over RCCL point-to-point, layerskip_amd/pipeline.py) until rank 0 calls `strategy.shutdown()`.

`HipPipelineSelfSpeculativeGenerationStrategy` decodes greedily (SSG:186-190 on the last rank's acceptance kernel);
`HipPipelineAutoRegressiveGenerationStrategy` is the same pipeline with zero speculations (one token per round trip:
ARG:26-80 over a model that does not fit one GPU).  Streamers (both protocols) and stopping criteria are served per step on
rank 0; `sample=True` and logits processors need full logits rows on rank 0 and are refused with a clear error.
"""
from __future__ import annotations

import datetime
import os
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .pipeline import PipelineSpeculativeDecoder, plan_partition
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
        # refused BEFORE any collective, so the other ranks stay in their serve loop
        if generation_config.sample:
            raise NotImplementedError("the layer pipeline decodes greedily (the acceptance kernel runs on the last rank): pass "
                                      "--sample False, or decode sample=True on one GPU")
        if logits_processors:
            raise NotImplementedError("logits processors need full logits rows on rank 0; the layer pipeline keeps them on the last rank")
        dec = self._decoder(model)
        dec.E = self._exit_layer(generation_config, model.config.num_hidden_layers)
        eos = [t for t in eos_token_ids if t is not None]

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

        res = dec.generate([int(t) for t in input_ids], eos, int(generation_config.max_steps), self._speculations(generation_config),
                           on_step=on_step if (streamer is not None or stopping_criteria) else None)
        self.last_steps = list(res.steps)
        return self._result(res)

    def _result(self, res) -> GenerationStrategyResult:
        matches = sum(n for _, n in res.steps)
        drafts = sum(td for td, _ in res.steps)
        return GenerationStrategyResult(predicted_tokens=res.predicted_tokens, acceptance_rate=matches / drafts)   # SSG:98 (ZeroDivisionError too)


class HipPipelineAutoRegressiveGenerationStrategy(HipPipelineSelfSpeculativeGenerationStrategy):
    """`AutoRegressiveGenerationStrategy` (ARG:25-80) over N GPUs: the same pipeline with zero speculations -- every round trip
    carries one row and yields one token.  Early-exit-only decoding (`exit_layer > 0`, ARG:44-51) is a one-GPU feature."""

    speculative = False

    def _speculations(self, cfg: GenerationConfig) -> int:
        return 0

    def _exit_layer(self, cfg: GenerationConfig, num_layers: int) -> int:
        if int(cfg.exit_layer) > 0 and int(cfg.exit_layer) != num_layers:
            raise NotImplementedError("early-exit-only autoregressive decoding runs on one GPU (the early layers are rank 0's)")
        return self.partition[0][1]

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
