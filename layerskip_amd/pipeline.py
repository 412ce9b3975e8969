"""Layer-range pipeline for self-speculative decoding across GPUs (one process per GPU).

The reference has no multi-GPU decoding path of its own: `generate.py:41-52` initialises a process group
and every rank but 0 exits; models that do not fit one device are spread by `device_map="auto"`
(`generate.py:62`), i.e. accelerate hooks that copy activations between devices, strictly sequentially.
This module is the MI355X-native form of that partition (BASELINE.json north_star):

* rank 0 owns the embedding, decoder layers `[0, b0)` with `b0 >= exit_layer`, and a copy of the final
  norm + lm_head: the whole draft loop (`forward_early` x S, LMU:213-276) is rank-local and DEVICE-RESIDENT --
  one asynchronous `lsk_draft_block` call, each argmax embedded into the next row on the device;
* the remaining layers are split in contiguous ranges over ranks 1..N-1; the verify block
  (`exit_query_cache || last draft` = T_d+1 hidden rows, LMU:364-383) travels rank to rank as ONE fixed-size message
  per hop -- S+2 rows of the engines' message buffer: row 0 a header {go, prompt length, rows, verified context length,
  draft ids}, rows 1.. the hidden rows -- by point-to-point `send/recv` straight from / into engine memory (RCCL
  over one xGMI link per hop: (S+2) x H bf16 = 64 KB at 7B, 230 KB at 70B -- latency-bound, no collective on the
  data path).  NO RANK READS THE HEADER ON THE HOST BEFORE IT HAS ENQUEUED THE STEP: the rollback it carries
  (`crop_past_key_values`, SSG:219-221: a counter write on every rank; each layer's KV lives only on its owner) is
  applied by a kernel (`lsk_pipeline_apply`), the launches of the rank's layer range are queued behind the receive,
  the block is forwarded, and only then does the host look at the header (go / stop, the exact context length) -- its
  launch overhead overlaps the wait for the message instead of following it;
* the last rank runs the final norm + lm_head + argmax AND the wavefront-ballot acceptance kernel (SSG:186-190, a
  drafted EOS ends the draft, SSG:146-148) against the header's draft ids (`lsk_pipeline_tail`) and returns a 96-byte
  result block {num_matches, num_drafts, next token, context length, emitted tokens} to rank 0;
* OPTIMISTIC OVERLAP: while the verify block of step k is in flight, rank 0 keeps drafting step k+1
  under the assumption that every draft is accepted and that the bonus token equals its own head's next guess.  If
  that is what the verify returns, step k+1's verify block is ready the moment step k's result arrives; otherwise the
  continuation is discarded (its early-layer KV entries sit beyond the verified length and are overwritten).  Greedy
  output is unchanged either way: the continuation rows are exactly what the next step would have computed.

* SAMPLING (`sample=True`, the reference's default, generator_base.py:39; acceptance SSG:191-199) keeps the same message
  flow.  Modified rejection sampling needs, per draft, the two scalars q_i(x_i) and p_i(x_i), and at the first rejection the
  rows q_n and p_n; p lives on rank 0, q on the last rank.  So the header carries the step's Philox offset and the S scalars
  p_i(x_i); the last rank draws its verify tokens, runs the acceptance test (`lsk_pipeline_tail_sampled`) and answers with
  the result words followed by ONE probability row, q_n (128 KB at V = 32 000, 513 KB at 128 256; one direct message to
  rank 0, not through the chain); rank 0 draws the residual token from max(q_n - p_n, 0) with its own p_n
  (`lsk_pipeline_residual`).  Same counters and comparisons as the one-GPU kernel: under the same (seed, offset) the
  pipeline's sampled generation is DRAW FOR DRAW `lsk_spec_generate_sampled`'s.
* LOGITS PROCESSORS (`no_repeat_ngram_size`, generator_base.py:77-85; called at SSG:138-139, :172-173) are host callables
  on full logits rows: rank 0 runs the draft loop row by row on its own head (`hip_strategies.slow_step`), and for the verify
  the last rank sends the logits rows back instead of running an acceptance kernel (`remote_verify`); the decisions are then
  the one-GPU slow path's, on rank 0.

One sequence is a serial draft -> verify chain, so the pipeline buys capacity (a model beyond one GPU's HBM) and
hides only what the optimistic guess gets right; independent requests scale as replicas (bench.py reports both).
All ranks call `generate` collectively: rank 0 broadcasts (prompt length, speculations, max_steps, eos ids), EVERY rank
sizes its engine for the whole generation up front, and the ranks agree that all of them could before the first block
moves (a capacity error surfaces on every rank at once, not as a hang of the others).  The stage backend is the
`HipEngine` API; tests drive the same protocol over gloo with a CPU backend.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

BUF_STEP = 0
BUF_BULK = 1
BUF_MSG = 2         # row 0 = header (int32 words), rows 1.. = the verify block
_MAX_ROWS = 16
_MAX_EOS = 1024    # LSK_MAX_EOS (include/layerskip_hip.h): eos + stop token ids of one generation
_RES_WORDS = 24     # of the result block: num_matches, num_drafts, next token, context length, emitted[17]
RES_PENDING, RES_ERROR = 21, 22     # sampled result block (csrc/lsk_sample.h): residual draw pending, protocol error
# header words (layerskip_amd/csrc/lsk_accept.h); a host reads the first HDR_WORDS back
HDR_MAGIC, HDR_GO, HDR_P, HDR_ROWS, HDR_KV, HDR_DRAFTS, HDR_MODE, HDR_OFF_LO, HDR_OFF_HI, HDR_WORDS = 0, 1, 2, 3, 4, 5, 21, 22, 23, 24
HDR_MAGIC_VALUE = 0x4C534B31
MODE_GREEDY, MODE_SAMPLED, MODE_LOGITS = 0, 1, 2      # what the last rank does with a verify block
_META_WORDS = 4 + 7  # (P, S, max_steps, n_eos), mode, five sampling words, "prompt rows wanted"; the eos ids follow in a broadcast of their own


@dataclass
class Sampling:
    """sample=True parameters of one generation (GenerationConfig temperature / top_k / top_p, generator_base.py:35-44) and its
    Philox stream: key = seed, step i draws at counter offset `offset + i` (the contract of lsk_spec_generate_sampled)."""
    temperature: float
    top_k: int
    top_p: float
    seed: int
    offset: int


def _f64_bits(x: float) -> int:
    import struct
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _bits_f64(v: int) -> float:
    import struct
    return struct.unpack("<d", struct.pack("<q", int(v)))[0]


def _u64_to_i64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def plan_partition(num_layers: int, exit_layer: int, world: int, balance: str = "draft") -> List[Tuple[int, int]]:
    """Contiguous layer ranges per rank.
    balance = "draft" (default): rank 0 gets exactly the early layers -- it already pays for the S draft passes over
    them -- and the late layers are spread evenly over the other ranks (shortest verify chain per step);
    balance = "memory": layers are spread evenly over ALL ranks by count (rank 0 never fewer than `exit_layer`), the
    capacity split of SURVEY.md 8e: llama2-13B on 2 GPUs = [0, 20) + [20, 40), llama2-70B on 8 = [0, 12) + 7 x 9..10."""
    if world == 1:
        return [(0, num_layers)]
    if not (1 <= exit_layer < num_layers):
        raise ValueError("exit_layer must be in [1, num_layers)")
    if balance not in ("draft", "memory"):
        raise ValueError("balance must be 'draft' or 'memory'")
    first = exit_layer if balance == "draft" else max(exit_layer, -(-num_layers // world))
    late = num_layers - first
    rest = world - 1
    if late < rest:
        raise ValueError(f"{late} late layers cannot be split over {rest} ranks")
    out = [(0, first)]
    start = first
    for r in range(rest):
        n = late // rest + (1 if r < late % rest else 0)
        out.append((start, start + n))
        start += n
    return out


@dataclass
class PipelineResult:
    predicted_tokens: List[int]
    acceptance_rate: Optional[float]
    steps: List[Tuple[int, int]]          # (num_drafts, num_matches) per step


class PipelineSpeculativeDecoder:
    """Greedy self-speculative decoding with the decoder layers sharded over `world` ranks."""

    def __init__(self, backend, rank: int, world: int, partition: Sequence[Tuple[int, int]], exit_layer: int,
                 group=None, comm_device: Optional[torch.device] = None, optimistic: bool = True):
        self.be = backend
        self.rank, self.world = rank, world
        self.partition = list(partition)
        self.lb, self.le = self.partition[rank]
        self.E = exit_layer
        self.group = group
        self.dev = torch.device(comm_device if comm_device is not None else backend.device)
        self.direct = torch.device(self.dev).type == torch.device(backend.device).type      # send / recv straight on the engine's rows
        self.optimistic = optimistic and world > 1
        if self.partition[0][0] != 0 or self.partition[0][1] < exit_layer:
            raise ValueError("rank 0 must own layers [0, exit_layer)")
        for (a, b), (c, d) in zip(self.partition, self.partition[1:]):
            if b != c:
                raise ValueError("layer ranges must be contiguous")
        self._stats = {"steps": 0, "optimistic_attempts": 0, "optimistic_hits": 0, "draft_s": 0.0, "verify_roundtrip_s": 0.0,
                       "hop_wait_s": 0.0, "hop_enqueue_s": 0.0, "hops": 0}

    def stats(self) -> dict:
        """Rank 0: steps, optimistic attempts / hits, draft and verify-round-trip time per step.  Ranks > 0: per hop, the time
        the host spent enqueueing the step (receive posted .. block forwarded) and the time it then waited for the message."""
        s = dict(self._stats)
        if s["steps"]:
            s["draft_ms_per_step"] = round(1e3 * s["draft_s"] / s["steps"], 3)
            s["verify_roundtrip_ms_per_step"] = round(1e3 * s["verify_roundtrip_s"] / s["steps"], 3)
            s["optimistic_hit_rate"] = round(s["optimistic_hits"] / max(1, s["optimistic_attempts"]), 4)
        if s["hops"]:
            s["hop_enqueue_ms"] = round(1e3 * s["hop_enqueue_s"] / s["hops"], 3)
            s["hop_wait_ms"] = round(1e3 * s["hop_wait_s"] / s["hops"], 3)
        for k in ("draft_s", "verify_roundtrip_s", "hop_wait_s", "hop_enqueue_s"):
            s.pop(k)
        return s

    # ------------------------------------------------------------------ comm helpers
    def warm_transport(self) -> float:
        """Collective, once per process group: one dummy message over every edge the protocol uses -- rank r -> r + 1 down the chain and
        last rank -> rank 0 for the result -- so that the transport's lazily created point-to-point channels (RCCL builds a communicator
        per peer pair at its first send / recv: tens of milliseconds) exist BEFORE the first timed block moves.  Returns the seconds it
        took on this rank."""
        if self.world == 1:
            return 0.0
        t0 = time.perf_counter()
        tok = torch.zeros(64, dtype=torch.int32, device=self.dev)
        if self.rank > 0:
            dist.recv(tok, src=self.rank - 1, group=self.group)
        if self.rank < self.world - 1:
            dist.send(tok, dst=self.rank + 1, group=self.group)
        if self.rank == self.world - 1:
            dist.send(tok, dst=0, group=self.group)
        if self.rank == 0:
            dist.recv(tok, src=self.world - 1, group=self.group)
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        return time.perf_counter() - t0

    def _rows_out(self, buffer: int, row_base: int, m: int, dst: int) -> None:
        view = self.be.rows_view(buffer, row_base, m)
        dist.send(view if self.direct else view.to(self.dev), dst=dst, group=self.group)

    def _rows_in(self, buffer: int, row_base: int, m: int, src: int) -> None:
        view = self.be.rows_view(buffer, row_base, m)
        if self.direct:
            dist.recv(view, src=src, group=self.group)
        else:
            t = torch.empty(view.shape, dtype=view.dtype, device=self.dev)
            dist.recv(t, src=src, group=self.group)
            self.be.write_rows(buffer, row_base, t)

    def _agree(self, prompt_ids, eos_token_ids, max_steps: int, S: int, mode: int = MODE_GREEDY, sampling: Optional[Sampling] = None,
               prompt_rows: bool = False):
        """Collective set-up: rank 0's (P, S, max_steps, eos ids) reach every rank, every rank sizes its engine for the WHOLE
        generation (rank 0's optimistic continuation writes up to 2S+2 positions past the verified length; the stop message
        makes the late ranks run one block on stale rows), and all ranks learn whether all of them could."""
        be = self.be
        meta = torch.zeros(_META_WORDS, dtype=torch.int64)
        eos: List[int] = []
        if self.rank == 0 and prompt_ids is None:
            meta[0] = -1                      # shutdown(): the serve loops of the other ranks end
        elif self.rank == 0:
            # (the reference folds any number of stop_token_ids into this list, generator_base.py:106: the ids travel in a broadcast of their
            # own, sized by the count in the meta words)
            eos = list(dict.fromkeys(int(t) for t in eos_token_ids if t is not None and 0 <= int(t) < be.vocab))
            meta[:4] = torch.tensor([len(prompt_ids), S, max_steps, len(eos)])
            meta[4] = mode
            if sampling is not None:
                meta[5:10] = torch.tensor([int(sampling.top_k), _u64_to_i64(sampling.seed), _u64_to_i64(sampling.offset),
                                           _f64_bits(sampling.temperature), _f64_bits(sampling.top_p)], dtype=torch.int64)
            meta[10] = 1 if prompt_rows else 0
        if self.world > 1:
            meta = meta.to(self.dev)
            dist.broadcast(meta, src=0, group=self.group)
            meta = meta.cpu()
        P, S, max_steps, n_eos = (int(v) for v in meta[:4].tolist())
        if P == -1:
            return None
        if self.world > 1 and n_eos > 0:
            ids = (torch.tensor(eos, dtype=torch.int64) if self.rank == 0 else torch.zeros(n_eos, dtype=torch.int64)).to(self.dev)
            dist.broadcast(ids, src=0, group=self.group)
            eos = [int(v) for v in ids.cpu().tolist()]
        tail = [int(v) for v in meta[4:].tolist()]
        mode = tail[0]
        # logits mode: do the FIRST block's P - 1 prompt rows come back too?  Only logits processors look at them (SSG:172-173 shows them every
        # input row); without the bit a sampled autoregressive run over a 2 048-token prompt and V = 128 256 built, sent and dropped 0.5 GB
        self._prompt_rows = bool(tail[6])
        sampling = None
        if mode == MODE_SAMPLED:
            sampling = Sampling(_bits_f64(tail[4]), tail[1], _bits_f64(tail[5]), tail[2] & ((1 << 64) - 1), tail[3] & ((1 << 64) - 1))
        err = None
        try:
            if n_eos > _MAX_EOS:
                raise ValueError(f"{n_eos} distinct eos / stop token ids; at most {_MAX_EOS}")
            if S + 1 > _MAX_ROWS:
                raise ValueError("num_speculations too large for the 16-row verify block")
            if max_steps < 1 or P < 1:
                raise ValueError("max_steps and the prompt length must be at least 1")
            if mode not in (MODE_GREEDY, MODE_SAMPLED, MODE_LOGITS):
                raise ValueError(f"unknown pipeline mode {mode}")
            if mode == MODE_SAMPLED and not hasattr(be, "pipeline_tail_sampled"):
                raise ValueError("this stage backend cannot sample")
            be.ensure_capacity(P + max_steps + 2 * S + 2 + _MAX_ROWS, P)
            be.reset()
            be.set_eos(eos)
        except Exception as exc:          # noqa: BLE001 -- reported on every rank below
            err = exc
        if self.world > 1:
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int64, device=self.dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 0:
                raise RuntimeError(f"pipeline set-up failed on rank {self.rank}: {err}" if err is not None
                                   else "pipeline set-up failed on another rank (capacity / configuration); nothing was started")
        elif err is not None:
            raise err
        return P, S, max_steps, eos, mode, sampling

    # ------------------------------------------------------------------ the late ranks: serve verify blocks until told to stop
    def _serve(self, P0: int, S: int, mode: int = MODE_GREEDY, sampling: Optional[Sampling] = None) -> None:
        be = self.be
        last = self.rank == self.world - 1
        bound = 0                       # host-side UPPER bound of the verified context before the step being served
        p = P0                          # new tokens in front of the block: the prompt on the first step, 1 afterwards
        step = 0                        # blocks served: a sampled step draws at Philox offset `sampling.offset + step`
        while True:
            t0 = time.perf_counter()
            if p > 1:
                self._rows_in(BUF_BULK, 0, p - 1, self.rank - 1)
            self._rows_in(BUF_MSG, 0, S + 2, self.rank - 1)
            be.pipeline_apply(bound)                         # the header's rollback, on the device
            if p > 1:
                be.run_bulk(p - 1, self.lb, self.le)
            be.run_layers(BUF_MSG, 1, S + 1, p - 1, self.lb, self.le)
            if not last:
                if p > 1:
                    self._rows_out(BUF_BULK, 0, p - 1, self.rank + 1)
                self._rows_out(BUF_MSG, 0, S + 2, self.rank + 1)
            elif mode == MODE_LOGITS:
                # logits processors decide on rank 0: the rows' logits go back instead of an acceptance result (LMU:386-387: the
                # reference's forward_remainder returns logits for every input row, the prompt rows of the first step included)
                rows = be.logits_rows(([(BUF_BULK, 0, p - 1)] if (p > 1 and self._prompt_rows) else []) + [(BUF_MSG, 1, S + 1)])
                dist.send(rows if self.direct else rows.to(self.dev), dst=0, group=self.group)
            elif mode == MODE_SAMPLED:
                res = be.pipeline_tail_sampled(S + 1, sampling.temperature, sampling.top_k, sampling.top_p, sampling.seed,
                                               sampling.offset + step)
                dist.send(res if self.direct else res.to(self.dev), dst=0, group=self.group)
            else:
                res = be.pipeline_tail(S + 1)                # head + argmax + acceptance kernel -> device result block
                dist.send(res[:_RES_WORDS] if self.direct else res[:_RES_WORDS].to(self.dev), dst=0, group=self.group)
            t1 = time.perf_counter()
            # only now the host looks at the header: everything above is already queued behind the receive
            hdr = be.header()
            t2 = time.perf_counter()
            self._stats["hops"] += 1
            self._stats["hop_enqueue_s"] += t1 - t0
            self._stats["hop_wait_s"] += t2 - t1
            if hdr[HDR_MAGIC] != HDR_MAGIC_VALUE:
                raise RuntimeError(f"rank {self.rank}: message without a header (protocol out of step)")
            if not hdr[HDR_GO]:
                be.set_kv_len(hdr[HDR_KV])                   # the final verified length, exactly
                return
            if hdr[HDR_P] != p:
                raise RuntimeError(f"rank {self.rank}: header says {hdr[HDR_P]} new tokens, expected {p}")
            bound = hdr[HDR_KV] + p + S                      # the next block's rollback cannot exceed this
            p = 1
            step += 1

    # ------------------------------------------------------------------ rank 0: ship one verify block / fetch its result
    def _ship(self, P: int, m: int, kv: int, row_base: int, offset: Optional[int] = None):
        """Rank 0's own late layers (if it owns any) over step rows [row_base, row_base + m) and the prompt rows, then the block
        into the message buffer.  One rank: the tail (head + acceptance kernel) runs here; otherwise the block leaves.
        offset: the Philox offset of a sampled step (the header then carries it and the drafts' own probabilities)."""
        be, E = self.be, self.E
        if self.le > E:
            if P > 1:
                be.run_bulk(P - 1, E, self.le)
            be.run_layers(BUF_STEP, row_base, m, P - 1, E, self.le)
        if self._mode == MODE_SAMPLED:
            be.pipeline_pack_sampled(1, P, row_base, m, kv, offset)
        else:
            be.pipeline_pack(1, P, row_base, m, kv)
        if self.world == 1:
            if self._mode == MODE_SAMPLED:
                sm = self._sampling
                return be.pipeline_tail_sampled(m, sm.temperature, sm.top_k, sm.top_p, sm.seed, offset)
            if self._mode == MODE_LOGITS:
                return be.logits_rows(([(BUF_BULK, 0, P - 1)] if (P > 1 and self._prompt_rows) else []) + [(BUF_MSG, 1, self._S + 1)])
            return be.pipeline_tail(m)
        if P > 1:
            self._rows_out(BUF_BULK, 0, P - 1, 1)
        self._rows_out(BUF_MSG, 0, self._S + 2, 1)
        self._inflight += 1                        # the last rank answers every message with one result block
        self._inflight_rows = ((P - 1) if self._prompt_rows else 0) + self._S + 1
        self._shipped += 1
        return None

    def _answer(self) -> torch.Tensor:
        """Receive the last rank's answer to the message in flight, as it comes: greedy 24 int32 words, sampled the result words +
        q_n, logits mode (new tokens - 1) + S + 1 rows of V logits in the model dtype."""
        be = self.be
        if self._mode == MODE_LOGITS:
            t = torch.zeros(self._inflight_rows, be.vocab, dtype=getattr(be, "dtype", torch.float32), device=self.dev)
        elif self._mode == MODE_SAMPLED:
            t = torch.zeros(be.pipeline_result_words(), dtype=torch.int32, device=self.dev)
        else:
            t = torch.zeros(_RES_WORDS, dtype=torch.int32, device=self.dev)
        dist.recv(t, src=self.world - 1, group=self.group)
        self._inflight -= 1
        return t

    def _result(self, local, row_base: int = 0, offset: Optional[int] = None) -> List[int]:
        if self._mode == MODE_SAMPLED:
            blk = local if local is not None else self._answer().to(self.be.device)
            self.be.pipeline_residual(blk, row_base, self._sampling.seed, offset)      # no-op unless a rejection left the draw pending
            res = [int(v) for v in blk[:_RES_WORDS].tolist()]
            if res[RES_ERROR]:
                raise RuntimeError("pipeline out of step: the last rank's Philox offset is not the header's")
            return res
        if local is not None:
            return [int(v) for v in local[:_RES_WORDS].tolist()]
        return [int(v) for v in self._answer().tolist()]

    def _stop(self, kv: int) -> None:
        """Rank 0: the stop message (header only: go = 0, the final verified length) and its answer."""
        if self._mode == MODE_SAMPLED:
            self.be.pipeline_pack_sampled(0, 1, 0, 1, kv, self._sampling.offset)
        else:
            self.be.pipeline_pack(0, 1, 0, 1, kv)
        # the late ranks expect the prompt rows in front of their FIRST message: a generation that ends before any block was shipped
        # (an error on rank 0 ahead of its first step) sends them too -- whatever they hold -- so that nobody is left in a receive
        p = self._P0 if self._shipped == 0 else 1
        if p > 1:
            self._rows_out(BUF_BULK, 0, p - 1, 1)
        self._inflight_rows = ((p - 1) if self._prompt_rows else 0) + self._S + 1
        self._rows_out(BUF_MSG, 0, self._S + 2, 1)
        self._inflight += 1
        self._answer()                             # the late ranks answer every message; this one is discarded

    # ------------------------------------------------------------------ whole generation (collective)
    def serve_forever(self) -> int:
        """Ranks > 0 of a long-lived deployment (the CLI drivers under torchrun): take part in one generation after the
        other until rank 0 calls `shutdown()`.  Returns the number of generations served.  (The reference's ranks > 0
        simply `exit()`, generate.py:49-51: it has no multi-GPU decoding path.)"""
        if self.rank == 0:
            raise RuntimeError("serve_forever is for ranks > 0; rank 0 calls generate() and, at the end, shutdown()")
        served = 0
        while self.generate(None, [], 0, 0) is not None:
            served += 1
        return served

    def shutdown(self) -> None:
        """Rank 0: end the other ranks' `serve_forever` loops (collective: one broadcast)."""
        if self.rank != 0:
            raise RuntimeError("shutdown is called by rank 0")
        if self.world > 1:
            self._agree(None, [], 0, 0)

    def generate(self, prompt_ids: Optional[Sequence[int]], eos_token_ids: Sequence[int], max_steps: int,
                 num_speculations: int, on_step=None, sampling: Optional[Sampling] = None, driver=None,
                 prompt_rows: bool = False) -> Optional[PipelineResult]:
        """Rank 0 passes the prompt and the settings; other ranks' arguments are ignored.  Mirrors SSG:32-99.
        on_step (rank 0): called after every speculation step with (draft tokens, number accepted, emitted tokens, next input
        token); a truthy return value ends the generation after that step (stopping criteria, SSG:92-95; streamers hang here too).
        sampling (rank 0): sample=True -- draws and modified rejection sampling on the devices (module docstring), draw for draw
        the one-GPU `lsk_spec_generate_sampled` under the same (seed, offset).
        driver (rank 0): logits processors -- `driver(self)` runs the generation itself on rank 0 (hip_strategies' slow path) and
        gets every verify's logits rows through `remote_verify`; its return value is this call's.  prompt_rows (with a driver): the first
        block's P - 1 prompt rows come back too -- what logits processors are shown (SSG:172-173); nobody else looks at them.
        Ranks > 0 get an empty result, or None when rank 0 shut the pipeline down instead of starting a generation."""
        if self.rank == 0 and prompt_ids is None:
            raise ValueError("rank 0 must pass the prompt")
        mode = MODE_LOGITS if driver is not None else (MODE_SAMPLED if sampling is not None else MODE_GREEDY)
        agreed = self._agree(prompt_ids, eos_token_ids, int(max_steps), int(num_speculations), mode, sampling,
                             bool(prompt_rows) and driver is not None)
        if agreed is None:
            return None
        P0, S, max_steps, eos, mode, sampling = agreed
        self._S, self._mode, self._sampling = S, mode, sampling
        if self.rank > 0:
            self._serve(P0, S, mode, sampling)
            return PipelineResult([], None, [])
        self._inflight = 0
        self._inflight_rows = S + 1
        self._kv_host = 0
        self._P0, self._shipped = P0, 0
        try:
            if driver is not None:
                out = driver(self)
                if self.world > 1:
                    self._stop(self._kv_host)
                return out
            return self._drive(prompt_ids, eos, max_steps, S, on_step)
        except BaseException:
            # An error on rank 0 mid-generation (an inconsistent result block, a failed launch) must not leave the other ranks
            # blocked in their next receive: collect the answers still in flight, send the stop message, then re-raise.  Best
            # effort -- if the transport itself is what failed, the process group's timeout ends the others.
            if self.world > 1:
                try:
                    while self._inflight > 0:
                        self._answer()
                    self._stop(self._kv_host)
                except Exception:       # noqa: BLE001
                    pass
            raise

    # ------------------------------------------------------------------ logits mode: rank 0's slow path asks for one verify at a time
    def remote_verify(self, P: int, m: int) -> torch.Tensor:
        """Rank 0, inside a `driver`: the late layers + final norm + lm_head of `forward_remainder` (LMU:364-387) over step rows
        [0, m) and the P - 1 prompt rows in front of them, wherever those layers live -> logits [(P - 1 if prompt rows were agreed) + m, V]
        in the model dtype on rank 0's device (the reference returns logits for EVERY input row; processors see them all, SSG:172-173)."""
        local = self._ship(P, m, self._kv_host, 0)
        rows = local if local is not None else self._answer()
        return rows.to(self.be.device)[: ((P - 1) if self._prompt_rows else 0) + m]

    def commit(self, kv: int) -> None:
        """Rank 0, inside a `driver`: the verified context length after a step (the next header carries it to the other ranks)."""
        self._kv_host = int(kv)
        self.be.set_kv_len(self._kv_host)

    def _drive(self, prompt_ids, eos, max_steps: int, S: int, on_step=None) -> PipelineResult:
        """Rank 0's loop (SSG:51-95)."""
        be, E = self.be, self.E
        out: List[int] = []
        steps: List[Tuple[int, int]] = []
        matches = gens = 0
        cur = [int(t) for t in prompt_ids]
        kv = 0                                     # verified context length (host mirror)
        cont = None                                # a ready optimistic continuation (its rows were moved down to row 0)
        # rows of one block: input + S drafts; a continuation lives in rows S+1 .. 2S+1 (+ one row for its own guess)
        room_cont = 2 * S + 2 <= _MAX_ROWS
        room_chain = 2 * S + 3 <= _MAX_ROWS
        sm = self._sampling if self._mode == MODE_SAMPLED else None
        step_i = 0                                 # a sampled step draws at Philox offset sm.offset + step_i (lsk_spec_generate_sampled's contract)
        while len(out) < max_steps:
            s_eff = max(0, min(S, max_steps - len(out) - 1))
            P = len(cur)
            t0 = time.perf_counter()
            fresh = not (cont is not None and s_eff == S)
            want_guess = False
            off = sm.offset + step_i if sm is not None else None
            if fresh and sm is not None:
                # (no optimistic continuation under sampling: the bonus token is a draw from the LAST rank's distribution)
                be.draft_block_sampled(cur, 0, s_eff + 1, P - 1, E, False, sm.temperature, sm.top_k, sm.top_p, sm.seed, off)
            elif fresh:
                want_guess = self.optimistic and s_eff == S and room_cont
                be.draft_block(cur, 0, s_eff + 1, P - 1, E, head_last=want_guess)
            # ship the block BEFORE looking at the drafts on the host: the drafted-EOS cut (SSG:146-148) and the prefix match are
            # the acceptance kernel's job on the last rank; the late ranks compute rows the cut drops, harmlessly
            local = self._ship(P, s_eff + 1, kv, 0, off)
            t1 = time.perf_counter()
            if fresh:
                toks = be.row_tokens(1, s_eff + (1 if want_guess else 0)) if (s_eff or want_guess) else []
                drafts, guess = toks[:s_eff], (toks[s_eff] if want_guess else None)
            else:
                drafts, guess = cont["tokens"], cont["guess"]
            cont = None
            no_eos_drafted = not any(t in eos for t in drafts)
            attempt = (self.world > 1 and guess is not None and len(drafts) == S and no_eos_drafted
                       and (max_steps - len(out) - (S + 1) - 1) >= S)
            if attempt:
                # step k+1, optimistically: input = the guessed bonus token (its embedding already sits in row S+1)
                be.draft_block(None, S + 1, S + 1, P - 1 + S + 1, E, head_last=room_chain)
                self._stats["optimistic_attempts"] += 1
            res = self._result(local, 0, off)
            step_i += 1
            t2 = time.perf_counter()
            n, td, nxt = res[0], res[1], res[2]
            emitted = res[4:4 + n + 1]
            if emitted != drafts[:n] + [nxt] or td > len(drafts):
                raise RuntimeError("pipeline result block inconsistent with the drafted tokens")
            kv += P + n
            self._kv_host = kv
            be.set_kv_len(kv)
            if attempt and n == S and nxt == guess:
                toks = be.row_tokens(S + 2, S + (1 if room_chain else 0))
                be.shift_rows(S + 1, 0, S + 1 + (1 if room_chain else 0))
                cont = {"tokens": toks[:S], "guess": toks[S] if room_chain else None}
                self._stats["optimistic_hits"] += 1
            steps.append((td, n))
            matches += n
            gens += td
            self._stats["steps"] += 1
            self._stats["draft_s"] += t1 - t0
            self._stats["verify_roundtrip_s"] += t2 - t1
            out.extend(emitted)
            # the callback sees every step, the one an EOS then cuts included (the reference feeds its streamer inside the step,
            # SSG:207-216, before the EOS check of SSG:82-91); its stop request ranks behind the EOS cut (SSG:92-95)
            stop = bool(on_step(drafts[:td], n, emitted, nxt)) if on_step is not None else False
            hit = [out.index(e) for e in eos if e in out]
            if hit:
                out = out[: hit[0]]
                break
            if stop:
                break
            cur = [nxt]
        if self.world > 1:
            self._stop(kv)
        rate = (matches / gens) if gens else None
        return PipelineResult(out, rate, steps)
