"""Layer-range pipeline for self-speculative decoding across GPUs (one process per GPU).

The reference has no multi-GPU decoding path of its own: `generate.py:41-52` initialises a process group
and every rank but 0 exits; models that do not fit one device are spread by `device_map="auto"`
(`generate.py:62`), i.e. accelerate hooks that copy activations between devices, strictly sequentially.
This module is the MI355X-native form of that capacity mode:

* rank 0 owns the embedding, decoder layers `[0, b0)` with `b0 >= exit_layer`, and a copy of the final
  norm + lm_head: the whole draft loop (`forward_early` x S, LMU:213-276) is rank-local, zero traffic;
* the remaining layers are split in contiguous ranges over ranks 1..N-1; the verify block
  (`exit_query_cache || last draft` = T_d+1 hidden rows, LMU:364-383) is streamed rank to rank with
  point-to-point `send/recv` (RCCL over one xGMI link per hop: (T_d+1) x H bf16 = 56 KB at 7B, 208 KB at
  70B -- latency-bound, no collective on the data path);
* the last rank runs the final norm + lm_head + argmax and returns the T_d+1 verified token ids (<= 64 B) to
  rank 0, which runs the greedy acceptance (SSG:186-190) and broadcasts the new KV length (the rollback of
  SSG:219-221 is a counter write on every rank; each layer's KV lives only on its owner).

One sequence is a strictly serial draft -> verify chain, so the pipeline buys capacity, not tokens/s (every
BASELINE config fits one 288 GB MI355X; bench.py's default multi-GPU mode is one replica per GPU).  All ranks
call `generate` collectively.  The stage backend is the `HipEngine` building-block API; tests drive the same
protocol over gloo with a CPU backend.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

BUF_STEP = 0
BUF_BULK = 1
_MAX_ROWS = 16


def plan_partition(num_layers: int, exit_layer: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous layer ranges per rank.  Rank 0 gets the early layers (it also pays for the S draft
    passes over them); the late layers are spread evenly over the other ranks."""
    if world == 1:
        return [(0, num_layers)]
    if not (1 <= exit_layer < num_layers):
        raise ValueError("exit_layer must be in [1, num_layers)")
    late = num_layers - exit_layer
    rest = world - 1
    if late < rest:
        raise ValueError(f"{late} late layers cannot be split over {rest} ranks")
    out = [(0, exit_layer)]
    start = exit_layer
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
                 group=None, comm_device: Optional[torch.device] = None):
        self.be = backend
        self.rank, self.world = rank, world
        self.partition = list(partition)
        self.lb, self.le = self.partition[rank]
        self.E = exit_layer
        self.group = group
        self.dev = comm_device if comm_device is not None else backend.device
        if self.partition[0][0] != 0 or self.partition[0][1] < exit_layer:
            raise ValueError("rank 0 must own layers [0, exit_layer)")
        for (a, b), (c, d) in zip(self.partition, self.partition[1:]):
            if b != c:
                raise ValueError("layer ranges must be contiguous")

    # ------------------------------------------------------------------ comm helpers
    def _send(self, t: torch.Tensor, dst: int) -> None:
        dist.send(t.contiguous(), dst=dst, group=self.group)

    def _recv(self, shape, dtype, src: int) -> torch.Tensor:
        t = torch.empty(shape, dtype=dtype, device=self.dev)
        dist.recv(t, src=src, group=self.group)
        return t

    def _bcast_ints(self, values: Sequence[int], n: int) -> List[int]:
        t = torch.zeros(n, dtype=torch.int64, device=self.dev)
        if self.rank == 0:
            t[: len(values)] = torch.tensor(list(values), dtype=torch.int64)
        if self.world > 1:
            dist.broadcast(t, src=0, group=self.group)
        return [int(v) for v in t.tolist()]

    def _rows_out(self, buffer: int, row_base: int, m: int, dst: int) -> None:
        for r0 in range(0, m, 256):
            k = min(256, m - r0)
            self._send(self.be.read_rows(buffer, row_base + r0, k).to(self.dev), dst)

    def _rows_in(self, buffer: int, row_base: int, m: int, src: int) -> None:
        for r0 in range(0, m, 256):
            k = min(256, m - r0)
            self.be.write_rows(buffer, row_base + r0, self._recv((k, self.be.hidden), getattr(self.be, "dtype", torch.bfloat16), src))

    # ------------------------------------------------------------------ one speculation step
    def _step(self, ids: Optional[List[int]], spec: int, eos: Sequence[int]):
        be, E = self.be, self.E
        last = self.world - 1
        # header: prompt_len, num_speculations of this step
        P, S = self._bcast_ints([len(ids), spec] if self.rank == 0 else [], 2)
        drafts: List[int] = []
        if self.rank == 0:
            if P > 1:
                be.embed_rows(ids[:-1], BUF_BULK, 0)
                be.run_bulk(P - 1, 0, E)
            tok = ids[-1]
            j = 0
            while True:                                   # draft loop, rank-local (SSG:127-148)
                be.embed_rows([tok], BUF_STEP, j)
                be.run_layers(BUF_STEP, j, 1, P - 1 + j, 0, E)
                if j >= S:
                    break
                tok = be.run_head(BUF_STEP, j, 1)[0]
                drafts.append(tok)
                j += 1
                if tok in eos:
                    be.embed_rows([tok], BUF_STEP, j)
                    be.run_layers(BUF_STEP, j, 1, P - 1 + j, 0, E)
                    break
        td = self._bcast_ints([len(drafts)] if self.rank == 0 else [], 1)[0]
        m = td + 1
        # verify, late layers: stream the block through the ranks (forward_remainder, LMU:364-383)
        lo = max(self.lb, E) if self.rank == 0 else self.lb
        if self.rank > 0:
            if P > 1:
                self._rows_in(BUF_BULK, 0, P - 1, self.rank - 1)
            self._rows_in(BUF_STEP, 0, m, self.rank - 1)
        if lo < self.le:
            if P > 1:
                be.run_bulk(P - 1, lo, self.le)
            be.run_layers(BUF_STEP, 0, m, P - 1, lo, self.le)
        if self.rank < last:
            if P > 1:
                self._rows_out(BUF_BULK, 0, P - 1, self.rank + 1)
            self._rows_out(BUF_STEP, 0, m, self.rank + 1)
        verified: List[int] = []
        if self.rank == last:
            verified = be.run_head(BUF_STEP, 0, m)
            if last != 0:
                self._send(torch.tensor(verified, dtype=torch.int64, device=self.dev), 0)
        if self.rank == 0 and last != 0:
            verified = [int(v) for v in self._recv((m,), torch.int64, last).tolist()]
        # greedy acceptance on rank 0 (SSG:186-190), rollback everywhere (SSG:219-221)
        n = 0
        if self.rank == 0:
            while n < td and drafts[n] == verified[n]:
                n += 1
        res = self._bcast_ints([n, verified[n]] if self.rank == 0 else [], 2)
        n, nxt = res
        be.set_kv_len(be.kv_len + P + n)
        emitted = (drafts[:n] + [nxt]) if self.rank == 0 else []
        return emitted, nxt, n, td

    # ------------------------------------------------------------------ whole generation (collective)
    def generate(self, prompt_ids: Optional[Sequence[int]], eos_token_ids: Sequence[int], max_steps: int,
                 num_speculations: int) -> PipelineResult:
        """Rank 0 passes the prompt; other ranks pass None.  Mirrors SSG:32-99 (greedy)."""
        if num_speculations + 1 > _MAX_ROWS:
            raise ValueError("num_speculations too large for the 16-row verify block")
        self.be.reset()
        out: List[int] = []
        steps: List[Tuple[int, int]] = []
        matches = gens = 0
        cur = list(prompt_ids) if self.rank == 0 else None
        while True:
            go = self._bcast_ints([1 if len(out) < max_steps else 0] if self.rank == 0 else [], 1)[0]
            if not go:
                break
            spec = min(num_speculations, max_steps - len(out) - 1) if self.rank == 0 else 0
            emitted, nxt, n, td = self._step(cur, max(0, spec), eos_token_ids)
            steps.append((td, n))
            matches += n
            gens += td
            stop = 0
            if self.rank == 0:
                out.extend(emitted)
                hit = [out.index(e) for e in eos_token_ids if e in out]
                if hit:
                    out = out[: hit[0]]
                    stop = 1
                cur = [nxt]
            if self._bcast_ints([stop] if self.rank == 0 else [], 1)[0]:
                break
        rate = (matches / gens) if gens else None
        return PipelineResult(out if self.rank == 0 else [], rate, steps)
