"""Layer-range pipeline for self-speculative decoding across GPUs (one process per GPU).

The reference has no multi-GPU decoding path of its own: `generate.py:41-52` initialises a process group
and every rank but 0 exits; models that do not fit one device are spread by `device_map="auto"`
(`generate.py:62`), i.e. accelerate hooks that copy activations between devices, strictly sequentially.
This module is the MI355X-native form of that partition (BASELINE.json north_star):

* rank 0 owns the embedding, decoder layers `[0, b0)` with `b0 >= exit_layer`, and a copy of the final
  norm + lm_head: the whole draft loop (`forward_early` x S, LMU:213-276) is rank-local and DEVICE-RESIDENT --
  one asynchronous `lsk_draft_block` call, each argmax embedded into the next row on the device, one host read of the
  S draft ids at the end (the reference: one sync and one upload per draft token, SSG:141,145);
* the remaining layers are split in contiguous ranges over ranks 1..N-1; the verify block
  (`exit_query_cache || last draft` = T_d+1 hidden rows, LMU:364-383) is streamed rank to rank with
  point-to-point `send/recv` straight from / into the engines' row buffers (RCCL over one xGMI link per hop:
  (T_d+1) x H bf16 = 56 KB at 7B, 208 KB at 70B -- latency-bound, no collective on the data path), preceded by ONE
  64-byte header {go, P, rows, verified context length} that also carries the previous step's rollback
  (`crop_past_key_values`, SSG:219-221, is a counter write on every rank; each layer's KV lives only on its owner);
* the last rank runs the final norm + lm_head + argmax and returns the T_d+1 verified token ids (<= 136 B) to
  rank 0, which runs the greedy acceptance (SSG:186-190);
* OPTIMISTIC OVERLAP: while the verify block of step k is in flight, rank 0 keeps drafting step k+1
  under the assumption that every draft is accepted and that the bonus token equals its own head's next guess.  If
  that is what the verify returns, step k+1's verify block is ready the moment step k's result arrives; otherwise the
  continuation is discarded (its early-layer KV entries sit beyond the verified length and are overwritten).  Greedy
  output is unchanged either way: the continuation rows are exactly what the next step would have computed.

One sequence is a serial draft -> verify chain, so the pipeline buys capacity (a model beyond one GPU's HBM) and
hides only what the optimistic guess gets right; independent requests scale as replicas (bench.py reports both).
All ranks call `generate` collectively.  The stage backend is the `HipEngine` API; tests drive the same protocol over
gloo with a CPU backend.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

BUF_STEP = 0
BUF_BULK = 1
_MAX_ROWS = 16
_HDR = 8            # int64 words of the per-step header


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
                 group=None, comm_device: Optional[torch.device] = None, optimistic: bool = True):
        self.be = backend
        self.rank, self.world = rank, world
        self.partition = list(partition)
        self.lb, self.le = self.partition[rank]
        self.E = exit_layer
        self.group = group
        self.dev = comm_device if comm_device is not None else backend.device
        self.direct = torch.device(self.dev).type == torch.device(backend.device).type      # send / recv straight on the engine's rows
        self.optimistic = optimistic and world > 1
        if self.partition[0][0] != 0 or self.partition[0][1] < exit_layer:
            raise ValueError("rank 0 must own layers [0, exit_layer)")
        for (a, b), (c, d) in zip(self.partition, self.partition[1:]):
            if b != c:
                raise ValueError("layer ranges must be contiguous")
        self._stats = {"steps": 0, "optimistic_attempts": 0, "optimistic_hits": 0, "draft_s": 0.0, "verify_roundtrip_s": 0.0}

    def stats(self) -> dict:
        s = dict(self._stats)
        if s["steps"]:
            s["draft_ms_per_step"] = round(1e3 * s["draft_s"] / s["steps"], 3)
            s["verify_roundtrip_ms_per_step"] = round(1e3 * s["verify_roundtrip_s"] / s["steps"], 3)
        s.pop("draft_s"), s.pop("verify_roundtrip_s")
        return s

    # ------------------------------------------------------------------ comm helpers
    def _send_ints(self, values: Sequence[int], dst: int) -> None:
        t = torch.zeros(max(_HDR, len(values)), dtype=torch.int64)
        t[: len(values)] = torch.tensor([int(v) for v in values], dtype=torch.int64)
        dist.send(t.to(self.dev), dst=dst, group=self.group)

    def _recv_ints(self, n: int, src: int) -> List[int]:
        t = torch.zeros(max(_HDR, n), dtype=torch.int64, device=self.dev)
        dist.recv(t, src=src, group=self.group)
        return [int(v) for v in t.tolist()[:n]]

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

    # ------------------------------------------------------------------ the late ranks: serve verify blocks until told to stop
    def _serve(self) -> None:
        be = self.be
        last = self.world - 1
        while True:
            go, P, m, kv = self._recv_ints(4, self.rank - 1)
            if self.rank < last:
                self._send_ints([go, P, m, kv], self.rank + 1)
            be.set_kv_len(kv)                      # the rollback of the previous step / the final verified length
            if not go:
                return
            if P > 1:
                self._rows_in(BUF_BULK, 0, P - 1, self.rank - 1)
            self._rows_in(BUF_STEP, 0, m, self.rank - 1)
            if P > 1:
                be.run_bulk(P - 1, self.lb, self.le)
            be.run_layers(BUF_STEP, 0, m, P - 1, self.lb, self.le)
            if self.rank < last:
                if P > 1:
                    self._rows_out(BUF_BULK, 0, P - 1, self.rank + 1)
                self._rows_out(BUF_STEP, 0, m, self.rank + 1)
            else:
                self._send_ints(be.run_head(BUF_STEP, 0, m), 0)

    # ------------------------------------------------------------------ rank 0: one verify round trip
    def _verify(self, P: int, m: int, kv: int, row_base: int) -> List[int]:
        """Late layers + head over step rows [row_base, row_base + m) (and the P-1 prompt rows): locally for the range this
        rank owns, then through the other ranks."""
        be, E = self.be, self.E
        if self.le > E:
            if P > 1:
                be.run_bulk(P - 1, E, self.le)
            be.run_layers(BUF_STEP, row_base, m, P - 1, E, self.le)
        if self.world == 1:
            return be.run_head(BUF_STEP, row_base, m)
        self._send_ints([1, P, m, kv], 1)
        if P > 1:
            self._rows_out(BUF_BULK, 0, P - 1, 1)
        self._rows_out(BUF_STEP, row_base, m, 1)
        return []          # the ids come back later: _collect

    def _collect(self, m: int) -> List[int]:
        return self._recv_ints(m, self.world - 1)

    # ------------------------------------------------------------------ whole generation (collective)
    def generate(self, prompt_ids: Optional[Sequence[int]], eos_token_ids: Sequence[int], max_steps: int,
                 num_speculations: int) -> PipelineResult:
        """Rank 0 passes the prompt; other ranks pass None.  Mirrors SSG:32-99 (greedy)."""
        if num_speculations + 1 > _MAX_ROWS:
            raise ValueError("num_speculations too large for the 16-row verify block")
        be, E, S = self.be, self.E, int(num_speculations)
        be.reset()
        if self.rank > 0:
            self._serve()
            return PipelineResult([], None, [])
        eos = [int(t) for t in eos_token_ids]
        out: List[int] = []
        steps: List[Tuple[int, int]] = []
        matches = gens = 0
        cur = [int(t) for t in prompt_ids]
        kv = 0                                     # verified context length (host mirror)
        cont = None                                # a ready optimistic continuation: {"tokens": [...], "guess": int | None}
        # rows of one block: input + S drafts; a continuation lives in rows S+1 .. 2S+1 (+ one row for its own guess)
        room_cont = 2 * S + 2 <= _MAX_ROWS
        room_chain = 2 * S + 3 <= _MAX_ROWS
        while len(out) < max_steps:
            s_eff = max(0, min(S, max_steps - len(out) - 1))
            P = len(cur)
            t0 = time.perf_counter()
            if cont is not None and s_eff == S:
                # the rows of this step were drafted while the previous verify was in flight (and moved down to row 0)
                drafts, guess = cont["tokens"], cont["guess"]
            else:
                want_guess = self.optimistic and s_eff == S and room_cont
                be.draft_block(cur, 0, s_eff + 1, P - 1, E, head_last=want_guess)
                toks = be.row_tokens(1, s_eff + (1 if want_guess else 0)) if (s_eff or want_guess) else []
                drafts, guess = toks[:s_eff], (toks[s_eff] if want_guess else None)
            cont = None
            td = next((i + 1 for i, t in enumerate(drafts) if t in eos), len(drafts))       # a drafted EOS ends the draft (SSG:146-148)
            drafts = drafts[:td]
            m = td + 1
            t1 = time.perf_counter()
            verified = self._verify(P, m, kv, 0)
            attempt = self.world > 1 and guess is not None and td == S and (max_steps - len(out) - (S + 1) - 1) >= S
            if attempt:
                # step k+1, optimistically: input = the guessed bonus token (its embedding already sits in row S+1)
                be.draft_block(None, S + 1, S + 1, P - 1 + S + 1, E, head_last=room_chain)
                self._stats["optimistic_attempts"] += 1
            if self.world > 1:
                verified = self._collect(m)
            t2 = time.perf_counter()
            n = 0
            while n < td and drafts[n] == verified[n]:
                n += 1
            nxt = verified[n]
            kv += P + n
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
            out.extend(drafts[:n] + [nxt])
            hit = [out.index(e) for e in eos if e in out]
            if hit:
                out = out[: hit[0]]
                break
            cur = [nxt]
        if self.world > 1:
            self._send_ints([0, 0, 0, kv], 1)
        rate = (matches / gens) if gens else None
        return PipelineResult(out, rate, steps)
