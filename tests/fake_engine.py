"""CPU stand-in for HipEngine used by the host-logic tests only: same method surface, arithmetic from
the reference-pinned oracle.  Lives under tests/ (the product never imports the oracle)."""
from typing import List, Sequence

import torch

from layerskip_amd.engine import StepResult
from oracle import llama_oracle as lo


class FakeEngine:
    def __init__(self, model):
        # fp32 arithmetic on the bf16-valued weights: torch's CPU bf16 GEMMs are not even reproducible across
        # two allocations of the same weights (alignment-dependent blocking), fp32 is (margins >> 1e-6)
        self.om = lo.OracleModel.from_hf(model, dtype=torch.float32)
        self.num_layers = self.om.num_layers
        self.vocab = self.om.embed.shape[0]
        self.device = torch.device("cpu")
        self.past = None
        self._kv_len = 0
        self.calls = []

    def ensure_capacity(self, total_tokens, prompt_len):
        self.calls.append(("ensure_capacity", total_tokens, prompt_len))

    def reset(self):
        self.past = None
        self._kv_len = 0

    @property
    def kv_len(self):
        return self._kv_len

    def set_kv_len(self, n):
        self._kv_len = n
        if self.past:
            self.past = lo.crop_past(self.past, n)

    def spec_step(self, input_ids: Sequence[int], num_speculations: int, exit_layer: int, eos: Sequence[int]) -> StepResult:
        ids = torch.tensor([list(input_ids)])
        # the oracle only needs len(input_ids_list) + len(output_ids) - 1 == kv_len + P + n for the crop
        fake_prompt = [0] * (self._kv_len + len(input_ids))
        with torch.inference_mode():
            new_in, out, past, n, td, tr = lo.single_step_speculation(
                self.om, ids, fake_prompt, [], num_speculations, self.past, list(eos), exit_layer)
        # crop inside the oracle used len(fake_prompt) + len(out) - 1 = kv_len + P + n
        self.past = past
        self._kv_len = self._kv_len + len(input_ids) + n
        return StepResult(n, td, int(new_in[0, 0]), self._kv_len, list(out), tr.draft_tokens, tr.verified_tokens)

    def ar_step(self, input_ids: Sequence[int], layer_end=None) -> int:
        ids = torch.tensor([list(input_ids)])
        with torch.inference_mode():
            if layer_end is None or layer_end == self.num_layers:
                r = lo.forward(self.om, ids, self.past)
            else:
                r = lo.forward_early(self.om, ids, self.past, layer_end, None)
        self.past = r.past
        self._kv_len += len(input_ids)
        return int(lo.decode_next_token_greedy(r.logits, token_idx=-1).item())


class FullFakeEngine:
    """CpuStageBackend (the building blocks the materialised-logits paths use) plus the fused `spec_step` / `ar_step`
    entry points, built FROM those blocks: the complete HipEngine surface the strategies touch, on the CPU oracle
    (fp32).  Lets the plugin run under the reference's own HuggingfaceLlamaGenerator without a GPU."""

    def __new__(cls, model, layer_range=None):
        from cpu_stage_backend import CpuStageBackend

        class _Impl(CpuStageBackend):
            dtype = torch.float32

            def spec_step(self, input_ids, num_speculations, exit_layer, eos):
                ids, S, E, L = list(input_ids), int(num_speculations), int(exit_layer), self.num_layers
                P = len(ids)
                if P > 1:
                    self.embed_rows(ids[:-1], 1, 0)
                    self.run_bulk(P - 1, 0, E)
                drafts, tok, j = [], ids[-1], 0
                while True:
                    self.embed_rows([tok], 0, j)
                    self.run_layers(0, j, 1, P - 1 + j, 0, E)
                    if j >= S:
                        break
                    tok = self.run_head(0, j, 1)[0]
                    drafts.append(tok)
                    j += 1
                    if tok in eos:
                        self.embed_rows([tok], 0, j)
                        self.run_layers(0, j, 1, P - 1 + j, 0, E)
                        break
                td = len(drafts)
                if P > 1:
                    self.run_bulk(P - 1, E, L)
                self.run_layers(0, 0, td + 1, P - 1, E, L)
                verified = self.run_head(0, 0, td + 1)
                n = 0
                while n < td and drafts[n] == verified[n]:
                    n += 1
                self.set_kv_len(self.kv_len + P + n)
                return StepResult(n, td, verified[n], self.kv_len, drafts[:n] + [verified[n]], drafts, verified)

            def ar_step(self, input_ids, layer_end=None):
                ids = list(input_ids)
                P, le = len(ids), int(layer_end or self.num_layers)
                if P > 1:
                    self.embed_rows(ids[:-1], 1, 0)
                    self.run_bulk(P - 1, 0, le)
                self.embed_rows(ids[-1:], 0, 0)
                self.run_layers(0, 0, 1, P - 1, 0, le)
                tok = self.run_head(0, 0, 1)[0]
                self.set_kv_len(self.kv_len + P)
                return tok

        return _Impl(model, layer_range=layer_range)
