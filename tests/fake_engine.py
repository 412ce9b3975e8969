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
