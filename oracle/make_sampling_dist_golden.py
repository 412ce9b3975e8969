"""TEST INFRASTRUCTURE.  Distribution fixtures of `sample=True` self-speculative decoding from the UNMODIFIED reference
(SSG:191-199, LMU:124-131) -- build container only (needs /root/reference):

    python oracle/make_sampling_dist_golden.py          # writes tests/golden/sampling/dist.json

Per case the reference's SelfSpeculativeGenerationStrategy runs N_RUNS generations (bf16 weights, torch.manual_seed(i)
before each); recorded: acceptance rate mean / std, output length histogram, per-step match histogram, the histogram of the
first and of the second emitted token (split in two halves of the runs: their mutual distance is the sampling noise the
GPU test calibrates against), and the reference's own nucleus (kept set of top_k_top_p_filtering) at the first position.
The HIP path draws different random numbers, so the comparison (tests/test_gpu_zz_sampling.py) is statistical."""
from __future__ import annotations

import collections
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sampling", "dist.json")
N_RUNS = 256

CASES = [
    # (name, fixture family, fixture name, temperature, top_k, top_p, num_speculations, max_steps)
    ("random_mha_nucleus", "legacy", "tiny_mha_s1", 0.12, 0, 0.9, 4, 10),
    ("random_mha_topk", "legacy", "tiny_mha_s1", 0.2, 50, 0.95, 4, 10),
    ("struct_gqa_hot", "struct", "tiny_gqa", 2.0, 0, 0.9, 6, 12),
]


def main():
    from conftest import build_case_model, build_struct_model, load_golden, load_struct
    ref = ref_shim.load_reference()
    lmu = ref.llama_model_utils
    out = []
    for name, family, fixture, temp, top_k, top_p, spec, max_steps in CASES:
        rec = load_golden(fixture) if family == "legacy" else load_struct(fixture)
        base = build_case_model(rec) if family == "legacy" else build_struct_model(rec)
        model = ref_shim.patch_model(copy.deepcopy(base))
        prompt, eos = rec["prompt"], [base.config.vocab_size]
        cfg = ref.generator_base.GenerationConfig(max_steps=max_steps, exit_layer=rec["exit_layer"], num_speculations=spec, sample=True,
                                                  temperature=temp, top_k=top_k, top_p=top_p)
        strat = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy()
        acc, lens = [], []
        first = [collections.Counter(), collections.Counter()]
        second = [collections.Counter(), collections.Counter()]
        for i in range(N_RUNS):
            torch.manual_seed(i)
            with torch.inference_mode():
                r = strat.generate_token_ids(model=model, input_ids=list(prompt), eos_token_ids=list(eos), generation_config=cfg)
            acc.append(r.acceptance_rate)
            lens.append(len(r.predicted_tokens))
            first[i % 2][r.predicted_tokens[0]] += 1
            if len(r.predicted_tokens) > 1:
                second[i % 2][r.predicted_tokens[1]] += 1
        with torch.inference_mode():
            logits = lmu.forward(model, torch.tensor([prompt]), None).logits[:, -1, :]
            warped = lmu.top_k_top_p_filtering(logits.clone() / temp, top_k=top_k, top_p=top_p)[0]
            probs = torch.softmax(warped.float(), -1)
        kept = torch.isfinite(warped).nonzero().flatten().tolist()
        mean = sum(acc) / N_RUNS
        std = (sum((a - mean) ** 2 for a in acc) / N_RUNS) ** 0.5
        out.append({
            "name": name, "family": family, "fixture": fixture, "temperature": temp, "top_k": top_k, "top_p": top_p,
            "num_speculations": spec, "max_steps": max_steps, "exit_layer": rec["exit_layer"], "n_runs": N_RUNS,
            "acceptance_mean": mean, "acceptance_std": std, "length_mean": sum(lens) / N_RUNS,
            "first_token_hist": [dict((str(k), v) for k, v in c.items()) for c in first],
            "second_token_hist": [dict((str(k), v) for k, v in c.items()) for c in second],
            "first_nucleus": kept, "first_nucleus_probs": [float(probs[i]) for i in kept],
        })
        print(f"{name}: acceptance {mean:.3f} +- {std:.3f}, mean length {sum(lens) / N_RUNS:.2f}, nucleus {len(kept)} tokens, "
              f"{len(first[0] + first[1])} distinct first tokens", flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
