"""TEST INFRASTRUCTURE.  Generates tests/golden/*.json from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py            # regenerate every fixture
    python oracle/make_golden.py --check    # re-run and compare with the committed files

For every synthetic case it
  1. builds the deterministic checkpoint (layerskip_amd/synthetic.py, CPU generator);
  2. runs the reference's own ``SelfSpeculativeGenerationStrategy`` and
     ``AutoRegressiveGenerationStrategy`` (greedy) through oracle/ref_shim.py, in bf16 (the
     dtype BASELINE.json quotes) and in fp32 on the same bf16-valued weights;
  3. runs the restatement oracle/llama_oracle.py on the same weights and REQUIRES identical
     token ids, per-step (num_drafts, num_matches) and bit-identical teacher-forced logits
     -- this is what pins the restatement to the reference;
  4. writes token ids, acceptance counters, per-token top-2 margins and a few logits rows.

BIG_CASES (``--only full7b_rand_512``; never part of a plain run -- 25+ GB of host memory, ~half an hour on 8 cores):
the BENCHMARKED checkpoint itself -- llama2-7B shape, ``build_model(seed=0, late_damping=0.03)`` from the CPU generator,
bench.py's prompt 0, 512-token prompt, exit_layer 8, 6 speculations -- through the unmodified reference in bf16 and in
fp32.  The one model object is converted in place (bf16 -> fp32 is exact), and besides the schema above the record carries
the per-step draft tokens, margins in bf16 ulp, and the top-32 logits of >= 64 rows of the bf16 trajectory (full depth and
early exit) with the reference's FP32 logits of the same positions beside them (``val_fp32``): what both bf16 runs -- the
reference's and the engine's -- approximate.
"""
from __future__ import annotations

import argparse
import copy
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from layerskip_amd import synthetic  # noqa: E402
from oracle import llama_oracle as lo  # noqa: E402
from oracle import ref_shim  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

CASES = {
    "tiny_mha_s0": synthetic.SyntheticCase("tiny-mha", seed=0, prompt_len=24, prompt_seed=0, max_steps=48),
    "tiny_mha_s1": synthetic.SyntheticCase("tiny-mha", seed=1, prompt_len=33, prompt_seed=1, max_steps=40,
                                           num_speculations=6, late_damping=0.05),
    "tiny_gqa_s0": synthetic.SyntheticCase("tiny-gqa", seed=0, prompt_len=37, prompt_seed=2, max_steps=40),
    "tiny_gqa_long": synthetic.SyntheticCase("tiny-gqa", seed=3, prompt_len=300, prompt_seed=3, max_steps=24,
                                             num_speculations=12),
    "tiny_d64_s0": synthetic.SyntheticCase("tiny-d64", seed=0, prompt_len=19, prompt_seed=4, max_steps=32),
    "small_wide_s0": synthetic.SyntheticCase("small-wide", seed=0, prompt_len=20, prompt_seed=5, max_steps=12),
    "slice7b_s0": synthetic.SyntheticCase("slice-7B", seed=0, prompt_len=40, prompt_seed=6, max_steps=20,
                                          late_damping=0.05),
}
# The benchmarked workload on the benchmarked checkpoint (bench.py: seed 0, late damping 0.03, prompt seed 0); --only selects it.
BIG_CASES = {
    "full7b_rand_512": synthetic.SyntheticCase("llama2-7B", seed=0, prompt_len=512, prompt_seed=0, max_steps=192,
                                               late_damping=0.03),
    # the other single-GPU BASELINE configs on bench.py's `other_configs` recipe (random init, late damping 0.03; CPU generator here):
    # llama3-8B (config #3: GQA 32:8, 128 256-entry vocabulary, theta 5e5) and llama3.2-1B (config #1's shape: d = 64, tied embeddings,
    # llama3 RoPE scaling)
    "full8b_rand_512": synthetic.SyntheticCase("llama3-8B", seed=0, prompt_len=512, prompt_seed=0, max_steps=128,
                                               late_damping=0.03),
    "full1b_rand_512": synthetic.SyntheticCase("llama3.2-1B", seed=0, prompt_len=512, prompt_seed=0, max_steps=192,
                                               late_damping=0.03),
}
# EOS cases are derived: the eos id is the first token at index >= k of the base case's fp32 output that has not occurred before.
EOS_CASES = {"tiny_mha_s0_eos": ("tiny_mha_s0", 3), "tiny_gqa_s0_eos": ("tiny_gqa_s0", 5), "tiny_mha_s1_eos": ("tiny_mha_s1", 10)}


def run_reference(ref, model, prompt, eos, case, strategy):
    gb = ref.generator_base
    cfg = gb.GenerationConfig(max_steps=case.max_steps, exit_layer=case.exit_layer,
                              num_speculations=case.num_speculations, sample=False,
                              generation_strategy=strategy)
    if strategy == "self_speculative":
        strat = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy()
    else:
        strat = ref.autoregressive_generator.AutoRegressiveGenerationStrategy()
        cfg.exit_layer = -1
    with torch.inference_mode():
        return strat.generate_token_ids(model=model, input_ids=list(prompt), eos_token_ids=list(eos),
                                        generation_config=cfg, logits_processors=None,
                                        stopping_criteria=None, streamer=None)


def topk_rows(logits: torch.Tensor, rows, k=8):
    out = []
    for r in rows:
        vals, idx = torch.topk(logits[r].float(), k)
        out.append({"row": int(r), "idx": idx.tolist(), "val": [float(v) for v in vals]})
    return out


def one_dtype(ref, model_bf16, case, prompt, eos, dtype):
    model = ref_shim.cast_parameters(copy.deepcopy(model_bf16), dtype)       # as from_pretrained(torch_dtype=dtype) does: buffers (inv_freq) untouched
    ref_shim.patch_model(model)
    ref_spec = run_reference(ref, model, prompt, eos, case, "self_speculative")
    ref_ar = run_reference(ref, model, prompt, eos, case, "autoregressive")
    om = lo.OracleModel.from_hf(model)
    with torch.inference_mode():
        mine_spec = lo.self_speculative_generate(om, list(prompt), list(eos), case.max_steps, case.exit_layer,
                                                 case.num_speculations)
        mine_ar = lo.autoregressive_generate(om, list(prompt), list(eos), case.max_steps)
        # --- pin the restatement to the reference ---
        assert mine_spec.predicted_tokens == ref_spec.predicted_tokens, "restated spec ids != reference"
        assert abs(mine_spec.acceptance_rate - ref_spec.acceptance_rate) == 0.0, "acceptance differs"
        assert mine_ar.predicted_tokens == ref_ar.predicted_tokens, "restated AR ids != reference"
        seq = list(prompt) + ref_spec.predicted_tokens
        ref_logits = ref.llama_model_utils.forward(model, torch.tensor([seq]), None).logits[0]
        my_logits = lo.teacher_forced_logits(om, seq)
        assert torch.equal(ref_logits, my_logits), "teacher-forced logits are not bit-identical"
    rows = sorted(set(r for r in [len(prompt) - 1, len(prompt), len(seq) // 2, len(seq) - 1] if r < len(seq)))
    return {
        "spec_tokens": ref_spec.predicted_tokens,
        "acceptance_rate": ref_spec.acceptance_rate,
        "steps": [[s.num_drafts, s.num_matches] for s in mine_spec.steps],
        "spec_margins": [round(m, 6) for m in mine_spec.margins],
        "ar_tokens": ref_ar.predicted_tokens,
        "ar_margins": [round(m, 6) for m in mine_ar.margins],
        "spec_equals_ar": ref_spec.predicted_tokens == ref_ar.predicted_tokens,
        "logits_topk": topk_rows(my_logits, rows),
    }


def logits_rows_big(logits: torch.Tensor, rows, k=32, exact=None):
    """Top-k entries of the given rows (bf16 values are exact as floats); `exact`: fp32 logits of the same sequence."""
    out = []
    for r in rows:
        row = logits[r].float()
        idx = sorted(torch.topk(row, k).indices.tolist())
        rec = {"row": int(r), "idx": idx, "val": [float(row[i]) for i in idx]}
        if exact is not None:
            rec["val_fp32"] = [float(exact[r, i]) for i in idx]
        out.append(rec)
    return out


def run_dtype_inplace(ref, model, case, prompt, eos, dtype):
    """One dtype of a multi-GB case on the ONE model object (converted in place).  Returns (record, oracle model)."""
    for prm in model.parameters():
        prm.data = prm.data.to(dtype)
    ref_shim.patch_model(model)
    t0 = time.time()
    ref_spec = run_reference(ref, model, prompt, eos, case, "self_speculative")
    t_spec = time.time() - t0
    t0 = time.time()
    ref_ar = run_reference(ref, model, prompt, eos, case, "autoregressive")
    t_ar = time.time() - t0
    om = lo.OracleModel.from_hf(model)
    with torch.inference_mode():
        mine_spec = lo.self_speculative_generate(om, list(prompt), list(eos), case.max_steps, case.exit_layer, case.num_speculations)
        mine_ar = lo.autoregressive_generate(om, list(prompt), list(eos), case.max_steps)
    assert mine_spec.predicted_tokens == ref_spec.predicted_tokens, "restated spec ids != reference"
    assert mine_spec.acceptance_rate == ref_spec.acceptance_rate, "acceptance differs"
    assert mine_ar.predicted_tokens == ref_ar.predicted_tokens, "restated AR ids != reference"
    rec = {
        "spec_tokens": ref_spec.predicted_tokens,
        "acceptance_rate": ref_spec.acceptance_rate,
        "steps": [[s.num_drafts, s.num_matches] for s in mine_spec.steps],
        "step_drafts": [s.draft_tokens for s in mine_spec.steps],
        "spec_margins": [round(m, 6) for m in mine_spec.margins],
        "spec_margins_ulp": [round(m, 3) for m in mine_spec.margins_ulp],
        "draft_margins_ulp": [round(m, 3) for m in mine_spec.draft_margins_ulp],
        "ar_tokens": ref_ar.predicted_tokens,
        "ar_margins": [round(m, 6) for m in mine_ar.margins],
        "spec_equals_ar": ref_spec.predicted_tokens == ref_ar.predicted_tokens,
        "reference_spec_seconds": round(t_spec, 2), "reference_ar_seconds": round(t_ar, 2),
    }
    return rec, om


def build_big_case(ref, name, case):
    """bf16 run first (its trajectory is THE sequence), then fp32 on the same object: the fp32 run's own generation, and the
    fp32 teacher-forced logits of the bf16 sequence (full depth and early exit)."""
    case = case.resolved()
    cfg = synthetic.make_config(case.shape)
    t0 = time.time()
    model = synthetic.build_model(cfg, seed=case.seed, exit_layer=case.exit_layer, late_damping=case.late_damping,
                                  dtype=torch.bfloat16, device="cpu")
    print(f"  {name}: checkpoint built in {time.time() - t0:.0f} s", flush=True)
    prompt = synthetic.make_prompt(cfg.vocab_size, case.prompt_len, case.prompt_seed)
    eos = [cfg.vocab_size]
    import transformers
    rec = {
        "name": name, "shape": case.shape, "seed": case.seed, "late_damping": case.late_damping,
        "exit_layer": case.exit_layer, "num_speculations": case.num_speculations,
        "prompt_len": case.prompt_len, "prompt_seed": case.prompt_seed, "max_steps": case.max_steps,
        "eos_token_ids": eos, "prompt": prompt, "attn_implementation": model.config._attn_implementation,
        "torch": torch.__version__, "transformers": transformers.__version__, "big": True,
    }
    lmu = ref.llama_model_utils
    b, om = run_dtype_inplace(ref, model, case, prompt, eos, torch.bfloat16)
    seq = list(prompt) + b["spec_tokens"]
    P, n = len(prompt), len(seq)
    with torch.inference_mode():
        ref_logits = lmu.forward(model, torch.tensor([seq]), None).logits[0]
        my_logits = lo.teacher_forced_logits(om, seq)
        assert torch.equal(ref_logits, my_logits), "teacher-forced logits are not bit-identical (bf16)"
        ref_early = lmu.forward_early(model, torch.tensor([seq]), None, case.exit_layer, None).logits[0]
        my_early = lo.forward_early(om, torch.tensor([seq]), None, case.exit_layer, None).logits[0]
        assert torch.equal(ref_early, my_early), "early-exit logits are not bit-identical (bf16)"
    # >= 64 rows: three prompt rows, then every third decision row of the generated part
    rows = sorted(set([0, P // 2] + list(range(P - 1, n, 3)) + [n - 1]))
    early_rows = rows[2::4]
    bf_full, bf_early = my_logits.clone(), my_early.clone()
    del om, ref_logits, my_logits, ref_early, my_early
    gc.collect()
    print(f"  {name} bf16: {len(b['spec_tokens'])} tokens, acceptance {b['acceptance_rate']:.3f}, spec==ar {b['spec_equals_ar']}, "
          f"reference spec {b['reference_spec_seconds']} s", flush=True)
    f, om = run_dtype_inplace(ref, model, case, prompt, eos, torch.float32)
    with torch.inference_mode():
        ex_full = lmu.forward(model, torch.tensor([seq]), None).logits[0]
        assert torch.equal(ex_full, lo.teacher_forced_logits(om, seq)), "teacher-forced logits are not bit-identical (fp32)"
        ex_early = lmu.forward_early(model, torch.tensor([seq]), None, case.exit_layer, None).logits[0]
    print(f"  {name} fp32: {len(f['spec_tokens'])} tokens, acceptance {f['acceptance_rate']:.3f}, spec==ar {f['spec_equals_ar']}", flush=True)
    b["logits_topk"] = logits_rows_big(bf_full, rows, 32, ex_full)
    b["early_logits_topk"] = logits_rows_big(bf_early, early_rows, 32, ex_early)
    # the reference's own bf16 run against its fp32 run, in bf16 ulp of the fp32 value (never finer than at |1.0|): the yardstick
    errs = []
    for r in b["logits_topk"]:
        for v, e in zip(r["val"], r["val_fp32"]):
            ulp = 2.0 ** (max(0, int(torch.floor(torch.log2(torch.tensor(abs(e) if abs(e) >= 1.0 else 1.0))).item())) - 7)
            errs.append(((v - e) / ulp, abs(v - e) / max(abs(e), 1e-9)))
    b["reference_bf16_vs_fp32"] = {"rms_ulp": (sum(x * x for x, _ in errs) / len(errs)) ** 0.5, "max_ulp": max(abs(x) for x, _ in errs),
                                   "max_rel": max(y for _, y in errs), "entries": len(errs)}
    f["first_divergence_from_bf16"] = next((i for i, (x, y) in enumerate(zip(f["spec_tokens"], b["spec_tokens"])) if x != y), None)
    rec["bf16"], rec["fp32"] = b, f
    del model, om
    gc.collect()
    return rec


def build_case(ref, name, case, eos=None):
    case = case.resolved()
    cfg = synthetic.make_config(case.shape)
    model = synthetic.build_model(cfg, seed=case.seed, exit_layer=case.exit_layer, late_damping=case.late_damping,
                                  dtype=torch.bfloat16, device="cpu")
    prompt = synthetic.make_prompt(cfg.vocab_size, case.prompt_len, case.prompt_seed)
    eos = [cfg.vocab_size] if eos is None else eos   # unreachable id -> always max_steps tokens
    rec = {
        "name": name, "shape": case.shape, "seed": case.seed, "late_damping": case.late_damping,
        "exit_layer": case.exit_layer, "num_speculations": case.num_speculations,
        "prompt_len": case.prompt_len, "prompt_seed": case.prompt_seed, "max_steps": case.max_steps,
        "eos_token_ids": eos, "prompt": prompt,
        "attn_implementation": model.config._attn_implementation,
        "torch": torch.__version__,
    }
    import transformers
    rec["transformers"] = transformers.__version__
    for dname, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        rec[dname] = one_dtype(ref, model, case, prompt, eos, dtype)
        print(f"  {name} {dname}: {len(rec[dname]['spec_tokens'])} tokens, acceptance "
              f"{rec[dname]['acceptance_rate']:.3f}, spec==ar {rec[dname]['spec_equals_ar']}, "
              f"min margin {min(rec[dname]['spec_margins'] or [0]):.4f}")
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    ref = ref_shim.load_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    records = {}
    for name, case in CASES.items():
        if args.only and args.only not in name:
            continue
        records[name] = build_case(ref, name, case)
    for name, case in BIG_CASES.items():
        if args.only and args.only == name:
            records[name] = build_big_case(ref, name, case)
    for name, (base, k) in EOS_CASES.items():
        if args.only and args.only not in name:
            continue
        if base not in records:
            records[base] = build_case(ref, base, CASES[base])
        toks = records[base]["fp32"]["spec_tokens"]
        k = next((i for i in range(k, len(toks)) if toks[i] not in toks[:i]), k)   # first occurrence => non-trivial cut
        eos_id = toks[k]
        records[name] = build_case(ref, name, CASES[base], eos=[eos_id])
    bad = 0
    for name, rec in records.items():
        path = os.path.join(GOLDEN_DIR, name + ".json")
        if args.check:
            with open(path) as f:
                old = json.load(f)
            same = all(old[d]["spec_tokens"] == rec[d]["spec_tokens"] and old[d]["ar_tokens"] == rec[d]["ar_tokens"]
                       and old[d]["steps"] == rec[d]["steps"] for d in ("bf16", "fp32"))
            print(("OK   " if same else "DIFF ") + name)
            bad += 0 if same else 1
        else:
            with open(path, "w") as f:
                json.dump(rec, f)
            print("wrote", path)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
