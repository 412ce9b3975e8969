"""The fp16 build of the engine (liblayerskip_hip_f16.so: the same kernels with elem_t = _Float16 and
v_mfma_f32_16x16x32_f16) against fixtures recorded from the unmodified reference in fp16 -- the dtype the reference's CLI hard-codes
(generate.py:59-64, `torch_dtype=torch.float16`).

* tests/golden/fp16/: four tiny random-init checkpoints (round 2; a first mismatch is accepted inside a near-tie);
* tests/golden/struct_fp16/ (oracle/make_golden_struct.py --dtype fp16): STRUCTURED checkpoints at BASELINE geometry -- the 4-layer
  slices with llama2-7B's / llama3-8B's / llama3.2-1B's projection, vocabulary and RoPE geometry and llama2-7B at FULL size (32
  layers, exit_layer 8, 6 speculations) -- on which every decision of the reference's fp16 run has a wide top-2 margin: ids, the
  per-step (num_drafts, num_matches) trace, the draft tokens and the autoregressive ids are asserted EQUAL, no tie branch, and the
  engine's fp16 logits along the reference trajectory must be as close to the reference's FP32 logits of the same weights as the
  reference's own fp16 run is (<= 1.1 x its rms error: the gate the bf16 suite uses, tests/test_gpu_struct_parity.py)."""
import json
import os

import pytest
import torch

from conftest import build_case_model, cast_parameters

pytestmark = pytest.mark.gpu

FP16_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp16")
NAMES = sorted(f[:-5] for f in os.listdir(FP16_DIR) if f.endswith(".json"))
TIE_TOL = 0.02          # fp16 logits of |value| < 8 have an ulp <= 0.008


@pytest.mark.parametrize("name", NAMES)
def test_fp16_engine_matches_reference_and_itself(gpu_device, name):
    from layerskip_amd import GenerationConfig
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    rec = json.load(open(os.path.join(FP16_DIR, name + ".json")))
    model = cast_parameters(build_case_model(rec), torch.float16).to(gpu_device)
    kw = dict(max_steps=rec["max_steps"], num_speculations=rec["num_speculations"], sample=False)
    spec = HipSelfSpeculativeGenerationStrategy().generate_token_ids(
        model, rec["prompt"], rec["eos_token_ids"],
        GenerationConfig(generation_strategy="self_speculative", exit_layer=rec["exit_layer"], **kw))
    ar = HipAutoRegressiveGenerationStrategy().generate_token_ids(          # exit_layer=-1: the full model (ARG:44-51)
        model, rec["prompt"], rec["eos_token_ids"], GenerationConfig(generation_strategy="autoregressive", exit_layer=-1, **kw))
    assert spec.predicted_tokens == ar.predicted_tokens            # row invariance holds for any element type
    gold = rec["fp16"]
    for i, (a, b) in enumerate(zip(spec.predicted_tokens, gold["spec_tokens"])):
        if a != b:
            assert gold["spec_margins"][i] < TIE_TOL, (name, i, a, b, gold["spec_margins"][i])
            break
    else:
        assert len(spec.predicted_tokens) == len(gold["spec_tokens"])


# ------------------------------------------------------------------------------------------------ BASELINE geometry, structured checkpoints
STRUCT16_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "struct_fp16")
STRUCT16 = sorted(f[:-5] for f in os.listdir(STRUCT16_DIR) if f.endswith(".json")) if os.path.isdir(STRUCT16_DIR) else []


def _fp16_ulp(value):
    import math
    return 2.0 ** (math.floor(math.log2(max(abs(value), 1.0))) - 10)     # never finer than at |1.0| (logits are sums of cancelling terms)


def test_the_baseline_geometries_have_fp16_fixtures():
    assert {"slice7b", "slice8b", "full7b"} <= set(STRUCT16)


@pytest.mark.parametrize("name", STRUCT16)
def test_fp16_struct_tokens_trace_and_logits_equal_reference(gpu_device, name):
    from conftest import build_struct_model
    from layerskip_amd import GenerationConfig
    from layerskip_amd.engine import BUF_BULK, get_engine
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    rec = json.load(open(os.path.join(STRUCT16_DIR, name + ".json")))
    gold = rec["fp16"]
    assert gold["min_margin_ulp"] >= 16 and gold["min_draft_margin_ulp"] >= 16        # (bf16 ulps: >= 128 fp16 ulps -- no tie branch below)
    model = cast_parameters(build_struct_model(rec, "cpu"), torch.float16).to(gpu_device)      # what `torch_dtype=torch.float16` does to a bf16 checkpoint
    kw = dict(max_steps=rec["max_steps"], num_speculations=rec["num_speculations"], sample=False)
    cfg_spec = GenerationConfig(generation_strategy="self_speculative", exit_layer=rec["exit_layer"], **kw)
    fused = HipSelfSpeculativeGenerationStrategy()
    res = fused.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg_spec)
    assert res.predicted_tokens == gold["spec_tokens"]
    assert [list(s) for s in fused.last_steps] == gold["steps"]
    assert res.acceptance_rate == gold["acceptance_rate"]
    eng = get_engine(model)
    assert eng.dtype == torch.float16
    drafts, inner = [], eng.spec_step

    def spy(*a, **k):
        r = inner(*a, **k)
        drafts.append(list(r.draft_tokens[: r.num_drafts]))
        return r

    eng.spec_step = spy
    try:
        res2 = HipSelfSpeculativeGenerationStrategy(fused_generate=False).generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg_spec)
    finally:
        del eng.spec_step
    assert res2.predicted_tokens == gold["spec_tokens"] and drafts == gold["step_drafts"]
    ar = HipAutoRegressiveGenerationStrategy().generate_token_ids(model, rec["prompt"], rec["eos_token_ids"],
                                                                GenerationConfig(generation_strategy="autoregressive", exit_layer=-1, **kw))
    assert ar.predicted_tokens == gold["ar_tokens"]
    # ---- logits along the reference trajectory: engine-fp16 and reference-fp16 against the reference's fp32 logits (fp16 ulps) ----
    seq = rec["prompt"] + gold["spec_tokens"]
    n = len(seq)
    eng.ensure_capacity(n + 4, n)
    e2 = r2 = 0.0
    e_max = r_max = 0.0
    cnt = within = 0
    for key, layer_end in (("logits", eng.num_layers), ("early_logits", rec["exit_layer"])):
        eng.reset()
        eng.embed_rows(seq, BUF_BULK, 0)
        eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, layer_end)
        for row in gold[key]:
            buf = torch.empty(1, eng.vocab, dtype=torch.float32, device=gpu_device)
            eng.run_head(BUF_BULK, row["row"], 1, logits=buf, want_tokens=False)
            mine = buf[0, row["idx"]].cpu().tolist()
            assert int(buf[0].argmax()) == max(zip(row["val"], row["idx"]))[1]
            for a, b, x in zip(mine, row["val"], row["val_fp32"]):
                u = _fp16_ulp(x)
                ee, er = abs(a - x) / u, abs(b - x) / u
                e2, r2, cnt = e2 + ee * ee, r2 + er * er, cnt + 1
                e_max, r_max = max(e_max, ee), max(r_max, er)
                within += int(abs(a - b) / u <= 1.0)
    eng.reset()
    strict = model.config.hidden_size >= 2048
    # 1.1 x: the bf16 suite's figure (tests/test_gpu_struct_parity.py), at every BASELINE geometry.  Round 5 had widened the full-size bound
    # to 1.15 x after measuring 0.492 vs 0.447 on the 640 entries then recorded (20 rows x 32 entries); round 6 looked for a rounding
    # point behind it (tools/diag_fp16.py, profiles/r06_diag_fp16_full7b.json) and found none: layer by layer the engine's error on the SAME
    # input equals the reference's fp16 run's (ratio 0.998-1.001 at all 32 layers, every stage >= 99 % bit-equal, attention closer to fp64),
    # and over ALL 3.6 M logits of the 112 rows the two are 0.39794 vs 0.39765 fp16 ulp rms.  The 32 entries of a row share that row's
    # hidden-state error, so 20 rows were ~20 samples of the ratio, not 640; the full-size fixture now records every generated position
    # (52 rows, oracle/make_golden_struct.py) and the common bound holds.
    tol = 1.1 if strict else 1.25
    msg = (f"{name}: vs the reference's fp32 logits, in fp16 ulp: engine rms {(e2 / cnt) ** 0.5:.3f} max {e_max:.2f}, reference-fp16 rms "
           f"{(r2 / cnt) ** 0.5:.3f} max {r_max:.2f}; {within}/{cnt} within 1 fp16 ulp of the reference's fp16 logits")
    print(msg)
    assert e2 <= tol * tol * r2 + 1e-9, msg
    # the worst single entry: within 1.5 x the reference's worst + 1 ulp -- but never asked to be below 4 fp16 ulp (0.004 at |logit| <= 1):
    # the maximum over ~1 500 entries is one rounding flip of one hidden element in front of an lm_head row of unit scale, and which of the
    # two fp16 computations draws it is chance (slice8b after round 6's regeneration: 1503 / 1504 entries within one ulp of the reference's
    # own fp16 logits, rms 0.262 vs 0.254, the one other entry 3.38 ulp from the fp32 value against a reference maximum of 1.54)
    assert e_max <= max(1.5 * r_max + 1.0, 4.0), msg
    if strict:
        assert within >= 0.97 * cnt, msg
