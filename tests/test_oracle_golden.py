"""The CPU restatement (oracle/llama_oracle.py) against the fixtures recorded from the UNMODIFIED
reference (oracle/make_golden.py).  In the build container -- where the fixtures were made -- the
match is bit-exact; on another host CPU the bf16 GEMM kernels may round differently, so a mismatch is
tolerated only at a near-tie of the recorded logits."""
import os

import pytest
import torch

from conftest import build_case_model, golden_names, load_golden
from oracle import llama_oracle as lo

SMALL = [n for n in golden_names() if not (n.startswith("small_wide") or n.startswith("slice7b"))]


def _run(rec, dtype):
    model = build_case_model(rec)
    om = lo.OracleModel.from_hf(model, dtype=dtype)
    with torch.inference_mode():
        spec = lo.self_speculative_generate(om, rec["prompt"], rec["eos_token_ids"], rec["max_steps"], rec["exit_layer"],
                                            rec["num_speculations"])
        ar = lo.autoregressive_generate(om, rec["prompt"], rec["eos_token_ids"], rec["max_steps"])
    return spec, ar


def _check(got, want, margins, what):
    for i, (a, b) in enumerate(zip(got, want)):
        if a != b:
            assert margins[i] < 0.02, f"{what}: token {i} differs at margin {margins[i]}"
            return
    assert len(got) == len(want), what


@pytest.mark.parametrize("name", SMALL)
def test_restatement_reproduces_reference_fp32(name):
    rec = load_golden(name)
    spec, ar = _run(rec, torch.float32)
    gold = rec["fp32"]
    _check(spec.predicted_tokens, gold["spec_tokens"], gold["spec_margins"] + [0.0], name + " spec")
    _check(ar.predicted_tokens, gold["ar_tokens"], gold["ar_margins"] + [0.0], name + " ar")
    if spec.predicted_tokens == gold["spec_tokens"]:
        assert [[s.num_drafts, s.num_matches] for s in spec.steps] == gold["steps"]
        assert spec.acceptance_rate == pytest.approx(gold["acceptance_rate"], abs=1e-12)


@pytest.mark.parametrize("name", SMALL[:3])
def test_restatement_reproduces_reference_bf16(name):
    rec = load_golden(name)
    spec, ar = _run(rec, torch.bfloat16)
    gold = rec["bf16"]
    _check(spec.predicted_tokens, gold["spec_tokens"], gold["spec_margins"] + [0.0], name + " spec")
    _check(ar.predicted_tokens, gold["ar_tokens"], gold["ar_margins"] + [0.0], name + " ar")


FP16_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp16")


@pytest.mark.parametrize("name", sorted(f[:-5] for f in os.listdir(FP16_DIR) if f.endswith(".json")))
def test_restatement_reproduces_reference_fp16(name):
    """fp16 (generate.py:63's dtype; the fp16 build of the engine is checked against these fixtures)."""
    import json
    rec = json.load(open(os.path.join(FP16_DIR, name + ".json")))
    spec, ar = _run(rec, torch.float16)
    gold = rec["fp16"]
    _check(spec.predicted_tokens, gold["spec_tokens"], gold["spec_margins"] + [0.0], name + " spec")
    _check(ar.predicted_tokens, gold["ar_tokens"], gold["ar_margins"] + [0.0], name + " ar")
    assert gold["spec_equals_ar"]          # in fp16 the reference is self-consistent on these cases


def test_reference_helper_semantics():
    """What the reference's own unit tests pin (tests/test_llama_model_utils.py:14-69), on the restatement."""
    m = lo.make_causal_mask(5, torch.float32, 3)
    assert tuple(m.shape) == (1, 1, 5, 8) and bool((m <= 0).all())
    assert float(m[0, 0, 0, 4]) == torch.finfo(torch.float32).min and float(m[0, 0, 4, 7]) == 0.0
    d = lo.decoder_mask(1, 9, torch.bfloat16, 8)
    assert tuple(d.shape) == (1, 1, 1, 9) and bool((d == 0).all())
    logits = torch.tensor([[[1.0, 2.0, 3.0]]])
    assert lo.decode_next_token_greedy(logits, token_idx=-1).tolist() == [2]
    tie = torch.tensor([[[0.5, 7.0, 7.0, 1.0]]])
    assert lo.decode_next_token_greedy(tie).tolist() == [[1]]          # first index wins ties


@pytest.mark.skipif(not os.path.isdir("/root/reference/self_speculation"), reason="reference tree not mounted")
def test_restatement_is_bit_identical_to_unmodified_reference():
    """Re-pins the oracle against the live reference (build container only)."""
    import copy
    from oracle import ref_shim
    ref = ref_shim.load_reference()
    rec = load_golden("tiny_mha_s1")
    base = build_case_model(rec)
    for dtype in (torch.float32, torch.bfloat16):
        model = ref_shim.patch_model(ref_shim.cast_parameters(copy.deepcopy(base), dtype))
        cfg = ref.generator_base.GenerationConfig(max_steps=20, exit_layer=rec["exit_layer"],
                                                  num_speculations=rec["num_speculations"], sample=False)
        with torch.inference_mode():
            want = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy().generate_token_ids(
                model=model, input_ids=list(rec["prompt"]), eos_token_ids=rec["eos_token_ids"], generation_config=cfg)
            got = lo.self_speculative_generate(lo.OracleModel.from_hf(model), rec["prompt"], rec["eos_token_ids"], 20,
                                               rec["exit_layer"], rec["num_speculations"])
            seq = rec["prompt"] + want.predicted_tokens
            a = ref.llama_model_utils.forward(model, torch.tensor([seq]), None).logits[0]
            b = lo.teacher_forced_logits(lo.OracleModel.from_hf(model), seq)
        assert got.predicted_tokens == want.predicted_tokens
        assert got.acceptance_rate == want.acceptance_rate
        assert torch.equal(a, b)
