"""CPU: the structured fixtures (tests/golden/struct/) against the restatement and against the checkpoint's own token
program.  The big shapes are covered on the GPU box only (tests/test_gpu_struct_parity.py); here the tiny ones."""
import os

import pytest
import torch

from conftest import build_struct_model, load_struct, struct_names
from oracle import llama_oracle as lo

TINY = [n for n in struct_names() if n.startswith("tiny")]


def test_every_fixture_records_healthy_margins():
    """What makes these fixtures decisive: the reference's own bf16 run never decided by less than 16 bf16 ulp."""
    names = struct_names()
    assert {"full1b", "full7b", "full8b", "full7b_512", "full13b", "slice70b", "slice7b", "slice8b", "slice13b", "slice1b", "tiny_mha", "tiny_gqa", "tiny_d64"} <= set(names)
    for name in names:
        rec = load_struct(name)
        b = rec["bf16"]
        assert b["min_margin_ulp"] >= 16 and b["min_draft_margin_ulp"] >= 16, name
        assert b["spec_equals_ar"], name
        assert sum(n + 1 for _, n in b["steps"]) >= len(b["spec_tokens"])
        assert 0.15 < b["acceptance_rate"] < 0.9, (name, b["acceptance_rate"])      # drafts are accepted AND rejected


@pytest.mark.parametrize("name", TINY)
def test_restatement_reproduces_struct_fixture(name):
    from layerskip_amd import synthetic
    rec = load_struct(name)
    model = build_struct_model(rec)
    om = lo.OracleModel.from_hf(model)
    with torch.inference_mode():
        spec = lo.self_speculative_generate(om, rec["prompt"], rec["eos_token_ids"], rec["max_steps"], rec["exit_layer"],
                                            rec["num_speculations"])
        ar = lo.autoregressive_generate(om, rec["prompt"], rec["eos_token_ids"], rec["max_steps"])
    gold = rec["bf16"]
    assert spec.predicted_tokens == gold["spec_tokens"]
    assert ar.predicted_tokens == gold["ar_tokens"]
    assert [[s.num_drafts, s.num_matches] for s in spec.steps] == gold["steps"]
    assert [s.draft_tokens for s in spec.steps] == gold["step_drafts"]
    assert spec.acceptance_rate == gold["acceptance_rate"]
    # the checkpoint does what it was built to do
    if rec["eos_token_ids"] == [model.config.vocab_size]:
        t, want = rec["prompt"][-1], []
        for _ in range(len(gold["spec_tokens"])):
            t = synthetic.struct_next_token(model.struct_program, t, True)
            want.append(t)
        assert want == gold["spec_tokens"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/self_speculation"), reason="reference tree not mounted")
def test_struct_fixture_is_what_the_unmodified_reference_produces():
    """Re-runs the live reference on one structured checkpoint (build container only)."""
    import copy
    from oracle import ref_shim
    ref = ref_shim.load_reference()
    rec = load_struct("tiny_gqa")
    model = ref_shim.patch_model(copy.deepcopy(build_struct_model(rec)))
    cfg = ref.generator_base.GenerationConfig(max_steps=rec["max_steps"], exit_layer=rec["exit_layer"],
                                              num_speculations=rec["num_speculations"], sample=False)
    with torch.inference_mode():
        want = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy().generate_token_ids(
            model=model, input_ids=list(rec["prompt"]), eos_token_ids=rec["eos_token_ids"], generation_config=cfg)
    assert want.predicted_tokens == rec["bf16"]["spec_tokens"]
    assert want.acceptance_rate == rec["bf16"]["acceptance_rate"]
