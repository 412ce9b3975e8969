"""The fp16 build of the engine (liblayerskip_hip_f16.so: the same kernels with elem_t = _Float16 and
v_mfma_f32_16x16x32_f16) against fixtures recorded from the unmodified reference in fp16 (tests/golden/fp16/).
Seen green on an MI355X at the start of round 2; fp16 models are accepted without a switch since."""
import json
import os

import pytest
import torch

from conftest import build_case_model

pytestmark = pytest.mark.gpu

FP16_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp16")
NAMES = sorted(f[:-5] for f in os.listdir(FP16_DIR) if f.endswith(".json"))
TIE_TOL = 0.02          # fp16 logits of |value| < 8 have an ulp <= 0.008


@pytest.mark.parametrize("name", NAMES)
def test_fp16_engine_matches_reference_and_itself(gpu_device, name):
    from layerskip_amd import GenerationConfig
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    rec = json.load(open(os.path.join(FP16_DIR, name + ".json")))
    model = build_case_model(rec).to(torch.float16).to(gpu_device)
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
