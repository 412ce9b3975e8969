"""Sampling path (`sample=True`, SSG:191-199 + LMU:124-131) "parity in distribution" (SURVEY.md 8f N2).

The strategies' materialised-logits path is driven on CPU through the oracle-backed stage backend and
compared with the UNMODIFIED reference's sampling run on the same weights.  The two consume random numbers in
different orders, so the comparison is statistical: mean acceptance rate and the distribution of the first
few generated tokens over many seeds.  Needs the reference tree (build container only)."""
import collections
import os

import pytest
import torch

from conftest import build_case_model, load_golden

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/self_speculation"), reason="reference tree not mounted")

N_RUNS = 160


def _tv(a, b):
    keys = set(a) | set(b)
    na, nb = sum(a.values()), sum(b.values())
    return 0.5 * sum(abs(a.get(k, 0) / na - b.get(k, 0) / nb) for k in keys)


@pytest.mark.parametrize("device_sampling", [False, True], ids=["host-sampling", "device-algorithm"])
def test_sampled_speculation_matches_reference_in_distribution(monkeypatch, device_sampling):
    """host-sampling: the strategies' default `sample=True` path (logits materialised, torch draws).
    device-algorithm: the orchestration of lsk_spec_step_sampled with the oracle's draw-for-draw model of the two
    sampling kernels (key-space thresholds, Philox + Gumbel-max, rejection step) -- the algorithm the device runs."""
    import copy
    from cpu_stage_backend import CpuStageBackend
    from layerskip_amd import GenerationConfig, hip_strategies
    from oracle import ref_shim
    ref = ref_shim.load_reference()
    rec = load_golden("tiny_mha_s1")
    base = build_case_model(rec).float()
    prompt, eos = rec["prompt"], rec["eos_token_ids"]
    E, S, T = rec["exit_layer"], 4, 10
    kw = dict(max_steps=T, exit_layer=E, num_speculations=S, sample=True, temperature=0.12, top_k=0, top_p=0.9)

    ref_model = ref_shim.patch_model(copy.deepcopy(base))
    ref_strat = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy()
    ref_cfg = ref.generator_base.GenerationConfig(**kw)
    ref_acc, ref_first, ref_len = [], collections.Counter(), []
    for i in range(N_RUNS):
        torch.manual_seed(1000 + i)
        with torch.inference_mode():
            r = ref_strat.generate_token_ids(model=ref_model, input_ids=list(prompt), eos_token_ids=list(eos),
                                             generation_config=ref_cfg)
        ref_acc.append(r.acceptance_rate)
        ref_first[r.predicted_tokens[0]] += 1
        ref_len.append(len(r.predicted_tokens))

    backend = CpuStageBackend(base)
    monkeypatch.setattr(hip_strategies, "get_engine", lambda model, **k: backend)
    mine = hip_strategies.HipSelfSpeculativeGenerationStrategy(device_sampling=device_sampling)
    if not device_sampling:
        monkeypatch.delattr(CpuStageBackend, "spec_step_sampled")      # the host path must not depend on it
    my_acc, my_first, my_len = [], collections.Counter(), []
    for i in range(N_RUNS):
        torch.manual_seed(5000 + i)
        with torch.inference_mode():
            r = mine.generate_token_ids(base, list(prompt), list(eos), GenerationConfig(**kw))
        assert 0 < len(r.predicted_tokens) <= T
        my_acc.append(r.acceptance_rate)
        my_first[r.predicted_tokens[0]] += 1
        my_len.append(len(r.predicted_tokens))

    ma, mb = sum(ref_acc) / N_RUNS, sum(my_acc) / N_RUNS
    sd = (sum((x - ma) ** 2 for x in ref_acc) / N_RUNS) ** 0.5
    # two means of N_RUNS samples each: 4 standard errors of the difference
    assert abs(ma - mb) < 4 * sd * (2 / N_RUNS) ** 0.5 + 0.01, (ma, mb, sd)

    # First token = one sample of the warped full-model distribution at the prompt (temperature, then top-p,
    # LMU:124-131).  That distribution is known exactly, so bucket the vocabulary into deciles of its
    # cumulative mass: both samplers must stay inside the nucleus and fill the deciles as that distribution says.
    with torch.inference_mode():
        logits = base(torch.tensor([prompt])).logits[0, -1].float() / kw["temperature"]
    probs = torch.softmax(logits, -1)
    order = torch.argsort(probs, descending=True)
    cum = torch.cumsum(probs[order], 0)
    keep = int((cum < kw["top_p"]).sum()) + 1                 # smallest prefix whose mass reaches top_p
    nucleus = order[:keep]
    p_n = probs[nucleus] / probs[nucleus].sum()
    edges = torch.cumsum(p_n, 0)
    decile = {int(t): min(9, int(float(edges[i] - p_n[i] / 2) * 10)) for i, t in enumerate(nucleus.tolist())}
    for name, hist in (("reference", ref_first), ("engine path", my_first)):
        outside = [t for t in hist if t not in decile]
        assert not outside, (name, "sampled outside the top-p nucleus", outside[:5])
    ref_dec, my_dec = collections.Counter(), collections.Counter()
    for t, c in ref_first.items():
        ref_dec[decile[t]] += c
    for t, c in my_first.items():
        my_dec[decile[t]] += c
    expect = collections.Counter()
    for i, t in enumerate(nucleus.tolist()):
        expect[decile[t]] += float(p_n[i]) * N_RUNS
    assert _tv(my_dec, expect) < 0.2, (my_dec, _tv(my_dec, expect))
    assert _tv(ref_dec, expect) < 0.2, (ref_dec, _tv(ref_dec, expect))
    # the number of emitted tokens per run is the other observable both share
    assert abs(sum(ref_len) - sum(my_len)) / N_RUNS < 0.5
