"""Greedy parity with the UNMODIFIED reference, token for token, with NO tie branch.

Fixtures: tests/golden/struct/*.json (oracle/make_golden_struct.py) -- the reference's own bf16 run on structured
synthetic checkpoints in which every decision (each draft-head argmax, each verify argmax) has a top-2 margin of at
least 16 bf16 ulp.  Shapes: tiny MHA / GQA / d=64 (+ llama3 RoPE scaling, tied embeddings), 4-layer slices with the
exact projection / vocabulary geometry of llama2-7B, llama3-8B (GQA, V = 128 256, theta = 5e5), llama2-13B (H = 5120)
and llama3.2-1B (d = 64, tied), and llama2-7B at FULL size (32 layers, exit_layer 8, 6 speculations).

Asserted: output ids == reference ids; the per-step (num_drafts, num_matches) trace and the draft tokens == reference;
acceptance rate equal; the autoregressive strategy's ids == reference; logits along the reference trajectory within
one bf16 ulp of the reference's bf16 logits on >= 99 % of the recorded entries and within two everywhere."""
import pytest
import torch

from conftest import bf16_ulp, build_struct_model, load_struct, struct_names

pytestmark = pytest.mark.gpu

_MODELS = {}


def _model(rec, device):
    key = (rec["shape"], rec["seed"], rec["exit_layer"], tuple(sorted(rec.get("knobs", {}).items())))
    if key not in _MODELS:
        _MODELS.clear()            # one model resident at a time
        torch.cuda.empty_cache()
        _MODELS[key] = build_struct_model(rec, device)
    return _MODELS[key]


def _cfg(rec, strategy):
    from layerskip_amd import GenerationConfig
    return GenerationConfig(max_steps=rec["max_steps"], exit_layer=rec["exit_layer"] if strategy == "self_speculative" else -1,
                            num_speculations=rec["num_speculations"], sample=False, generation_strategy=strategy)


@pytest.mark.parametrize("name", struct_names())
def test_tokens_and_trace_equal_reference(gpu_device, name):
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    rec = load_struct(name)
    gold = rec["bf16"]
    model = _model(rec, gpu_device)
    # ---- fused path: the whole generation as one C-ABI call ----
    spec = HipSelfSpeculativeGenerationStrategy()
    res = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, "self_speculative"))
    assert res.predicted_tokens == gold["spec_tokens"]
    assert [list(s) for s in spec.last_steps] == gold["steps"]
    assert res.acceptance_rate == gold["acceptance_rate"]
    # ---- one C-ABI call per step (single_step_speculation): the draft tokens too ----
    stepwise = HipSelfSpeculativeGenerationStrategy(fused_generate=False)
    from layerskip_amd.engine import get_engine
    eng = get_engine(model)
    drafts = []
    inner = eng.spec_step

    def spy(*a, **kw):
        r = inner(*a, **kw)
        drafts.append(list(r.draft_tokens[: r.num_drafts]))
        return r

    eng.spec_step = spy
    try:
        res2 = stepwise.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, "self_speculative"))
    finally:
        del eng.spec_step
    assert res2.predicted_tokens == gold["spec_tokens"]
    assert res2.acceptance_rate == gold["acceptance_rate"]
    assert drafts == gold["step_drafts"]
    # ---- autoregressive strategy ----
    ar = HipAutoRegressiveGenerationStrategy().generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, "autoregressive"))
    assert ar.predicted_tokens == gold["ar_tokens"]


def _ulp_report(mine, ref_vals):
    """(#entries, #within 1 ulp, max error in ulp); ulp of the REFERENCE value, never finer than at |1.0|."""
    n = within = 0
    worst = 0.0
    for a, b in zip(mine, ref_vals):
        u = bf16_ulp(max(abs(b), 1.0))
        e = abs(a - b) / u
        n += 1
        within += int(e <= 1.0)
        worst = max(worst, e)
    return n, within, worst


@pytest.mark.parametrize("name", [n for n in struct_names() if not n.endswith("_eos")])
def test_teacher_forced_logits_within_one_ulp(gpu_device, name):
    """Engine logits along the REFERENCE trajectory (prompt + reference output) vs the reference's bf16 logits: full
    depth (forward, LMU:155-209) and early exit (forward_early, LMU:213-276)."""
    from layerskip_amd.engine import BUF_BULK, get_engine
    rec = load_struct(name)
    gold = rec["bf16"]
    model = _model(rec, gpu_device)
    eng = get_engine(model)
    seq = rec["prompt"] + gold["spec_tokens"]
    n = len(seq)
    eng.ensure_capacity(n + 4, n)
    total = ok = 0
    worst = 0.0
    for key, layer_end in (("logits", eng.num_layers), ("early_logits", rec["exit_layer"])):
        eng.reset()
        eng.embed_rows(seq, BUF_BULK, 0)
        eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, layer_end)
        for row in gold[key]:
            buf = torch.empty(1, eng.vocab, dtype=torch.float32, device=gpu_device)
            eng.run_head(BUF_BULK, row["row"], 1, logits=buf, want_tokens=False)
            mine = buf[0, row["idx"]].cpu().tolist()
            a, b, w = _ulp_report(mine, row["val"])
            total, ok, worst = total + a, ok + b, max(worst, w)
            # and the decision itself
            assert int(buf[0].argmax()) == max(zip(row["val"], row["idx"]))[1]
    eng.reset()
    assert ok >= 0.99 * total, f"{name}: {ok}/{total} recorded logits within 1 bf16 ulp (worst {worst:.2f} ulp)"
    assert worst <= 2.0, f"{name}: worst recorded logit is {worst:.2f} bf16 ulp away from the reference"


def test_prefill_kernels_follow_the_reference_too(gpu_device):
    """The same ulp gate with the prompt rows going through the MFMA-tiled prefill kernels (run_bulk) instead of 16-row
    passes of the decode kernels: 300-token prompt."""
    from layerskip_amd.engine import BUF_BULK, get_engine
    rec = load_struct("tiny_gqa_long")
    gold = rec["bf16"]
    model = _model(rec, gpu_device)
    eng = get_engine(model)
    seq = rec["prompt"] + gold["spec_tokens"]
    n = len(seq)
    eng.ensure_capacity(n + 4, n)
    eng.reset()
    eng.embed_rows(seq, BUF_BULK, 0)
    eng.run_bulk(n, 0, eng.num_layers)
    total = ok = 0
    worst = 0.0
    for row in gold["logits"]:
        buf = torch.empty(1, eng.vocab, dtype=torch.float32, device=gpu_device)
        eng.run_head(BUF_BULK, row["row"], 1, logits=buf, want_tokens=False)
        a, b, w = _ulp_report(buf[0, row["idx"]].cpu().tolist(), row["val"])
        total, ok, worst = total + a, ok + b, max(worst, w)
    eng.reset()
    assert ok >= 0.99 * total and worst <= 2.0, (ok, total, worst)
