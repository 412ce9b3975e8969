"""End-to-end parity of the HIP path against the reference-pinned golden fixtures (tests/golden/).

Token ids are compared with the bf16 run of the UNMODIFIED reference recorded by
oracle/make_golden.py.  bf16 greedy decoding is only defined up to near-ties: where the reference's
own top-2 logit margin is below ``TIE_TOL`` a different (equally valid) argmax is accepted and the
comparison continues teacher-forced.  A mismatch at a healthy margin is a failure.
"""
import pytest
import torch

from conftest import build_case_model, golden_names, load_golden

pytestmark = pytest.mark.gpu

ACC_TOL = 0.15          # acceptance rate: a handful of near-tie draft tokens may flip (few tens of drafts per case)
TIE_TOL = 0.05          # logits units; bf16 ulp at |logit| ~ 2..4 is 0.016..0.031
LOGIT_ATOL = 0.08       # teacher-forced logits, engine bf16 vs reference bf16 (both carry bf16 noise)


def _strategies():
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    return HipSelfSpeculativeGenerationStrategy(), HipAutoRegressiveGenerationStrategy()


def _config(rec, strategy):
    from layerskip_amd import GenerationConfig
    return GenerationConfig(max_steps=rec["max_steps"], exit_layer=rec["exit_layer"],
                            num_speculations=rec["num_speculations"], sample=False, generation_strategy=strategy)


_MODELS = {}


def _model(rec, device):
    key = (rec["shape"], rec["seed"], rec["late_damping"], rec["exit_layer"])
    if key not in _MODELS:
        _MODELS.clear()            # one model resident at a time
        _MODELS[key] = build_case_model(rec, device)
    return _MODELS[key]


def _first_mismatch(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


# The two random-init fixtures at BASELINE / wide geometry are superseded on the GPU by fixtures that assert MORE on the same geometry -- the
# structured family (tests/golden/struct/slice7b, small_wide: ids, traces, draft tokens, logits with no tie branch) and the full-size
# random-init family (tests/test_gpu_rand7b_parity.py) -- and cost ~27 s of the driver's GPU suite; the CPU suite still pins them to the
# oracle (tests/test_oracle_golden.py).
SUPERSEDED = ("slice7b_s0", "small_wide_s0")
NAMES = [n for n in golden_names() if n not in SUPERSEDED]


@pytest.mark.parametrize("name", NAMES)
def test_spec_tokens_match_reference(gpu_device, name):
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    spec, _ = _strategies()
    res = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _config(rec, "self_speculative"))
    gold = rec["bf16"]
    i = _first_mismatch(res.predicted_tokens, gold["spec_tokens"])
    if i is None:
        # identical output; the acceptance COUNT also depends on the draft head's own argmaxes, which have
        # near-ties of their own (not recorded in the fixtures): equal up to a few flipped drafts
        assert abs(res.acceptance_rate - gold["acceptance_rate"]) < ACC_TOL
        return
    margins = gold["spec_margins"]
    assert i < len(margins), f"{name}: length differs without a token mismatch"
    assert margins[i] < TIE_TOL, (f"{name}: token {i} differs ({res.predicted_tokens[i]} vs {gold['spec_tokens'][i]}) "
                                  f"at a healthy reference margin {margins[i]:.4f}")


@pytest.mark.parametrize("name", NAMES)
def test_spec_equals_autoregressive_in_engine(gpu_device, name):
    """The reference's own correctness criterion (correctness.py:82-88), bit-exact here by construction."""
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    spec, ar = _strategies()
    a = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _config(rec, "self_speculative"))
    cfg = _config(rec, "autoregressive")
    cfg.exit_layer = -1
    b = ar.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg)
    assert a.predicted_tokens == b.predicted_tokens
    assert len(a.predicted_tokens) <= rec["max_steps"]


@pytest.mark.parametrize("name", [n for n in NAMES if not n.endswith("_eos")])
def test_teacher_forced_logits(gpu_device, name):
    """Engine logits along the REFERENCE trajectory vs the recorded reference logits rows."""
    from layerskip_amd.engine import BUF_BULK, BUF_STEP, get_engine
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    eng = get_engine(model)
    gold = rec["bf16"]
    seq = rec["prompt"] + gold["spec_tokens"]
    eng.ensure_capacity(len(seq) + 4, len(seq))
    eng.reset()
    n = len(seq)
    eng.embed_rows(seq, BUF_BULK, 0)
    eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, eng.num_layers)
    logits = torch.empty(n, eng.vocab, dtype=torch.float32, device=gpu_device)
    for r0 in range(0, n, 16):
        m = min(16, n - r0)
        eng.run_head(BUF_BULK, r0, m, logits=logits[r0:r0 + m], want_tokens=False)
    torch.cuda.synchronize()
    worst = 0.0
    for row in gold["logits_topk"]:
        mine = logits[row["row"], row["idx"]].cpu()
        ref = torch.tensor(row["val"])
        worst = max(worst, (mine - ref).abs().max().item())
    assert worst <= LOGIT_ATOL, f"{name}: teacher-forced logits differ by {worst}"
    # argmax along the reference trajectory: every healthy-margin position must agree
    P = len(rec["prompt"])
    pred = logits.argmax(-1).cpu().tolist()
    for i, tok in enumerate(gold["spec_tokens"]):
        if gold["spec_margins"][i] >= TIE_TOL:
            assert pred[P - 1 + i] == tok, f"{name}: position {i} margin {gold['spec_margins'][i]}"


def test_step_api_matches_reference_shape(gpu_device):
    """single_step_speculation keeps the reference's signature / 5-tuple (SSG:102-120, :223-229) and the
    invariants the reference's own test pins (tests/test_self_speculation_generator.py:37-66)."""
    rec = load_golden("tiny_mha_s0")
    model = _model(rec, gpu_device)
    spec, _ = _strategies()
    ids = rec["prompt"]
    out = spec.single_step_speculation(
        model=model, input_ids_list=ids, input_ids=torch.tensor([ids]), output_ids=[], num_speculations=1,
        past_key_values=None, exit_layer=rec["exit_layer"], eos_token_ids=[rec["eos_token_ids"][0]], calls=0,
        sample=False, temperature=0.7, top_k=50, top_p=0.95)
    next_ids, output_ids, past, n_matches, n_spec = out
    assert n_spec == 1 and 0 <= n_matches <= n_spec
    assert tuple(next_ids.shape) == (1, 1)
    assert len(output_ids) == n_matches + 1
    assert past.length == len(ids) + len(output_ids) - 1


@pytest.mark.parametrize("name", ["tiny_mha_s1", "tiny_gqa_long"])
def test_slow_path_equals_fast_path(gpu_device, name):
    """A no-op logits processor forces the materialised-logits path; greedy ids must not change
    (short prompt: 16-row prefill passes; 300-token prompt: the MFMA prefill kernels in both paths)."""
    import transformers
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    spec, ar = _strategies()
    fast = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _config(rec, "self_speculative"))

    class Identity(transformers.LogitsProcessor):
        def __call__(self, input_ids, scores):
            return scores

    procs = transformers.LogitsProcessorList([Identity()])
    slow = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _config(rec, "self_speculative"),
                                   logits_processors=procs)
    assert slow.predicted_tokens == fast.predicted_tokens
    assert slow.acceptance_rate == fast.acceptance_rate
    cfg = _config(rec, "autoregressive")
    cfg.exit_layer = -1
    cfg.max_steps = 12
    a = ar.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg)
    b = ar.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg, logits_processors=procs)
    assert a.predicted_tokens == b.predicted_tokens


def test_sampling_path_runs(gpu_device):
    """sample=True keeps the API contract (SSG:191-199); distributional parity is a later row."""
    from layerskip_amd import GenerationConfig
    rec = load_golden("tiny_mha_s0")
    model = _model(rec, gpu_device)
    spec, _ = _strategies()
    torch.manual_seed(0)
    cfg = GenerationConfig(max_steps=12, exit_layer=rec["exit_layer"], num_speculations=3, sample=True,
                           temperature=0.8, top_k=0, top_p=0.9, generation_strategy="self_speculative")
    res = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg)
    assert 0 < len(res.predicted_tokens) <= 12
    assert 0.0 <= res.acceptance_rate <= 1.0


def test_pipeline_decoder_single_rank_equals_fused_path(gpu_device):
    """layerskip_amd/pipeline.py drives HipEngine through its building-block API; with one rank (no
    communication) it must reproduce the fused lsk_spec_step path token for token."""
    from layerskip_amd.engine import get_engine
    from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
    rec = load_golden("tiny_gqa_s0")
    model = _model(rec, gpu_device)
    spec, _ = _strategies()
    fast = spec.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _config(rec, "self_speculative"))
    eng = get_engine(model)
    dec = PipelineSpeculativeDecoder(eng, 0, 1, plan_partition(eng.num_layers, rec["exit_layer"], 1), rec["exit_layer"])
    res = dec.generate(rec["prompt"], rec["eos_token_ids"], rec["max_steps"], rec["num_speculations"])
    assert res.predicted_tokens == fast.predicted_tokens
    assert res.acceptance_rate == pytest.approx(fast.acceptance_rate, abs=1e-12)


@pytest.mark.parametrize("prompt_len,spec", [(1, 4), (2, 15), (17, 15), (16, 1), (130, 7)])
def test_edge_shapes_spec_equals_ar_and_oracle(gpu_device, prompt_len, spec):
    """Edge cases of the step geometry: 1-token prompt (no prefill rows), the 16-row verify block
    (num_speculations = 15), prompts that end exactly on / just past a 16-row chunk and a KV page."""
    from layerskip_amd import GenerationConfig, synthetic
    from oracle import llama_oracle as lo
    cfg = synthetic.make_config("tiny-gqa")
    model_cpu = synthetic.build_model(cfg, seed=11, exit_layer=3, late_damping=0.05)
    prompt = synthetic.make_prompt(cfg.vocab_size, prompt_len, 40 + prompt_len)
    eos = [cfg.vocab_size]
    om = lo.OracleModel.from_hf(model_cpu, dtype=torch.float32)
    with torch.inference_mode():
        want = lo.self_speculative_generate(om, prompt, eos, 30, 3, spec)
    model = model_cpu.to(gpu_device)
    spec_s, ar_s = _strategies()
    gen = GenerationConfig(max_steps=30, exit_layer=3, num_speculations=spec, sample=False)
    a = spec_s.generate_token_ids(model, prompt, eos, gen)
    b = ar_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=30, exit_layer=-1, sample=False))
    assert a.predicted_tokens == b.predicted_tokens and len(a.predicted_tokens) == 30
    i = _first_mismatch(a.predicted_tokens, want.predicted_tokens)
    if i is not None:
        assert want.margins[i] < TIE_TOL, f"token {i} differs at fp32-oracle margin {want.margins[i]}"
    else:
        assert abs(a.acceptance_rate - want.acceptance_rate) < ACC_TOL


def test_early_exit_only_decoding(gpu_device):
    """AutoRegressive strategy with exit_layer > 0 = forward_early-only decoding (ARG:44-51)."""
    from layerskip_amd import GenerationConfig, synthetic
    from oracle import llama_oracle as lo
    cfg = synthetic.make_config("tiny-mha")
    model_cpu = synthetic.build_model(cfg, seed=2, exit_layer=2, late_damping=0.1)
    prompt = synthetic.make_prompt(cfg.vocab_size, 23, 7)
    om = lo.OracleModel.from_hf(model_cpu, dtype=torch.float32)
    with torch.inference_mode():
        want = lo.autoregressive_generate(om, prompt, [cfg.vocab_size], 16, exit_layer=2)
    _, ar = _strategies()
    got = ar.generate_token_ids(model_cpu.to(gpu_device), prompt, [cfg.vocab_size],
                                GenerationConfig(max_steps=16, exit_layer=2, sample=False))
    i = _first_mismatch(got.predicted_tokens, want.predicted_tokens)
    assert i is None or want.margins[i] < TIE_TOL


def test_full_size_7b_properties(gpu_device):
    """BASELINE.json's headline shape (llama2-7B: L32 H4096 I11008 V32000, exit_layer 8, 6 speculations) at
    full size, through size-independent properties (the CPU oracle needs minutes here; bench.py's
    cpu_baseline leg does the bounded oracle comparison):
      * speculative output == autoregressive output (the reference's correctness.py criterion),
      * a rerun is bit-identical (no run-to-run nondeterminism in any kernel),
      * acceptance bookkeeping is consistent: sum(n + 1) tokens emitted, len <= max_steps."""
    from layerskip_amd import GenerationConfig, synthetic
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~30 GB of free HBM")
    _MODELS.clear()
    cfg = synthetic.make_config("llama2-7B")
    model = synthetic.build_model(cfg, seed=0, exit_layer=8, late_damping=0.03, device=gpu_device, gen_device=gpu_device)
    prompt = synthetic.make_prompt(cfg.vocab_size, 512, 123)
    eos = [cfg.vocab_size]
    spec, ar = _strategies()
    gen = GenerationConfig(max_steps=160, exit_layer=8, num_speculations=6, sample=False)
    steps = []
    inner = spec.single_step_speculation

    def spy(**kw):
        r = inner(**kw)
        steps.append((r[4], r[3]))
        return r

    spec.single_step_speculation = spy
    a = spec.generate_token_ids(model, prompt, eos, gen)
    del spec.single_step_speculation
    a2 = spec.generate_token_ids(model, prompt, eos, gen)
    b = ar.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=160, exit_layer=-1, sample=False))
    assert a.predicted_tokens == a2.predicted_tokens
    assert a.predicted_tokens == b.predicted_tokens
    assert len(a.predicted_tokens) == 160
    assert sum(n + 1 for _, n in steps) == 160
    assert all(0 <= n <= td <= 6 for td, n in steps)
    assert a.acceptance_rate == pytest.approx(sum(n for _, n in steps) / sum(td for td, _ in steps))
    assert 0.2 < a.acceptance_rate < 0.95
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["tiny_mha_s1", "tiny_gqa_long", "tiny_gqa_s0_eos", "tiny_mha_s1_eos"])
def test_pipelined_generate_equals_stepwise(gpu_device, name):
    """lsk_spec_generate (whole generation in one call, steps pipelined on the stream) vs one lsk_spec_step
    call per step driven from Python: same tokens, same acceptance counters, same per-step trace."""
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    fused = HipSelfSpeculativeGenerationStrategy(fused_generate=True)
    stepwise = HipSelfSpeculativeGenerationStrategy(fused_generate=False)
    trace = []
    inner = stepwise.single_step_speculation

    def spy(**kw):
        r = inner(**kw)
        trace.append((r[4], r[3]))
        return r

    stepwise.single_step_speculation = spy
    cfg = _config(rec, "self_speculative")
    eos = rec["eos_token_ids"]
    if name.endswith("_eos"):
        # make sure the cut is exercised on THIS engine's trajectory: take a token it really emits
        free = fused.generate_token_ids(model, rec["prompt"], [model.config.vocab_size], cfg).predicted_tokens
        k = next(i for i in range(5, len(free)) if free[i] not in free[:i])
        eos = [free[k], model.config.vocab_size + 5]
    a = fused.generate_token_ids(model, rec["prompt"], eos, cfg)
    b = stepwise.generate_token_ids(model, rec["prompt"], eos, cfg)
    assert a.predicted_tokens == b.predicted_tokens
    assert a.acceptance_rate == b.acceptance_rate
    if not name.endswith("_eos"):
        assert [tuple(s) for s in fused.last_steps] == trace
    else:
        assert eos[0] not in a.predicted_tokens and len(a.predicted_tokens) < rec["max_steps"]
        assert [tuple(s) for s in fused.last_steps] == trace[: len(fused.last_steps)]
    # the engine is left consistent: a further generation gives the same answer
    c = fused.generate_token_ids(model, rec["prompt"], eos, cfg)
    assert c.predicted_tokens == a.predicted_tokens


@pytest.mark.parametrize("name", ["tiny_mha_s0", "tiny_gqa_long"])
def test_fused_autoregressive_generate_equals_stepwise(gpu_device, name):
    """lsk_ar_generate (device-resident token feedback, EOS checked every 8 tokens) vs one lsk_ar_step per token."""
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy
    rec = load_golden(name)
    model = _model(rec, gpu_device)
    cfg = _config(rec, "autoregressive")
    cfg.exit_layer = -1
    fused = HipAutoRegressiveGenerationStrategy(fused_generate=True)
    stepwise = HipAutoRegressiveGenerationStrategy(fused_generate=False)
    free = fused.generate_token_ids(model, rec["prompt"], [model.config.vocab_size], cfg).predicted_tokens
    assert free == stepwise.generate_token_ids(model, rec["prompt"], [model.config.vocab_size], cfg).predicted_tokens
    assert len(free) == rec["max_steps"]
    k = next(i for i in range(3, len(free)) if free[i] not in free[:i])
    eos = [free[k]]
    a = fused.generate_token_ids(model, rec["prompt"], eos, cfg).predicted_tokens
    b = stepwise.generate_token_ids(model, rec["prompt"], eos, cfg).predicted_tokens
    assert a == b == free[:k]
    cfg.exit_layer = rec["exit_layer"]            # early-exit-only decoding through the same call
    assert (fused.generate_token_ids(model, rec["prompt"], eos, cfg).predicted_tokens
            == stepwise.generate_token_ids(model, rec["prompt"], eos, cfg).predicted_tokens)


def test_long_context_many_pages(gpu_device):
    """A context of > 8 KV pages (the page-partial combine walks pages in groups of 8) and a prompt longer than
    the default prefill buffer: speculative == autoregressive in the engine, and both follow the fp32 oracle."""
    from layerskip_amd import GenerationConfig, synthetic
    from oracle import llama_oracle as lo
    cfg = synthetic.make_config("tiny-mha")
    model_cpu = synthetic.build_model(cfg, seed=21, exit_layer=2, late_damping=0.05)
    prompt = synthetic.make_prompt(cfg.vocab_size, 1190, 77)
    eos = [cfg.vocab_size]
    om = lo.OracleModel.from_hf(model_cpu, dtype=torch.float32)
    with torch.inference_mode():
        want = lo.self_speculative_generate(om, prompt, eos, 36, 2, 5)
    model = model_cpu.to(gpu_device)
    spec_s, ar_s = _strategies()
    a = spec_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=36, exit_layer=2, num_speculations=5, sample=False))
    b = ar_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=36, exit_layer=-1, sample=False))
    assert a.predicted_tokens == b.predicted_tokens and len(a.predicted_tokens) == 36
    i = _first_mismatch(a.predicted_tokens, want.predicted_tokens)
    assert i is None or want.margins[i] < TIE_TOL, (i, want.margins[i])


def test_more_than_15_speculations(gpu_device):
    """num_speculations beyond the 16-row fused verify block: host-walked 16-row passes, same kernels."""
    from layerskip_amd import GenerationConfig, synthetic
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=13, exit_layer=3, late_damping=0.02).to(gpu_device)
    prompt = synthetic.make_prompt(cfg.vocab_size, 45, 5)
    eos = [cfg.vocab_size]
    spec_s, ar_s = _strategies()
    a = spec_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=50, exit_layer=3, num_speculations=21, sample=False))
    b = ar_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=50, exit_layer=-1, sample=False))
    c = spec_s.generate_token_ids(model, prompt, eos, GenerationConfig(max_steps=50, exit_layer=3, num_speculations=6, sample=False))
    assert a.predicted_tokens == b.predicted_tokens == c.predicted_tokens
    assert 0.0 <= a.acceptance_rate <= 1.0


def test_engine_is_released_with_its_model(gpu_device):
    """`get_engine` maps model -> engine weakly and the engine holds no strong reference back: dropping the model
    frees the engine (packed weights, KV pool) instead of pinning both in HBM."""
    import gc
    import weakref
    from layerskip_amd import engine as engine_mod
    rec = load_golden("tiny_mha_s0")
    model = build_case_model(rec).to(gpu_device)
    eng = engine_mod.get_engine(model, max_ctx=256, max_prompt=64)
    assert eng.model is model
    out, _, _, _ = eng.spec_generate(rec["prompt"], rec["num_speculations"], rec["exit_layer"], rec["eos_token_ids"], 8)
    assert len(out) > 0
    ref = weakref.ref(eng)
    del eng, model
    gc.collect()
    assert ref() is None


def test_engine_follows_weight_changes_on_the_same_model_object(gpu_device):
    """The engine streams packed COPIES of the projections; the reference reads live weights.  After load_state_dict (or
    any in-place edit) on the same model object the next generation must use the new weights: get_engine() compares the
    (address, version) fingerprint taken at pack time and re-packs."""
    from conftest import build_struct_model, load_struct
    from layerskip_amd import GenerationConfig
    from layerskip_amd.engine import get_engine
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    rec_a, rec_b = load_struct("tiny_gqa"), load_struct("tiny_gqa_spec15")        # same shape, different seeds
    model = build_struct_model(rec_a, gpu_device)
    other = build_struct_model(rec_b, gpu_device)
    strat = HipSelfSpeculativeGenerationStrategy()
    cfg = GenerationConfig(max_steps=rec_a["max_steps"], exit_layer=rec_a["exit_layer"], num_speculations=rec_a["num_speculations"], sample=False)
    a = strat.generate_token_ids(model, rec_a["prompt"], rec_a["eos_token_ids"], cfg)
    assert a.predicted_tokens == rec_a["bf16"]["spec_tokens"]
    eng = get_engine(model)
    assert not eng.weights_changed(model)
    model.load_state_dict(other.state_dict())                 # in-place copy_ into the same Parameters: versions bump
    assert eng.weights_changed(model)
    cfg_b = GenerationConfig(max_steps=rec_b["max_steps"], exit_layer=rec_b["exit_layer"], num_speculations=rec_b["num_speculations"], sample=False)
    b = strat.generate_token_ids(model, rec_b["prompt"], rec_b["eos_token_ids"], cfg_b)
    assert get_engine(model) is eng and not eng.weights_changed(model)
    assert b.predicted_tokens == rec_b["bf16"]["spec_tokens"]
    assert b.acceptance_rate == rec_b["bf16"]["acceptance_rate"]


def test_engine_notices_edits_through_dot_data(gpu_device):
    """`p.data += delta` (PEFT's default LoRA merge) moves neither a Parameter's address nor its version counter: the sampled
    content checksum of the fingerprint (lsk_engine_weights_checksum) is what catches it.  After such a merge on the same model
    object the next generation must come from the merged weights, like the reference's (it reads live weights)."""
    from conftest import build_struct_model, load_struct
    from layerskip_amd import GenerationConfig
    from layerskip_amd.engine import get_engine
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    rec_a, rec_b = load_struct("tiny_gqa"), load_struct("tiny_gqa_spec15")        # same shape, different seeds
    model = build_struct_model(rec_a, gpu_device)
    other = build_struct_model(rec_b, gpu_device)
    strat = HipSelfSpeculativeGenerationStrategy()
    cfg = GenerationConfig(max_steps=rec_a["max_steps"], exit_layer=rec_a["exit_layer"], num_speculations=rec_a["num_speculations"], sample=False)
    assert strat.generate_token_ids(model, rec_a["prompt"], rec_a["eos_token_ids"], cfg).predicted_tokens == rec_a["bf16"]["spec_tokens"]
    eng = get_engine(model)
    versions = [p._version for p in model.parameters()]
    ptrs = [p.data_ptr() for p in model.parameters()]
    with torch.no_grad():
        for p, q in zip(model.parameters(), other.parameters()):
            p.data.copy_(q.data)                              # through .data: no version bump, same storage
    assert versions == [p._version for p in model.parameters()] and ptrs == [p.data_ptr() for p in model.parameters()]
    assert eng.weights_changed(model)
    cfg_b = GenerationConfig(max_steps=rec_b["max_steps"], exit_layer=rec_b["exit_layer"], num_speculations=rec_b["num_speculations"], sample=False)
    b = strat.generate_token_ids(model, rec_b["prompt"], rec_b["eos_token_ids"], cfg_b)
    assert get_engine(model) is eng and not eng.weights_changed(model)
    assert b.predicted_tokens == rec_b["bf16"]["spec_tokens"] and b.acceptance_rate == rec_b["bf16"]["acceptance_rate"]
    # a single-tensor merge (one projection of one layer, `+=` through .data) is seen too
    with torch.no_grad():
        model.model.layers[1].mlp.down_proj.weight.data += 0.125
    assert eng.weights_changed(model)
