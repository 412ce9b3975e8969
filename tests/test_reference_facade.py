"""The plugin under the reference's OWN objects (build container only: needs /root/reference).

`HuggingfaceLlamaGenerator` (generator_base.py:65-130) -- the facade generate.py / benchmark.py / correctness.py / eval.py
drive -- is imported unmodified through oracle/ref_shim.py and handed `HipSelfSpeculativeGenerationStrategy` /
`HipAutoRegressiveGenerationStrategy` exactly where it takes the reference's strategies, with the reference's own
`GenerationConfig`.  The engine behind the plugin is the CPU stand-in of tests/fake_engine.py (same method surface as
HipEngine, arithmetic from the reference-pinned oracle in fp32), so what is tested is the HOST side of the drop-in: the
call contract, the result objects, EOS / stop handling, logits processors, stopping criteria and the streamer protocol --
each compared with what the reference's own strategy does under the same facade on the same weights.
Also replays the invariants of the reference's unit tests (tests/test_self_speculation_generator.py:37-79,
tests/test_autoregressive_generator.py:37-61)."""
import copy
import os

import pytest
import torch

from conftest import build_struct_model, load_struct

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/self_speculation"), reason="reference tree not mounted")


class FakeTokenizer:
    """Whitespace 'tokenizer' over the checkpoint's active vocabulary: enough of the HF interface for the facade."""
    eos_token_id = 2

    def __init__(self, active):
        self.active = list(active)

    def _id(self, word):
        return self.active[sum(ord(c) for c in word) % len(self.active)]

    def __call__(self, prompt, return_tensors="pt", add_special_tokens=True):
        ids = ([1] if add_special_tokens else []) + [self._id(w) for w in prompt.split()]
        return {"input_ids": torch.tensor([ids])}

    def encode(self, text):
        return [1] + [self._id(w) for w in text.split()]

    def decode(self, ids, **kw):
        return " ".join(f"<{int(i)}>" for i in ids)


@pytest.fixture(scope="module")
def world():
    from fake_engine import FullFakeEngine
    from oracle import ref_shim
    ref = ref_shim.load_reference()
    rec = load_struct("tiny_gqa")
    base = build_struct_model(rec).float()                      # fp32 on the bf16-valued weights: reproducible on any CPU
    ref_model = ref_shim.patch_model(copy.deepcopy(base))
    tok = FakeTokenizer(base.struct_program["active"])
    # BOS (id 1) is outside the active vocabulary; give it an active token's embedding so the token program applies
    return {"ref": ref, "rec": rec, "base": base, "ref_model": ref_model, "tok": tok, "engine": FullFakeEngine(base)}


@pytest.fixture()
def patched(world, monkeypatch):
    from layerskip_amd import hip_strategies
    monkeypatch.setattr(hip_strategies, "get_engine", lambda model, **k: world["engine"])
    return world


def _facades(w, strategy_name):
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    ref = w["ref"]
    gb = ref.generator_base
    if strategy_name == "self_speculative":
        theirs, mine = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy(), HipSelfSpeculativeGenerationStrategy()
    else:
        theirs, mine = ref.autoregressive_generator.AutoRegressiveGenerationStrategy(), HipAutoRegressiveGenerationStrategy()
    return (gb.HuggingfaceLlamaGenerator(tokenizer=w["tok"], model=w["ref_model"], generation_strategy=theirs),
            gb.HuggingfaceLlamaGenerator(tokenizer=w["tok"], model=w["base"], generation_strategy=mine))


PROMPT = "the quick brown fox jumps over the lazy dog and keeps running through the forest"


@pytest.mark.parametrize("strategy,kw", [
    ("self_speculative", dict(exit_layer=3, num_speculations=6)),
    ("self_speculative", dict(exit_layer=3, num_speculations=2, no_repeat_ngram_size=3)),      # a real HF logits processor
    ("autoregressive", dict(exit_layer=-1)),
    ("autoregressive", dict(exit_layer=3)),                                                       # early-exit-only decoding
    ("autoregressive", dict(exit_layer=-1, no_repeat_ngram_size=2)),
])
def test_facade_gives_the_same_generation_result(patched, strategy, kw):
    w = patched
    gb = w["ref"].generator_base
    cfg = gb.GenerationConfig(max_steps=24, sample=False, generation_strategy=strategy, **kw)
    theirs, mine = _facades(w, strategy)
    a = theirs.generate(prompt=PROMPT, generation_config=cfg)
    b = mine.generate(prompt=PROMPT, generation_config=cfg)
    assert isinstance(b, gb.GenerationResult)
    assert b.decoded_prediction == a.decoded_prediction
    assert b.num_tokens_generated == a.num_tokens_generated == 24
    assert b.generation_strategy_result.predicted_tokens == a.generation_strategy_result.predicted_tokens
    assert b.generation_strategy_result.acceptance_rate == a.generation_strategy_result.acceptance_rate
    assert b.total_time > 0 and b.tokens_per_second > 0 and b.time_per_token > 0


def test_facade_eos_and_stop_token_ids(patched):
    """stop_token_ids are folded into eos_token_ids by the facade (GB:106); the first one hit truncates the output."""
    w = patched
    gb = w["ref"].generator_base
    theirs, mine = _facades(w, "self_speculative")
    free = theirs.generate(prompt=PROMPT, generation_config=gb.GenerationConfig(max_steps=24, sample=False, exit_layer=3,
                                                                               num_speculations=6)).generation_strategy_result.predicted_tokens
    stop = free[7]
    cfg = gb.GenerationConfig(max_steps=24, sample=False, exit_layer=3, num_speculations=6, stop_token_ids=[stop])
    a, b = theirs.generate(prompt=PROMPT, generation_config=cfg), mine.generate(prompt=PROMPT, generation_config=cfg)
    assert a.generation_strategy_result.predicted_tokens == free[: free.index(stop)]
    assert b.generation_strategy_result.predicted_tokens == a.generation_strategy_result.predicted_tokens
    assert b.generation_strategy_result.acceptance_rate == a.generation_strategy_result.acceptance_rate


def test_replay_reference_unit_tests_self_speculation(patched):
    """tests/test_self_speculation_generator.py:37-79, with the reference's own strategy run beside the plugin."""
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    w = patched
    ref = w["ref"]
    cfg = ref.generator_base.GenerationConfig(max_steps=4, exit_layer=3, num_speculations=4)
    tok = w["tok"]
    for strategy, model in ((ref.self_speculation_generator.SelfSpeculativeGenerationStrategy(), w["ref_model"]),
                            (HipSelfSpeculativeGenerationStrategy(), w["base"])):
        input_ids = torch.tensor([[tok.encode("my")[1]]])
        with torch.inference_mode():
            _, output_ids, _, matches, specs = strategy.single_step_speculation(
                model=model, input_ids=input_ids, input_ids_list=input_ids.tolist(), output_ids=[], num_speculations=1,
                past_key_values=None, eos_token_ids=[tok.eos_token_id], calls=0, exit_layer=cfg.exit_layer, sample=cfg.sample,
                temperature=cfg.temperature, top_k=cfg.top_k, top_p=cfg.top_p)
        assert matches <= specs
        assert len(output_ids) == matches + 1
    outs = []
    for strategy, model in ((ref.self_speculation_generator.SelfSpeculativeGenerationStrategy(), w["ref_model"]),
                            (HipSelfSpeculativeGenerationStrategy(), w["base"])):
        ids = [tok.encode("my")[1], tok.encode("name")[1], tok.encode("is")[1]]
        logits_processor = lambda inputs, logits: torch.log(torch.softmax(logits, dim=-1))      # noqa: E731 (as in the reference test)
        cfg.sample = False
        with torch.inference_mode():
            result = strategy.generate_token_ids(model, ids, [tok.eos_token_id], cfg, logits_processors=logits_processor)
        assert len(result.predicted_tokens) > 0
        assert tok.eos_token_id in result.predicted_tokens or len(result.predicted_tokens) == cfg.max_steps
        outs.append((result.predicted_tokens, result.acceptance_rate))
    assert outs[0] == outs[1]


def test_replay_reference_unit_tests_autoregressive(patched):
    """tests/test_autoregressive_generator.py:37-61."""
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy
    w = patched
    ref = w["ref"]
    tok = w["tok"]
    cfg = ref.generator_base.GenerationConfig(max_steps=8, sample=False)
    ids = [tok.encode("my")[1], tok.encode("name")[1], tok.encode("is")[1]]
    outs = []
    for strategy, model in ((ref.autoregressive_generator.AutoRegressiveGenerationStrategy(), w["ref_model"]),
                            (HipAutoRegressiveGenerationStrategy(), w["base"])):
        with torch.inference_mode():
            stop_now = strategy.generate_token_ids(model, ids, [tok.eos_token_id], cfg,
                                                   stopping_criteria=lambda inputs, scores: torch.tensor([True]))
            assert len(stop_now.predicted_tokens) == 0
            result = strategy.generate_token_ids(model, ids, [tok.eos_token_id], cfg,
                                                 logits_processors=lambda inputs, logits: torch.log(torch.softmax(logits, dim=-1)))
        assert len(result.predicted_tokens) > 0
        assert tok.eos_token_id in result.predicted_tokens or len(result.predicted_tokens) == cfg.max_steps
        outs.append(result.predicted_tokens)
    assert outs[0] == outs[1]


def test_speculative_streamer_protocol(patched):
    """SpeculativeTextStreamer sees the same put(draft, is_draft=True) / delete / put(accepted) / put(next) sequence
    (SSG:158-161, :207-213); a plain TextStreamer the same put(LongTensor) calls (SSG:214-216, ARG:63-64)."""
    import importlib
    from layerskip_amd.hip_strategies import HipAutoRegressiveGenerationStrategy, HipSelfSpeculativeGenerationStrategy
    import transformers
    w = patched
    ref = w["ref"]
    streamer_mod = importlib.import_module("self_speculation.speculative_streamer")

    class Recording(streamer_mod.SpeculativeTextStreamer):
        def __init__(self, tokenizer):
            super().__init__(tokenizer)
            self.log = []

        def put(self, value, is_draft=False):
            self.log.append(("put", [int(t) for t in torch.as_tensor(value).reshape(-1).tolist()], bool(is_draft)))

        def delete(self, num_tokens, is_draft=False):
            self.log.append(("delete", int(num_tokens)))

    class Plain(transformers.TextStreamer):
        def __init__(self, tokenizer):
            super().__init__(tokenizer)
            self.log = []

        def put(self, value):
            self.log.append(("put", [int(t) for t in torch.as_tensor(value).reshape(-1).tolist()], str(torch.as_tensor(value).dtype)))

    ids = w["tok"].encode(PROMPT)
    cfg = ref.generator_base.GenerationConfig(max_steps=14, exit_layer=3, num_speculations=5, sample=False)
    for cls in (Recording, Plain):
        logs = []
        for strategy, model in ((ref.self_speculation_generator.SelfSpeculativeGenerationStrategy(), w["ref_model"]),
                                (HipSelfSpeculativeGenerationStrategy(), w["base"])):
            st = cls(w["tok"])
            with torch.inference_mode():
                strategy.generate_token_ids(model, ids, [w["tok"].eos_token_id], cfg, streamer=st)
            logs.append(st.log)
        assert logs[0] == logs[1] and len(logs[0]) > 3
    logs = []
    for strategy, model in ((ref.autoregressive_generator.AutoRegressiveGenerationStrategy(), w["ref_model"]),
                            (HipAutoRegressiveGenerationStrategy(), w["base"])):
        st = Plain(w["tok"])
        with torch.inference_mode():
            strategy.generate_token_ids(model, ids, [w["tok"].eos_token_id], ref.generator_base.GenerationConfig(max_steps=6, sample=False), streamer=st)
        logs.append([entry[:2] for entry in st.log])
    assert logs[0] == logs[1] and len(logs[0]) == 6


def test_sampling_through_the_facade(patched):
    """The CLI default (sample=True, generator_base.py:39) runs through the device-sampling orchestration and keeps the contract."""
    w = patched
    gb = w["ref"].generator_base
    _, mine = _facades(w, "self_speculative")
    torch.manual_seed(3)
    res = mine.generate(prompt=PROMPT, generation_config=gb.GenerationConfig(max_steps=10, exit_layer=3, num_speculations=3, temperature=1.5))
    assert 0 < res.num_tokens_generated <= 10
    assert 0.0 <= res.generation_strategy_result.acceptance_rate <= 1.0


def test_cli_drivers_report_the_reference_metric_keys(patched, monkeypatch, capsys, tmp_path):
    """benchmark.py: the four means of benchmark.py:95-117; correctness.py: errors / error_pct (correctness.py:82-88)."""
    import importlib.util
    import json
    import sys
    from conftest import ROOT

    def load(name):     # THIS repo's driver, not the reference's namesake (ref_shim put /root/reference first on sys.path)
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        monkeypatch.setitem(sys.modules, name, mod)
        spec.loader.exec_module(mod)
        return mod

    benchmark = load("benchmark")
    correctness = load("correctness")
    from layerskip_amd import GenerationConfig
    from layerskip_amd.cli import common
    w = patched
    gen = GenerationConfig(max_steps=8, exit_layer=3, num_speculations=4, sample=False, generation_strategy="self_speculative")
    metrics = benchmark.benchmark(w["base"], None, benchmark.BenchmarkArguments(num_samples=2), gen,
                                  common.SyntheticArguments(prompt_len=12, device="cpu"), seed=0)
    assert set(metrics) == {"acceptance_rate", "total_time", "time_per_token", "tokens_per_second"}
    assert all(set(v) == {"mean"} and v["mean"] > 0 for v in metrics.values())
    monkeypatch.setattr(common, "load_model_and_tokenizer", lambda args, syn, exit_layer, *rest: (w["base"], None))
    monkeypatch.setattr(correctness, "load_model_and_tokenizer", lambda args, syn, exit_layer, *rest: (w["base"], None))
    monkeypatch.setattr(sys, "argv", ["correctness.py", "--model", "synthetic:tiny-gqa", "--num_samples", "2", "--prompt_len", "12",
                                      "--device", "cpu", "--max_steps", "8", "--exit_layer", "3", "--num_speculations", "4",
                                      "--output_dir", str(tmp_path)])
    assert correctness.main() == 0          # the exit code (the script passes it to sys.exit)
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["errors"] == 0 and out["error_pct"] == 0 and out["num_samples"] == 2


def test_sweep_driver_writes_the_reference_csv(patched, monkeypatch, tmp_path):
    """sweep.py (reference sweep.py:47-65): one CSV row per (exit_layer, num_speculations) grid point with the reference's
    column names, rewritten after every point."""
    import csv
    import importlib.util
    import sys
    from conftest import ROOT
    w = patched

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        monkeypatch.setitem(sys.modules, name, mod)
        spec.loader.exec_module(mod)
        return mod

    load("benchmark")
    sweep = load("sweep")
    monkeypatch.setattr(sweep, "load_model_and_tokenizer", lambda args, syn, exit_layer, *rest: (w["base"], None))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(sys, "argv", ["sweep.py", "--model", "synthetic:tiny-gqa", "--num_samples", "1", "--prompt_len", "10", "--device", "cpu",
                                      "--max_steps", "6", "--exit_layer_first", "3", "--exit_layer_last", "3", "--num_speculations_first", "2",
                                      "--num_speculations_last", "4", "--num_speculations_step", "2", "--output_dir", str(tmp_path)])
    sweep.main()
    import glob
    files = glob.glob(str(tmp_path / "sweep_*.csv"))
    assert len(files) == 1
    rows = list(csv.DictReader(open(files[0])))
    assert [(r["exit_layer"], r["num_speculations"]) for r in rows] == [("3", "2"), ("3", "4")]
    assert list(rows[0]) == ["exit_layer", "num_speculations", "acceptance_rate", "total_time", "time_per_token", "tokens_per_second"]
    assert all(float(r["tokens_per_second"]) > 0 for r in rows)


def test_generate_repl_driver(patched, monkeypatch, capsys):
    """generate.py (reference generate.py:95-161): the REPL decodes a prompt of token ids and prints the reference's footer."""
    import builtins
    import importlib.util
    import sys
    from conftest import ROOT
    w = patched
    spec = importlib.util.spec_from_file_location("lsk_generate_cli", os.path.join(ROOT, "generate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "load_model_and_tokenizer", lambda args, syn, exit_layer, *rest: (w["base"], None))
    ids = w["tok"].encode(PROMPT)
    lines = iter([" ".join(str(i) for i in ids), "exit"])
    monkeypatch.setattr(builtins, "input", lambda prompt="": next(lines))
    monkeypatch.setattr(sys, "argv", ["generate.py", "--model", "synthetic:tiny-gqa", "--device", "cpu", "--max_steps", "8", "--exit_layer", "3",
                                      "--num_speculations", "4", "--sample", "False", "--generation_strategy", "self_speculative"])
    mod.main()
    out = capsys.readouterr().out
    assert "Tokens per second:" in out and "Acceptance rate:" in out and "Time per token:" in out
    toks = [int(t) for t in out.split("[", 1)[1].split("]", 1)[0].split(",")]
    assert len(toks) == 8
