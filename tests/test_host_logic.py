"""Host-side strategy logic on CPU: the HIP strategies driven by an oracle-backed fake engine must
reproduce the reference fixtures exactly (same arithmetic => exact), including EOS truncation, the
max_steps clamp of the speculation count, acceptance accounting and the reference's error behaviour."""
import pytest
import torch

from conftest import build_case_model, golden_names, load_golden
from fake_engine import FakeEngine

from layerskip_amd import GenerationConfig, hip_strategies


@pytest.fixture()
def fake(monkeypatch):
    engines = {}

    def get_engine(model, **kw):
        if id(model) not in engines:
            engines[id(model)] = FakeEngine(model)
        return engines[id(model)]

    monkeypatch.setattr(hip_strategies, "get_engine", get_engine)
    return engines


def _cfg(rec, **kw):
    base = dict(max_steps=rec["max_steps"], exit_layer=rec["exit_layer"], num_speculations=rec["num_speculations"],
                sample=False, generation_strategy="self_speculative")
    base.update(kw)
    return GenerationConfig(**base)


CASES = [n for n in golden_names() if n.startswith("tiny_mha") or n.endswith("_eos") or n == "tiny_d64_s0"]


@pytest.mark.parametrize("name", CASES)
def test_speculative_strategy_reproduces_reference(fake, name):
    rec = load_golden(name)
    model = build_case_model(rec)
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    res = strat.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec), logits_processors=[],
                                   stopping_criteria=[])
    gold = rec["fp32"]
    assert res.predicted_tokens == gold["spec_tokens"]
    assert res.acceptance_rate == pytest.approx(gold["acceptance_rate"], abs=1e-12)
    assert len(res.predicted_tokens) <= rec["max_steps"]
    for e in rec["eos_token_ids"]:
        assert e not in res.predicted_tokens


@pytest.mark.parametrize("name", CASES[:2])
def test_autoregressive_strategy_reproduces_reference(fake, name):
    rec = load_golden(name)
    model = build_case_model(rec)
    strat = hip_strategies.HipAutoRegressiveGenerationStrategy()
    res = strat.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, exit_layer=-1))
    assert res.predicted_tokens == rec["fp32"]["ar_tokens"]
    assert res.acceptance_rate is None


def test_speculation_count_is_clamped_by_max_steps(fake):
    rec = load_golden("tiny_mha_s0")
    model = build_case_model(rec)
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    seen = []
    inner = strat.single_step_speculation

    def spy(**kw):
        seen.append((len(kw["output_ids"]), kw["num_speculations"]))
        return inner(**kw)

    strat.single_step_speculation = spy
    res = strat.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, max_steps=7))
    assert len(res.predicted_tokens) == 7
    for n_out, s in seen:
        assert s == min(rec["num_speculations"], 7 - n_out - 1)       # SSG:63-66


def test_no_draft_ever_raises_zero_division_like_the_reference(monkeypatch):
    """acceptance_rate = matches / drafts (SSG:98) is evaluated unguarded, as in the reference."""
    from layerskip_amd.engine import StepResult

    class NoDraftEngine:
        num_layers, vocab, kv_len = 6, 512, 0

        def ensure_capacity(self, *a):
            pass

        def reset(self):
            self.kv_len = 0

        def set_kv_len(self, n):
            self.kv_len = n

        def spec_step(self, ids, s, e, eos):
            self.kv_len += len(ids)
            return StepResult(0, 0, 7, self.kv_len, [7], [], [7])

    eng = NoDraftEngine()
    monkeypatch.setattr(hip_strategies, "get_engine", lambda model, **kw: eng)
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    cfg = GenerationConfig(max_steps=3, exit_layer=2, num_speculations=0, sample=False)
    with pytest.raises(ZeroDivisionError):
        strat.generate_token_ids(object(), [5, 6], [99], cfg)


def test_stopping_criteria_and_streamer_are_honoured(fake):
    rec = load_golden("tiny_mha_s0")
    model = build_case_model(rec)

    class Stop:
        def __init__(self):
            self.calls = 0
            self.shapes = []

        def __call__(self, input_ids, scores=None):
            self.calls += 1
            self.shapes.append(tuple(input_ids.shape))
            return torch.tensor([True])

    class Streamer:
        def __init__(self):
            self.tokens = []

        def put(self, value):
            self.tokens.extend(int(v) for v in value.reshape(-1).tolist())

    import transformers
    stop, streamer = Stop(), Streamer()
    crit = transformers.StoppingCriteriaList([stop])
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    res = strat.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec), stopping_criteria=crit,
                                   streamer=streamer)
    assert stop.calls == 1 and stop.shapes == [(1, 1)]                # evaluated on the next-input tensor (SSG:94)
    assert res.predicted_tokens == rec["fp32"]["spec_tokens"][: len(res.predicted_tokens)]
    assert streamer.tokens == res.predicted_tokens
    ar = hip_strategies.HipAutoRegressiveGenerationStrategy()
    out = ar.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, exit_layer=-1), stopping_criteria=crit)
    assert out.predicted_tokens == []                                    # reference tests/test_autoregressive_generator.py:37-47


def test_argument_validation(fake):
    rec = load_golden("tiny_mha_s0")
    model = build_case_model(rec)
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    with pytest.raises(ValueError):
        strat.generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], _cfg(rec, exit_layer=99))


def test_eos_list_is_never_silently_truncated():
    from layerskip_amd._lib import LSK_MAX_EOS, LskError
    from layerskip_amd.engine import HipEngine
    eng = object.__new__(HipEngine)          # no device needed for the host-side list handling
    eng.vocab = 100
    eos, arr = eng._eos_array([2, 1000, -1, 7])
    assert eos == [2, 7] and list(arr)[:2] == [2, 7]         # ids outside the vocabulary can never be produced
    assert eng._eos_array([])[0] == []
    # the reference folds any number of stop_token_ids into the list (generator_base.py:106): repeats collapse (first occurrence kept,
    # the list ORDER decides which id truncates the output), a dozen ids are nothing special, and only more than LSK_MAX_EOS DISTINCT
    # in-vocabulary ids is an error
    assert eng._eos_array([7, 2, 7, 2, 9])[0] == [7, 2, 9]
    assert eng._eos_array(list(range(12)))[0] == list(range(12))
    eng.vocab = 4 * LSK_MAX_EOS
    assert len(eng._eos_array(list(range(LSK_MAX_EOS)))[0]) == LSK_MAX_EOS
    with pytest.raises(LskError):
        eng._eos_array(list(range(LSK_MAX_EOS + 1)))


def test_engine_status_checks_use_the_engines_own_library():
    """Every C-ABI call of HipEngine goes through `_ck`, which reads the message of the library that engine loaded
    (two libraries -- bf16 and fp16 -- can be resident).  Exercised with a stub library: no device needed."""
    import ctypes
    from layerskip_amd._lib import LskError
    from layerskip_amd.engine import HipEngine

    class StubLib:
        def __init__(self):
            self.fail = False

        def lsk_engine_get_kv_len(self, handle, out):
            ctypes.cast(out, ctypes.POINTER(ctypes.c_int32))[0] = 41
            return 1 if self.fail else 0

        def lsk_engine_set_option(self, handle, option, value):
            return 1 if self.fail else 0

        def lsk_last_error(self):
            return b"stub says no"

    eng = object.__new__(HipEngine)
    eng.lib, eng._handle, eng._options = StubLib(), ctypes.c_void_p(None), {}
    assert eng.kv_len == 41
    eng.set_option(3, 1)
    eng.lib.fail = True
    with pytest.raises(LskError, match="stub says no"):
        eng.set_option(3, 1)
    with pytest.raises(LskError, match="stub says no"):
        _ = eng.kv_len


def test_single_step_extends_the_callers_output_list_in_place(fake):
    """The reference's step extends the list it was handed and returns that same object (self_speculation_generator.py:204-205,
    :223-229): a caller holding the list sees the new tokens without reading the return value."""
    import torch
    rec = load_golden("tiny_mha_s0")
    model = build_case_model(rec)
    strat = hip_strategies.HipSelfSpeculativeGenerationStrategy()
    mine = [7, 7]                      # (whatever the caller already collected stays in front)
    ids = rec["prompt"]
    nxt, out, past, n, td = strat.single_step_speculation(
        model=model, input_ids_list=ids, input_ids=torch.tensor([ids]), output_ids=mine, num_speculations=3, past_key_values=None,
        exit_layer=rec["exit_layer"], eos_token_ids=rec["eos_token_ids"], calls=0, sample=False)
    assert out is mine and len(mine) == 2 + n + 1 and mine[:2] == [7, 7]
    assert mine[2:] == rec["fp32"]["spec_tokens"][: n + 1]
