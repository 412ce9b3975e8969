"""The reference-style drivers and the slow path's real users ON THE HIP ENGINE.

* benchmark.py / correctness.py / sweep.py (reference benchmark.py:155-205, correctness.py:70-88, sweep.py:47-65): their
  `main()` on `synthetic:tiny-gqa` on cuda:0 -- the four metric means, errors == 0, the sweep CSV with the reference's columns.
* the callables the facade hands the strategies (generator_base.py:77-95): a real HF `NoRepeatNGramLogitsProcessor`, a logits
  processor that changes decisions, a stopping criterion that fires mid-generation (SSG:92-95, ARG:68-71), and both streamer
  protocols (`SpeculativeTextStreamer`: SSG:158-161, :207-213; plain `TextStreamer`: SSG:214-216, ARG:63-64) -- each run through
  `HipEngine` and compared, call for call, with the same strategy code over the CPU stand-in of tests/fake_engine.py (oracle
  arithmetic).  tests/test_reference_facade.py shows (build container) that this stand-in run equals the UNMODIFIED reference's
  under the same callables, so the chain reference == oracle stand-in == HIP engine is closed on the GPU box.
Structured checkpoints: every decision has a wide top-2 margin, so bf16 on the GPU and fp32 on the CPU decide alike."""
import csv
import glob
import importlib.util
import json
import os
import sys

import pytest
import torch

from conftest import ROOT, build_struct_model, load_struct

pytestmark = pytest.mark.gpu


def _load_driver(name, monkeypatch):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, name, mod)
    spec.loader.exec_module(mod)
    return mod


COMMON = ["--model", "synthetic:tiny-gqa", "--prompt_len", "40", "--device", "cuda:0", "--max_steps", "24", "--exit_layer", "3",
          "--num_speculations", "4", "--late_damping", "0.1"]


def test_benchmark_driver_on_the_engine(gpu_device, monkeypatch, capsys, tmp_path):
    benchmark = _load_driver("benchmark", monkeypatch)
    monkeypatch.setattr(sys, "argv", ["benchmark.py"] + COMMON + ["--num_samples", "3", "--generation_strategy", "self_speculative",
                                      "--sample", "False", "--output_dir", str(tmp_path)])
    benchmark.main()
    metrics = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert set(metrics) == {"acceptance_rate", "total_time", "time_per_token", "tokens_per_second"}      # benchmark.py:95-117
    assert all(set(v) == {"mean"} for v in metrics.values())
    assert 0.0 < metrics["acceptance_rate"]["mean"] <= 1.0 and metrics["tokens_per_second"]["mean"] > 0
    assert abs(metrics["time_per_token"]["mean"] * 24 - metrics["total_time"]["mean"]) < 1e-6 * 24 + 1e-9
    dumped = glob.glob(str(tmp_path / "benchmark_*.json"))
    assert len(dumped) == 1 and json.load(open(dumped[0]))["metrics"] == metrics
    # the CLI default: sample=True (generator_base.py:39) through the device-sampling path, and the autoregressive baseline
    for extra in (["--generation_strategy", "self_speculative"], ["--generation_strategy", "autoregressive", "--sample", "False"]):
        monkeypatch.setattr(sys, "argv", ["benchmark.py"] + COMMON + ["--num_samples", "2", "--output_dir", str(tmp_path)] + extra)
        benchmark.main()
        m = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert m["tokens_per_second"]["mean"] > 0
        if "autoregressive" in extra:
            assert m["acceptance_rate"]["mean"] == 0.0                   # benchmark.py:78-79: 0 for AR runs


def test_correctness_driver_on_the_engine(gpu_device, monkeypatch, capsys, tmp_path):
    _load_driver("benchmark", monkeypatch)
    correctness = _load_driver("correctness", monkeypatch)
    monkeypatch.setattr(sys, "argv", ["correctness.py"] + COMMON + ["--num_samples", "4", "--output_dir", str(tmp_path)])
    assert correctness.main() == 0          # the exit code (the script passes it to sys.exit)
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out == {"errors": 0, "error_pct": 0.0, "num_samples": 4}              # correctness.py:82-88
    dumped = glob.glob(str(tmp_path / "correctness_*.json"))
    assert len(dumped) == 1 and json.load(open(dumped[0]))["errors"] == 0


def test_sweep_driver_on_the_engine(gpu_device, monkeypatch, tmp_path):
    _load_driver("benchmark", monkeypatch)
    sweep = _load_driver("sweep", monkeypatch)
    monkeypatch.setattr(sys, "argv", ["sweep.py"] + COMMON + ["--num_samples", "2", "--exit_layer_first", "2", "--exit_layer_last", "4",
                                      "--exit_layer_step", "2", "--num_speculations_first", "2", "--num_speculations_last", "6",
                                      "--num_speculations_step", "4", "--output_dir", str(tmp_path)])
    sweep.main()
    files = glob.glob(str(tmp_path / "sweep_*.csv"))
    assert len(files) == 1
    rows = list(csv.DictReader(open(files[0])))
    assert [(r["exit_layer"], r["num_speculations"]) for r in rows] == [("2", "2"), ("2", "6"), ("4", "2"), ("4", "6")]
    assert list(rows[0]) == ["exit_layer", "num_speculations", "acceptance_rate", "total_time", "time_per_token", "tokens_per_second"]   # sweep.py:54-61
    assert all(float(r["tokens_per_second"]) > 0 and float(r["total_time"]) > 0 for r in rows)


# --------------------------------------------------------------------------------------------------------------------------
# the slow path's real users: HipEngine vs the oracle stand-in under the same callables
# --------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pair(gpu_device):
    """(record, GPU model, CPU stand-in engine over the same weights)."""
    from fake_engine import FullFakeEngine
    rec = load_struct("tiny_gqa")
    cpu = build_struct_model(rec)
    return rec, build_struct_model(rec, gpu_device), FullFakeEngine(cpu.float()), cpu


def _run(pair, monkeypatch, which, strategy, cfg, **kw):
    """One generation through HipEngine (which = "hip") or through the oracle stand-in (which = "oracle")."""
    from layerskip_amd import hip_strategies
    rec, gpu_model, fake, cpu = pair
    cls = hip_strategies.HipSelfSpeculativeGenerationStrategy if strategy == "self_speculative" else hip_strategies.HipAutoRegressiveGenerationStrategy
    with monkeypatch.context() as mp:
        if which == "oracle":
            mp.setattr(hip_strategies, "get_engine", lambda model, **k: fake)
        strat = cls()
        with torch.inference_mode():
            return strat.generate_token_ids(gpu_model if which == "hip" else cpu, rec["prompt"], rec["eos_token_ids"], cfg, **kw)


def _cfg(rec, strategy, **kw):
    from layerskip_amd import GenerationConfig
    return GenerationConfig(max_steps=32, exit_layer=rec["exit_layer"] if strategy == "self_speculative" else -1,
                            num_speculations=rec["num_speculations"], sample=False, generation_strategy=strategy, **kw)


class ForceEvery:
    """A logits processor that CHANGES decisions: on every `period`-th call it lifts one token of the active vocabulary far
    above everything else in the last row (wide margin by construction).  Calls are counted, so drafts, verify rows and the
    autoregressive rows are hit at positions that depend on the call sequence -- which must therefore be the reference's."""

    def __init__(self, tokens, period=3):
        self.tokens, self.period, self.calls, self.shapes = list(tokens), period, 0, []

    def __call__(self, input_ids, scores):
        self.calls += 1
        self.shapes.append((tuple(input_ids.shape), tuple(scores.shape)))
        if self.calls % self.period == 0:
            scores = scores.clone()
            scores[:, -1, self.tokens[(self.calls // self.period) % len(self.tokens)]] += 60.0
        return scores


@pytest.mark.parametrize("strategy", ["self_speculative", "autoregressive"])
def test_real_logits_processors_on_the_engine(pair, monkeypatch, strategy):
    import transformers
    rec = pair[0]
    active = pair[3].struct_program["active"]
    outs, shapes = [], []
    for which in ("hip", "oracle"):
        force = ForceEvery(active[5:25])
        procs = transformers.LogitsProcessorList([transformers.NoRepeatNGramLogitsProcessor(3), force])      # generator_base.py:77-85
        res = _run(pair, monkeypatch, which, strategy, _cfg(rec, strategy, no_repeat_ngram_size=3), logits_processors=procs)
        outs.append((res.predicted_tokens, res.acceptance_rate))
        shapes.append(force.shapes)
    assert outs[0] == outs[1] and len(outs[0][0]) == 32
    assert shapes[0] == shapes[1]                       # same rows shown to the processors, call for call (SSG:138-139, :172-173)
    free = _run(pair, monkeypatch, "hip", strategy, _cfg(rec, strategy))
    assert (free.predicted_tokens, free.acceptance_rate) != outs[0]      # the processor did change the generation (drafts forced
                                                                         # off the token program are rejected: the acceptance rate moves)


@pytest.mark.parametrize("strategy", ["self_speculative", "autoregressive"])
def test_stopping_criterion_fires_mid_generation(pair, monkeypatch, strategy):
    """`stopping_criteria(input_ids, scores=None)` is evaluated on the NEXT-INPUT tensor after every step (SSG:92-95, ARG:68-71)."""
    import transformers
    rec = pair[0]
    class StopOn(transformers.StoppingCriteria):
        def __init__(self, target):
            self.seen, self.target = [], target

        def __call__(self, input_ids, scores, **kw):
            self.seen.append([int(t) for t in input_ids.reshape(-1).tolist()])
            return torch.tensor([bool((input_ids[:, -1] == self.target).all())])

    never = StopOn(-1)                                   # what the criterion is shown in a full run: the next-input token of every step
    free = _run(pair, monkeypatch, "hip", strategy, _cfg(rec, strategy), stopping_criteria=transformers.StoppingCriteriaList([never])).predicted_tokens
    assert len(free) == 32 and len(never.seen) >= 6
    target = never.seen[4][-1]
    outs, seen = [], []
    for which in ("hip", "oracle"):
        crit = StopOn(target)
        res = _run(pair, monkeypatch, which, strategy, _cfg(rec, strategy), stopping_criteria=transformers.StoppingCriteriaList([crit]))
        outs.append((res.predicted_tokens, res.acceptance_rate))
        seen.append(crit.seen)
    assert outs[0] == outs[1] and seen[0] == seen[1]
    assert 0 < len(outs[0][0]) < 32 and outs[0][0] == free[: len(outs[0][0])]          # cut short, a prefix of the free run


def test_both_streamer_protocols_on_the_engine(pair, monkeypatch):
    rec = pair[0]

    class SpeculativeStandIn:
        """The protocol of the reference's SpeculativeTextStreamer (speculative_streamer.py:31-66): put(value, is_draft) / delete(n)."""

        def __init__(self):
            self.log = []

        def put(self, value, is_draft=False):
            self.log.append(("put", [int(t) for t in torch.as_tensor(value).reshape(-1).tolist()], bool(is_draft)))

        def delete(self, num_tokens, is_draft=False):
            self.log.append(("delete", int(num_tokens)))

        def end(self):
            self.log.append(("end",))

    class PlainStandIn:
        """transformers.TextStreamer's protocol: put(LongTensor) only."""

        def __init__(self):
            self.log = []

        def put(self, value):
            self.log.append(("put", [int(t) for t in torch.as_tensor(value).reshape(-1).tolist()], str(torch.as_tensor(value).dtype)))

        def end(self):
            self.log.append(("end",))

    for cls in (SpeculativeStandIn, PlainStandIn):
        logs, outs = [], []
        for which in ("hip", "oracle"):
            st = cls()
            res = _run(pair, monkeypatch, which, "self_speculative", _cfg(rec, "self_speculative"), streamer=st)
            logs.append(st.log)
            outs.append(res.predicted_tokens)
        assert logs[0] == logs[1] and len(logs[0]) > 6 and outs[0] == outs[1] == rec["bf16"]["spec_tokens"][:32]
    logs = []
    for which in ("hip", "oracle"):
        st = PlainStandIn()
        _run(pair, monkeypatch, which, "autoregressive", _cfg(rec, "autoregressive"), streamer=st)
        logs.append(st.log)
    assert logs[0] == logs[1] and len(logs[0]) == 32
