"""The real-checkpoint / tokenizer branch of the CLI drivers and the multi-process (torchrun-style) path, offline on the CPU.

No checkpoint or tokenizer can be downloaded here, so the fixture IS a checkpoint directory: a tiny `LlamaConfig` model
written by `save_pretrained` (sharded safetensors + index) next to a real `PreTrainedTokenizerFast` (WordLevel, built offline).
Covered:
  * `layerskip_amd.checkpoint.load_layer_range`: only the rank's layers are materialised (reference generate.py:59-64's
    `device_map="auto"` becomes one process per GPU);
  * `benchmark.main()` under a 2-process torchrun-style launch (gloo, CPU stage backend) == the single-process run on the same
    checkpoint: same token ids, same metrics keys, same acceptance (the reference's ranks > 0 `exit()`, generate.py:49-51);
  * `--dataset custom_jsonl` with the reference's `prompt` / `response` text rows and `--template` (reference data.py:175-185),
    and the `{"input_ids"}` rows kept as an extra format;
  * `generate.main()` with a text prompt, the tokenizer's BOS, a `TextStreamer`;
  * `correctness.main()` under the 2-process launch (speculative == autoregressive through ONE serve loop).
The engine stand-ins are tests/fake_engine.py / tests/cpu_stage_backend.py (the oracle's arithmetic in fp32): what is tested is the
host side.  tests/test_gpu_drivers.py runs the same drivers on the HIP engine.
"""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, make_wordlevel_tokenizer

COMMON = ["--device", "cpu", "--max_steps", "20", "--exit_layer", "3", "--num_speculations", "4", "--sample", "False"]
TEMPLATE = "w9 {message} w10"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from layerskip_amd import synthetic
    path = str(tmp_path_factory.mktemp("ckpt"))
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=5, exit_layer=3, late_damping=0.05)
    model.save_pretrained(path, safe_serialization=True, max_shard_size="2MB")     # several shard files + the index
    make_wordlevel_tokenizer(cfg.vocab_size, path)
    g = torch.Generator().manual_seed(7)
    rows = []
    for i in range(4):
        ids = torch.randint(4, cfg.vocab_size, (10 + 3 * i,), generator=g).tolist()
        rows.append({"prompt": " ".join(f"w{t}" for t in ids), "response": "w5 w6"})
    data = os.path.join(path, "prompts.jsonl")
    with open(data, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    ids_data = os.path.join(path, "prompts_ids.jsonl")
    with open(ids_data, "w") as f:
        for r in rows:
            f.write(json.dumps({"input_ids": [1, 9] + [int(w[1:]) for w in r["prompt"].split()] + [10]}) + "\n")
    return {"path": path, "model": model, "data": data, "ids_data": ids_data, "cfg": cfg}


def test_checkpoint_is_sharded_and_a_layer_range_loads_only_its_layers(ckpt):
    from layerskip_amd.checkpoint import load_layer_range
    assert os.path.exists(os.path.join(ckpt["path"], "model.safetensors.index.json"))
    full = ckpt["model"]
    part = load_layer_range(ckpt["path"], (2, 4), device="cpu", dtype=torch.bfloat16)
    assert part.loaded_layer_range == (2, 4)
    for i, layer in enumerate(part.model.layers):
        w = layer.self_attn.q_proj.weight
        if 2 <= i < 4:
            assert w.device.type == "cpu"
            for name, prm in layer.named_parameters():
                assert torch.equal(prm, dict(full.model.layers[i].named_parameters())[name]), (i, name)
        else:
            assert w.device.type == "meta"                       # no storage at all for the other ranks' layers
    for name in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"):
        assert torch.equal(dict(part.named_parameters())[name], dict(full.named_parameters())[name])
    whole = load_layer_range(ckpt["path"], None, device="cpu")
    assert all(torch.equal(a, b) for (_, a), (_, b) in zip(sorted(whole.named_parameters()), sorted(full.named_parameters())))
    with pytest.raises(ValueError):
        load_layer_range(ckpt["path"], (3, 9), device="cpu")


def test_a_device_map_auto_model_is_refused_with_directions(ckpt):
    from layerskip_amd import _lib
    from layerskip_amd.engine import HipEngine
    ckpt["model"].hf_device_map = {"model.layers.0": 0, "model.layers.1": 1}
    try:
        with pytest.raises(_lib.LskError, match="device_map"):
            HipEngine(ckpt["model"])
    finally:
        del ckpt["model"].hf_device_map


def _cpu_backend(model, layer_range, **kw):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_stage_backend import CpuStageBackend
    return CpuStageBackend(model, layer_range=layer_range)


def _rank_worker(rank, world, port, queue, driver, argv):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world)})
    torch.set_num_threads(2)
    import importlib
    import torch.distributed as dist
    mod = importlib.import_module(driver)
    try:
        out = mod.main(argv, backend_factory=_cpu_backend)
        if rank == 0:
            extra = getattr(getattr(mod, "benchmark", None), "last_outputs", None)
            queue.put((out, extra))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _launch(world, driver, argv):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, queue, driver, argv)) for r in range(world)]
    for p in procs:
        p.start()
    out = queue.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, f"a rank exited with {p.exitcode}"
    return out


def _single_process(ckpt, monkeypatch, driver_name, argv, **main_kwargs):
    """The same driver in ONE process: the plugin over the CPU stand-in engine."""
    import importlib.util
    from fake_engine import FullFakeEngine
    from layerskip_amd import hip_strategies
    engines = {}

    def get_engine(model, **kw):
        if id(model) not in engines:
            engines[id(model)] = FullFakeEngine(model.float())
        return engines[id(model)]

    monkeypatch.setattr(hip_strategies, "get_engine", get_engine)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    spec = importlib.util.spec_from_file_location(driver_name, os.path.join(ROOT, driver_name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, driver_name, mod)
    spec.loader.exec_module(mod)
    return mod, mod.main(argv, **main_kwargs)


@pytest.mark.parametrize("world", [2, 3])
def test_benchmark_main_under_a_torchrun_style_launch_equals_the_single_process_run(ckpt, monkeypatch, tmp_path, world):
    argv = ["--model", ckpt["path"], "--dataset", "custom_jsonl", "--data_path", ckpt["data"], "--template", TEMPLATE, "--num_samples", "3",
            "--generation_strategy", "self_speculative", "--output_dir", str(tmp_path)] + COMMON
    mod, single = _single_process(ckpt, monkeypatch, "benchmark", argv)
    single_ids = mod.benchmark.last_outputs
    assert set(single) == {"acceptance_rate", "total_time", "time_per_token", "tokens_per_second"}      # benchmark.py:95-117
    assert len(single_ids) == 3 and all(len(t) == 20 for t in single_ids)
    multi, multi_ids = _launch(world, "benchmark", argv)
    assert multi_ids == single_ids
    assert multi["acceptance_rate"]["mean"] == pytest.approx(single["acceptance_rate"]["mean"], abs=1e-12)
    assert multi["tokens_per_second"]["mean"] > 0
    dumped = [f for f in os.listdir(tmp_path) if f.startswith("benchmark_")]
    assert len(dumped) == 2                                      # one file per run, written by rank 0 only
    with_ranges = [json.load(open(os.path.join(tmp_path, f))) for f in dumped]
    assert sum("layer_ranges" in d for d in with_ranges) == 1


def test_custom_jsonl_text_rows_are_templated_and_tokenised_like_the_reference(ckpt, monkeypatch, tmp_path):
    """reference data.py:175-185 + generator_base.py:104: template.format(message=prompt), tokenizer(..., add_special_tokens=True)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("benchmark", os.path.join(ROOT, "benchmark.py"))
    benchmark = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, "benchmark", benchmark)
    spec.loader.exec_module(benchmark)
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(ckpt["path"])
    b = benchmark.BenchmarkArguments(dataset="custom_jsonl", data_path=ckpt["data"], num_samples=4, random_shuffle=False, template=TEMPLATE)
    text = benchmark.load_prompts(b, ckpt["cfg"].vocab_size, 0, 0, tok)
    b_ids = benchmark.BenchmarkArguments(dataset="custom_jsonl", data_path=ckpt["ids_data"], num_samples=4, random_shuffle=False)
    ids = benchmark.load_prompts(b_ids, ckpt["cfg"].vocab_size, 0, 0, None)
    assert text == ids                                           # <s> w9 ...message... w10
    assert all(p[0] == 1 and p[1] == 9 and p[-1] == 10 for p in text)
    with pytest.raises(ValueError, match="tokenizer"):
        benchmark.load_prompts(b, ckpt["cfg"].vocab_size, 0, 0, None)
    if os.path.isdir("/root/reference"):                         # the reference's own loader on the same file (build container only)
        sys.path.insert(0, os.path.join(ROOT))
        from oracle import ref_shim
        ref_shim.load_reference()
        try:
            import data as ref_data
        except Exception:                                       # noqa: BLE001 -- data.py imports `datasets`, present here; anything else: skip
            return
        examples = ref_data.prepare_custom(ckpt["data"], template=TEMPLATE)
        theirs = [tok(e.input, return_tensors="pt", add_special_tokens=True)["input_ids"].tolist()[0] for e in examples]
        assert theirs == text


def test_generate_main_with_a_text_prompt_and_a_text_streamer(ckpt, monkeypatch, capsys):
    argv = ["--model", ckpt["path"], "--generation_strategy", "self_speculative"] + COMMON
    prompt = "w11 w250 w37 w999 w4 w5"
    _, res = _single_process(ckpt, monkeypatch, "generate", argv, lines=[prompt])
    out = capsys.readouterr().out
    assert len(res) == 1 and res[0].num_tokens_generated == 20
    toks = res[0].generation_strategy_result.predicted_tokens
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(ckpt["path"])
    assert res[0].decoded_prediction == tok.decode(toks)        # generator_base.py:119-121
    assert res[0].decoded_prediction in out                     # printed by the driver; the TextStreamer printed the words as they came
    assert "Tokens per second:" in out and "Acceptance rate:" in out
    # the prompt went through the tokenizer WITH its BOS (generator_base.py:104): the same ids given directly give the same tokens
    from layerskip_amd import GenerationConfig, TokenGenerator
    from layerskip_amd.checkpoint import load_layer_range
    from layerskip_amd.cli.common import make_strategy
    ids = tok(prompt, return_tensors="pt", add_special_tokens=True)["input_ids"].tolist()[0]
    assert ids[0] == tok.bos_token_id == 1 and len(ids) == 7
    cfg = GenerationConfig(max_steps=20, exit_layer=3, num_speculations=4, sample=False, generation_strategy="self_speculative")
    model = load_layer_range(ckpt["path"], None, device="cpu")
    direct = TokenGenerator(tok, model, make_strategy(cfg)).generate_from_ids(ids, [tok.eos_token_id], cfg)
    assert direct.generation_strategy_result.predicted_tokens == toks


def test_stop_words_become_a_stop_string_criterion(ckpt, monkeypatch):
    """generator_base.py:87-95: `stop_words` -> `StopStringCriteria(tokenizer, stop_words)`, evaluated on the step's next input
    token (SSG:92-95).  The generation ends after the first step whose last token spells a stop word."""
    import transformers
    from fake_engine import FullFakeEngine
    from layerskip_amd import GenerationConfig, TokenGenerator, hip_strategies
    from layerskip_amd.checkpoint import load_layer_range
    from layerskip_amd.cli.common import make_strategy
    tok = transformers.AutoTokenizer.from_pretrained(ckpt["path"])
    model = load_layer_range(ckpt["path"], None, device="cpu")
    engine = FullFakeEngine(model.float())
    monkeypatch.setattr(hip_strategies, "get_engine", lambda m, **k: engine)
    cfg = GenerationConfig(max_steps=20, exit_layer=3, num_speculations=4, sample=False, generation_strategy="self_speculative")
    strat = make_strategy(cfg)
    steps = []
    inner = strat.single_step_speculation

    def spy(**kw):
        r = inner(**kw)
        steps.append(list(r[1]))
        return r

    gen = TokenGenerator(tok, model, strat)
    prompt = "w11 w250 w37 w999 w4 w5"
    # a free run through the step path (an always-false criterion keeps it off the fused call) to learn the step boundaries
    strat.single_step_speculation = spy
    free = gen.generate(prompt, cfg, streamer=transformers.TextStreamer(tok, skip_prompt=True)).generation_strategy_result.predicted_tokens
    assert len(steps) >= 3
    last_of_step_2 = steps[1][-1]
    steps.clear()
    cfg_stop = GenerationConfig(max_steps=20, exit_layer=3, num_speculations=4, sample=False, generation_strategy="self_speculative",
                                stop_words=[tok.convert_ids_to_tokens(last_of_step_2)])
    assert isinstance(gen._criteria(cfg_stop)[0], transformers.StopStringCriteria)
    stopped = gen.generate(prompt, cfg_stop).generation_strategy_result.predicted_tokens
    first_hit = next(i for i, st in enumerate(steps) if tok.convert_ids_to_tokens(last_of_step_2) in tok.convert_ids_to_tokens(st[-1]))
    assert stopped == steps[first_hit] and stopped == free[: len(stopped)] and len(stopped) < 20


def test_correctness_main_under_a_torchrun_style_launch(ckpt, tmp_path):
    argv = ["--model", ckpt["path"], "--dataset", "custom_jsonl", "--data_path", ckpt["ids_data"], "--num_samples", "2",
            "--output_dir", str(tmp_path)] + COMMON
    code, _ = _launch(2, "correctness", argv)
    assert code == 0                                             # speculative == autoregressive on every sample (correctness.py:82-88)
    out = [json.load(open(os.path.join(tmp_path, f))) for f in os.listdir(tmp_path) if f.startswith("correctness_")]
    assert out == [{"errors": 0, "error_pct": 0.0, "num_samples": 2}]


def test_tied_embeddings_checkpoint_loads_with_the_head_tied(tmp_path):
    """llama3.2-1B's layout (config #1): `tie_word_embeddings` checkpoints store the embedding only; the loader ties the head to it."""
    from layerskip_amd import synthetic
    from layerskip_amd.checkpoint import load_layer_range
    cfg = synthetic.make_config("tiny-d64")
    assert cfg.tie_word_embeddings
    model = synthetic.build_model(cfg, seed=2, exit_layer=2, late_damping=0.1)
    path = str(tmp_path / "tied")
    model.save_pretrained(path, safe_serialization=True)
    part = load_layer_range(path, (1, 3), device="cpu")
    assert part.lm_head.weight.data_ptr() == part.model.embed_tokens.weight.data_ptr()
    assert torch.equal(part.lm_head.weight, model.model.embed_tokens.weight)
    assert part.model.layers[0].mlp.up_proj.weight.device.type == "meta" and part.model.layers[2].mlp.up_proj.weight.device.type == "cpu"
    # and the partial model decodes like the whole one through the CPU stage backend (rank-style: layers [1, 3) only hold weights)
    assert torch.equal(part.model.layers[1].self_attn.q_proj.weight, model.model.layers[1].self_attn.q_proj.weight)


def test_loader_error_paths_name_the_problem(tmp_path, ckpt):
    """What a user gets for the checkpoints the loader cannot read: a directory without safetensors files (the reference's
    `from_pretrained` would try .bin files), a non-Llama model type, a shard that lacks a tensor the rank owns; and the single-file
    layout (no index) loads like the sharded one."""
    import shutil
    from safetensors.torch import load_file, save_file
    from layerskip_amd.checkpoint import load_layer_range
    empty = tmp_path / "empty"
    empty.mkdir()
    shutil.copy(os.path.join(ckpt["path"], "config.json"), empty / "config.json")
    with pytest.raises(FileNotFoundError, match="safetensors"):
        load_layer_range(str(empty), None, device="cpu")

    # one file, no index: same tensors
    single = tmp_path / "single"
    ckpt["model"].save_pretrained(str(single), safe_serialization=True)          # default shard size: one model.safetensors
    assert os.path.exists(single / "model.safetensors") and not os.path.exists(single / "model.safetensors.index.json")
    a = load_layer_range(str(single), (0, 2), device="cpu")
    b = load_layer_range(ckpt["path"], (0, 2), device="cpu")
    assert torch.equal(a.model.layers[1].mlp.down_proj.weight, b.model.layers[1].mlp.down_proj.weight)
    assert a.model.layers[3].mlp.down_proj.weight.device.type == "meta"

    # a tensor of an OWNED layer is gone: named in the error; a rank that does not own the layer is not affected
    broken = tmp_path / "broken"
    shutil.copytree(single, broken)
    tensors = load_file(str(broken / "model.safetensors"))
    del tensors["model.layers.1.self_attn.k_proj.weight"]
    save_file(tensors, str(broken / "model.safetensors"), metadata={"format": "pt"})
    with pytest.raises(KeyError, match="model.layers.1.self_attn.k_proj.weight"):
        load_layer_range(str(broken), (0, 2), device="cpu")
    ok = load_layer_range(str(broken), (2, 4), device="cpu")
    assert ok.model.layers[2].self_attn.k_proj.weight.device.type == "cpu"

    # not a Llama decoder
    other = tmp_path / "other"
    shutil.copytree(single, other)
    cfg = json.load(open(other / "config.json"))
    cfg["model_type"] = "gpt2"
    cfg["architectures"] = ["GPT2LMHeadModel"]
    json.dump(cfg, open(other / "config.json", "w"))
    with pytest.raises((ValueError, KeyError)):
        load_layer_range(str(other), None, device="cpu")


def test_a_hub_id_resolves_to_its_cached_snapshot_and_a_middle_rank_holds_neither_embedding_nor_head(ckpt, tmp_path, monkeypatch):
    """`--model facebook/layerskip-llama2-7B` is how the reference is used (generate.py:59-64, README): a name that is not a directory is
    resolved to the snapshot directory of the hub cache (offline: a cache lookup).  And a middle rank of a pipeline materialises its
    layers and the final norm only: the embedding and the lm_head (2 x 2.1 GB at llama3-70B) stay on the meta device."""
    import shutil
    import huggingface_hub
    from layerskip_amd.checkpoint import load_layer_range, resolve_checkpoint_dir
    cache = tmp_path / "hub"
    snap = cache / "models--acme--tiny-layerskip" / "snapshots" / "0123abcd"
    snap.parent.mkdir(parents=True)
    shutil.copytree(ckpt["path"], snap)
    refs = cache / "models--acme--tiny-layerskip" / "refs"
    refs.mkdir()
    (refs / "main").write_text("0123abcd")
    monkeypatch.setattr(huggingface_hub.constants, "HF_HUB_CACHE", str(cache))
    monkeypatch.setattr(huggingface_hub.constants, "HF_HUB_OFFLINE", True)
    assert os.path.samefile(resolve_checkpoint_dir("acme/tiny-layerskip"), snap)
    assert resolve_checkpoint_dir(ckpt["path"]) == ckpt["path"]
    part = load_layer_range("acme/tiny-layerskip", (2, 4), device="cpu", embed=False, head=False)        # a middle rank
    assert part.model.embed_tokens.weight.is_meta and part.lm_head.weight.is_meta
    assert not part.model.norm.weight.is_meta and not part.model.layers[2].mlp.up_proj.weight.is_meta
    assert torch.equal(part.model.layers[3].self_attn.o_proj.weight, ckpt["model"].model.layers[3].self_attn.o_proj.weight)
    last = load_layer_range("acme/tiny-layerskip", (4, 6), device="cpu", embed=False, head=True)          # the last rank: the verify head
    assert last.model.embed_tokens.weight.is_meta and torch.equal(last.lm_head.weight, ckpt["model"].lm_head.weight)
    with pytest.raises(FileNotFoundError, match="neither a checkpoint directory nor a hub id"):
        resolve_checkpoint_dir("acme/not-in-the-cache")
    # the CLI's loader takes the same name (tokenizer included)
    from layerskip_amd.cli.common import Arguments, SyntheticArguments, load_model_and_tokenizer
    model, tok = load_model_and_tokenizer(Arguments(model="acme/tiny-layerskip"), SyntheticArguments(device="cpu"), 3)
    assert tok is not None and torch.equal(model.lm_head.weight, ckpt["model"].lm_head.weight)
