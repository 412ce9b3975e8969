"""`-m gpu` twin of tests/test_cli_checkpoint.py: the CLI drivers on a REAL checkpoint directory (sharded safetensors written by
`save_pretrained` + an offline-built `PreTrainedTokenizerFast`) through the HIP engine, single process and under a
torchrun-style multi-process launch (every rank loads ITS layer range with layerskip_amd.checkpoint.load_layer_range and owns a
HipEngine; on this 1-GPU box the ranks share device 0 and exchange rows through gloo, on a multi-GPU node the same launch line
runs one rank per GPU over RCCL -- `init_distributed` decides)."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, make_wordlevel_tokenizer

pytestmark = pytest.mark.gpu

COMMON = ["--device", "cuda:0", "--max_steps", "24", "--exit_layer", "3", "--num_speculations", "5", "--sample", "False"]


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    """A structured tiny checkpoint (every decision of a greedy run has a wide margin: token-exact comparisons are meaningful)."""
    from layerskip_amd import synthetic
    path = str(tmp_path_factory.mktemp("ckpt_gpu"))
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_structured_model(cfg, seed=4, exit_layer=3, override_frac=0.3)
    model.save_pretrained(path, safe_serialization=True, max_shard_size="2MB")
    make_wordlevel_tokenizer(cfg.vocab_size, path)
    data = os.path.join(path, "prompts.jsonl")
    with open(data, "w") as f:
        for i in range(3):
            ids = synthetic.make_struct_prompt(model.struct_program, 14 + 5 * i, i)
            f.write(json.dumps({"prompt": " ".join(f"w{t}" for t in ids), "response": ""}) + "\n")
    return {"path": path, "data": data, "cfg": cfg}


def _rank_worker(rank, world, port, queue, driver, argv):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import importlib
    import torch.distributed as dist
    mod = importlib.import_module(driver)
    try:
        out = mod.main(argv)
        if rank == 0:
            queue.put((out, getattr(getattr(mod, "benchmark", None), "last_outputs", None)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _launch(world, driver, argv):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, queue, driver, argv)) for r in range(world)]
    for p in procs:
        p.start()
    out = queue.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def _load_driver(name, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setitem(sys.modules, name, mod)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("world", [2, 3])
def test_benchmark_main_on_a_real_checkpoint_one_process_and_torchrun_style(gpu_device, ckpt, monkeypatch, tmp_path, world):
    argv = ["--model", ckpt["path"], "--dataset", "custom_jsonl", "--data_path", ckpt["data"], "--num_samples", "3", "--random_shuffle", "False",
            "--generation_strategy", "self_speculative", "--output_dir", str(tmp_path)] + COMMON
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    benchmark = _load_driver("benchmark", monkeypatch)
    single = benchmark.main(argv)                                 # from_pretrained-style loading, tokenizer, text prompts: ONE HipEngine
    single_ids = benchmark.benchmark.last_outputs
    assert len(single_ids) == 3 and all(len(t) == 24 for t in single_ids)
    assert 0.0 < single["acceptance_rate"]["mean"] < 1.0
    multi, multi_ids = _launch(world, "benchmark", argv)          # the same command line, one process per rank
    assert multi_ids == single_ids
    assert multi["acceptance_rate"]["mean"] == pytest.approx(single["acceptance_rate"]["mean"], abs=1e-12)
    assert set(multi) == {"acceptance_rate", "total_time", "time_per_token", "tokens_per_second"}


@pytest.mark.parametrize("flags", [["--sample", "True", "--temperature", "0.7", "--top_k", "50", "--top_p", "0.95"],
                                   ["--sample", "False", "--no_repeat_ngram_size", "2"],
                                   ["--sample", "True", "--no_repeat_ngram_size", "3", "--top_k", "0"]])
def test_the_reference_readme_flags_run_on_the_pipeline(gpu_device, ckpt, monkeypatch, tmp_path, flags):
    """The reference's README command lines pass `--sample True` (its default, generator_base.py:39) and benchmark.py's
    no_repeat_ngram_size (generator_base.py:77-85): on the layer pipeline they give the one-process engine's tokens -- draw for draw
    under the same --seed (sampled acceptance split over the ranks), logits processors on rank 0 with the last rank's logits rows."""
    argv = ["--model", ckpt["path"], "--dataset", "custom_jsonl", "--data_path", ckpt["data"], "--num_samples", "3", "--random_shuffle", "False",
            "--generation_strategy", "self_speculative", "--output_dir", str(tmp_path), "--seed", "7"] + [a for a in COMMON if a not in ("--sample", "False")] + flags
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    benchmark = _load_driver("benchmark", monkeypatch)
    single = benchmark.main(argv)
    single_ids = benchmark.benchmark.last_outputs
    assert len(single_ids) == 3 and all(len(t) == 24 for t in single_ids)
    multi, multi_ids = _launch(2, "benchmark", argv)
    assert multi_ids == single_ids
    assert multi["acceptance_rate"]["mean"] == pytest.approx(single["acceptance_rate"]["mean"], abs=1e-12)


def test_correctness_main_torchrun_style_and_generate_repl_on_the_engine(gpu_device, ckpt, monkeypatch, capsys, tmp_path):
    argv = ["--model", ckpt["path"], "--dataset", "custom_jsonl", "--data_path", ckpt["data"], "--num_samples", "2", "--output_dir", str(tmp_path)] + COMMON
    code, _ = _launch(2, "correctness", argv)
    assert code == 0
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    generate = _load_driver("generate", monkeypatch)
    prompt = open(ckpt["data"]).readline()
    prompt = json.loads(prompt)["prompt"]
    res = generate.main(["--model", ckpt["path"], "--generation_strategy", "self_speculative"] + COMMON, lines=[prompt])
    out = capsys.readouterr().out
    assert len(res) == 1 and res[0].num_tokens_generated == 24 and res[0].decoded_prediction in out
    # stop_words -> StopStringCriteria on the engine's step path (generator_base.py:87-95)
    import transformers
    from layerskip_amd import GenerationConfig, TokenGenerator
    from layerskip_amd.checkpoint import load_layer_range
    from layerskip_amd.cli.common import make_strategy
    tok = transformers.AutoTokenizer.from_pretrained(ckpt["path"])
    model = load_layer_range(ckpt["path"], None, device=gpu_device)
    cfg = GenerationConfig(max_steps=24, exit_layer=3, num_speculations=5, sample=False, generation_strategy="self_speculative")
    gen = TokenGenerator(tok, model, make_strategy(cfg))
    free = gen.generate(prompt, cfg).generation_strategy_result.predicted_tokens
    assert free == res[0].generation_strategy_result.predicted_tokens
    word = tok.convert_ids_to_tokens(free[-1])
    cfg_stop = GenerationConfig(max_steps=24, exit_layer=3, num_speculations=5, sample=False, generation_strategy="self_speculative", stop_words=[word])
    stopped = gen.generate(prompt, cfg_stop).generation_strategy_result.predicted_tokens
    assert stopped == free[: len(stopped)] and 0 < len(stopped) <= 24
