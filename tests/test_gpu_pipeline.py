"""The layer-range pipeline with REAL HIP engines: two / three processes sharing the one GPU of the box (gloo for the
host-side exchange; on a multi-GPU node the same protocol runs over RCCL with the rows sent straight from the engines'
buffers), against the single-process fused engine.  Covers the device-resident draft block, the header protocol and
the optimistic continuation (always right on a checkpoint whose drafts are all accepted, discarded on one that rejects)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, queue, override_frac):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from layerskip_amd import synthetic
        from layerskip_amd.engine import HipEngine
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        dev = torch.device("cuda:0")
        cfg = synthetic.make_config("tiny-gqa")
        E, S = 3, 6
        part = plan_partition(cfg.num_hidden_layers, E, world)
        model = synthetic.build_structured_model(cfg, seed=4, exit_layer=E, override_frac=override_frac, layer_range=part[rank], device=dev)
        eng = HipEngine(model, max_ctx=512, max_prompt=64, layer_range=part[rank])
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E, comm_device=torch.device("cpu"))
        prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 40, S)
        if rank == 0:
            queue.put((res.predicted_tokens, res.acceptance_rate, res.steps, dec.stats()))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,override_frac", [(2, 0.0), (2, 0.3), (3, 0.3)])
def test_pipeline_of_hip_engines_equals_the_fused_engine(gpu_device, world, override_frac):
    from layerskip_amd import GenerationConfig, synthetic
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue, override_frac)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, rate, steps, stats = queue.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_structured_model(cfg, seed=4, exit_layer=3, override_frac=override_frac, device=gpu_device)
    prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
    strat = HipSelfSpeculativeGenerationStrategy()
    want = strat.generate_token_ids(model, prompt, [cfg.vocab_size],
                                    GenerationConfig(max_steps=40, exit_layer=3, num_speculations=6, sample=False))
    assert tokens == want.predicted_tokens
    assert [tuple(s) for s in steps] == [tuple(s) for s in strat.last_steps]
    assert rate == want.acceptance_rate
    if override_frac == 0.0:
        assert stats["optimistic_attempts"] >= 3 and stats["optimistic_hits"] == stats["optimistic_attempts"]
    else:
        assert stats["optimistic_attempts"] > stats["optimistic_hits"]


SAMPLING = dict(temperature=0.8, top_k=50, top_p=0.9, seed=20240917, offset=(1 << 45) + 3)


def _sampled_worker(rank, world, port, queue, shape, E, S, max_steps, top_k):
    """sample=True on the pipeline (the reference's default, generator_base.py:39): rank 0 drafts with draws, the last rank runs the
    acceptance test on the header's p_i(x_i) and returns q_n, rank 0 draws the residual token."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from layerskip_amd import synthetic
        from layerskip_amd.engine import HipEngine
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, Sampling, plan_partition
        dev = torch.device("cuda:0")
        cfg = synthetic.make_config(shape)
        part = plan_partition(cfg.num_hidden_layers, E, world)
        model = synthetic.build_model(cfg, seed=2, exit_layer=E, late_damping=0.2, dtype=torch.bfloat16, device=dev, gen_device="cpu", layer_range=part[rank])
        eng = HipEngine(model, max_ctx=512, max_prompt=64, layer_range=part[rank])
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E, comm_device=torch.device("cpu"))
        prompt = synthetic.make_prompt(cfg.vocab_size, 23, 5)
        sm = dict(SAMPLING, top_k=top_k)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], max_steps, S, sampling=Sampling(**sm) if rank == 0 else None)
        if rank == 0:
            queue.put((res.predicted_tokens, res.steps))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,E,S,top_k", [(2, "tiny-gqa", 3, 6, 50), (3, "tiny-gqa", 3, 6, 0), (2, "slice-1B", 2, 4, 0), (2, "slice-1B", 2, 4, 40)])
def test_sampled_pipeline_of_hip_engines_is_draw_for_draw_the_fused_engine(gpu_device, world, shape, E, S, top_k):
    """Same (seed, offset) -> the SAME tokens and per-step (drafts, matches) as lsk_spec_generate_sampled on one engine: the split
    acceptance (scalars forward, one probability row back) makes the same draws and comparisons as lsk_accept_sampled_kernel.
    slice-1B: V = 128 256, the multi-workgroup histogram form of the draw (csrc/lsk_sample.h), with and without top-k."""
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    max_steps = 40
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_sampled_worker, args=(r, world, port, queue, shape, E, S, max_steps, top_k)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, steps = queue.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=2, exit_layer=E, late_damping=0.2, dtype=torch.bfloat16, device=gpu_device, gen_device="cpu")
    eng = HipEngine(model, max_ctx=512, max_prompt=64)
    prompt = synthetic.make_prompt(cfg.vocab_size, 23, 5)
    sm = dict(SAMPLING, top_k=top_k)
    want, matches, drafts, want_steps = eng.spec_generate_sampled(prompt, S, E, [cfg.vocab_size], max_steps, sm["temperature"], sm["top_k"],
                                                                  sm["top_p"], sm["seed"], sm["offset"])
    eng.close()
    assert tokens == want
    assert [tuple(s) for s in steps] == [tuple(s) for s in want_steps]
    assert any(n < td for td, n in want_steps) and any(n == td and td > 0 for td, n in want_steps)   # rejections AND full acceptances occurred


def _fullsize_worker(rank, world, port, queue, model_name, prompt_len, max_steps, sampled):
    """BASELINE config #4 at FULL size: every rank materialises only its layer range (device generator: the same bits in every
    process), releases the unpacked originals as it packs them, and the ranks share device 0 over gloo on a 1-GPU box (one rank per
    device over RCCL when the box has them: tools/pp_identity.py is the same run as a measurement tool)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import datetime
    one_per_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if one_per_gpu else 0)
    torch.cuda.set_device(dev)
    if one_per_gpu:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=10))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    try:
        from layerskip_amd import synthetic
        from layerskip_amd.engine import HipEngine
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, Sampling, plan_partition
        cfg = synthetic.make_config(model_name)
        E, S = synthetic.default_exit_layer(model_name), synthetic.default_num_speculations(model_name)
        part = plan_partition(cfg.num_hidden_layers, E, world, balance="memory")          # [0, 20) + [20, 40)
        model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=dev, gen_device=dev,
                                      layer_range=part[rank])
        eng = HipEngine(model, max_ctx=prompt_len + max_steps + 2 * S + 32, max_prompt=prompt_len, layer_range=part[rank], release_weights=True)
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E, comm_device=dev if one_per_gpu else torch.device("cpu"))
        prompt = synthetic.make_prompt(cfg.vocab_size, prompt_len, 0) if rank == 0 else None
        sm = Sampling(**SAMPLING) if (sampled and rank == 0) else None
        res = dec.generate(prompt, [cfg.vocab_size], max_steps, S, sampling=sm)
        if rank == 0:
            queue.put((res.predicted_tokens, res.steps, [list(p) for p in part]))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sampled", [False, True])
def test_llama2_13b_at_full_size_on_two_pipeline_ranks_equals_one_engine(gpu_device, sampled):
    """BASELINE config #4 (llama2-13B, exit_layer 10, 8 speculations, 2 ranks: [0, 20) + [20, 40)) at FULL size, 128 new tokens of a
    random-init checkpoint whose drafts ARE rejected (acceptance ~0.6): ids and the per-step (drafts, matches) trace identical to ONE
    fused engine holding all 40 layers -- greedy, and under sample=True (the reference's default flags) draw for draw."""
    free, _ = torch.cuda.mem_get_info()
    if free < 64 * 2 ** 30:
        pytest.skip("needs 64 GB of free HBM (2 x 13 GB of packed layer ranges, then 26 GB for the one-engine run)")
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    name, prompt_len, max_steps, world = "llama2-13B", 96, 128, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_fullsize_worker, args=(r, world, port, queue, name, prompt_len, max_steps, sampled)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, steps, part = queue.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert part == [[0, 20], [20, 40]]
    cfg = synthetic.make_config(name)
    E, S = synthetic.default_exit_layer(name), synthetic.default_num_speculations(name)
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=prompt_len + max_steps + 2 * S + 32, max_prompt=prompt_len, release_weights=True)
    prompt = synthetic.make_prompt(cfg.vocab_size, prompt_len, 0)
    if sampled:
        sm = SAMPLING
        want, _, _, want_steps = eng.spec_generate_sampled(prompt, S, E, [cfg.vocab_size], max_steps, sm["temperature"], sm["top_k"], sm["top_p"],
                                                           sm["seed"], sm["offset"])
    else:
        want, _, _, want_steps = eng.spec_generate(prompt, S, E, [cfg.vocab_size], max_steps)
    eng.close()
    del eng, model
    torch.cuda.empty_cache()
    assert len(want) == max_steps and tokens == want
    assert [tuple(t) for t in steps] == [tuple(t) for t in want_steps]
    assert any(n < td for td, n in want_steps), "the run must contain rejected drafts"
    assert all(td <= S for td, _ in want_steps) and max(td for td, _ in want_steps) == S        # 9-row verify blocks


def _fullsize_both_worker(rank, world, port, queue, model_name, prompt_len, max_steps):
    """`_fullsize_worker` for a checkpoint whose build dominates: ONE model / engine per rank, a greedy generation and then a sampled one
    through the same decoder (the serve loop of the late ranks takes the mode from each generation's set-up broadcast)."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import datetime
    one_per_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if one_per_gpu else 0)
    torch.cuda.set_device(dev)
    if one_per_gpu:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=15))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=15))
    try:
        from layerskip_amd import synthetic
        from layerskip_amd.engine import HipEngine
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, Sampling, plan_partition
        cfg = synthetic.make_config(model_name)
        E, S = synthetic.default_exit_layer(model_name), synthetic.default_num_speculations(model_name)
        part = plan_partition(cfg.num_hidden_layers, E, world, balance="memory")          # [0, 12) + 7 x 9..10 at 70B / 8
        t0 = time.time()
        model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=dev, gen_device=dev,
                                      layer_range=part[rank])
        eng = HipEngine(model, max_ctx=prompt_len + max_steps + 2 * S + 32, max_prompt=prompt_len, layer_range=part[rank], release_weights=True)
        torch.cuda.synchronize()
        build_s = time.time() - t0
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E, comm_device=dev if one_per_gpu else torch.device("cpu"))
        dec.warm_transport()
        prompt = synthetic.make_prompt(cfg.vocab_size, prompt_len, 0) if rank == 0 else None
        runs = []
        for sampled in (False, True):
            sm = Sampling(**SAMPLING) if (sampled and rank == 0) else None
            dist.barrier()
            t0 = time.perf_counter()
            res = dec.generate(prompt, [cfg.vocab_size], max_steps, S, sampling=sm)
            runs.append({"tokens": res.predicted_tokens, "steps": [list(t) for t in res.steps], "seconds": time.perf_counter() - t0})
        stats = [None] * world
        dist.all_gather_object(stats, dict(dec.stats(), build_s=round(build_s, 2), packed_gb=round(sum(
            t.numel() for pk in eng._packed if pk is not None for t in pk[:4]) / 2 ** 30, 2)))
        if rank == 0:
            queue.put((runs, [list(p) for p in part], stats, "nccl" if one_per_gpu else "gloo (ranks share device 0)"))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_llama2_70b_at_full_size_on_eight_pipeline_ranks_equals_one_engine(gpu_device):
    """BASELINE config #5 (llama2-70B, exit_layer 12, 12 speculations -- 13-row verify blocks -- on 8 ranks: [0, 12) + 7 x 9..10 layers;
    the reference runs it through `device_map="auto"`, generate.py:59-64, with its default sample=True, generator_base.py:39) at FULL
    size: 96 new tokens of a random-init checkpoint whose drafts ARE rejected, greedy AND sampled, ids and per-step (drafts, matches)
    traces identical to ONE engine holding all 80 layers.  Seven of the eight ranks hold neither the embedding's consumer nor a head; the
    sampled protocol's q_n row crosses them.  The evidence block goes to gpurun_out/ (copied to profiles/r06_pp_70B_8ranks_one_gpu.json)."""
    import json
    import time
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2 ** 30:
        pytest.skip("needs 150 GB of free HBM (8 x 17-22 GB of packed layer ranges, then 140 GB for the one-engine run)")
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    name, prompt_len, max_steps, world = "llama2-70B", 96, 96, 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    t_spawn = time.time()
    procs = [ctx.Process(target=_fullsize_both_worker, args=(r, world, port, queue, name, prompt_len, max_steps)) for r in range(world)]
    for p in procs:
        p.start()
    runs, part, stats, transport = queue.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    pipeline_wall_s = time.time() - t_spawn
    assert part[0] == [0, 12] and len(part) == 8 and part[-1][1] == 80 and all(9 <= b - a <= 10 for a, b in part[1:])
    cfg = synthetic.make_config(name)
    E, S = synthetic.default_exit_layer(name), synthetic.default_num_speculations(name)
    assert (E, S) == (12, 12)
    t0 = time.time()
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=0.03, dtype=torch.bfloat16, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=prompt_len + max_steps + 2 * S + 32, max_prompt=prompt_len, release_weights=True)
    torch.cuda.synchronize()
    one_build_s = time.time() - t0
    prompt = synthetic.make_prompt(cfg.vocab_size, prompt_len, 0)
    t0 = time.perf_counter()
    want_g, _, _, steps_g = eng.spec_generate(prompt, S, E, [cfg.vocab_size], max_steps)
    one_greedy_s = time.perf_counter() - t0
    sm = SAMPLING
    t0 = time.perf_counter()
    want_s, _, _, steps_s = eng.spec_generate_sampled(prompt, S, E, [cfg.vocab_size], max_steps, sm["temperature"], sm["top_k"], sm["top_p"],
                                                      sm["seed"], sm["offset"])
    one_sampled_s = time.perf_counter() - t0
    eng.close()
    del eng, model
    torch.cuda.empty_cache()
    for run, want, want_steps, label in ((runs[0], want_g, steps_g, "greedy"), (runs[1], want_s, steps_s, "sampled")):
        assert len(want) == max_steps and run["tokens"] == want, label
        assert [tuple(t) for t in run["steps"]] == [tuple(t) for t in want_steps], label
        assert any(n < td for td, n in want_steps), f"{label}: the run must contain rejected drafts"
        assert max(td for td, _ in want_steps) == S, f"{label}: 13-row verify blocks"
    assert all(st["hops"] == len(steps_g) + 1 + len(steps_s) + 1 for st in stats[1:])       # every late rank served every block + the two stop messages
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r06_pp_70B_8ranks_one_gpu.json"), "w") as f:
            json.dump({"model": name, "world": world, "partition": part, "transport": transport, "prompt_len": prompt_len, "new_tokens": max_steps,
                       "exit_layer": E, "num_speculations": S,
                       "greedy": {"identical_to_one_engine": True, "steps": len(steps_g), "acceptance": round(sum(n for _, n in steps_g) / sum(td for td, _ in steps_g), 4),
                                  "pipeline_tokens_per_s": round(max_steps / runs[0]["seconds"], 1), "one_engine_tokens_per_s": round(max_steps / one_greedy_s, 1)},
                       "sampled": {"identical_to_one_engine": True, "steps": len(steps_s), "acceptance": round(sum(n for _, n in steps_s) / sum(td for td, _ in steps_s), 4),
                                   "sampling": {k: SAMPLING[k] for k in ("temperature", "top_k", "top_p")},
                                   "pipeline_tokens_per_s": round(max_steps / runs[1]["seconds"], 1), "one_engine_tokens_per_s": round(max_steps / one_sampled_s, 1)},
                       "per_rank": stats, "pipeline_wall_s_incl_spawn_and_build": round(pipeline_wall_s, 1), "one_engine_build_s": round(one_build_s, 1),
                       "note": "tests/test_gpu_pipeline.py::test_llama2_70b_at_full_size_on_eight_pipeline_ranks_equals_one_engine"}, f, indent=1)
    except OSError:
        pass


def _nccl_worker(rank, world, port, queue):
    """One rank per DEVICE, backend nccl (= RCCL): rows go straight from / into the engines' message buffers over xGMI."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    import datetime
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=5))
    try:
        from layerskip_amd import synthetic
        from layerskip_amd.engine import HipEngine
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        cfg = synthetic.make_config("tiny-gqa")
        E, S = 3, 6
        part = plan_partition(cfg.num_hidden_layers, E, world)
        model = synthetic.build_structured_model(cfg, seed=4, exit_layer=E, override_frac=0.3, layer_range=part[rank], device=dev)
        eng = HipEngine(model, max_ctx=512, max_prompt=64, layer_range=part[rank])
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E)          # comm device = the engine's device: direct send / recv
        prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 40, S)
        stats = [None] * world
        dist.all_gather_object(stats, dec.stats())
        if rank == 0:
            queue.put((res.predicted_tokens, res.acceptance_rate, res.steps, stats))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: one pipeline rank per device over RCCL")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_pipeline_over_rccl_one_rank_per_gpu(world):
    """The transport the gloo twins stand in for, the first time a multi-GPU node runs this suite: backend nccl, one rank per
    device, rows sent straight from the engines' buffers -- token identity with the fused single-GPU engine."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"{world} GPUs needed")
    from layerskip_amd import GenerationConfig, synthetic
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    if world - 1 > synthetic.make_config("tiny-gqa").num_hidden_layers - 3:
        pytest.skip("more ranks than late layers in the tiny checkpoint")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, rate, steps, stats = queue.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = synthetic.make_config("tiny-gqa")
    dev = torch.device("cuda:0")
    model = synthetic.build_structured_model(cfg, seed=4, exit_layer=3, override_frac=0.3, device=dev)
    prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
    strat = HipSelfSpeculativeGenerationStrategy()
    want = strat.generate_token_ids(model, prompt, [cfg.vocab_size], GenerationConfig(max_steps=40, exit_layer=3, num_speculations=6, sample=False))
    assert tokens == want.predicted_tokens and rate == want.acceptance_rate
    assert [tuple(s) for s in steps] == [tuple(s) for s in strat.last_steps]
    assert all(st["hops"] == len(steps) + 1 for st in stats[1:])          # every late rank served every block and the stop message


def test_sampled_pipeline_kernels_flag_a_protocol_out_of_step_and_check_their_buffers(gpu_device):
    """One rank, the sampled pipeline's C-ABI calls by hand: (1) the last rank's acceptance kernel refuses a block whose header carries
    another Philox offset than its own step counter says (result word 22; rank 0 raises on it -- a message out of step must not pass
    as a draw); (2) with matching offsets the block is consistent and `lsk_pipeline_residual` leaves a block that is not pending
    untouched; (3) a result buffer that is too small is an error, not an overrun."""
    import ctypes
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import HipEngine
    from layerskip_amd.pipeline import RES_ERROR, RES_PENDING
    cfg = synthetic.make_config("tiny-gqa")
    E, S = 3, 5
    model = synthetic.build_model(cfg, seed=2, exit_layer=E, late_damping=0.2, dtype=torch.bfloat16, device=gpu_device, gen_device="cpu")
    eng = HipEngine(model, max_ctx=512, max_prompt=64)
    eng.set_eos([cfg.vocab_size - 1])
    prompt = synthetic.make_prompt(cfg.vocab_size, 17, 1)
    P = len(prompt)
    T, K, TP, seed, off = 0.8, 0, 0.9, 99, 1234
    for header_off, tail_off, want_error in ((off, off + 1, 1), (off, off, 0)):
        eng.reset()
        eng.draft_block_sampled(prompt, 0, S + 1, P - 1, E, False, T, K, TP, seed, header_off)
        eng.run_bulk(P - 1, E, eng.num_layers)
        eng.run_layers(0, 0, S + 1, P - 1, E, eng.num_layers)
        eng.pipeline_pack_sampled(1, P, 0, S + 1, 0, header_off)
        blk = eng.pipeline_tail_sampled(S + 1, T, K, TP, seed, tail_off).clone()
        words = [int(v) for v in blk[:24].tolist()]
        assert words[RES_ERROR] == want_error
        if want_error:
            continue
        n, td = words[0], words[1]
        assert 0 <= n <= td <= S and words[3] == P + n
        drafts = eng.row_tokens(1, S)
        assert words[4:4 + n] == drafts[:n]
        assert words[RES_PENDING] == (1 if n < td else 0) and (words[2] == -1) == (n < td)
        before = blk.clone()
        eng.pipeline_residual(blk, 0, seed, tail_off)
        after = [int(v) for v in blk[:24].tolist()]
        if n < td:
            assert after[RES_PENDING] == 0 and 0 <= after[2] < cfg.vocab_size and after[4 + n] == after[2]
        else:
            assert torch.equal(blk, before)                     # not pending: a no-op
        q = blk[64:64 + cfg.vocab_size].view(torch.float32)
        assert abs(float(q.sum()) - 1.0) < 1e-3 and float(q.min()) >= 0.0      # q_n is a probability row
    small = torch.zeros(64, dtype=torch.int32, device=gpu_device)
    scratch = eng._sampling_scratch()
    rc = eng.lib.lsk_pipeline_tail_sampled(eng._handle, S + 1, ctypes.c_float(T), K, ctypes.c_float(TP), seed, off, scratch.data_ptr(), scratch.numel(),
                                           small.data_ptr(), small.numel(), eng._stream)
    assert rc != 0 and b"result block" in eng.lib.lsk_last_error()
    with pytest.raises(_lib.LskError):
        eng.pipeline_residual(torch.zeros(8, dtype=torch.int32, device=gpu_device), 0, seed, off)
    eng.close()
