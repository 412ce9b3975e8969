"""The N>1 path on CPU: the layer-range pipeline protocol (layerskip_amd/pipeline.py) over gloo with
world_size 2 and 3, each rank owning only its layers, against the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _struct_worker(rank, world, port, queue, override_frac, optimistic):
    """Structured checkpoint: override_frac = 0 -> every draft accepted and every optimistic guess right (the
    continuation path runs on every step); 0.3 -> rejections force the continuation to be discarded."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_stage_backend import CpuStageBackend
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        cfg = synthetic.make_config("tiny-gqa")
        E, S = 3, 6
        part = plan_partition(cfg.num_hidden_layers, E, world)
        model = synthetic.build_structured_model(cfg, seed=4, exit_layer=E, override_frac=override_frac, layer_range=part[rank]).float()
        be = CpuStageBackend(model, layer_range=part[rank])
        dec = PipelineSpeculativeDecoder(be, rank, world, part, E, optimistic=optimistic)
        prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 40, S)
        if rank == 0:
            queue.put((res.predicted_tokens, res.acceptance_rate, res.steps, dec.stats()))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, queue):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_stage_backend import CpuStageBackend
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        cfg = synthetic.make_config("tiny-mha")
        E, S = 2, 4
        part = plan_partition(cfg.num_hidden_layers, E, world)
        model = synthetic.build_model(cfg, seed=1, exit_layer=E, late_damping=0.05, layer_range=part[rank])
        be = CpuStageBackend(model, layer_range=part[rank])
        dec = PipelineSpeculativeDecoder(be, rank, world, part, E)
        assert dec.warm_transport() >= 0.0          # every point-to-point channel of the protocol opened before the first block moves
        prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 18, S)
        # an EOS case: the 5th token of the free run becomes the eos id
        eos = None
        if rank == 0:
            eos = res.predicted_tokens[5]
        t = torch.tensor([eos if eos is not None else 0])
        dist.broadcast(t, src=0)
        # ... listed LAST of twelve ids, the other eleven never produced (the reference folds any number of stop_token_ids into the list,
        # generator_base.py:106): the ids travel to the late ranks in a broadcast of their own, sized by the count in the set-up words
        eos_list = [cfg.vocab_size - 1 - i for i in range(11)] + [int(t.item())]
        res2 = dec.generate(prompt if rank == 0 else None, eos_list if rank == 0 else [], 18, S)
        if rank == 0:
            queue.put((res.predicted_tokens, res.acceptance_rate, res.steps, res2.predicted_tokens, int(t.item())))
    finally:
        dist.destroy_process_group()


def _reference():
    from cpu_stage_backend import CpuStageBackend  # noqa: F401
    from layerskip_amd import synthetic
    from oracle import llama_oracle as lo
    cfg = synthetic.make_config("tiny-mha")
    model = synthetic.build_model(cfg, seed=1, exit_layer=2, late_damping=0.05)
    om = lo.OracleModel.from_hf(model, dtype=torch.float32)
    prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
    with torch.inference_mode():
        return om, prompt, lo.self_speculative_generate(om, prompt, [cfg.vocab_size], 18, 2, 4)


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_matches_single_process(world):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, rate, steps, tokens_eos, eos = queue.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    om, prompt, want = _reference()
    assert tokens == want.predicted_tokens
    assert rate == pytest.approx(want.acceptance_rate, abs=1e-12)
    assert [list(s) for s in steps] == [[s.num_drafts, s.num_matches] for s in want.steps]
    from oracle import llama_oracle as lo
    with torch.inference_mode():
        eos_list = [om.embed.shape[0] - 1 - i for i in range(11)] + [eos]
        assert not any(t in want.predicted_tokens for t in eos_list[:11])
        want_eos = lo.self_speculative_generate(om, prompt, eos_list, 18, 2, 4)
    assert tokens_eos == want_eos.predicted_tokens
    assert eos not in tokens_eos and len(tokens_eos) < len(tokens)


@pytest.mark.parametrize("world,override_frac,optimistic", [(2, 0.0, True), (3, 0.3, True), (2, 0.3, False)])
def test_optimistic_overlap_is_output_preserving(world, override_frac, optimistic):
    """Rank 0 drafts step k+1 while step k's verify block is in flight.  With a checkpoint whose drafts are always
    accepted the continuation is used on every step; with one that rejects, it is discarded -- tokens and the per-step
    (num_drafts, num_matches) trace equal the single-process oracle in both."""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_struct_worker, args=(r, world, port, queue, override_frac, optimistic)) for r in range(world)]
    for p in procs:
        p.start()
    tokens, rate, steps, stats = queue.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from layerskip_amd import synthetic
    from oracle import llama_oracle as lo
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_structured_model(cfg, seed=4, exit_layer=3, override_frac=override_frac).float()
    prompt = synthetic.make_struct_prompt(model.struct_program, 19, 2)
    with torch.inference_mode():
        want = lo.self_speculative_generate(lo.OracleModel.from_hf(model), prompt, [cfg.vocab_size], 40, 3, 6)
    assert tokens == want.predicted_tokens
    assert [list(s) for s in steps] == [[s.num_drafts, s.num_matches] for s in want.steps]
    assert rate == pytest.approx(want.acceptance_rate, abs=1e-12)
    if not optimistic:
        assert stats["optimistic_attempts"] == 0
    elif override_frac == 0.0:
        assert want.acceptance_rate == 1.0
        assert stats["optimistic_attempts"] >= 3 and stats["optimistic_hits"] == stats["optimistic_attempts"]
    else:
        assert 0 < want.acceptance_rate < 1.0
        assert stats["optimistic_attempts"] > stats["optimistic_hits"]          # a forced rejection discarded a continuation


def _capacity_worker(rank, world, port, queue):
    """Rank 1's backend cannot hold the generation: the set-up agreement must make EVERY rank raise before any block moves."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from cpu_stage_backend import CpuStageBackend
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        cfg = synthetic.make_config("tiny-mha")
        part = plan_partition(cfg.num_hidden_layers, 2, world)
        model = synthetic.build_model(cfg, seed=1, exit_layer=2, late_damping=0.05, layer_range=part[rank])
        be = CpuStageBackend(model, layer_range=part[rank])
        if rank == 1:
            def refuse(total_tokens, prompt_len):
                raise RuntimeError(f"cannot hold {total_tokens} tokens")
            be.ensure_capacity = refuse
        dec = PipelineSpeculativeDecoder(be, rank, world, part, 2)
        try:
            dec.generate(synthetic.make_prompt(cfg.vocab_size, 21, 3) if rank == 0 else None, [cfg.vocab_size], 18, 4)
            queue.put((rank, "no error"))
        except RuntimeError as exc:
            queue.put((rank, str(exc)))
    finally:
        dist.destroy_process_group()


def test_a_capacity_error_surfaces_on_every_rank_not_as_a_hang():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_capacity_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(queue.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "cannot hold" in got[1]
    assert "another rank" in got[0] and "another rank" in got[2]


def test_partition_plan():
    from layerskip_amd.pipeline import plan_partition
    assert plan_partition(40, 10, 2, balance="memory") == [(0, 20), (20, 40)]            # llama2-13B on two GPUs
    p70 = plan_partition(80, 12, 8, balance="memory")
    assert p70[0] == (0, 12) and p70[-1][1] == 80 and {b - a for a, b in p70[1:]} <= {9, 10}
    assert plan_partition(32, 8, 1) == [(0, 32)]
    assert plan_partition(40, 10, 2) == [(0, 10), (10, 40)]
    p = plan_partition(80, 12, 8)
    assert p[0] == (0, 12) and p[-1][1] == 80 and all(b == c for (_, b), (c, _) in zip(p, p[1:]))
    assert max(b - a for a, b in p[1:]) - min(b - a for a, b in p[1:]) <= 1
    with pytest.raises(ValueError):
        plan_partition(6, 5, 4)
