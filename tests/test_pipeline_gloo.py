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
        prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 18, S)
        # an EOS case: the 5th token of the free run becomes the eos id
        eos = None
        if rank == 0:
            eos = res.predicted_tokens[5]
        t = torch.tensor([eos if eos is not None else 0])
        dist.broadcast(t, src=0)
        res2 = dec.generate(prompt if rank == 0 else None, [int(t.item())], 18, S)
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
        want_eos = lo.self_speculative_generate(om, prompt, [eos], 18, 2, 4)
    assert tokens_eos == want_eos.predicted_tokens
    assert eos not in tokens_eos


def test_partition_plan():
    from layerskip_amd.pipeline import plan_partition
    assert plan_partition(32, 8, 1) == [(0, 32)]
    assert plan_partition(40, 10, 2) == [(0, 10), (10, 40)]
    p = plan_partition(80, 12, 8)
    assert p[0] == (0, 12) and p[-1][1] == 80 and all(b == c for (_, b), (c, _) in zip(p, p[1:]))
    assert max(b - a for a, b in p[1:]) - min(b - a for a, b in p[1:]) <= 1
    with pytest.raises(ValueError):
        plan_partition(6, 5, 4)
