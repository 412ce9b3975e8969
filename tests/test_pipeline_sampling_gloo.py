"""sample=True and logits processors on the layer pipeline (layerskip_amd/pipeline.py, pipeline_strategy.py) over gloo, world 2 and 3,
each rank owning only its layers (CPU stage backend: the oracle's arithmetic and its draw-for-draw model of the sampling kernels).

* sampled fast path: the header's p_i(x_i), the last rank's acceptance test + q_n, rank 0's residual draw -- DRAW FOR DRAW the
  one-process orchestration of lsk_spec_step_sampled under the same (seed, offset) (reference SSG:191-199, generator_base.py:39);
* slow path: logits processors (generator_base.py:77-85; SSG:138-139, :172-173), greedy and sampled, and sampled / early-exit
  autoregressive decoding (ARG:44-51) -- token for token the one-process strategies on the same torch seed.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

SAMPLING = dict(temperature=0.8, top_k=40, top_p=0.9, seed=1234567, offset=(1 << 40) + 17)
E, S, MAX_STEPS = 2, 4, 22


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))


def _model(layer_range=None):
    from layerskip_amd import synthetic
    cfg = synthetic.make_config("tiny-mha")
    return cfg, synthetic.build_model(cfg, seed=1, exit_layer=E, late_damping=0.2, layer_range=layer_range)


# ------------------------------------------------------------------------------------------------ decoder level: the sampled protocol
def _sampled_worker(rank, world, port, queue, eos_case):
    _setup(rank, world, port)
    try:
        from cpu_stage_backend import CpuStageBackend
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, Sampling, plan_partition
        cfg = synthetic.make_config("tiny-mha")
        part = plan_partition(cfg.num_hidden_layers, E, world)
        _, model = _model(part[rank])
        dec = PipelineSpeculativeDecoder(CpuStageBackend(model, layer_range=part[rank]), rank, world, part, E)
        prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
        res = dec.generate(prompt if rank == 0 else None, [eos_case if eos_case is not None else cfg.vocab_size], MAX_STEPS, S,
                           sampling=Sampling(**SAMPLING) if rank == 0 else None)
        if rank == 0:
            queue.put((res.predicted_tokens, res.steps, dec.stats()))
    finally:
        dist.destroy_process_group()


def _one_process_sampled(eos):
    """lsk_spec_generate_sampled's contract on one backend: step i draws at offset + i (CpuStageBackend.spec_step_sampled is the
    oracle's orchestration of lsk_spec_step_sampled)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_stage_backend import CpuStageBackend
    from layerskip_amd import synthetic
    cfg, model = _model()
    be = CpuStageBackend(model)
    prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
    out, steps, cur, i = [], [], list(prompt), 0
    sm = SAMPLING
    while len(out) < MAX_STEPS:
        s_eff = max(0, min(S, MAX_STEPS - len(out) - 1))
        r = be.spec_step_sampled(cur, s_eff, E, eos, sm["temperature"], sm["top_k"], sm["top_p"], sm["seed"], sm["offset"] + i)
        i += 1
        out.extend(r.emitted)
        steps.append((r.num_drafts, r.num_matches))
        hit = [out.index(e) for e in eos if e in out]
        if hit:
            out = out[: hit[0]]
            break
        cur = [r.next_token]
    return out, steps


def _run(world, target, args):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, queue) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _q
    import time
    deadline = time.monotonic() + 600
    got = None
    while got is None:
        try:
            got = queue.get(timeout=2)
        except _q.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.monotonic() > deadline:          # a rank died: do not wait out the others' collective timeouts
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"a pipeline rank failed (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world", [2, 3])
def test_sampled_pipeline_is_draw_for_draw_the_one_process_step(world):
    want_tokens, want_steps = _one_process_sampled([512])
    assert any(n < td for td, n in want_steps), "the case must contain rejections (the residual draw on rank 0)"
    assert any(n == td and td > 0 for td, n in want_steps), "... and fully accepted steps (the bonus token of the last rank)"
    tokens, steps, stats = _run(world, _sampled_worker, (None,))
    assert tokens == want_tokens
    assert [tuple(s) for s in steps] == want_steps
    assert stats["optimistic_attempts"] == 0           # no continuation under sampling: the bonus token is the last rank's draw
    # an EOS case: a token of the free run becomes the eos id (the drafted-EOS cut of the acceptance test, the output cut on rank 0)
    eos = want_tokens[6]
    want_eos, want_eos_steps = _one_process_sampled([eos])
    tokens, steps, _ = _run(world, _sampled_worker, (eos,))
    assert tokens == want_eos and eos not in tokens
    assert [tuple(s) for s in steps] == want_eos_steps


def test_split_acceptance_equals_the_one_kernel_acceptance():
    """oracle: device_accept (lsk_accept_sampled_kernel) == device_accept_test on the p_i(x_i) scalars (last rank) + device_residual
    with the rows (rank 0): the decomposition the pipeline rests on."""
    import numpy as np
    from oracle import sampling_oracle as so
    rng = np.random.default_rng(5)
    V, T = 256, 5
    for trial in range(40):
        p_draft = [so._softmax(rng.normal(size=V).astype(np.float32) * 2) for _ in range(T)]
        p_verify = [so._softmax((np.log(p_draft[i % T]) if i < T else rng.normal(size=V)).astype(np.float32)
                                + rng.normal(size=V).astype(np.float32) * (0.3 + trial % 3)) for i in range(T + 1)]
        drafts = [int(rng.integers(V)) for _ in range(T)]
        verified = [int(rng.integers(V)) for _ in range(T + 1)]
        eos = [drafts[3]] if trial % 4 == 0 else []
        n, td, tok = so.device_accept(drafts, verified, p_draft, p_verify, eos, 99, 1000 + trial)
        n2, td2 = so.device_accept_test(drafts, [p_draft[i][t] for i, t in enumerate(drafts)], p_verify, eos, 99, 1000 + trial)
        assert (n, td) == (n2, td2)
        tok2 = verified[td] if n == td else so.device_residual(p_verify[n], p_draft[n], drafts[n], 99, 1000 + trial)
        assert tok == tok2


# ------------------------------------------------------------------------------------------------ strategy level: every reference flag
def _cases():
    """(name, strategy, GenerationConfig kwargs, logits processors?)"""
    return [
        ("sampled", "self_speculative", dict(sample=True, temperature=0.7, top_k=50, top_p=0.95), False),
        ("ngram_greedy", "self_speculative", dict(sample=False), True),
        ("ngram_sampled", "self_speculative", dict(sample=True, temperature=0.9, top_k=0, top_p=0.9), True),
        ("ar_sampled", "autoregressive", dict(sample=True, temperature=0.7, top_k=50, top_p=0.95, exit_layer=-1), False),
        ("ar_ngram", "autoregressive", dict(sample=False, exit_layer=-1), True),
        ("ar_early_exit", "autoregressive", dict(sample=False, exit_layer=E), False),
    ]


def _processors(on: bool):
    import transformers
    if not on:
        return None
    return transformers.LogitsProcessorList([transformers.NoRepeatNGramLogitsProcessor(2)])      # generator_base.py:77-85


def _gen_cfg(kw):
    from layerskip_amd import GenerationConfig
    base = dict(max_steps=16, exit_layer=E, num_speculations=S)
    base.update(kw)
    return GenerationConfig(**base)


def _strategy_worker(rank, world, port, queue):
    _setup(rank, world, port)
    try:
        from fake_engine import FullFakeEngine
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import plan_partition
        from layerskip_amd.pipeline_strategy import PIPELINE_STRATEGIES, DistContext
        cfg = synthetic.make_config("tiny-mha")
        part = plan_partition(cfg.num_hidden_layers, E, world)
        _, model = _model(part[rank])
        ctx = DistContext(rank, world, rank, torch.device("cpu"), "gloo", torch.device("cpu"))
        factory = lambda m, lr, **kw: FullFakeEngine(m, layer_range=lr)      # the stage backend + the fused ar_step the rank-local early-exit run uses
        strategies = {name: cls(ctx, part, backend_factory=factory) for name, cls in PIPELINE_STRATEGIES.items()}
        if rank > 0:
            strategies["self_speculative"].serve(model)
            return
        prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
        out = {}
        try:
            for name, strat, kw, procs in _cases():
                torch.manual_seed(11)
                res = strategies[strat].generate_token_ids(model, prompt, [cfg.vocab_size], _gen_cfg(kw), logits_processors=_processors(procs))
                out[name] = (res.predicted_tokens, res.acceptance_rate)
        finally:
            strategies["self_speculative"].shutdown()
        queue.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_strategies_sampled_and_with_processors_equal_the_one_process_strategies(world, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_engine import FullFakeEngine
    from layerskip_amd import hip_strategies, synthetic
    got = _run(world, _strategy_worker, ())
    cfg, model = _model()
    engine = FullFakeEngine(model)
    monkeypatch.setattr(hip_strategies, "get_engine", lambda m, **kw: engine)
    prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)
    for name, strat, kw, procs in _cases():
        torch.manual_seed(11)
        want = hip_strategies.STRATEGIES[strat]().generate_token_ids(model, prompt, [cfg.vocab_size], _gen_cfg(kw),
                                                                   logits_processors=_processors(procs))
        assert got[name][0] == want.predicted_tokens, name
        assert got[name][1] == want.acceptance_rate, name
        assert len(want.predicted_tokens) == 16


def test_a_block_whose_philox_offset_is_not_the_last_ranks_is_refused():
    """The last rank counts steps on its own; a header that carries another Philox offset (a protocol out of step) sets the result
    block's error word and rank 0 raises instead of emitting a token drawn from the wrong stream."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_stage_backend import CpuStageBackend
    from layerskip_amd import synthetic
    from layerskip_amd.pipeline import PipelineSpeculativeDecoder, Sampling
    cfg, model = _model()
    be = CpuStageBackend(model)
    dec = PipelineSpeculativeDecoder(be, 0, 1, [(0, cfg.num_hidden_layers)], E)
    inner = be.pipeline_tail_sampled
    be.pipeline_tail_sampled = lambda m, t, k, p, seed, off: inner(m, t, k, p, seed, off + 7)
    with pytest.raises(RuntimeError, match="out of step"):
        dec.generate(synthetic.make_prompt(cfg.vocab_size, 9, 3), [cfg.vocab_size], 8, S, sampling=Sampling(**SAMPLING))
    be.pipeline_tail_sampled = inner
    ok = dec.generate(synthetic.make_prompt(cfg.vocab_size, 9, 3), [cfg.vocab_size], 8, S, sampling=Sampling(**SAMPLING))
    assert len(ok.predicted_tokens) == 8                       # and the decoder is usable again afterwards


def _early_error_worker(rank, world, port, queue):
    _setup(rank, world, port)
    try:
        from cpu_stage_backend import CpuStageBackend
        from layerskip_amd import synthetic
        from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
        cfg = synthetic.make_config("tiny-mha")
        part = plan_partition(cfg.num_hidden_layers, E, world)
        _, model = _model(part[rank])
        dec = PipelineSpeculativeDecoder(CpuStageBackend(model, layer_range=part[rank]), rank, world, part, E)
        prompt = synthetic.make_prompt(cfg.vocab_size, 21, 3)

        def boom(_dec):
            raise ValueError("a logits processor failed before the first verify")

        outcome = "no error"
        try:
            dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 12, S, driver=boom if rank == 0 else None)
        except ValueError as exc:
            outcome = str(exc)
        # the pipeline is still in step: a normal generation follows on the same decoder
        res = dec.generate(prompt if rank == 0 else None, [cfg.vocab_size], 12, S)
        queue_item = (rank, outcome, res.predicted_tokens)
        gathered = [None] * world
        dist.all_gather_object(gathered, queue_item)
        if rank == 0:
            queue.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_an_error_on_rank_0_before_its_first_block_leaves_nobody_in_a_receive(world):
    """Rank 0 fails ahead of its first step (here: the slow path's driver raises): the stop message then carries the prompt rows the
    late ranks expect in front of their first message, every rank returns, and the next generation on the same decoders is correct."""
    got = _run(world, _early_error_worker, ())
    by_rank = {r: (o, t) for r, o, t in got}
    assert "logits processor failed" in by_rank[0][0]
    assert all(by_rank[r][0] == "no error" for r in range(1, world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from layerskip_amd import synthetic
    from oracle import llama_oracle as lo
    cfg, model = _model()
    with torch.inference_mode():
        want = lo.self_speculative_generate(lo.OracleModel.from_hf(model.float()), synthetic.make_prompt(cfg.vocab_size, 21, 3), [cfg.vocab_size], 12, E, S)
    assert by_rank[0][1] == want.predicted_tokens
