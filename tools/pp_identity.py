"""The layer-range pipeline at FULL model size on whatever GPUs the box has (measurement / evidence tool, not a test):

    python tools/pp_identity.py --model llama2-13B --world 2 [--balance memory] [--max-steps 48] [--out x.json]
    python tools/pp_identity.py --model llama2-70B --world 8 --max-steps 24

Spawns `world` ranks (one process each).  With at least `world` GPUs every rank takes its own device and the rows travel over
RCCL; on a smaller box (the 1-GPU development boxes) all ranks share device 0 and exchange rows through gloo -- the SAME protocol,
partition and kernels, only the transport differs.  Every rank materialises ONLY its layer range of the random-init checkpoint
(device generator: the same bits in every process) and releases the unpacked originals as it packs them (llama2-70B: 140 GB of
packed weights across the ranks).  Rank 0 reports tokens/s, the per-hop host times and the generated ids; then -- if the model
fits one GPU -- the parent decodes the same prompt with ONE fused engine and checks that the ids, the acceptance rate and the
per-step trace are identical.  BASELINE configs #4 (13B / 2 ranks) and #5 (70B / 8 ranks)."""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _worker(rank, world, port, queue, args):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import datetime
    import torch.distributed as dist
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    from layerskip_amd.pipeline import PipelineSpeculativeDecoder, plan_partition
    one_per_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if one_per_gpu else 0)
    torch.cuda.set_device(dev)
    if one_per_gpu:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=20))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=20))
    try:
        cfg = synthetic.make_config(args.model)
        E, S = synthetic.default_exit_layer(args.model), synthetic.default_num_speculations(args.model)
        part = plan_partition(cfg.num_hidden_layers, E, world, balance=args.balance)
        t0 = time.time()
        model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=args.late_damping, dtype=torch.bfloat16, device=dev, gen_device=dev,
                                      layer_range=part[rank])
        eng = HipEngine(model, max_ctx=args.prompt_len + args.max_steps + 2 * S + 32, max_prompt=args.prompt_len, layer_range=part[rank],
                        release_weights=True)
        torch.cuda.synchronize()
        build_s = time.time() - t0
        dec = PipelineSpeculativeDecoder(eng, rank, world, part, E, comm_device=dev if one_per_gpu else torch.device("cpu"))
        prompt = synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 0) if rank == 0 else None
        dec.generate(synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 1000) if rank == 0 else None, [cfg.vocab_size], 16, S)     # warm-up
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = dec.generate(prompt, [cfg.vocab_size], args.max_steps, S)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stats = [None] * world
        dist.all_gather_object(stats, dec.stats())
        mem = [None] * world
        dist.all_gather_object(mem, round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2))
        if rank == 0:
            queue.put({"tokens": res.predicted_tokens, "acceptance_rate": res.acceptance_rate, "steps": [list(s) for s in res.steps],
                       "tokens_per_s": round(len(res.predicted_tokens) / dt, 2), "seconds": round(dt, 3), "build_s": round(build_s, 1),
                       "layer_ranges": [list(p) for p in part], "transport": "RCCL point-to-point, one rank per GPU" if one_per_gpu else
                       "gloo, ranks SHARING device 0 (plumbing run: the ranks' kernels take turns on one GPU)",
                       "hbm_gib_per_rank": mem,
                       "hops": [{"rank": r, "hop_enqueue_ms": st.get("hop_enqueue_ms"), "hop_wait_ms": st.get("hop_wait_ms"), "blocks": st.get("hops")}
                                for r, st in enumerate(stats) if r > 0],
                       "rank0": {k: stats[0].get(k) for k in ("steps", "draft_ms_per_step", "verify_roundtrip_ms_per_step", "optimistic_hit_rate")}})
        eng.close()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-13B")
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--balance", default="memory", choices=["draft", "memory"])
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--max-steps", type=int, default=48)
    ap.add_argument("--late-damping", type=float, default=0.03)
    ap.add_argument("--no-single", action="store_true", help="skip the one-engine comparison run")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, args.world, port, queue, args)) for r in range(args.world)]
    for p in procs:
        p.start()
    out = queue.get(timeout=3000)
    for p in procs:
        p.join(timeout=600)
    out = {"model": args.model, "world": args.world, "gpus_on_this_box": torch.cuda.device_count(), **out}
    if not args.no_single:
        from layerskip_amd import GenerationConfig, synthetic
        from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
        cfg = synthetic.make_config(args.model)
        E, S = synthetic.default_exit_layer(args.model), synthetic.default_num_speculations(args.model)
        dev = torch.device("cuda", 0)
        model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=args.late_damping, dtype=torch.bfloat16, device=dev, gen_device=dev)
        strat = HipSelfSpeculativeGenerationStrategy(engine_kwargs={"max_ctx": args.prompt_len + args.max_steps + S + 16, "max_prompt": args.prompt_len,
                                                                    "release_weights": True})
        gen = GenerationConfig(max_steps=args.max_steps, exit_layer=E, num_speculations=S, sample=False, generation_strategy="self_speculative")
        strat.generate_token_ids(model, synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 1000), [cfg.vocab_size], gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        want = strat.generate_token_ids(model, synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 0), [cfg.vocab_size], gen)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["single_engine"] = {"tokens_per_s": round(len(want.predicted_tokens) / dt, 2),
                                "identical_tokens": want.predicted_tokens == out["tokens"],
                                "identical_acceptance": want.acceptance_rate == out["acceptance_rate"],
                                "identical_step_trace": [list(s) for s in strat.last_steps] == out["steps"]}
    out["n_tokens"] = len(out.pop("tokens"))
    out["n_steps"] = len(out.pop("steps"))
    print(json.dumps(out), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
