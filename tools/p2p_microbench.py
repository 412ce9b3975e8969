"""Point-to-point latency of the layer pipeline's messages, on the transport the pipeline uses (measurement tool).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/p2p_microbench.py
    python tools/p2p_microbench.py --world 2            # spawns the ranks itself

One rank per GPU over RCCL (backend nccl) when the node has a device per rank; otherwise the ranks share device 0 and the messages go
through gloo from host copies -- the same two transports `layerskip_amd.pipeline_strategy.init_distributed` chooses between.  Timed: a
ping-pong rank 0 <-> rank 1 of the pipeline's message sizes ((S + 2) x H elements of 2 bytes: 64 KB at llama2-7B / S = 6, 100 KB at
llama2-13B / S = 8, 224 KB at llama2-70B / S = 12; the 96-byte greedy result block; the sampled result block with its V fp32
probability row), from / into views of one large device buffer (as the engines' message buffers are views of their workspace).
Half the round trip = one hop: what `hop_wait_ms` of bench.py's `pipeline.hops[]` is to be read against."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

SIZES = [("greedy result block", 96), ("7B message (S=6)", 8 * 4096 * 2), ("13B message (S=8)", 10 * 5120 * 2), ("70B message (S=12)", 14 * 8192 * 2),
         ("sampled result, V=32000", (64 + 32000) * 4), ("sampled result, V=128256", (64 + 128256) * 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.world}", "--master-addr",
                                 "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    one_per_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    backend = "nccl" if one_per_gpu else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(rank if one_per_gpu else 0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo")
    dev = torch.device("cuda", rank) if one_per_gpu else torch.device("cpu")
    pool = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    peer = 1 - rank if rank < 2 else None
    rows = []
    for name, nbytes in SIZES:
        view = pool[4096: 4096 + nbytes]
        if peer is None:
            continue
        for _ in range(10):                                   # warm-up: the peer channel exists after the first exchange
            (dist.send(view, dst=peer), dist.recv(view, src=peer)) if rank == 0 else (dist.recv(view, src=peer), dist.send(view, dst=peer))
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            if rank == 0:
                dist.send(view, dst=peer)
                dist.recv(view, src=peer)
            else:
                dist.recv(view, src=peer)
                dist.send(view, dst=peer)
            if dev.type == "cuda":
                torch.cuda.synchronize()                      # the pipeline's host reads a header / a result after every message
        dt = time.perf_counter() - t0
        rows.append({"message": name, "bytes": nbytes, "one_hop_us": round(1e6 * dt / args.iters / 2, 2)})
    dist.barrier()
    if rank == 0:
        out = {"backend": backend, "world": world, "transport": "RCCL over xGMI, one rank per GPU" if one_per_gpu else "gloo, host copies (ranks share a GPU / no GPU)",
               "iters": args.iters, "hops": rows}
        print(json.dumps(out, indent=1))
        if args.out:
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
