"""TEST / MEASUREMENT INFRASTRUCTURE -- NOT PRODUCT CODE.

The reference's own CPU path, timed (north_star: "next to the reference's CPU HF-transformers path"):
the UNMODIFIED /root/reference SelfSpeculativeGenerationStrategy / AutoRegressiveGenerationStrategy (through oracle/ref_shim.py)
on the benchmark's synthetic llama2-7B-shaped bf16 weights and prompt shape, on the cores of THIS box -- and the
reference-pinned restatement (oracle/llama_oracle.py, bench.py's `cpu_baseline` kind "port") on the same weights right after,
so the file shows both what the reference does on a CPU and that the port which travels to the GPU box is a fair stand-in.
Timing bracket = generator_base.py:107-118 (wall clock around generate_token_ids).  /root/reference exists only in the build
container, so this runs here (no GPU needed); the output is committed under profiles/.

    python oracle/time_reference_cpu.py [--model llama2-7B] [--prompt-len 512] [--new-tokens 64] [--out profiles/...json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from layerskip_amd import synthetic  # noqa: E402
from oracle import llama_oracle as lo  # noqa: E402
from oracle import ref_shim  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7B")
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--ar-tokens", type=int, default=24)
    ap.add_argument("--late-damping", type=float, default=0.03)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = synthetic.make_config(args.model)
    E, S = synthetic.default_exit_layer(args.model), synthetic.default_num_speculations(args.model)
    t0 = time.time()
    model = synthetic.build_model(cfg, seed=0, exit_layer=E, late_damping=args.late_damping, dtype=torch.bfloat16)
    build_s = time.time() - t0
    ref = ref_shim.load_reference()
    ref_shim.patch_model(model)
    gb = ref.generator_base
    prompt = synthetic.make_prompt(cfg.vocab_size, args.prompt_len, 4242)      # bench.py's cpu_baseline prompt
    eos = [cfg.vocab_size]
    out = {"model": args.model, "dtype": "bf16", "exit_layer": E, "num_speculations": S, "prompt_len": args.prompt_len,
           "cores": os.cpu_count(), "threads": args.threads, "torch": torch.__version__, "model_build_s": round(build_s, 1),
           "weights": f"synthetic.build_model(seed=0, late_damping={args.late_damping}) generated on the CPU (bench.py draws them on the GPU: same "
                      "distribution, different values)"}

    def timed(fn):
        t = time.time()
        r = fn()
        return r, time.time() - t

    with torch.inference_mode():
        # ---- the unmodified reference ----
        spec_cfg = gb.GenerationConfig(max_steps=args.new_tokens, exit_layer=E, num_speculations=S, sample=False,
                                       generation_strategy="self_speculative")
        strat = ref.self_speculation_generator.SelfSpeculativeGenerationStrategy()
        r_spec, t_spec = timed(lambda: strat.generate_token_ids(model=model, input_ids=list(prompt), eos_token_ids=eos,
                                                                generation_config=spec_cfg, logits_processors=None,
                                                                stopping_criteria=None, streamer=None))
        ar_cfg = gb.GenerationConfig(max_steps=args.ar_tokens, exit_layer=-1, num_speculations=-1, sample=False,
                                     generation_strategy="autoregressive")
        ar = ref.autoregressive_generator.AutoRegressiveGenerationStrategy()
        r_ar, t_ar = timed(lambda: ar.generate_token_ids(model=model, input_ids=list(prompt), eos_token_ids=eos,
                                                         generation_config=ar_cfg, logits_processors=None,
                                                         stopping_criteria=None, streamer=None))
        out["reference"] = {
            "self_speculative": {"tokens": len(r_spec.predicted_tokens), "seconds": round(t_spec, 2),
                                 "tokens_per_s": round(len(r_spec.predicted_tokens) / t_spec, 3),
                                 "acceptance_rate": round(float(r_spec.acceptance_rate), 4)},
            "autoregressive": {"tokens": len(r_ar.predicted_tokens), "seconds": round(t_ar, 2),
                               "tokens_per_s": round(len(r_ar.predicted_tokens) / t_ar, 3)},
            "what": "UNMODIFIED /root/reference strategies through oracle/ref_shim.py, wall clock around generate_token_ids"}
        # ---- the restatement bench.py times on the GPU box ----
        om = lo.OracleModel.from_hf(model)
        m_spec, t_mspec = timed(lambda: lo.self_speculative_generate(om, list(prompt), eos, args.new_tokens, E, S))
        m_ar, t_mar = timed(lambda: lo.autoregressive_generate(om, list(prompt), eos, args.ar_tokens))
        out["port"] = {
            "self_speculative": {"tokens": len(m_spec.predicted_tokens), "seconds": round(t_mspec, 2),
                                 "tokens_per_s": round(len(m_spec.predicted_tokens) / t_mspec, 3),
                                 "acceptance_rate": round(float(m_spec.acceptance_rate), 4)},
            "autoregressive": {"tokens": len(m_ar.predicted_tokens), "seconds": round(t_mar, 2),
                               "tokens_per_s": round(len(m_ar.predicted_tokens) / t_mar, 3)},
            "what": "oracle/llama_oracle.py (bench.py cpu_baseline kind 'port') on the same model object"}
        out["port_equals_reference"] = {"self_speculative_ids": m_spec.predicted_tokens == r_spec.predicted_tokens,
                                        "acceptance": float(m_spec.acceptance_rate) == float(r_spec.acceptance_rate),
                                        "autoregressive_ids": m_ar.predicted_tokens == r_ar.predicted_tokens}
        out["port_over_reference_speed"] = round(out["port"]["self_speculative"]["tokens_per_s"] /
                                                 out["reference"]["self_speculative"]["tokens_per_s"], 3)
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
