"""TEST INFRASTRUCTURE.  fp16 fixtures (the dtype the reference's generate.py hard-codes, generate.py:63) from the
UNMODIFIED reference, for the fp16 build of the engine (liblayerskip_hip_f16.so).  Same procedure as
oracle/make_golden.py (the restatement must reproduce the reference bit for bit before anything is written):

    python oracle/make_golden_fp16.py            # writes tests/golden/fp16/*.json
    python oracle/make_golden_fp16.py --check
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from layerskip_amd import synthetic  # noqa: E402
from oracle import make_golden as mg  # noqa: E402
from oracle import ref_shim  # noqa: E402

OUT_DIR = os.path.join(mg.GOLDEN_DIR, "fp16")
NAMES = ["tiny_mha_s0", "tiny_gqa_s0", "tiny_d64_s0", "tiny_mha_s1"]


def build(ref, name):
    case = mg.CASES[name].resolved()
    cfg = synthetic.make_config(case.shape)
    model = synthetic.build_model(cfg, seed=case.seed, exit_layer=case.exit_layer, late_damping=case.late_damping,
                                  dtype=torch.bfloat16, device="cpu")       # same weight VALUES as the bf16 fixtures
    prompt = synthetic.make_prompt(cfg.vocab_size, case.prompt_len, case.prompt_seed)
    eos = [cfg.vocab_size]
    rec = {"name": name, "shape": case.shape, "seed": case.seed, "late_damping": case.late_damping,
           "exit_layer": case.exit_layer, "num_speculations": case.num_speculations, "prompt_len": case.prompt_len,
           "prompt_seed": case.prompt_seed, "max_steps": case.max_steps, "eos_token_ids": eos, "prompt": prompt,
           "torch": torch.__version__}
    rec["fp16"] = mg.one_dtype(ref, model, case, prompt, eos, torch.float16)
    print(f"  {name} fp16: acceptance {rec['fp16']['acceptance_rate']:.3f}, spec==ar {rec['fp16']['spec_equals_ar']}, "
          f"min margin {min(rec['fp16']['spec_margins'] or [0]):.4f}")
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    ref = ref_shim.load_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    bad = 0
    for name in NAMES:
        rec = build(ref, name)
        path = os.path.join(OUT_DIR, name + ".json")
        if args.check:
            old = json.load(open(path))
            same = old["fp16"]["spec_tokens"] == rec["fp16"]["spec_tokens"] and old["fp16"]["ar_tokens"] == rec["fp16"]["ar_tokens"]
            print(("OK   " if same else "DIFF ") + name)
            bad += 0 if same else 1
        else:
            json.dump(rec, open(path, "w"))
            print("wrote", path)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
