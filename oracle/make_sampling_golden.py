"""Golden vectors for oracle/sampling_oracle.py from the UNMODIFIED reference functions (build container only).

    python oracle/make_sampling_golden.py            # writes tests/golden/sampling/cases.json
    python oracle/make_sampling_golden.py --check    # recompute and compare with the committed file
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sampling", "cases.json")

CASES = [   # (vocab, logit scale, temperature, top_k, top_p)
    (512, 1.0, 0.7, 50, 0.95),      # the reference's GenerationConfig defaults
    (512, 3.0, 0.6, 0, 0.9),        # BASELINE.md's sampling settings (top_k disabled)
    (1000, 0.3, 1.0, 0, 1.0),       # nothing filtered
    (768, 5.0, 0.2, 5, 0.5),        # sharp: nucleus of a few tokens
    (300, 2.0, 1.3, 400, 0.99),     # top_k larger than the vocabulary
    (640, 8.0, 0.05, 0, 0.3),       # nucleus of one token: min_tokens_to_keep binds
]


def build():
    ref = ref_shim.load_reference()
    lmu, ssg = ref.llama_model_utils, ref.self_speculation_generator
    out = []
    for idx, (v, scale, temp, top_k, top_p) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + idx)
        draft_logits = (torch.randn(v, generator=g) * scale).to(torch.bfloat16).float()      # logits arrive bf16-rounded
        verify_logits = (draft_logits + torch.randn(v, generator=g) * scale * 0.5).to(torch.bfloat16).float()
        rec = {"vocab": v, "temperature": temp, "top_k": top_k, "top_p": top_p,
               "draft_logits": draft_logits.tolist(), "verify_logits": verify_logits.tolist()}
        for name, lg in (("draft", draft_logits), ("verify", verify_logits)):
            warped = lmu.top_k_top_p_filtering(lg[None, :].clone() / temp, top_k=top_k, top_p=top_p)[0]
            torch.manual_seed(0)
            _, probs = lmu.decode_next_token(logits=lg[None, None, :].clone(), token_idx=None, sample=True, temperature=temp,
                                             top_k=top_k, top_p=top_p)
            rec[name + "_kept"] = torch.isfinite(warped).nonzero().flatten().tolist()
            rec[name + "_probs"] = probs.reshape(-1).tolist()
        pd, pv = torch.tensor(rec["draft_probs"]), torch.tensor(rec["verify_probs"])
        rec["residual"] = ssg.max_fn(pv - pd).tolist()
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    recs = build()
    if args.check:
        old = json.load(open(OUT))
        assert len(old) == len(recs)
        for a, b in zip(old, recs):
            for k in a:
                assert np.allclose(np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64), rtol=0, atol=1e-7), k
        print("sampling golden: up to date")
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(recs, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
