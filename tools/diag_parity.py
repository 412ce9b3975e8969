"""GPU diagnostic (not a test): how close is the engine to the fp32 / bf16 runs of the reference?"""
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import build_case_model, golden_names, load_golden  # noqa: E402

from layerskip_amd import GenerationConfig  # noqa: E402
from layerskip_amd.engine import BUF_BULK, get_engine  # noqa: E402
from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy  # noqa: E402


def first_mismatch(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None


rows = []
for name in golden_names():
    rec = load_golden(name)
    model = build_case_model(rec, "cuda:0")
    cfg = GenerationConfig(max_steps=rec["max_steps"], exit_layer=rec["exit_layer"], num_speculations=rec["num_speculations"],
                           sample=False)
    res = HipSelfSpeculativeGenerationStrategy().generate_token_ids(model, rec["prompt"], rec["eos_token_ids"], cfg)
    out = {"name": name, "n": len(res.predicted_tokens), "acc": round(res.acceptance_rate, 3)}
    for d in ("fp32", "bf16"):
        g = rec[d]
        i = first_mismatch(res.predicted_tokens, g["spec_tokens"])
        out[d] = {"first_mismatch": i, "margin": None if i is None or i >= len(g["spec_margins"]) else g["spec_margins"][i],
                  "acc": round(g["acceptance_rate"], 3)}
        # teacher-forced logits error on the recorded rows
        eng = get_engine(model)
        seq = rec["prompt"] + g["spec_tokens"]
        if len(seq) > len(rec["prompt"]):
            eng.ensure_capacity(len(seq) + 4, len(seq))
            eng.reset()
            eng.embed_rows(seq, BUF_BULK, 0)
            eng.run_layers_chunked(BUF_BULK, 0, len(seq), 0, 0, eng.num_layers)
            worst = 0.0
            for row in g["logits_topk"]:
                lg = torch.empty(1, eng.vocab, dtype=torch.float32, device="cuda:0")
                eng.run_head(BUF_BULK, row["row"], 1, logits=lg, want_tokens=False)
                torch.cuda.synchronize()
                worst = max(worst, (lg[0, row["idx"]].cpu() - torch.tensor(row["val"])).abs().max().item())
            out[d]["logit_err"] = round(worst, 5)
    rows.append(out)
    print(json.dumps(out))
    del model
