"""GPU diagnostic (not a test): where the fp16 build of the engine is less accurate than the reference's own fp16 run.

    python tools/diag_fp16.py [struct_fp16 fixture name, default full7b] [--out x.json]

VERDICT round 5, weak #1a: on `full7b` the engine's fp16 logits sit 0.492 fp16 ulp (rms) from the reference's fp32 logits, the
reference's own fp16 run 0.447 -- in bf16 the engine is the closer one.  A consistent 10 % is a rounding point, so:

1. the fp16 MFMA against SUBNORMAL operands (fp16 runs out of exponent at 6.1e-5; bf16 never does): does
   v_mfma_f32_16x16x32_f16 keep them (x subnormal, w subnormal, both) -- through the skinny projection kernel of the fp16 test library;
2. layer by layer on the fixture's teacher-forced sequence: the ENGINE's fp16 rows of layer l-1 are fed to (a) the engine's layer l,
   (b) the oracle's layer l in fp16 on the host CPU (= the reference's arithmetic, oracle/llama_oracle.py is pinned bit for bit to it),
   (c) the oracle's layer l in fp32 (the truth for that input); reported per layer: rms and max of (a - c) and (b - c) in fp16 ulp of c;
3. for the layers where the engine loses most: the same split for the attention half (h + o_proj(attention)) and the MLP half.
"""
import argparse
import ctypes
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import build_struct_model  # noqa: E402
from layerskip_amd import _lib  # noqa: E402
from layerskip_amd.engine import BUF_BULK, get_engine  # noqa: E402
from oracle import llama_oracle as lo  # noqa: E402


def ulp16(t, floor):
    return torch.pow(2.0, torch.floor(torch.log2(t.abs().clamp_min(floor))) - 10)


def err_stats(x, truth, floor):
    e = (x.double() - truth.double()).abs() / ulp16(truth.double(), floor)
    return float(e.pow(2).mean().sqrt()), float(e.max())


def subnormal_probe(dev):
    import lsk_test_lib
    lib = _lib.load(dtype="fp16")
    tl = lsk_test_lib.load(dtype="fp16")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    k, n, m = 256, 64, 4
    out = {}
    sub = 503 * 2.0 ** -24            # 3.0e-5: an fp16 subnormal (below 2^-14 = 6.1e-5), exactly representable
    for tag, xv, wv in (("x normal, w subnormal", 1.0, sub), ("x subnormal, w normal", sub, 1.0), ("x normal, w normal", 0.5, 0.25),
                        ("x subnormal, w 1024", sub, 1024.0)):
        x = torch.full((m, k), xv, dtype=torch.float16, device=dev)
        w = torch.full((n, k), wv, dtype=torch.float16, device=dev)
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.lsk_packed_bytes(n, k, ctypes.byref(nbytes)), lib)
        wp = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
        _lib.check(lib.lsk_pack_linear(w.data_ptr(), n, k, w.stride(0), wp.data_ptr(), 0, 1, 0, st), lib)
        y = torch.full((m, n), float("nan"), dtype=torch.float32, device=dev)
        lsk_test_lib.check(tl.lsk_test_gemm(x.data_ptr(), m, k, wp.data_ptr(), n, None, ctypes.c_float(1e-5), y.data_ptr(), 0, st), tl)
        torch.cuda.synchronize()
        want = k * float(x[0, 0].double()) * float(w[0, 0].double())
        out[tag] = {"got": float(y[0, 0]), "want": want, "kept": abs(float(y[0, 0]) - want) <= 1e-6 * abs(want)}
        print(f"  MFMA f16 {tag:26s}: got {float(y[0, 0]):.6e} want {want:.6e}", flush=True)
    # the conversion f32 -> f16 in the subnormal range (the P tile of the attention kernel, every epilogue)
    v = torch.tensor([sub, 1e-6, 2.0 ** -25 * 3], dtype=torch.float32, device=dev)
    print("  torch f32 -> f16 of", v.tolist(), "=", v.to(torch.float16).float().tolist(), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", nargs="?", default="full7b")
    ap.add_argument("--out", default=None)
    ap.add_argument("--layers", type=int, default=0, help="only the first N layers (0 = all)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    report = {"subnormal_probe": subnormal_probe(dev)}
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "struct_fp16", args.name + ".json")))
    model_cpu = build_struct_model(rec, "cpu")                   # bf16 values
    om16 = lo.OracleModel.from_hf(model_cpu, dtype=torch.float16)
    om32 = lo.OracleModel.from_hf(model_cpu, dtype=torch.float32)
    model = build_struct_model(rec, "cpu").to(torch.float16).to(dev)
    eng = get_engine(model)
    seq = rec["prompt"] + rec["fp16"]["spec_tokens"]
    n = len(seq)
    eng.ensure_capacity(n + 4, n)
    eng.reset()
    eng.embed_rows(seq, BUF_BULK, 0)
    h_prev = eng.read_rows(BUF_BULK, 0, n).cpu()
    pos = torch.arange(n).unsqueeze(0)
    mask16 = lo.decoder_mask(n, n, torch.float16, 0)
    mask32 = lo.decoder_mask(n, n, torch.float32, 0)
    L = args.layers or eng.num_layers
    rows = []
    print(f"== {args.name}: {n} rows, {L} layers; errors in fp16 ulp of the fp32 result for the SAME (engine) input rows", flush=True)
    with torch.inference_mode():
        for l in range(L):
            eng.run_layers_chunked(BUF_BULK, 0, n, 0, l, l + 1)
            h_eng = eng.read_rows(BUF_BULK, 0, n).cpu()
            t32, _ = lo.decoder_layer(om32, om32.layers[l], h_prev.float()[None], mask32, pos, None)
            r16, _ = lo.decoder_layer(om16, om16.layers[l], h_prev[None], mask16, pos, None)
            # the two halves, same input: attention half = h + o_proj(attn(norm(h)))
            def halves(om, h, mask):
                lw = om.layers[l]
                b, m, _ = h.shape
                x = lo.rms_norm(h, lw.input_norm, om.eps)
                q = F.linear(x, lw.q).view(b, m, -1, om.head_dim).transpose(1, 2)
                k = F.linear(x, lw.k).view(b, m, -1, om.head_dim).transpose(1, 2)
                v = F.linear(x, lw.v).view(b, m, -1, om.head_dim).transpose(1, 2)
                cos, sin = lo.rope_cos_sin(om.inv_freq, om.attention_scaling, pos, h.dtype)
                q, k = lo.apply_rope(q, k, cos, sin)
                a = lo.attention_core(om, q, k, v, mask).reshape(b, m, -1)
                return a, h + F.linear(a, lw.o)
            a32, mid32 = halves(om32, h_prev.float()[None], mask32)
            a16, mid16 = halves(om16, h_prev[None], mask16)
            floor = float(t32.pow(2).mean().sqrt()) / 8
            e_rms, e_max = err_stats(h_eng, t32[0], floor)
            r_rms, r_max = err_stats(r16[0].float(), t32[0], floor)
            fa = float(a32.pow(2).mean().sqrt()) / 8
            ra_rms, _ = err_stats(a16[0].float(), a32[0], fa)
            rm_rms, _ = err_stats(mid16[0].float(), mid32[0], floor)
            rows.append({"layer": l, "engine_rms": e_rms, "engine_max": e_max, "reference_fp16_rms": r_rms, "reference_fp16_max": r_max,
                         "reference_attn_out_rms": ra_rms, "reference_mid_rms": rm_rms})
            print(f"  layer {l:2d}: engine rms {e_rms:.3f} max {e_max:6.2f} | reference-fp16 rms {r_rms:.3f} max {r_max:6.2f} | ratio {e_rms / max(r_rms, 1e-9):.3f}"
                  f" | ref attention-out rms {ra_rms:.3f}, ref mid rms {rm_rms:.3f}", flush=True)
            h_prev = h_eng
    report["layers"] = rows
    tot_e = sum(r["engine_rms"] ** 2 for r in rows) ** 0.5
    tot_r = sum(r["reference_fp16_rms"] ** 2 for r in rows) ** 0.5
    print(f"  root-sum-square over layers: engine {tot_e:.3f}, reference-fp16 {tot_r:.3f}, ratio {tot_e / tot_r:.3f}", flush=True)
    report["rss"] = {"engine": tot_e, "reference_fp16": tot_r}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
