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
import math
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
    # accumulation accuracy of the projection kernel (MFMA products summed in fp32, 8 waves' partial sums, K-chunks): fp32-out GEMM on
    # random operands against fp64, in units of eps32 * sum|a||b| -- both element types (the bf16 library is the control)
    for dname, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        l2, t2 = _lib.load(dtype=dname), lsk_test_lib.load(dtype=dname)
        g = torch.Generator().manual_seed(3)
        for k2, outlier in ((4096, False), (4096, True), (11008, False)):
            m2, n2 = 8, 256
            x = torch.randn(m2, k2, generator=g)
            if outlier:
                x[:, ::97] *= 64.0
            x = x.to(dt).to(dev)
            w = (torch.randn(n2, k2, generator=g) * 0.02).to(dt).to(dev)
            nbytes = ctypes.c_size_t(0)
            _lib.check(l2.lsk_packed_bytes(n2, k2, ctypes.byref(nbytes)), l2)
            wp = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
            _lib.check(l2.lsk_pack_linear(w.data_ptr(), n2, k2, w.stride(0), wp.data_ptr(), 0, 1, 0, st), l2)
            y = torch.full((m2, n2), float("nan"), dtype=torch.float32, device=dev)
            lsk_test_lib.check(t2.lsk_test_gemm(x.data_ptr(), m2, k2, wp.data_ptr(), n2, None, ctypes.c_float(1e-5), y.data_ptr(), 0, st), t2)
            torch.cuda.synchronize()
            ref = x.double() @ w.double().t()
            mag = x.double().abs() @ w.double().abs().t()
            tref = (x.float() @ w.float().t()).double()
            e = ((y.double() - ref).abs() / mag).max().item() / 2.0 ** -24
            e_t = ((tref - ref).abs() / mag).max().item() / 2.0 ** -24
            rel = ((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            rel_t = ((tref - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            out[f"gemm accuracy {dname} k={k2} outliers={outlier}"] = {"max_err_over_eps32_sum_abs": e, "torch_fp32_matmul_same": e_t, "rms_rel": rel, "torch_rms_rel": rel_t}
            print(f"  GEMM {dname} K={k2} outliers={outlier}: max err / (eps32 * sum|a||b|) = {e:.3f} (torch fp32 matmul {e_t:.3f}); rms rel err {rel:.2e} (torch {rel_t:.2e})", flush=True)
    # the conversion f32 -> f16 in the subnormal range (the P tile of the attention kernel, every epilogue)
    v = torch.tensor([sub, 1e-6, 2.0 ** -25 * 3], dtype=torch.float32, device=dev)
    print("  torch f32 -> f16 of", v.tolist(), "=", v.to(torch.float16).float().tolist(), flush=True)
    return out


def stage_pass(eng, om, l, x_in, n, dev):
    """The five launches of decoder layer l in isolation (liblayerskip_hip_test_f16.so: exactly the engine's kernels), each fed the
    REFERENCE's fp16 intermediates of the stage before it, against (a) the reference's fp16 result of the stage and (b) the stage in
    fp64 on the same fp16 inputs without any rounding inside."""
    import lsk_test_lib
    tl = lsk_test_lib.load(dtype="fp16")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lw = om.layers[l]
    hd, nh, nkv, H = om.head_dim, om.n_heads, om.n_kv_heads, x_in.shape[1]
    qdim, kvdim = nh * hd, nkv * hd
    wqkv, wo, wgu, wdown, n1, n2 = eng._packed[l]
    cos_t, sin_t = eng._buffers["cos"], eng._buffers["sin"]
    pos = torch.arange(n).unsqueeze(0)
    f16 = torch.float16
    out = {}

    def cmp(tag, mine, ref16, truth64):
        mine, ref16 = mine.cpu(), ref16.cpu()
        floor = float(truth64.pow(2).mean().sqrt()) / 8
        e_rms, e_max = err_stats(mine, truth64, floor)
        r_rms, r_max = err_stats(ref16, truth64, floor)
        same = float((mine == ref16).float().mean())
        d = (mine.double() - ref16.double()).abs() / ulp16(truth64, floor)
        out[tag] = {"engine_rms": e_rms, "engine_max": e_max, "reference_rms": r_rms, "reference_max": r_max, "bit_equal_share": same,
                    "max_ulp_between_them": float(d.max())}
        print(f"    {tag:34s} engine rms {e_rms:.3f} max {e_max:6.2f} | reference-fp16 rms {r_rms:.3f} max {r_max:6.2f} | bit-equal {same:.4f}, "
              f"worst engine-vs-reference {float(d.max()):.2f} ulp", flush=True)

    print(f"  -- stages of layer {l} (fp16 ulp of the stage's fp64 result on the SAME fp16 inputs)", flush=True)
    with torch.inference_mode():
        # ---- the reference's fp16 intermediates ----
        x16 = x_in[None]
        xn = lo.rms_norm(x16, lw.input_norm, om.eps)
        q = F.linear(xn, lw.q).view(1, n, -1, hd).transpose(1, 2)
        k = F.linear(xn, lw.k).view(1, n, -1, hd).transpose(1, 2)
        v = F.linear(xn, lw.v).view(1, n, -1, hd).transpose(1, 2)
        cos, sin = lo.rope_cos_sin(om.inv_freq, om.attention_scaling, pos, f16)
        qr, kr = lo.apply_rope(q, k, cos, sin)
        mask = lo.decoder_mask(n, n, f16, 0)
        a16 = lo.attention_core(om, qr, kr, v, mask).reshape(1, n, -1)
        mid16 = x16 + F.linear(a16, lw.o)
        xn2 = lo.rms_norm(mid16, lw.post_norm, om.eps)
        act16 = F.silu(F.linear(xn2, lw.gate)) * F.linear(xn2, lw.up)
        out16 = mid16 + F.linear(act16, lw.down)
        # ---- fp64 statements of each stage on the fp16 inputs the engine stage gets ----
        d = torch.float64
        x64 = x_in.double()
        xn64 = x64 * torch.rsqrt(x64.pow(2).mean(-1, keepdim=True) + om.eps) * lw.input_norm.double()
        def rope64(t):          # t [n, heads, hd]
            c, s_ = cos[0].double()[:, None, :], sin[0].double()[:, None, :]
            rot = torch.cat((-t[..., hd // 2:], t[..., : hd // 2]), dim=-1)
            return t * c + rot * s_
        q64 = rope64((xn64 @ lw.q.double().t()).view(n, nh, hd)).reshape(n, qdim)
        k64 = rope64((xn64 @ lw.k.double().t()).view(n, nkv, hd)).reshape(n, kvdim)
        v64 = (xn64 @ lw.v.double().t())
        # ---- stage A: q/k/v + RoPE + KV append ----
        n_pages = (n + 127) // 128 + 1
        kpool = torch.zeros(n_pages, nkv, 128, hd, dtype=f16, device=dev)
        vpool = torch.zeros(n_pages, nkv, hd, 128, dtype=f16, device=dev)
        table = torch.arange(n_pages, dtype=torch.int32, device=dev)
        zero = torch.zeros(1, dtype=torch.int32, device=dev)
        xd = x_in.to(dev)
        q_eng = torch.zeros(n, qdim, dtype=f16, device=dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            lsk_test_lib.check(tl.lsk_test_qkv(xd[r0:r0 + m].data_ptr(), m, H, wqkv.data_ptr(), n1.data_ptr(), ctypes.c_float(om.eps), nh, nkv, hd,
                                               cos_t.data_ptr(), sin_t.data_ptr(), zero.data_ptr(), r0, table.data_ptr(), q_eng[r0:r0 + m].data_ptr(),
                                               kpool.data_ptr(), vpool.data_ptr(), st), tl)
        torch.cuda.synchronize()
        k_eng = torch.stack([kpool[p // 128, :, p % 128, :] for p in range(n)]).reshape(n, kvdim)
        v_eng = torch.stack([vpool[p // 128, :, :, p % 128] for p in range(n)]).reshape(n, kvdim)
        cmp("A q (norm + proj + RoPE)", q_eng, qr[0].transpose(0, 1).reshape(n, qdim), q64)
        cmp("A k (norm + proj + RoPE)", k_eng, kr[0].transpose(0, 1).reshape(n, kvdim), k64)
        cmp("A v (norm + proj)", v_eng, v[0].transpose(0, 1).reshape(n, kvdim), v64)
        # ---- stage B: attention on the REFERENCE's q / k / v ----
        g = nh // nkv
        qd, kd, vd = qr[0].double(), kr[0].double().repeat_interleave(g, 0), v[0].double().repeat_interleave(g, 0)
        sc = torch.einsum("hmd,hcd->hmc", qd, kd) / hd ** 0.5
        sc = sc.masked_fill(torch.arange(n)[None, None, :] > torch.arange(n)[None, :, None], float("-inf"))
        a64 = torch.einsum("hmc,hcd->mhd", torch.softmax(sc, -1), vd).reshape(n, qdim)
        kpool.zero_(); vpool.zero_()
        for p in range(n):
            kpool[p // 128, :, p % 128, :] = kr[0, :, p, :].to(dev)
            vpool[p // 128, :, :, p % 128] = v[0, :, p, :].to(dev)
        nb = ctypes.c_size_t(0)
        lsk_test_lib.check(tl.lsk_test_attention_scratch_bytes(nh, hd, n_pages, ctypes.byref(nb)), tl)
        scratch = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
        qdev = qr[0].transpose(0, 1).reshape(n, qdim).contiguous().to(dev)
        a_eng = torch.zeros(n, qdim, dtype=f16, device=dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            kvl = torch.tensor([r0], dtype=torch.int32, device=dev)
            lsk_test_lib.check(tl.lsk_test_attention(qdev[r0:r0 + m].data_ptr(), m, nh, nkv, hd, kpool.data_ptr(), vpool.data_ptr(), table.data_ptr(),
                                                     n_pages, kvl.data_ptr(), r0, 0, scratch.data_ptr(), nb.value, a_eng[r0:r0 + m].data_ptr(), 0, st), tl)
            torch.cuda.synchronize()
        cmp("B attention", a_eng, a16[0], a64)
        # ---- stage C: o_proj + residual on the reference's attention rows ----
        mid64 = x64 + a16[0].double() @ lw.o.double().t()
        h_io = x_in.clone().to(dev)
        ad = a16[0].contiguous().to(dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            lsk_test_lib.check(tl.lsk_test_resid(ad[r0:r0 + m].data_ptr(), m, qdim, wo.data_ptr(), H, h_io[r0:r0 + m].data_ptr(), st), tl)
        torch.cuda.synchronize()
        cmp("C o_proj + residual", h_io, mid16[0], mid64)
        # ---- stage D: post-attention norm + gate/up + SiLU * up on the reference's mid rows ----
        m64 = mid16[0].double()
        xn2_64 = m64 * torch.rsqrt(m64.pow(2).mean(-1, keepdim=True) + om.eps) * lw.post_norm.double()
        g64 = xn2_64 @ lw.gate.double().t()
        act64 = g64 * torch.sigmoid(g64) * (xn2_64 @ lw.up.double().t())
        I = lw.gate.shape[0]
        act_eng = torch.zeros(n, I, dtype=f16, device=dev)
        md = mid16[0].contiguous().to(dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            lsk_test_lib.check(tl.lsk_test_swiglu(md[r0:r0 + m].data_ptr(), m, H, wgu.data_ptr(), n2.data_ptr(), ctypes.c_float(om.eps), I,
                                                  act_eng[r0:r0 + m].data_ptr(), st), tl)
        torch.cuda.synchronize()
        cmp("D norm + gate/up + SiLU*up", act_eng, act16[0], act64)
        # ---- stage E: down_proj + residual on the reference's activation rows ----
        out64 = m64 + act16[0].double() @ lw.down.double().t()
        h_io = mid16[0].clone().to(dev)
        acd = act16[0].contiguous().to(dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            lsk_test_lib.check(tl.lsk_test_resid(acd[r0:r0 + m].data_ptr(), m, I, wdown.data_ptr(), H, h_io[r0:r0 + m].data_ptr(), st), tl)
        torch.cuda.synchronize()
        cmp("E down_proj + residual", h_io, out16[0], out64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", nargs="?", default="full7b")
    ap.add_argument("--out", default=None)
    ap.add_argument("--layers", type=int, default=0, help="only the first N layers (0 = all)")
    ap.add_argument("--stages", default="1,12", help="layers whose five stages are run in isolation through the test library")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    report = {"subnormal_probe": subnormal_probe(dev)}
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "struct_fp16", args.name + ".json")))
    # what `torch_dtype=torch.float16` does to the bf16 checkpoint (generate.py:59-64): the PARAMETERS become fp16, the rotary inv_freq
    # buffer stays fp32 -- `model.to(float16)` would round it (the first version of this tool did, like rounds 2-5's fixture recipe for
    # checkpoints below 2e9 parameters: the "rounding point" behind the fp16 gate's 1.10 was there, not in a kernel)
    from conftest import cast_parameters
    model_cpu = cast_parameters(build_struct_model(rec, "cpu"), torch.float16)
    om16 = lo.OracleModel.from_hf(model_cpu)
    om32 = lo.OracleModel.from_hf(model_cpu, dtype=torch.float32)
    print("  inv_freq dtype of the converted model:", model_cpu.model.rotary_emb.inv_freq.dtype, flush=True)
    model = cast_parameters(build_struct_model(rec, "cpu"), torch.float16).to(dev)
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
    layer_inputs = {}
    print(f"== {args.name}: {n} rows, {L} layers; errors in fp16 ulp of the fp32 result for the SAME (engine) input rows", flush=True)
    with torch.inference_mode():
        for l in range(L):
            eng.run_layers_chunked(BUF_BULK, 0, n, 0, l, l + 1)
            h_eng = eng.read_rows(BUF_BULK, 0, n).cpu()
            layer_inputs[l] = h_prev
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
    if L == eng.num_layers:
        # ---- end to end: the engine's own 32-layer trajectory and its head against the reference's fp16 run and the fp32 truth, on EVERY
        #      vocabulary entry of every row (the test's gate looks at 32 entries of 20 rows: rms ratios of 640 samples scatter by ~3 %) ----
        with torch.inference_mode():
            t_log = lo.teacher_forced_logits(om32, seq).double()
            r_log = lo.teacher_forced_logits(om16, seq).double()
        e_log = torch.empty(n, eng.vocab, dtype=torch.float32, device=dev)
        for r0 in range(0, n, 16):
            m = min(16, n - r0)
            eng.run_head(BUF_BULK, r0, m, logits=e_log[r0:r0 + m], want_tokens=False)
        e_log = e_log.cpu().double()
        u = torch.pow(2.0, torch.floor(torch.log2(t_log.abs().clamp_min(1.0))) - 10)
        ee, er = (e_log - t_log).abs() / u, (r_log - t_log).abs() / u
        top = torch.topk(t_log, 16, dim=-1).indices
        et, rt = ee.gather(1, top), er.gather(1, top)
        gen = slice(len(rec["prompt"]) - 1, n)
        end = {"all_entries": {"engine_rms": float(ee.pow(2).mean().sqrt()), "reference_rms": float(er.pow(2).mean().sqrt())},
               "top16": {"engine_rms": float(et.pow(2).mean().sqrt()), "reference_rms": float(rt.pow(2).mean().sqrt())},
               "top16_generated_rows": {"engine_rms": float(et[gen].pow(2).mean().sqrt()), "reference_rms": float(rt[gen].pow(2).mean().sqrt())},
               "engine_vs_reference_bit_equal_share": float((e_log == r_log).float().mean())}
        print("  end to end, logits in fp16 ulp of the fp32 logits:", json.dumps(end), flush=True)
        report["end_to_end"] = end
        # ---- the test's own statistic (tests/test_gpu_zz_fp16.py) on the fixture's recorded entries, split by what it mixes: full-depth /
        #      early-exit rows, top-16 / strided entries; against the fixture's fp32 values AND against this tool's fp32 oracle ----
        gate = {}
        for key, layer_end in (("logits", eng.num_layers), ("early_logits", rec["exit_layer"])):
            eng.reset()
            eng.embed_rows(seq, BUF_BULK, 0)
            eng.run_layers_chunked(BUF_BULK, 0, n, 0, 0, layer_end)
            with torch.inference_mode():
                if key == "logits":
                    o32, o16 = t_log, r_log
                else:
                    o32 = lo.forward_early(om32, torch.tensor([seq]), None, rec["exit_layer"], None).logits[0].double()
                    o16 = lo.forward_early(om16, torch.tensor([seq]), None, rec["exit_layer"], None).logits[0].double()
            acc = {}
            for row in rec["fp16"][key]:
                buf = torch.empty(1, eng.vocab, dtype=torch.float32, device=dev)
                eng.run_head(BUF_BULK, row["row"], 1, logits=buf, want_tokens=False)
                mine = buf[0].cpu().double()
                order = sorted(range(len(row["idx"])), key=lambda i: -row["val"][i])
                top = set(order[:16])
                for i, (idx, b, x) in enumerate(zip(row["idx"], row["val"], row["val_fp32"])):
                    cls = "top16" if i in top else "strided"
                    a = float(mine[idx])
                    u = 2.0 ** (math.floor(math.log2(max(abs(x), 1.0))) - 10)
                    u2 = 2.0 ** (math.floor(math.log2(max(abs(float(o32[row["row"], idx])), 1.0))) - 10)
                    d = acc.setdefault(cls, [0.0] * 8)
                    d[0] += ((a - x) / u) ** 2; d[1] += ((b - x) / u) ** 2; d[2] += 1
                    d[3] += ((a - float(o32[row["row"], idx])) / u2) ** 2; d[4] += ((float(o16[row["row"], idx]) - float(o32[row["row"], idx])) / u2) ** 2
                    d[5] += float(float(o16[row["row"], idx]) == b); d[6] += ((x - float(o32[row["row"], idx])) / u) ** 2
            gate[key] = {c: {"n": int(d[2]), "engine_vs_fixture_fp32": (d[0] / d[2]) ** 0.5, "reference_vs_fixture_fp32": (d[1] / d[2]) ** 0.5,
                             "engine_vs_tool_fp32": (d[3] / d[2]) ** 0.5, "tool_fp16_vs_tool_fp32": (d[4] / d[2]) ** 0.5,
                             "tool_fp16_equals_fixture_fp16_share": d[5] / d[2], "fixture_fp32_vs_tool_fp32": (d[6] / d[2]) ** 0.5} for c, d in acc.items()}
            print(f"  fixture gate, {key}:", json.dumps(gate[key]), flush=True)
        report["fixture_gate"] = gate
    for l in [int(v) for v in args.stages.split(",") if v != ""]:
        if l in layer_inputs:
            report[f"stages_layer_{l}"] = stage_pass(eng, om16, l, layer_inputs[l], n, dev)
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
