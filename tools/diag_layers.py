"""GPU diagnostic (not a test): where, layer by layer and stage by stage, the engine's bf16 values leave the
reference-pinned restatement (oracle/llama_oracle.py on the host CPU).

    python tools/diag_layers.py tiny_gqa_long [more struct fixture names]

Per layer l: the oracle's decoder layer is fed the ENGINE's hidden rows of layer l-1 (teacher-forced single pass), so each
line shows the error ONE layer adds: fraction of elements within 1 bf16 ulp of the oracle, worst element in ulp.  Then, for
the worst layer, the stages in isolation through the lsk_test_* exports on the oracle's own intermediate tensors."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import build_struct_model, load_struct  # noqa: E402
from layerskip_amd import _lib  # noqa: E402
from layerskip_amd.engine import BUF_BULK, get_engine  # noqa: E402
from oracle import llama_oracle as lo  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def ulp_of(t):
    return torch.pow(2.0, torch.floor(torch.log2(t.abs().clamp_min(1e-30))) - 7)


def report(tag, mine, ref, scale_floor=None):
    mine, ref = mine.float().cpu(), ref.float().cpu()
    base = ref.abs()
    if scale_floor is not None:
        base = base.clamp_min(scale_floor)
    u = ulp_of(base)
    e = (mine - ref).abs() / u
    print(f"  {tag:34s} within1 {float((e <= 1).float().mean()):.4f}  exact {float((e == 0).float().mean()):.4f}  "
          f"worst {float(e.max()):7.2f} ulp  max|d| {float((mine - ref).abs().max()):.5f}  rms(ref) {float(ref.pow(2).mean().sqrt()):.4f}", flush=True)
    return e


def main():
    dev = torch.device("cuda:0")
    for name in sys.argv[1:] or ["tiny_gqa_long"]:
        rec = load_struct(name)
        model_cpu = build_struct_model(rec)
        om = lo.OracleModel.from_hf(model_cpu)
        model = build_struct_model(rec, dev)
        eng = get_engine(model)
        seq = rec["prompt"] + rec["bf16"]["spec_tokens"]
        n = len(seq)
        eng.ensure_capacity(n + 4, n)
        eng.reset()
        eng.embed_rows(seq, BUF_BULK, 0)
        ids = torch.tensor([seq])
        h_prev = eng.read_rows(BUF_BULK, 0, n).cpu()
        print(f"== {name}: {n} rows, {eng.num_layers} layers", flush=True)
        assert torch.equal(h_prev, F.embedding(ids, om.embed)[0])
        mask = lo.decoder_mask(n, n, torch.bfloat16, 0)
        pos = torch.arange(n).unsqueeze(0)
        worst_layer, worst_val = 0, -1.0
        hs = [h_prev]
        with torch.inference_mode():
            for l in range(eng.num_layers):
                eng.run_layers_chunked(BUF_BULK, 0, n, 0, l, l + 1)
                h_eng = eng.read_rows(BUF_BULK, 0, n).cpu()
                h_ref, _ = lo.decoder_layer(om, om.layers[l], hs[-1][None], mask, pos, None)
                e = report(f"layer {l} (fed engine rows)", h_eng, h_ref[0], scale_floor=float(h_ref.float().pow(2).mean().sqrt()) / 8)
                if float(e.max()) > worst_val:
                    worst_layer, worst_val = l, float(e.max())
                hs.append(h_eng)
            # ---- stages of the worst layer in isolation, on the oracle's intermediates ----
            l = worst_layer
            lw = om.layers[l]
            x_in = hs[l][None]
            print(f"  -- stages of layer {l}", flush=True)
            xn = lo.rms_norm(x_in, lw.input_norm, om.eps)
            q = F.linear(xn, lw.q).view(1, n, -1, om.head_dim).transpose(1, 2)
            k = F.linear(xn, lw.k).view(1, n, -1, om.head_dim).transpose(1, 2)
            v = F.linear(xn, lw.v).view(1, n, -1, om.head_dim).transpose(1, 2)
            cos, sin = lo.rope_cos_sin(om.inv_freq, om.attention_scaling, pos, torch.bfloat16)
            qr, kr = lo.apply_rope(q, k, cos, sin)
            a = lo.attention_core(om, qr, kr, v, mask)                     # [1, n, heads, hd]
            a2 = a.reshape(1, n, -1)
            # fp64 attention on the same bf16 q/k/v: what an exact kernel would give
            g = om.n_heads // om.n_kv_heads
            sc = torch.einsum("hmd,hcd->hmc", qr[0].double(), kr[0].double().repeat_interleave(g, 0)) / om.head_dim ** 0.5
            sc = sc.masked_fill(torch.arange(n)[None, None, :] > torch.arange(n)[None, :, None], float("-inf"))
            a64 = torch.einsum("hmc,hcd->mhd", torch.softmax(sc, -1), v[0].double().repeat_interleave(g, 0)).reshape(n, -1)
            report("oracle SDPA vs fp64 attention", a2[0], a64.float(), scale_floor=float(a64.pow(2).mean().sqrt()) / 8)
            # engine attention kernel on the oracle's q / k / v
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import lsk_test_lib
            lib = lsk_test_lib.load()
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            n_pages = (n + 127) // 128 + 1
            kpool = torch.zeros(n_pages, om.n_kv_heads, 128, om.head_dim, dtype=torch.bfloat16)
            vpool = torch.zeros(n_pages, om.n_kv_heads, om.head_dim, 128, dtype=torch.bfloat16)
            for p in range(n):
                kpool[p // 128, :, p % 128, :] = kr[0, :, p, :]
                vpool[p // 128, :, :, p % 128] = v[0, :, p, :]
            kpool, vpool = kpool.to(dev), vpool.to(dev)
            table = torch.arange(n_pages, dtype=torch.int32, device=dev)
            nb = ctypes.c_size_t(0)
            lsk_test_lib.check(lib.lsk_test_attention_scratch_bytes(om.n_heads, om.head_dim, n_pages, ctypes.byref(nb)))
            scratch = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
            qd = qr[0].transpose(0, 1).reshape(n, -1).contiguous().to(dev)
            outs = []
            for r0 in range(0, n, 16):
                m = min(16, n - r0)
                out = torch.zeros(m, om.n_heads * om.head_dim, dtype=torch.bfloat16, device=dev)
                kvl = torch.tensor([r0], dtype=torch.int32, device=dev)
                lsk_test_lib.check(lib.lsk_test_attention(qd[r0:r0 + m].data_ptr(), m, om.n_heads, om.n_kv_heads, om.head_dim, kpool.data_ptr(),
                                                  vpool.data_ptr(), table.data_ptr(), n_pages, kvl.data_ptr(), r0, 0, scratch.data_ptr(),
                                                  nb.value, out.data_ptr(), 0, st))
                torch.cuda.synchronize()
                outs.append(out.cpu())
            a_eng = torch.cat(outs)
            fl = float(a64.pow(2).mean().sqrt()) / 8
            report("engine attention vs fp64", a_eng, a64.float(), scale_floor=fl)
            report("engine attention vs oracle SDPA", a_eng, a2[0], scale_floor=fl)
        eng.reset()
        del eng, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
