"""Isolated, BIT-EXACT checks of the fused epilogues and tight checks of the attention kernels, through the C-ABI
(`lsk_test_*`, include/layerskip_hip_test.h, liblayerskip_hip_test.so), against plain torch statements of the HF ops they replace.

The projections use one-hot ("permutation") weights and activations whose RMS statistics are exact, so the GEMM and
the RMSNorm are exact and the epilogue under test must reproduce torch's bf16 arithmetic bit for bit:
  EPI_QKV     q * cos + rotate_half(q) * sin in bf16 (modeling_llama.py:138-160), half-split layout, llama3 RoPE
              scaling, K page [slot][d] / V^T page [d][slot] addressing across a page boundary and a shuffled block table;
  EPI_SWIGLU  bf16(silu(gate)) * up (modeling_llama.py:174-176);
  EPI_RESID   h + Linear(x) in bf16 (modeling_llama.py:317,323);
  EPI_HEAD    torch.argmax's lowest-index tie-break with DUPLICATED lm_head rows placed in one tile, in two tiles of one
              workgroup, in different workgroups and at the ragged end of the vocabulary (llama_model_utils.py:120-122).
Attention: softmax_fp32(Q K^T / sqrt(d)) V on the same bf16 Q / K / V, rows 1 / 7 / 16, contexts straddling 1, 2 and 9
pages, MHA and GQA, d = 128 and 64, unwritten KV slots poisoned with NaN.
"""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
PAGE = 128


def _lib():
    from layerskip_amd import _lib
    return _lib.load(), _lib


def _tlib():
    """liblayerskip_hip_test.so (include/layerskip_hip_test.h): the engine's kernels on caller-owned buffers."""
    import lsk_test_lib
    return lsk_test_lib.load(), lsk_test_lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack_into(dst, w, tile_offset=0, tile_stride=1, rope_hd=0):
    lib, L = _lib()
    L.check(lib.lsk_pack_linear(w.data_ptr(), w.shape[0], w.shape[1], w.stride(0), dst.data_ptr(), tile_offset, tile_stride,
                                rope_hd, _stream()))


def _packed(n_rows, k, dev):
    lib, L = _lib()
    nbytes = ctypes.c_size_t(0)
    L.check(lib.lsk_packed_bytes(n_rows, k, ctypes.byref(nbytes)))
    return torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)


def _exact_rows(m, k, g):
    """Rows with mean(x^2) == 1 exactly (k/2 entries of 0.5, 3k/8 of 1, k/8 of 2, random signs and order): the RMS
    statistics are exact in any summation order and x * rsqrt(1 + eps) rounds back to x in bf16."""
    assert k % 8 == 0
    mags = torch.cat([torch.full((k // 2,), 0.5), torch.full((3 * k // 8,), 1.0), torch.full((k // 8,), 2.0)])
    rows = []
    for _ in range(m):
        p = torch.randperm(k, generator=g)
        s = torch.randint(0, 2, (k,), generator=g) * 2 - 1
        rows.append(mags[p] * s)
    return torch.stack(rows).to(BF)


def _gain(k, g):
    vals = torch.tensor([0.5, 0.75, 1.0, 1.25, 1.5, 2.0])
    return vals[torch.randint(0, len(vals), (k,), generator=g)].to(BF)


def _one_hot_rows(n, k, g, signs=True):
    """nn.Linear weight [n][k] whose row j picks input feature sel[j] (times +-1 or +-2): exact dot products."""
    sel = torch.randint(0, k, (n,), generator=g)
    scale = torch.tensor([1.0, -1.0, 2.0, -2.0, 0.5])[torch.randint(0, 5 if signs else 1, (n,), generator=g)]
    w = torch.zeros(n, k)
    w[torch.arange(n), sel] = scale
    return w.to(BF), sel, scale


def _rmsnorm_bf16(x, gain, eps):
    x32 = x.float()
    xn = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)).to(BF)
    return gain * xn


def _rope_table(shape_name, length, dev):
    """cos / sin exactly as LlamaRotaryEmbedding.forward returns them (fp32 -> bf16), [pos][head_dim]."""
    from layerskip_amd import synthetic
    import transformers
    cfg = synthetic.make_config(shape_name)
    rot = transformers.models.llama.modeling_llama.LlamaRotaryEmbedding(cfg)
    pos = torch.arange(length)[None, :]
    cos, sin = rot(torch.zeros(1, 1, dtype=BF), pos)
    return cos[0].to(dev), sin[0].to(dev)          # [length][hd] bf16


@pytest.mark.parametrize("shape_name,n_heads,n_kv,hd,m,kv_len", [
    ("tiny-gqa", 4, 2, 128, 16, 120),      # rows 120..135 straddle the page boundary at 128; theta 5e5
    ("tiny-mha", 2, 2, 128, 7, 0),         # first positions (cos = 1, sin = 0 at position 0)
    ("tiny-d64", 8, 2, 64, 5, 253),        # head_dim 64, llama3 frequency scaling, second page boundary
    ("tiny-gqa", 4, 2, 128, 1, 1151),      # one row on the LAST slot of page 8
    ("tiny-gqa", 4, 2, 128, 9, 124),       # the 10-row template (llama2-13B's verify pass), across a page boundary
    ("tiny-mha", 2, 2, 128, 13, 500),      # the 13-row template (llama2-70B's verify pass)
])
def test_qkv_rope_kv_append_bit_exact(gpu_device, shape_name, n_heads, n_kv, hd, m, kv_len):
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb
    lib, L = _lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(1000 * m + kv_len + hd)
    H = 512
    eps = 1e-5
    qdim, kvdim = n_heads * hd, n_kv * hd
    x = _exact_rows(m, H, g)
    gain = _gain(H, g)
    wq, _, _ = _one_hot_rows(qdim, H, g)
    wk, _, _ = _one_hot_rows(kvdim, H, g)
    wv, _, _ = _one_hot_rows(kvdim, H, g)
    # ---- torch statement (CPU bf16 ops) ----
    xn = _rmsnorm_bf16(x, gain, eps)
    assert torch.equal(xn.float(), (x.float() * gain.float()))          # the construction: the norm is exact
    q = (xn.float() @ wq.float().t()).to(BF).view(1, m, n_heads, hd).transpose(1, 2)
    k = (xn.float() @ wk.float().t()).to(BF).view(1, m, n_kv, hd).transpose(1, 2)
    v = (xn.float() @ wv.float().t()).to(BF).view(m, n_kv, hd)
    n_pages = 10
    max_ctx = n_pages * PAGE
    cos_t, sin_t = _rope_table(shape_name, max_ctx, "cpu")
    pos = torch.arange(kv_len, kv_len + m)
    qr, kr = apply_rotary_pos_emb(q, k, cos_t[pos][None], sin_t[pos][None])
    qr = qr.transpose(1, 2).reshape(m, qdim)
    kr = kr.transpose(1, 2).reshape(m, n_kv, hd)
    # ---- kernel ----
    wp = _packed(qdim + 2 * kvdim, H, dev)
    _pack_into(wp, wq.to(dev), 0, 1, hd)
    _pack_into(wp, wk.to(dev), qdim // 16, 1, hd)
    _pack_into(wp, wv.to(dev), (qdim + kvdim) // 16, 1, 0)
    table = torch.randperm(n_pages, generator=g).to(torch.int32)
    kpool = torch.full((n_pages, n_kv, PAGE, hd), 7.0, dtype=BF, device=dev)
    vpool = torch.full((n_pages, n_kv, hd, PAGE), 7.0, dtype=BF, device=dev)
    q_out = torch.zeros(m, qdim, dtype=BF, device=dev)
    kvl = torch.tensor([kv_len], dtype=torch.int32, device=dev)
    cos_h = cos_t[:, : hd // 2].contiguous().to(dev)
    sin_h = sin_t[:, : hd // 2].contiguous().to(dev)
    xd, gd, td = x.to(dev), gain.to(dev), table.to(dev)
    _tlib()[1].check(_tlib()[0].lsk_test_qkv(xd.data_ptr(), m, H, wp.data_ptr(), gd.data_ptr(), eps, n_heads, n_kv, hd, cos_h.data_ptr(),
                             sin_h.data_ptr(), kvl.data_ptr(), 0, td.data_ptr(), q_out.data_ptr(), kpool.data_ptr(),
                             vpool.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(q_out.cpu(), qr), "q rows (RoPE) differ from apply_rotary_pos_emb in bf16"
    kp, vp = kpool.cpu(), vpool.cpu()
    touched = torch.zeros(n_pages, PAGE, dtype=torch.bool)
    for i in range(m):
        p = kv_len + i
        page, slot = int(table[p // PAGE]), p % PAGE
        touched[page, slot] = True
        assert torch.equal(kp[page, :, slot, :], kr[i]), f"K row of position {p}"
        assert torch.equal(vp[page, :, :, slot], v[i]), f"V^T column of position {p}"
    # nothing else in the pool was written
    assert bool((kp.permute(0, 2, 1, 3)[~touched] == 7.0).all()) and bool((vp.permute(0, 3, 1, 2)[~touched] == 7.0).all())


@pytest.mark.parametrize("m,H,I", [(1, 512, 1408), (7, 4096, 1024), (16, 5120, 256), (9, 5120, 512), (13, 8192, 256)])
def test_swiglu_epilogue_bit_exact(gpu_device, m, H, I):
    lib, L = _lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(m + H + I)
    eps = 1e-5
    x = _exact_rows(m, H, g)
    gain = _gain(H, g)
    wg, _, _ = _one_hot_rows(I, H, g)
    wu, _, _ = _one_hot_rows(I, H, g)
    xn = _rmsnorm_bf16(x, gain, eps)
    gate = (xn.float() @ wg.float().t()).to(BF)
    up = (xn.float() @ wu.float().t()).to(BF)
    want = torch.nn.functional.silu(gate) * up                      # bf16 tensors: silu rounds, the product rounds
    wp = _packed(2 * I, H, dev)
    _pack_into(wp, wg.to(dev), 0, 2, 0)
    _pack_into(wp, wu.to(dev), 1, 2, 0)
    act = torch.zeros(m, I, dtype=BF, device=dev)
    xd, gd = x.to(dev), gain.to(dev)
    _tlib()[1].check(_tlib()[0].lsk_test_swiglu(xd.data_ptr(), m, H, wp.data_ptr(), gd.data_ptr(), eps, I, act.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(act.cpu(), want)
    assert gate.unique().numel() > 10                                # the check is not vacuous


@pytest.mark.parametrize("m,K,N", [(1, 4096, 4096), (7, 11008, 512), (16, 1408, 256), (9, 13824, 256), (13, 5120, 512)])
def test_residual_epilogue_bit_exact(gpu_device, m, K, N):
    lib, L = _lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(m + K + N)
    x = torch.randn(m, K, generator=g).to(BF)
    h = (3.0 * torch.randn(m, N, generator=g)).to(BF)
    w, _, _ = _one_hot_rows(N, K, g)
    want = h + (x.float() @ w.float().t()).to(BF)                    # bf16 + bf16 -> bf16
    wp = _packed(N, K, dev)
    _pack_into(wp, w.to(dev))
    hd_, xd = h.to(dev).clone(), x.to(dev)
    _tlib()[1].check(_tlib()[0].lsk_test_resid(xd.data_ptr(), m, K, wp.data_ptr(), N, hd_.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(hd_.cpu(), want)


@pytest.mark.parametrize("m,V,target_wgs", [(1, 32000, 0), (7, 32000, 0), (16, 1000, 0), (3, 128256, 0), (7, 32000, 37),
                                            (5, 50, 0), (13, 128256, 0), (1, 128256, 100)])
def test_lm_head_exact_ties_resolve_to_lowest_index(gpu_device, m, V, target_wgs):
    """Duplicated lm_head rows give EXACTLY equal logits; torch.argmax (CPU) returns the first one (LMU:121)."""
    lib, L = _lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(V + m)
    H = 256
    eps = 1e-5
    x = torch.randn(m, H, generator=g).to(BF)
    gain = (1 + 0.1 * torch.randn(H, generator=g)).to(BF)
    w = (0.02 * torch.randn(V, H, generator=g)).to(BF)
    xn = _rmsnorm_bf16(x, gain, eps)
    # row r of the activations gets its own winner direction, copied into several vocabulary rows
    tiles = (V + 15) // 16
    wg_tiles = max(1, -(-tiles // (target_wgs or 256)))      # (K is one chunk: an lm_head workgroup owns any number of tiles)
    base = [
        [5, 9],                                              # same tile
        [16 * 3 + 2, 16 * 4 + 2],                            # neighbouring tiles (same workgroup when it owns > 1 tile)
        [16 * wg_tiles * 2 + 1, 16 * wg_tiles * 7 + 1],      # different workgroups
        [7, V - 1],                                          # first tile and the ragged last tile
        [V - 4, V - 3, V - 2],                               # three copies at the end
        [16 * wg_tiles - 1, 16 * wg_tiles, 16 * wg_tiles * 3 + 15],   # across a workgroup boundary
        [0, V // 2, V - 6],
    ]
    used = set()
    placements = []
    for r in range(m):
        ids = sorted(set((i + 3 * (r // len(base))) for i in base[r % len(base)] if 0 <= i + 3 * (r // len(base)) < V) - used)
        if not ids:
            ids = [next(i for i in range(V) if i not in used)]
        used |= set(ids)
        placements.append(ids)
    want = []
    for r in range(m):
        ids = placements[r]
        for i in ids:
            w[i] = (xn[r].float() * 0.25).to(BF)
        want.append(ids[0])
    assert sum(len(p) > 1 for p in placements) >= min(m, 3)
    # the torch statement: bf16 logits, first max wins
    logits_ref = (xn.float() @ w.float().t()).to(BF)
    wp = _packed(V, H, dev)
    _pack_into(wp, w.to(dev))
    nb = ctypes.c_size_t(0)
    _tlib()[1].check(_tlib()[0].lsk_test_head_scratch_bytes(V, ctypes.byref(nb)))
    scratch = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    ld = (V + 3) // 4 * 4
    logits = torch.zeros(m, ld, dtype=torch.float32, device=dev)
    toks = torch.full((m,), -1, dtype=torch.int32, device=dev)
    xd, gd = x.to(dev), gain.to(dev)
    _tlib()[1].check(_tlib()[0].lsk_test_head(xd.data_ptr(), m, H, wp.data_ptr(), gd.data_ptr(), eps, V, target_wgs, scratch.data_ptr(),
                              logits.data_ptr(), ld, toks.data_ptr(), _stream()))
    torch.cuda.synchronize()
    got_logits = logits[:, :V].cpu()
    for r in range(m):
        ids = placements[r]
        vals = got_logits[r, ids]
        assert bool((vals == vals[0]).all()), "duplicated rows must give bit-identical logits"
        assert float(vals[0]) == float(got_logits[r].max()), "the duplicated row was built to be the maximum"
    assert toks.cpu().tolist() == want
    assert torch.argmax(got_logits, dim=-1).tolist() == want            # torch.argmax on the kernel's own logits agrees
    # and the logits themselves are the bf16-rounded products (<= 1 ulp: the accumulation order differs from torch's)
    ulp = torch.pow(2.0, torch.floor(torch.log2(logits_ref.float().abs().clamp_min(2.0 ** -10))) - 7)
    assert bool(((got_logits - logits_ref.float()).abs() <= ulp).all())


def _bf16_ulp(t):
    return torch.pow(2.0, torch.floor(torch.log2(t.abs().clamp_min(1e-30))) - 7)


def _fill_pools(k, v, table, n_pages, n_kv, hd, dev, poison):
    """k, v: [ctx][n_kv][hd] bf16 -> K pages [page][kv][slot][d], V^T pages [page][kv][d][slot] (physical order)."""
    fill = float("nan") if poison else 0.0
    kpool = torch.full((n_pages, n_kv, PAGE, hd), fill, dtype=BF)
    vpool = torch.full((n_pages, n_kv, hd, PAGE), fill, dtype=BF)
    for p in range(k.shape[0]):
        page, slot = int(table[p // PAGE]), p % PAGE
        kpool[page, :, slot, :] = k[p]
        vpool[page, :, :, slot] = v[p]
    return kpool.to(dev), vpool.to(dev)


_GEOMS = [(1, 0), (1, 127), (7, 125), (16, 120), (16, 250), (7, 1140), (1, 1151), (16, 1136), (9, 700), (13, 1000)]   # 9 / 13 rows: the 8-wide merges
_ATTN_CASES = [(mode, nh, nkv, hd, m, kv) for mode in (0, 1, 2) for (nh, nkv, hd) in [(4, 4, 128), (8, 2, 128), (8, 2, 64)]
               for (m, kv) in _GEOMS]
_ATTN_CASES += [(mode, 32, 8, 128, m, kv) for mode in (0, 1, 2) for (m, kv) in [(7, 1140), (16, 120), (1, 127)]]   # llama3-8B heads


@pytest.mark.parametrize("mode,n_heads,n_kv,hd,m,kv_len", _ATTN_CASES)
def test_attention_matches_fp32_softmax(gpu_device, mode, n_heads, n_kv, hd, m, kv_len):
    """Rows at positions kv_len .. kv_len+m-1 (their own K/V already appended, as the engine does) against
    softmax_fp32(q k^T / sqrt(d) + causal) v in float64 on the SAME bf16 tensors.  Unwritten slots hold NaN."""
    lib, L = _lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(n_heads * 131 + hd + 17 * m + kv_len)
    ctx = kv_len + m
    n_pages = 10
    q = torch.randn(m, n_heads, hd, generator=g).to(BF)
    k = torch.randn(ctx, n_kv, hd, generator=g).to(BF)
    v = torch.randn(ctx, n_kv, hd, generator=g).to(BF)
    # a few sharply peaked rows too (large scores)
    q[0] = (q[0].float() * 3).to(BF)
    table = torch.randperm(n_pages, generator=g).to(torch.int32)
    kpool, vpool = _fill_pools(k, v, table, n_pages, n_kv, hd, dev, poison=True)
    group = n_heads // n_kv
    qd = q.double()
    kd = k.double().repeat_interleave(group, dim=1)
    vd = v.double().repeat_interleave(group, dim=1)
    scores = torch.einsum("mhd,chd->hmc", qd, kd) / math.sqrt(hd)
    keys = torch.arange(ctx)[None, None, :]
    rows = (kv_len + torch.arange(m))[None, :, None]
    scores = scores.masked_fill(keys > rows, float("-inf"))
    ref = torch.einsum("hmc,chd->mhd", torch.softmax(scores, dim=-1), vd).reshape(m, n_heads * hd)
    nb = ctypes.c_size_t(0)
    _tlib()[1].check(_tlib()[0].lsk_test_attention_scratch_bytes(n_heads, hd, n_pages, ctypes.byref(nb)))
    scratch = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
    out = torch.full((m, n_heads * hd), float("nan"), dtype=BF, device=dev)
    kvl = torch.tensor([kv_len], dtype=torch.int32, device=dev)
    qdev, td = q.reshape(m, n_heads * hd).to(dev), table.to(dev)
    _tlib()[1].check(_tlib()[0].lsk_test_attention(qdev.data_ptr(), m, n_heads, n_kv, hd, kpool.data_ptr(), vpool.data_ptr(), td.data_ptr(),
                                   n_pages, kvl.data_ptr(), kv_len, 0, scratch.data_ptr(), nb.value, out.data_ptr(), mode,
                                   _stream()))
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert bool(torch.isfinite(got).all()), "NaN in unwritten KV slots leaked into the output"
    # P is rounded to bf16 before P V (as HF's eager path and torch's flash kernels do), so the result carries that noise
    # on top of its own rounding.  Unit: one bf16 ulp of the element, never finer than the ulp of the head vector's RMS
    # (an element that cancels to ~0 inherits the noise of the terms it is made of; o_proj consumes the whole vector).
    ref_h = ref.view(m, n_heads, hd)
    rms = ref_h.pow(2).mean(-1, keepdim=True).sqrt().expand_as(ref_h).reshape(m, -1)
    ulp = _bf16_ulp(torch.maximum(ref.abs(), rms))
    err = (got - ref).abs() / ulp
    frac = float((err <= 1).double().mean())
    assert frac >= 0.99, f"only {frac:.4f} of the outputs within 1 bf16 ulp"
    assert float(err.max()) <= 2.0, f"max error {float(err.max()):.2f} ulp"
    # and no less accurate than torch's own bf16 SDPA on the same tensors (the kernel the reference runs on a CPU)
    mask = torch.zeros(m, ctx).masked_fill(keys[0] > rows[0], float("-inf")).to(BF)[None, None]
    sd = torch.nn.functional.scaled_dot_product_attention(
        q.transpose(0, 1)[None], k.repeat_interleave(group, dim=1).transpose(0, 1)[None],
        v.repeat_interleave(group, dim=1).transpose(0, 1)[None], attn_mask=mask, scale=1.0 / math.sqrt(hd))
    sd = sd[0].transpose(0, 1).reshape(m, -1).double()
    ulp_e = _bf16_ulp(torch.maximum(ref.abs(), rms / 8))
    e_mine, e_sdpa = (got - ref).abs() / ulp_e, (sd - ref).abs() / ulp_e
    assert float((e_mine <= 1).double().mean()) >= float((e_sdpa <= 1).double().mean()) - 0.02, \
        (float((e_mine <= 1).double().mean()), float((e_sdpa <= 1).double().mean()))
    assert float(e_mine.pow(2).mean().sqrt()) <= 1.25 * float(e_sdpa.pow(2).mean().sqrt()) + 0.05
