"""Single-kernel parity on the GPU: every HIP kernel against a plain fp32 torch statement of the op."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from layerskip_amd import _lib
    return _lib.load(), _lib


def _tlib():
    """liblayerskip_hip_test.so (include/layerskip_hip_test.h): the engine's kernels on caller-owned buffers."""
    import lsk_test_lib
    return lsk_test_lib.load(), lsk_test_lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack(w, rope_hd=0):
    lib, L = _lib()
    n, k = w.shape
    nbytes = ctypes.c_size_t(0)
    L.check(lib.lsk_packed_bytes(n, k, ctypes.byref(nbytes)))
    dst = torch.zeros(nbytes.value, dtype=torch.uint8, device=w.device)
    L.check(lib.lsk_pack_linear(w.data_ptr(), n, k, w.stride(0), dst.data_ptr(), 0, 1, rope_hd, _stream()))
    return dst


def _gemm(x, wp, n, norm_w=None, eps=1e-5, target_wgs=0):
    lib, L = _lib()
    m, k = x.shape
    y = torch.full((m, n), float("nan"), dtype=torch.float32, device=x.device)
    _tlib()[1].check(_tlib()[0].lsk_test_gemm(x.data_ptr(), m, k, wp.data_ptr(), n, None if norm_w is None else norm_w.data_ptr(),
                              eps, y.data_ptr(), target_wgs, _stream()))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("m,k,n", [(1, 256, 64), (7, 4096, 512), (16, 4096, 1040), (3, 704, 256), (13, 11008, 320),
                                   (5, 8192, 48), (16, 1408, 1000), (9, 5120, 272), (10, 13824, 64), (11, 4096, 128)])
def test_skinny_gemm_matches_fp32(gpu_device, m, k, n):
    """A = asymmetric random (transpose / row-swap detecting); fp32 accumulate => tight tolerance."""
    g = torch.Generator().manual_seed(m * 1000 + k + n)
    x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(gpu_device)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(gpu_device)
    y = _gemm(x, _pack(w), n)
    ref = x.float() @ w.float().t()
    assert torch.isfinite(y).all()
    err = (y - ref).abs().max().item()
    assert err <= 1e-3, f"max abs err {err}"


def test_skinny_gemm_identity_weight(gpu_device):
    """W = I picks x back out exactly: catches any fragment-layout permutation."""
    k = n = 256
    x = torch.arange(4 * k, dtype=torch.float32).reshape(4, k).div(64.0).to(torch.bfloat16).to(gpu_device)
    w = torch.eye(n, k, dtype=torch.bfloat16, device=gpu_device)
    y = _gemm(x, _pack(w), n)
    assert torch.equal(y, x.float())


def test_skinny_gemm_rmsnorm_prologue(gpu_device):
    g = torch.Generator().manual_seed(7)
    m, k, n = 6, 4096, 256
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(gpu_device)
    w = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16).to(gpu_device)
    nw = (1 + 0.1 * torch.randn(k, generator=g)).to(torch.bfloat16).to(gpu_device)
    eps = 1e-5
    y = _gemm(x, _pack(w), n, norm_w=nw, eps=eps)
    x32 = x.float()
    xn = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16)
    xn = (nw * xn)                      # bf16 * bf16 -> bf16, as LlamaRMSNorm does
    ref = xn.float() @ w.float().t()
    # the normalised activations may differ by one bf16 ulp where fp32 statistics round differently
    err = (y - ref).abs().max().item()
    assert err <= 2e-2, err
    frac_exact = ((y - ref).abs() <= 1e-3).float().mean().item()
    assert frac_exact > 0.9, frac_exact


def test_skinny_gemm_row_invariance(gpu_device):
    """Row r of an M-row pass is bit-identical to the same row in a 1-row pass (and any grid size)."""
    g = torch.Generator().manual_seed(11)
    k, n = 11008, 512
    x = torch.randn(9, k, generator=g).to(torch.bfloat16).to(gpu_device)
    w = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16).to(gpu_device)
    wp = _pack(w)
    y9 = _gemm(x, wp, n)
    for r in (0, 4, 8):
        y1 = _gemm(x[r:r + 1].contiguous(), wp, n)
        assert torch.equal(y1[0], y9[r])
    y9b = _gemm(x, wp, n, target_wgs=7)
    assert torch.equal(y9, y9b)
    # every row template (1 | 8 | 10 | 13 | 16 rows, csrc/lsk_launch.h) gives the same bits for the same row
    x16 = torch.randn(16, k, generator=g).to(torch.bfloat16).to(gpu_device)
    y16 = _gemm(x16, wp, n)
    for m in (2, 8, 9, 10, 11, 13, 14):
        ym = _gemm(x16[:m].contiguous(), wp, n)
        assert torch.equal(ym, y16[:m]), m


@pytest.mark.parametrize("k", [2048, 4096, 5120, 8192, 1408])
def test_skinny_gemm_row_invariance_with_the_rmsnorm_prologue(gpu_device, k):
    """The RMSNorm-fused launches at every wave count / chunking `lsk_gemm_waves` can pick (csrc/lsk_gemm.h: four waves and 2048-feature
    chunks at K <= 2048 and K > 4096 -- one, three and four chunks here, one of them ragged --, eight waves and ONE 4096-feature chunk at
    K = 4096, where 2..8 rows take the split prologue and 1 / 9+ rows do not): the wave count is a function of the projection and K, never
    of the row count, so row r of every row template has the bits of the same row passed alone, whatever the grid."""
    g = torch.Generator().manual_seed(31 + k)
    n = 272
    x16 = torch.randn(16, k, generator=g).to(torch.bfloat16).to(gpu_device)
    w = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16).to(gpu_device)
    gain = (1 + 0.1 * torch.randn(k, generator=g)).to(torch.bfloat16).to(gpu_device)
    wp = _pack(w)
    y16 = _gemm(x16, wp, n, norm_w=gain)
    assert torch.isfinite(y16).all()
    for r in (0, 7, 15):
        assert torch.equal(_gemm(x16[r:r + 1].contiguous(), wp, n, norm_w=gain)[0], y16[r]), r
    for m in (2, 7, 8, 9, 10, 13):
        assert torch.equal(_gemm(x16[:m].contiguous(), wp, n, norm_w=gain), y16[:m]), m
    assert torch.equal(_gemm(x16[:7].contiguous(), wp, n, norm_w=gain, target_wgs=5), y16[:7])


@pytest.mark.parametrize("drafts,verified,eos,expect", [
    ([5, 6, 7, 8], [5, 6, 9, 8, 1], [], (2, 4)),
    ([5, 6, 7, 8], [5, 6, 7, 8, 1], [], (4, 4)),
    ([5, 6, 7, 8], [4, 6, 7, 8, 1], [], (0, 4)),
    ([], [3], [], (0, 0)),
    ([5, 2, 7, 8], [5, 2, 7, 8, 1], [2], (2, 2)),     # drafted EOS ends the draft (SSG:146-148)
    ([5, 2, 7, 8], [9, 2, 7, 8, 1], [2, 11], (0, 2)),
    (list(range(15)), list(range(15)) + [99], [], (15, 15)),
    # the reference folds any number of stop_token_ids into the eos list (generator_base.py:106): 12 ids, the hit in the last place;
    # 70 and 1024 ids (more than one 64-id chunk of the ballot scan), the hit in the second / last chunk
    ([5, 6, 7, 8], [5, 6, 7, 8, 1], [100 + i for i in range(11)] + [7], (3, 3)),
    ([5, 6, 7, 8], [5, 9, 7, 8, 1], [100 + i for i in range(11)] + [7], (1, 3)),
    ([5, 6, 7, 8], [5, 6, 7, 8, 1], [100 + i for i in range(69)] + [8], (4, 4)),
    ([5, 6, 7, 8, 9, 10], [5, 6, 7, 8, 9, 10, 1], [100 + i for i in range(1023)] + [6], (2, 2)),
    ([5, 6, 7, 8], [5, 6, 7, 8, 1], [100 + i for i in range(1024)], (4, 4)),
])
def test_accept_kernel(gpu_device, drafts, verified, eos, expect):
    lib, L = _lib()
    d = torch.tensor(drafts + [0], dtype=torch.int32, device=gpu_device)
    v = torch.tensor(verified, dtype=torch.int32, device=gpu_device)
    e = torch.tensor(eos + [0], dtype=torch.int32, device=gpu_device)
    res = torch.full((64,), -1, dtype=torch.int32, device=gpu_device)
    _tlib()[1].check(_tlib()[0].lsk_test_accept(d.data_ptr(), v.data_ptr(), len(drafts), e.data_ptr(), len(eos), res.data_ptr(), _stream()))
    torch.cuda.synchronize()
    r = res.tolist()
    n, td = expect
    assert (r[0], r[1]) == (n, td)
    assert r[2] == verified[n]
    assert r[4:4 + n + 1] == drafts[:n] + [verified[n]]
    # the reference expression, SSG:186-190
    dt = torch.tensor([drafts[:td]])
    vt = torch.tensor([verified[:td + 1]])
    ref_n = int(((~(dt == vt[:, :-1])).cumsum(dim=-1) < 1).sum().item())
    assert ref_n == n


def test_prefill_kernels_match_decode_kernels(gpu_device):
    """The MFMA-tiled prefill path (lsk_gemm_big + bulk RMSNorm) against 16-row passes of the decode
    kernels on the same rows: same rounding points, different accumulation order -> equal up to rare
    one-ulp bf16 flips in the hidden states."""
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_BULK, HipEngine
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=2, exit_layer=3, late_damping=0.1).to(gpu_device)
    eng = HipEngine(model, max_ctx=512, max_prompt=300)
    ids = synthetic.make_prompt(cfg.vocab_size, 203, 77)
    outs = []
    for threshold in (1 << 30, 1):
        eng.set_option(_lib.LSK_OPT_BIG_THRESHOLD, threshold)
        eng.reset()
        eng.embed_rows(ids, BUF_BULK, 0)
        eng.run_bulk(len(ids), 0, eng.num_layers)
        outs.append(eng.read_rows(BUF_BULK, 0, len(ids)).float())
    torch.cuda.synchronize()
    a, b = outs
    assert torch.isfinite(b).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 0.03 * scale
    assert (a - b).abs().mean().item() <= 0.004 * scale
    eng.close()


@pytest.mark.parametrize("shape,lengths", [("slice-7B", (255, 767, 1023, 1279, 2047)), ("slice-70B", (383, 1023, 1600)), ("slice-1B", (511, 1023, 2047))])
def test_every_prefill_tile_shape_matches_the_decode_kernels(gpu_device, shape, lengths):
    """The prefill launches pick their tile shape from the workgroup counts a prompt length gives (csrc/lsk_engine.hip: launch_big_qkv,
    launch_big_resid; csrc/lsk_launch.h: two or three row tiles in the attention kernel): tiny models only ever reach the smallest shapes.
    Real widths at prompt lengths on both sides of every threshold -- 64 x 192 / 128 x 192 / 128 x 256 / 128 x 384 q/k/v tiles (the
    ragged last panel of llama2-70B's 640 tiles included), the 32- / 64- / 128-row K-split and the 128 x 256 pinned residual tiles, both
    attention forms, d = 64 -- against 16-row passes of the decode kernels on the same rows: same rounding points, another summation
    order."""
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_BULK, HipEngine
    cfg = synthetic.make_config(shape, num_hidden_layers=2)
    model = synthetic.build_model(cfg, seed=3, exit_layer=1, late_damping=0.1, dtype=torch.bfloat16, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=max(lengths) + 64, max_prompt=max(lengths) + 1)
    for n in lengths:
        ids = synthetic.make_prompt(cfg.vocab_size, n, 100 + n)
        outs = []
        for threshold in (1 << 30, 1):
            eng.set_option(_lib.LSK_OPT_BIG_THRESHOLD, threshold)
            eng.reset()
            eng.embed_rows(ids, BUF_BULK, 0)
            eng.run_bulk(n, 0, eng.num_layers)
            outs.append(eng.read_rows(BUF_BULK, 0, n).float())
        torch.cuda.synchronize()
        a, b = outs
        assert torch.isfinite(b).all(), n
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 0.03 * scale, n
        assert (a - b).abs().mean().item() <= 0.004 * scale, n
    eng.close()
    del model


def test_fused_attention_combine_equals_two_kernel_form(gpu_device):
    """In-launch last-arriver combine (write-through partials + ticket) vs the separate combine kernel:
    bit-identical hidden states, for decode rows and a verify block, under repeated launches."""
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_BULK, BUF_STEP, HipEngine
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=4, exit_layer=3, late_damping=0.1).to(gpu_device)
    eng = HipEngine(model, max_ctx=1024, max_prompt=600)
    ids = synthetic.make_prompt(cfg.vocab_size, 530, 5)
    outs = []
    for fused in (0, 1):
        eng.set_option(_lib.LSK_OPT_FUSED_ATTN, fused)
        eng.reset()
        eng.embed_rows(ids[:-9], BUF_BULK, 0)
        eng.run_bulk(len(ids) - 9, 0, eng.num_layers)
        eng.embed_rows(ids[-9:], BUF_STEP, 0)
        for _ in range(3):          # repeated launches exercise the self-resetting tickets
            eng.embed_rows(ids[-9:], BUF_STEP, 0)
            eng.run_layers(BUF_STEP, 0, 9, len(ids) - 9, 0, eng.num_layers)
        outs.append((eng.read_rows(BUF_BULK, 0, len(ids) - 9).clone(), eng.read_rows(BUF_STEP, 0, 9).clone()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    eng.close()


def test_paged_kv_block_table_indirection(gpu_device):
    """The KV pool is paged: a permuted logical->physical block table must give bit-identical results
    (prefill kernels, decode kernels and attention all go through the table)."""
    from layerskip_amd import synthetic
    from layerskip_amd.engine import BUF_BULK, BUF_STEP, HipEngine
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=9, exit_layer=3, late_damping=0.1).to(gpu_device)
    eng = HipEngine(model, max_ctx=1024, max_prompt=400)
    n_pages = eng.max_ctx // eng.page_size
    ids = synthetic.make_prompt(cfg.vocab_size, 300, 12)
    outs = []
    for table in (list(range(n_pages)), [(5 * i + 3) % n_pages for i in range(n_pages)]):
        assert sorted(table) == list(range(n_pages))
        eng.set_block_table(table)
        eng.reset()
        eng.embed_rows(ids[:290], BUF_BULK, 0)
        eng.run_bulk(290, 0, eng.num_layers)
        eng.set_kv_len(290)
        eng.embed_rows(ids[290:297], BUF_STEP, 0)
        eng.run_layers(BUF_STEP, 0, 7, 0, 0, eng.num_layers)
        toks = eng.run_head(BUF_STEP, 0, 7)
        outs.append((eng.read_rows(BUF_STEP, 0, 7).clone(), toks))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    eng.close()


@pytest.mark.parametrize("graph_steps", [0, 1])
def test_generation_is_identical_under_a_permuted_block_table(gpu_device, graph_steps):
    """The attention launch skips the table read while the table is the identity (a flag in its arguments, also inside captured
    steps): switching to a permuted table and back must take the other path each time and give the same ids."""
    from layerskip_amd import GenerationConfig, _lib, synthetic
    from layerskip_amd.engine import get_engine
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    cfg = synthetic.make_config("tiny-gqa")
    model = synthetic.build_model(cfg, seed=21, exit_layer=3, late_damping=0.1).to(gpu_device)
    eng = get_engine(model, max_ctx=1024, max_prompt=400)
    eng.set_option(_lib.LSK_OPT_GRAPH_STEPS, graph_steps)
    n_pages = eng.max_ctx // eng.page_size
    prompt = synthetic.make_prompt(cfg.vocab_size, 300, 4)
    gen = GenerationConfig(max_steps=96, exit_layer=3, num_speculations=4, sample=False)
    strat = HipSelfSpeculativeGenerationStrategy()
    outs = []
    for table in (list(range(n_pages)), [(3 * i + 1) % n_pages for i in range(n_pages)], list(range(n_pages))):
        eng.set_block_table(table)
        outs.append(strat.generate_token_ids(model, prompt, [cfg.vocab_size], gen).predicted_tokens)
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == 96


def test_engine_grows_context_on_demand(gpu_device):
    from layerskip_amd import GenerationConfig, synthetic
    from layerskip_amd.engine import get_engine
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    cfg = synthetic.make_config("tiny-mha")
    model = synthetic.build_model(cfg, seed=3, exit_layer=2, late_damping=0.1).to(gpu_device)
    eng = get_engine(model, max_ctx=128, max_prompt=16)
    strat = HipSelfSpeculativeGenerationStrategy()
    prompt = synthetic.make_prompt(cfg.vocab_size, 200, 1)            # longer than both initial limits
    gen = GenerationConfig(max_steps=40, exit_layer=2, num_speculations=4, sample=False)
    a = strat.generate_token_ids(model, prompt, [cfg.vocab_size], gen)
    assert eng.max_ctx >= 200 + 40 and eng.max_prompt >= 200
    b = strat.generate_token_ids(model, prompt, [cfg.vocab_size], gen)
    assert a.predicted_tokens == b.predicted_tokens and len(a.predicted_tokens) == 40


@pytest.mark.parametrize("shape,n", [("tiny-gqa", 301), ("tiny-d64", 150), ("tiny-mha", 64)])
def test_flash_prefill_attention_matches_decode_attention(gpu_device, shape, n):
    """lsk_attn_prefill_kernel (one launch per layer, online softmax over all visible pages) vs rows/16
    launches of the split-KV decode kernel inside the same MFMA prefill path: same rounding points, different
    summation order -> hidden states equal up to bf16-ulp noise; logits rows nearly identical."""
    from layerskip_amd import _lib, synthetic
    from layerskip_amd.engine import BUF_BULK, HipEngine
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=5, exit_layer=2, late_damping=0.1).to(gpu_device)
    eng = HipEngine(model, max_ctx=512, max_prompt=320)
    eng.set_option(_lib.LSK_OPT_BIG_THRESHOLD, 1)
    ids = synthetic.make_prompt(cfg.vocab_size, n, 21)
    outs = []
    for flash in (0, 1):
        eng.set_option(_lib.LSK_OPT_FLASH_PREFILL, flash)
        eng.reset()
        eng.embed_rows(ids, BUF_BULK, 0)
        eng.run_bulk(n, 0, eng.num_layers)
        outs.append(eng.read_rows(BUF_BULK, 0, n).float())
    torch.cuda.synchronize()
    a, b = outs
    assert torch.isfinite(b).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 0.03 * scale
    assert (a - b).abs().mean().item() <= 0.003 * scale
    eng.close()
