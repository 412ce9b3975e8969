"""Device-side sampling (lsk_sample.h): the two kernels against the draw-for-draw model in
oracle/sampling_oracle.py, and the whole `sample=True` path -- the strategies' default since round 2 -- against
(a) distribution fixtures recorded from the UNMODIFIED reference (oracle/make_sampling_dist_golden.py, 256 seeds per case)
and (b) the host-sampling path on the same GPU."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, build_case_model, load_golden

pytestmark = pytest.mark.gpu
DIST = json.load(open(os.path.join(GOLDEN_DIR, "sampling", "dist.json")))

CASES = json.load(open(os.path.join(GOLDEN_DIR, "sampling", "cases.json")))
SHAPE_OF_VOCAB = {512: "tiny-mha", 1000: "tiny-gqa", 768: "tiny-d64"}


def _engine(vocab, gpu_device):
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    shape = SHAPE_OF_VOCAB[vocab]
    cfg = synthetic.make_config(shape)
    model = synthetic.build_model(cfg, seed=0, exit_layer=synthetic.default_exit_layer(shape), late_damping=0.1).to(gpu_device)
    return model, HipEngine(model, max_ctx=512, max_prompt=64)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_sample_rows_matches_the_model_draw_for_draw(gpu_device, idx):
    from oracle import sampling_oracle as so
    rec = CASES[idx]
    model, eng = _engine(rec["vocab"], gpu_device)
    rows = np.stack([np.asarray(rec["draft_logits"], dtype=np.float32), np.asarray(rec["verify_logits"], dtype=np.float32)])
    logits = torch.tensor(rows, device=gpu_device)
    mismatches = total = 0
    for offset in range(40):
        toks, probs = eng.sample_rows(logits, rec["temperature"], rec["top_k"], rec["top_p"], seed=4242, offset=offset, tag0=5)
        toks, probs = toks.cpu().tolist(), probs.cpu().numpy()
        for r in range(2):
            want_tok, want_probs = so.device_sample_row(rows[r], rec["temperature"], rec["top_k"], rec["top_p"], 4242, offset, 5 + r)
            assert ((probs[r] > 0) == (want_probs > 0)).all()
            assert np.allclose(probs[r], want_probs, rtol=0, atol=2e-6)
            assert want_probs[toks[r]] > 0
            total += 1
            mismatches += int(toks[r] != want_tok)      # fast-math log/exp can flip a near-tie of two Gumbel scores
    assert mismatches <= max(1, total // 25), (mismatches, total)
    eng.close()


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.6, 0, 0.9), (1.0, 50, 0.95), (0.8, 0, 1.0), (0.7, 400, 0.5)])
def test_sample_rows_of_a_128k_vocabulary_match_the_model(gpu_device, temperature, top_k, top_p):
    """The llama3 vocabulary (128 256 logits per row) takes the multi-workgroup form of the sampler -- row maximum, a full-resolution
    mass histogram over the 65 536 keys, one scan for K / Z / P, per-workgroup Gumbel-max, pick -- against the same model as the
    one-workgroup form: the kept set, the probabilities and the draws; and the histogram must be clean again after every draw
    (the second and third draws of the loop would show a leftover)."""
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    from oracle import sampling_oracle as so
    cfg = synthetic.make_config("slice-1B")
    assert cfg.vocab_size > 32768
    model = synthetic.build_model(cfg, seed=0, exit_layer=2, late_damping=0.1, dtype=torch.bfloat16, device=gpu_device, gen_device=gpu_device)
    eng = HipEngine(model, max_ctx=256, max_prompt=32)
    g = torch.Generator().manual_seed(77)
    V = cfg.vocab_size
    rows = torch.stack([(torch.randn(V, generator=g) * 2.0), (torch.randn(V, generator=g) * 3.5 - 1.0), (torch.randn(V, generator=g) * 0.7)])
    rows = rows.to(torch.bfloat16).float()                   # logits are bf16-exact, as the lm_head produces them
    rows[1, 777] = rows[1].max() + 4.0                        # one row with a dominant token
    ld = (V + 3) // 4 * 4
    logits = torch.zeros(3, ld, dtype=torch.float32)
    logits[:, :V] = rows
    logits = logits.to(gpu_device)
    rows_np = rows.numpy()
    warped = [so.device_warp(rows_np[r], temperature, top_k, top_p) for r in range(3)]
    mismatches = total = 0
    for offset in range(6):
        toks, probs = eng.sample_rows(logits, temperature, top_k, top_p, seed=99, offset=offset, tag0=3)
        toks, probs = toks.cpu().tolist(), probs.cpu().numpy()
        for r in range(3):
            keep, want_probs = warped[r]
            assert ((probs[r, :V] > 0) == (want_probs > 0)).all(), (r, offset, int((probs[r, :V] > 0).sum()), int(keep.sum()))
            assert np.allclose(probs[r, :V], want_probs, rtol=0, atol=2e-6)
            x = rows_np[r]
            z = ((x - x.max()) * np.float32(1.0 / temperature)).astype(np.float32)
            u = so.device_uniforms(V, 3 + r, 99, offset)
            score = np.where(keep, z.astype(np.float64) - np.log(-np.log(u.astype(np.float64))), -np.inf)
            assert want_probs[toks[r]] > 0
            total += 1
            mismatches += int(toks[r] != int(np.argmax(score)))
    assert mismatches <= 1, (mismatches, total)
    eng.close()


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.7, 5, 1.0), (1.0, 0, 0.6), (0.9, 12, 0.8)])
def test_fp16_library_filters_at_fp16_resolution(gpu_device, temperature, top_k, top_p):
    """The -DLSK_ELEM_F16 library: logits are fp16 values, whose 10-bit mantissa the bf16-shaped key (upper half of the fp32 pattern)
    would fold 8 : 1 -- top-k / top-p thresholds would then act on coarser buckets than the reference's warpers.
    The key is the fp16 pattern itself: rows whose leading logits differ ONLY below bf16 resolution must be cut exactly where
    HF's warpers cut them (all values distinct, so tie handling does not enter)."""
    import transformers
    from layerskip_amd import synthetic
    from layerskip_amd.engine import HipEngine
    from oracle import sampling_oracle as so
    cfg = synthetic.make_config("tiny-mha")
    model = synthetic.build_model(cfg, seed=0, exit_layer=2, late_damping=0.1, dtype=torch.float16).to(gpu_device)
    eng = HipEngine(model, max_ctx=256, max_prompt=32)
    assert eng.dtype == torch.float16
    V = cfg.vocab_size
    g = torch.Generator().manual_seed(5)
    rows = (torch.randn(2, V, generator=g) * 1.5).to(torch.float16)
    # the 24 leading logits of row 0: consecutive fp16 values around 6.0 (spacing 2^-8 * 4 = 0.0039..): eight of them share every
    # bf16-resolution bucket
    base = torch.tensor(6.0, dtype=torch.float16).view(torch.int16).item()
    lead = torch.arange(base, base + 24, dtype=torch.int16).view(torch.float16)
    perm = torch.randperm(V, generator=g)[:24]
    rows[0, perm] = lead
    assert rows[0].unique().numel() >= V - 40
    rows_np = rows.float().numpy()
    logits = rows.float().to(gpu_device).contiguous()
    toks, probs = eng.sample_rows(logits, temperature, top_k, top_p, seed=7, offset=0, tag0=1)
    probs = probs.cpu().numpy()
    for r in range(2):
        keep, want = so.device_warp(rows_np[r], temperature, top_k, top_p, dtype="fp16")
        assert ((probs[r] > 0) == keep).all(), (r, int((probs[r] > 0).sum()), int(keep.sum()))
        assert np.allclose(probs[r], want, rtol=0, atol=2e-6)
        # and the kept set is HF's: the reference's own warpers on the same fp16-valued logits (llama_model_utils.py:75-107)
        x = torch.tensor(rows_np[r:r + 1]) / temperature
        if top_k > 0:
            x = transformers.TopKLogitsWarper(top_k=top_k, filter_value=-float("inf"), min_tokens_to_keep=1)(None, x)
        if 0 <= top_p <= 1.0:
            x = transformers.TopPLogitsWarper(top_p=top_p, filter_value=-float("inf"), min_tokens_to_keep=1)(None, x)
        hf_keep = torch.isfinite(x[0]).numpy()
        diff = int((hf_keep != keep).sum())
        assert diff <= 1, f"row {r}: kept set differs from HF's warpers in {diff} tokens"      # (<= 1: fp32 vs fp64 mass at the top-p boundary)
        if top_k > 0 and top_p >= 1.0:
            assert int(keep.sum()) == top_k                    # distinct values: exactly k survive -- a bf16-shaped key would keep up to 8x
    eng.close()


def test_accept_sampled_kernel_matches_the_model(gpu_device):
    import lsk_test_lib
    from oracle import sampling_oracle as so
    lib = lsk_test_lib.load()
    rng = np.random.default_rng(3)
    v, ld = 640, 640
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for trial in range(60):
        td = int(rng.integers(0, 7))
        pd = [so.probabilities(rng.normal(size=v).astype(np.float32) * 2) for _ in range(max(td, 1))]
        pv = [so.probabilities(rng.normal(size=v).astype(np.float32) * 2) for _ in range(td + 1)]
        drafts = [int(rng.choice(v, p=p.astype(np.float64) / p.astype(np.float64).sum())) for p in pd[:td]]
        verified = [int(np.argmax(p)) for p in pv]
        eos = [drafts[1]] if (td >= 3 and trial % 5 == 0) else []
        d_dev = torch.tensor([-1] + drafts + [0] * (17 - td), dtype=torch.int32, device=gpu_device)
        v_dev = torch.tensor(verified + [0] * (17 - len(verified)), dtype=torch.int32, device=gpu_device)
        e_dev = torch.tensor(eos + [0] * (8 - len(eos)), dtype=torch.int32, device=gpu_device)
        pd_dev = torch.tensor(np.stack(pd), device=gpu_device)
        pv_dev = torch.tensor(np.stack(pv), device=gpu_device)
        res = torch.zeros(64, dtype=torch.int32, device=gpu_device)
        lsk_test_lib.check(lib.lsk_test_accept_sampled(d_dev.data_ptr() + 4, v_dev.data_ptr(), td, e_dev.data_ptr(), len(eos), pd_dev.data_ptr(),
                                               pv_dev.data_ptr(), ld, v, 77, trial, res.data_ptr(), st))
        torch.cuda.synchronize()
        r = res.cpu().tolist()
        n, ntd, tok = so.device_accept(drafts, verified, pd, pv, eos, seed=77, offset=trial)
        assert (r[0], r[1]) == (n, ntd), (trial, r[:4], n, ntd)
        if n == ntd:
            assert r[2] == tok
        else:
            assert pv[n][r[2]] > pd[n][r[2]]
        assert r[4:4 + n] == drafts[:n] and r[4 + n] == r[2]


def _tv(a, b):
    na, nb = sum(a.values()), sum(b.values())
    return 0.5 * sum(abs(a.get(k, 0) / na - b.get(k, 0) / nb) for k in set(a) | set(b))


@pytest.mark.parametrize("idx", range(len(DIST)))
def test_device_sampling_matches_the_unmodified_reference_in_distribution(gpu_device, idx):
    """The HIP `sample=True` path (fused lsk_spec_generate_sampled AND one lsk_spec_step_sampled per step) over 256 seeds
    against what the reference produced over 256 seeds on the same weights (bf16): acceptance rate, output length,
    first- and second-token histograms, and no token outside the reference's own nucleus."""
    from conftest import build_struct_model, load_struct
    from layerskip_amd import GenerationConfig
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    case = DIST[idx]
    rec = load_golden(case["fixture"]) if case["family"] == "legacy" else load_struct(case["fixture"])
    model = (build_case_model(rec) if case["family"] == "legacy" else build_struct_model(rec)).to(gpu_device)
    cfg = GenerationConfig(max_steps=case["max_steps"], exit_layer=case["exit_layer"], num_speculations=case["num_speculations"],
                           sample=True, temperature=case["temperature"], top_k=case["top_k"], top_p=case["top_p"])
    eos = [model.config.vocab_size]
    n_runs = case["n_runs"]
    ref_first = [{int(k): v for k, v in h.items()} for h in case["first_token_hist"]]
    ref_second = [{int(k): v for k, v in h.items()} for h in case["second_token_hist"]]
    merged = lambda hs: {k: hs[0].get(k, 0) + hs[1].get(k, 0) for k in set(hs[0]) | set(hs[1])}   # noqa: E731
    nucleus = set(case["first_nucleus"])
    # tokens whose reference probability is within a few bf16 ulp of the nucleus boundary may fall on either side
    pmin = min(case["first_nucleus_probs"])
    for fused in (True, False):
        strat = HipSelfSpeculativeGenerationStrategy(fused_generate=fused)
        assert strat.device_sampling
        acc, lens, first, second = [], [], {}, {}
        for i in range(n_runs):
            torch.manual_seed(10_000 + i)
            r = strat.generate_token_ids(model, list(rec["prompt"]), eos, cfg)
            assert 0 < len(r.predicted_tokens) <= case["max_steps"]
            acc.append(r.acceptance_rate)
            lens.append(len(r.predicted_tokens))
            first[r.predicted_tokens[0]] = first.get(r.predicted_tokens[0], 0) + 1
            if len(r.predicted_tokens) > 1:
                second[r.predicted_tokens[1]] = second.get(r.predicted_tokens[1], 0) + 1
        mean = float(np.mean(acc))
        se = (case["acceptance_std"] ** 2 / n_runs + float(np.var(acc)) / n_runs) ** 0.5
        assert abs(mean - case["acceptance_mean"]) < 4 * se + 0.01, (fused, mean, case["acceptance_mean"], se)
        assert abs(float(np.mean(lens)) - case["length_mean"]) < 0.5
        outside = {t: c for t, c in first.items() if t not in nucleus}
        assert sum(outside.values()) <= 0.02 * n_runs, (fused, "first tokens outside the reference's nucleus", outside, pmin)
        # the reference's two half-samples differ from each other by pure sampling noise: that is the yardstick
        for mine, halves, what in ((first, ref_first, "first"), (second, ref_second, "second")):
            noise = _tv(halves[0], halves[1])
            tv = _tv(mine, merged(halves))
            assert tv < max(0.12, 1.25 * noise), (fused, what, tv, noise)
        # reproducibility contract: the same torch seed gives the same generation
        torch.manual_seed(10_000)
        a = strat.generate_token_ids(model, list(rec["prompt"]), eos, cfg).predicted_tokens
        torch.manual_seed(10_000)
        b = strat.generate_token_ids(model, list(rec["prompt"]), eos, cfg).predicted_tokens
        c = HipSelfSpeculativeGenerationStrategy(fused_generate=fused)
        torch.manual_seed(10_000)
        assert a == b == c.generate_token_ids(model, list(rec["prompt"]), eos, cfg).predicted_tokens


def test_device_sampled_generation_matches_the_host_path_in_distribution(gpu_device):
    from layerskip_amd import GenerationConfig
    from layerskip_amd.hip_strategies import HipSelfSpeculativeGenerationStrategy
    rec = load_golden("tiny_mha_s1")
    model = build_case_model(rec).to(gpu_device)
    kw = dict(max_steps=10, exit_layer=rec["exit_layer"], num_speculations=4, sample=True, temperature=0.12, top_k=0, top_p=0.9)
    host, dev = HipSelfSpeculativeGenerationStrategy(device_sampling=False), HipSelfSpeculativeGenerationStrategy(device_sampling=True)
    acc = {"host": [], "dev": []}
    first = {"host": {}, "dev": {}}
    n_runs = 120
    for i in range(n_runs):
        for name, strat in (("host", host), ("dev", dev)):
            torch.manual_seed(100 + i)
            r = strat.generate_token_ids(model, list(rec["prompt"]), list(rec["eos_token_ids"]), GenerationConfig(**kw))
            assert 0 < len(r.predicted_tokens) <= 10
            acc[name].append(r.acceptance_rate)
            first[name][r.predicted_tokens[0]] = first[name].get(r.predicted_tokens[0], 0) + 1
    ma, mb = np.mean(acc["host"]), np.mean(acc["dev"])
    sd = np.std(acc["host"])
    assert abs(ma - mb) < 4 * sd * (2 / n_runs) ** 0.5 + 0.01, (ma, mb, sd)
    keys = set(first["host"]) | set(first["dev"])
    tv = 0.5 * sum(abs(first["host"].get(k, 0) - first["dev"].get(k, 0)) / n_runs for k in keys)
    assert tv < 0.25, tv
