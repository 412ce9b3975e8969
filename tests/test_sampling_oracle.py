"""oracle/sampling_oracle.py against vectors produced by the reference's own sampling functions
(oracle/make_sampling_golden.py -> tests/golden/sampling/cases.json), plus the properties speculative sampling
rests on.  This is the checker for a device-side sampling path; CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import sampling_oracle as so

CASES = json.load(open(os.path.join(GOLDEN_DIR, "sampling", "cases.json")))


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_warp_and_probabilities_match_reference(idx):
    rec = CASES[idx]
    for name in ("draft", "verify"):
        logits = np.asarray(rec[name + "_logits"], dtype=np.float32)
        warped = so.warp_logits(logits, rec["temperature"], rec["top_k"], rec["top_p"])
        kept = set(np.flatnonzero(np.isfinite(warped)).tolist())
        gold_kept = set(rec[name + "_kept"])
        probs = so.probabilities(warped)
        gold = np.asarray(rec[name + "_probs"], dtype=np.float32)
        # bf16-rounded logits have exact ties; which member of a tie group straddling the nucleus boundary survives
        # is the sort's tie order (torch.sort is not stable, a device kernel has its own): sets may differ only
        # inside ONE group of equal logits, and the kept VALUES must be the same multiset
        assert len(kept) == len(gold_kept)
        odd = kept ^ gold_kept
        assert len({float(logits[i]) for i in odd}) <= 1, (idx, name, sorted(odd))
        assert sorted(float(logits[i]) for i in kept) == sorted(float(logits[i]) for i in gold_kept)
        assert np.allclose(np.sort(probs), np.sort(gold), rtol=0, atol=2e-7)
        same = np.asarray([i not in odd for i in range(len(logits))])
        assert np.allclose(probs[same], gold[same], rtol=0, atol=2e-7), float(np.abs(probs - gold)[same].max())
        assert abs(float(probs.sum()) - 1.0) < 1e-5


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_residual_matches_reference_and_preserves_the_target_distribution(idx):
    rec = CASES[idx]
    pd = np.asarray(rec["draft_probs"], dtype=np.float32)
    pv = np.asarray(rec["verify_probs"], dtype=np.float32)
    res = so.residual(pv, pd)
    assert np.allclose(res, np.asarray(rec["residual"], dtype=np.float32), rtol=0, atol=2e-7)
    # draw from the draft, keep with min(1, q/p), else redraw from the residual: the emitted token follows q
    emitted = so.emitted_distribution(pd, pv)
    assert np.abs(emitted - pv.astype(np.float64)).max() < 5e-6
    assert abs(emitted.sum() - 1.0) < 5e-6


def test_accept_step_follows_the_reference_loop():
    rng = np.random.default_rng(0)
    v = 16
    pd = [so.probabilities(rng.normal(size=v).astype(np.float32)) for _ in range(3)]
    pv = [so.probabilities(rng.normal(size=v).astype(np.float32)) for _ in range(4)]
    drafts = [int(np.argmax(p)) for p in pd]
    # every uniform below the ratio: all drafts kept, the bonus token comes from the last verify row
    n, tok = so.accept_step(drafts, pd, pv, [0.0, 0.0, 0.0], 0.5, bonus_token=7)
    assert (n, tok) == (3, 7)
    n, tok = so.accept_step(drafts, pd, pv, [0.0, 0.0, 0.0], 0.0)
    assert n == 3 and tok == int(np.flatnonzero(pv[3] > 0)[0])
    # a uniform of ~1 rejects unless q >= p; the replacement comes from max_fn(q - p) of THAT row and never is a
    # token whose verify probability does not exceed its draft probability
    ratios = [min(1.0, float(pv[i][t]) / float(pd[i][t])) for i, t in enumerate(drafts)]
    first_reject = next((i for i, r in enumerate(ratios) if r < 1.0), None)
    n, tok = so.accept_step(drafts, pd, pv, [0.999999] * 3, 0.37)
    if first_reject is None:
        assert n == 3
    else:
        assert n == first_reject
        assert pv[first_reject][tok] > pd[first_reject][tok]
    # identical distributions: everything is accepted whatever the uniforms (ratio == 1)
    n, _ = so.accept_step(drafts, pd, pd + [pv[3]], [0.99, 0.5, 0.01], 0.2)
    assert n == 3


def test_inverse_cdf_is_a_categorical_draw():
    p = np.asarray([0.0, 0.25, 0.0, 0.5, 0.25], dtype=np.float32)
    assert so.inverse_cdf(p, 0.0) == 1
    assert so.inverse_cdf(p, 0.24) == 1
    assert so.inverse_cdf(p, 0.26) == 3
    assert so.inverse_cdf(p, 0.76) == 4
    assert so.inverse_cdf(p, 0.999999) == 4
    us = (np.arange(20000) + 0.5) / 20000
    counts = np.bincount([so.inverse_cdf(p, u) for u in us], minlength=5) / 20000
    assert np.abs(counts - p).max() < 1e-3


# ---- the model of the device algorithm (lsk_sample.h) against the reference-pinned functions -------------------
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        got = so.philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1])
        assert [int(np.asarray(w).reshape(-1)[0]) for w in got] == want
    u = so.device_uniforms(100000, 3, seed=1234, offset=7)
    assert 0.0 < u.min() and u.max() < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.005 and abs(float(u.var()) - 1.0 / 12.0) < 0.002
    assert not np.array_equal(u[:16], so.device_uniforms(16, 4, seed=1234, offset=7))       # the tag separates streams
    assert not np.array_equal(u[:16], so.device_uniforms(16, 3, seed=1234, offset=8))       # so does the offset


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_device_thresholds_keep_what_the_reference_keeps(idx):
    """Key-space thresholds (device) vs sort + cumulative sum (HF): same kept set except that a group of EQUAL logits
    straddling the nucleus boundary is kept whole on the device and split by sort order in HF."""
    rec = CASES[idx]
    for name in ("draft", "verify"):
        logits = np.asarray(rec[name + "_logits"], dtype=np.float32)
        keep, probs = so.device_warp(logits, rec["temperature"], rec["top_k"], rec["top_p"])
        kept, gold_kept = set(np.flatnonzero(keep).tolist()), set(rec[name + "_kept"])
        assert gold_kept <= kept, sorted(gold_kept - kept)
        extra = kept - gold_kept
        assert len({float(logits[i]) for i in extra}) <= 1
        if extra:       # the extra members tie with the smallest logit HF kept
            assert float(logits[next(iter(extra))]) == min(float(logits[i]) for i in gold_kept)
        gold = np.asarray(rec[name + "_probs"], dtype=np.float64)
        if not extra:
            assert np.allclose(probs, gold, rtol=0, atol=3e-7)
        else:           # same shape, renormalised over the slightly larger set
            scale = gold[list(gold_kept)].sum() / probs[list(gold_kept)].astype(np.float64).sum()
            assert np.allclose(probs[list(gold_kept)] * scale, gold[list(gold_kept)], rtol=0, atol=3e-7)


def test_gumbel_max_draw_follows_the_warped_distribution():
    rec = CASES[3]                                       # sharp case: a nucleus of a few tokens
    logits = np.asarray(rec["verify_logits"], dtype=np.float32)
    keep, probs = so.device_warp(logits, rec["temperature"], rec["top_k"], rec["top_p"])
    n = 4000
    counts = np.zeros_like(probs, dtype=np.float64)
    for off in range(n):
        tok, _ = so.device_sample_row(logits, rec["temperature"], rec["top_k"], rec["top_p"], seed=99, offset=off, tag=0)
        assert keep[tok]
        counts[tok] += 1
    tv = 0.5 * np.abs(counts / n - probs).sum()
    assert tv < 0.03, tv


def test_device_accept_is_the_reference_loop():
    rng = np.random.default_rng(5)
    v = 64
    for trial in range(200):
        td = int(rng.integers(1, 6))
        pd = [so.probabilities(rng.normal(size=v).astype(np.float32) * 2) for _ in range(td)]
        pv = [so.probabilities(rng.normal(size=v).astype(np.float32) * 2) for _ in range(td + 1)]
        drafts = [int(rng.choice(v, p=p.astype(np.float64) / p.astype(np.float64).sum())) for p in pd]
        verified = [int(np.argmax(p)) for p in pv]
        n, ntd, tok = so.device_accept(drafts, verified, pd, pv, eos=[], seed=11, offset=trial)
        u = so.u01(so.philox4x32_10(np.arange(td), so.TAG_ACCEPT, trial, 0, 11, 0)[0])
        n_ref, _ = so.accept_step(drafts, pd, pv, [float(x) for x in u], 0.5, bonus_token=verified[td])
        assert (n, ntd) == (n_ref, td)
        if n == td:
            assert tok == verified[td]
        else:
            assert pv[n][tok] > pd[n][tok]
    # a drafted EOS ends the draft: later drafts are neither tested nor counted
    pd = [so.probabilities(rng.normal(size=v).astype(np.float32)) for _ in range(3)]
    n, ntd, _ = so.device_accept([5, 9, 7], [5, 9, 7, 1], pd, pd + [pd[0]], eos=[9], seed=1, offset=0)
    assert ntd == 2 and n == 2


@pytest.mark.parametrize("vocab,top_k", [(40000, 0), (128256, 0), (50000, 64), (2000, 0)])
def test_the_histogram_form_of_the_device_sampler_keeps_the_same_set(vocab, top_k):
    """oracle.device_warp_histogram (how lsk_sample.h treats vocabularies of more than 32 768 entries: exact integer masses per key,
    two 256-bin levels or a full histogram, no search) against oracle.device_warp (thresholds by bisection over float sums): the same
    K / P definitions, so the same kept set and the same probabilities -- on random rows of several shapes, a row with a dominant token,
    a row of many exact ties, and the extreme settings of top_p."""
    rng = np.random.default_rng(vocab + top_k)
    import torch
    rows = []
    for scale, shift in ((2.0, 0.0), (4.0, -1.0), (0.5, 3.0)):
        r = torch.tensor(rng.standard_normal(vocab) * scale + shift).to(torch.bfloat16).float().numpy()
        rows.append(r)
    dom = rows[0].copy(); dom[17] = dom.max() + 6.0; rows.append(dom)
    ties = np.round(rows[1] * 2) / 2; rows.append(torch.tensor(ties).to(torch.bfloat16).float().numpy())
    checked = 0
    for x in rows:
        for temperature, top_p in ((0.6, 0.9), (1.0, 0.5), (0.8, 0.999), (1.3, 0.05), (0.7, 1.0), (0.9, 0.0)):
            keep, probs = so.device_warp(x, temperature, top_k, top_p)
            keep_h, probs_h, _ = so.device_warp_histogram(x, temperature, top_k, top_p)
            assert (keep == keep_h).all(), (vocab, temperature, top_p, int(keep.sum()), int(keep_h.sum()))
            assert np.allclose(probs, probs_h, rtol=0, atol=2e-6)
            checked += 1
    assert checked == 30
