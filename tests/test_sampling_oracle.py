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
