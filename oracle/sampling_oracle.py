"""CPU restatement of the SAMPLING side of the reference's speculation step -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product never does.

What is restated (deterministic parts; the random draws are inputs):
  * warp_logits       : `decode_next_token`'s `logits / temperature` followed by `top_k_top_p_filtering`
                        (reference self_speculation/llama_model_utils.py:72-107, :124-126), i.e. transformers'
                        TopKLogitsWarper then TopPLogitsWarper (third-party dependency, version pinned by this image:
                        transformers 4.x; their published algorithm is restated here):
                          top-k : remove every logit strictly below the k-th largest;
                          top-p : sort ascending, softmax, cumulative sum; remove the prefix whose cumulative mass is
                                  <= 1 - top_p, never the last `min_tokens_to_keep` entries;
  * probabilities     : softmax of the warped logits (llama_model_utils.py:127);
  * residual          : `max_fn(p_verify - p_draft)` (self_speculation_generator.py:27-29): negatives clamped to 0,
                        divided by (sum + 1e-6);
  * accept_step       : the modified rejection sampling loop of single_step_speculation
                        (self_speculation_generator.py:191-199): draft i is kept when u_i < min(1, q(x_i) / p(x_i));
                        at the first rejection the verified token is drawn from the residual distribution; when all
                        drafts are kept the extra token is the one sampled from the last verify row.
Parity is pinned by oracle/make_sampling_golden.py: vectors produced by the UNMODIFIED reference functions in the
build container, stored under tests/golden/sampling/, checked by tests/test_sampling_oracle.py.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

NEG_INF = -np.inf


def _softmax(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    m = np.max(x[np.isfinite(x)]) if np.isfinite(x).any() else 0.0
    e = np.where(np.isfinite(x), np.exp((x - m).astype(np.float32)), np.float32(0.0)).astype(np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)


def warp_logits(logits: np.ndarray, temperature: float, top_k: int, top_p: float, min_tokens_to_keep: int = 1) -> np.ndarray:
    """One row of logits -> the filtered logits the reference feeds to softmax (removed entries = -inf)."""
    x = (logits.astype(np.float32) / np.float32(temperature)).astype(np.float32)
    if top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), x.shape[-1])
        kth = np.sort(x)[-k]
        x = np.where(x < kth, np.float32(NEG_INF), x)
    if 0 <= top_p <= 1.0:
        order = np.argsort(x, kind="stable")                       # ascending, like torch.sort(descending=False)
        probs = _softmax(x[order])
        cum = np.cumsum(probs, dtype=np.float32)
        remove = cum <= np.float32(1.0 - top_p)
        remove[-min_tokens_to_keep:] = False
        out = x.copy()
        out[order[remove]] = np.float32(NEG_INF)
        x = out
    return x


def probabilities(warped: np.ndarray) -> np.ndarray:
    return _softmax(warped)


def residual(p_verify: np.ndarray, p_draft: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    d = (p_verify.astype(np.float32) - p_draft.astype(np.float32)).astype(np.float32)
    d = np.where(d > 0, d, np.float32(0.0)).astype(np.float32)
    return (d / (d.sum(dtype=np.float32) + np.float32(eps))).astype(np.float32)


def inverse_cdf(p: np.ndarray, u: float) -> int:
    """The categorical draw as a function of one uniform (what a device kernel with its own counter RNG does;
    torch.multinomial consumes its generator differently, so draws are compared in distribution, not by value)."""
    c = np.cumsum(p.astype(np.float64))
    return int(min(np.searchsorted(c, u * c[-1], side="right"), p.shape[0] - 1))


def accept_step(draft_tokens: Sequence[int], p_draft: Sequence[np.ndarray], p_verify: Sequence[np.ndarray],
                uniforms: Sequence[float], resample_u: float, bonus_token: Optional[int] = None) -> Tuple[int, int]:
    """(number_of_matches, token emitted after the kept drafts).  `p_verify` has len(draft_tokens) + 1 rows; the
    bonus token (all drafts kept) is the reference's sample from the last verify row, passed in by the caller or
    drawn here from `resample_u`."""
    n = 0
    for i, tok in enumerate(draft_tokens):
        q, p = float(p_verify[i][tok]), float(p_draft[i][tok])
        if uniforms[i] < min(1.0, q / p):
            n += 1
        else:
            return n, inverse_cdf(residual(p_verify[i], p_draft[i]), resample_u)
    if bonus_token is None:
        bonus_token = inverse_cdf(p_verify[len(draft_tokens)], resample_u)
    return n, int(bonus_token)


def emitted_distribution(p_draft_row: np.ndarray, p_verify_row: np.ndarray) -> np.ndarray:
    """Property the whole scheme rests on (Leviathan et al. 2023, Chen et al. 2023): drawing x ~ p_draft, keeping it
    with probability min(1, q/p) and otherwise drawing from max_fn(q - p) emits a token distributed as q (up to the
    reference's 1e-6 regulariser).  Returned in closed form for tests."""
    p, q = p_draft_row.astype(np.float64), p_verify_row.astype(np.float64)
    keep = np.minimum(p, q)                                   # P(draw x and keep it)
    reject_mass = 1.0 - keep.sum()
    return keep + reject_mass * residual(p_verify_row, p_draft_row).astype(np.float64)
