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
  * device_*          : models of the DEVICE kernels (lsk_sample.h) -- their thresholds in the 16-bit key space, their Philox
                        stream, their Gumbel-max draw -- so that kernel and model can be compared draw for draw;
                        device_warp_histogram restates the large-vocabulary form (exact integer mass histograms).
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


# ----------------------------------------------------------------------------------------------------------------
# Model of the DEVICE algorithm (layerskip_amd/csrc/lsk_sample.h), draw for draw: same Philox4x32-10 stream, same
# 16-bit key thresholds, same Gumbel-max draws.  The kernels are checked against this; this is checked against the
# reference-pinned functions above (kept sets, probabilities) and against the closed-form distributions.
# ----------------------------------------------------------------------------------------------------------------
TAG_VERIFY, TAG_ACCEPT, TAG_RESIDUAL = 32, 64, 96
_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11; the Random123 known-answer vectors are in the tests)."""
    c = [np.asarray(v, dtype=np.uint64) & _M32 for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0) & _M32, np.uint64(k1) & _M32
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & _M32, p1 & _M32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & _M32, p0 & _M32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return [v.astype(np.uint32) for v in c]


def u01(bits: np.ndarray) -> np.ndarray:
    # 23 bits: (bits >> 9) + 0.5 is exact in fp32, so the result lies strictly inside (0, 1)
    return ((bits >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)


def device_uniforms(n: int, tag: int, seed: int, offset: int) -> np.ndarray:
    """u_i of element i (counter = (i // 4, tag, offset), word i % 4)."""
    groups = (n + 3) // 4
    words = philox4x32_10(np.arange(groups), tag, offset & 0xFFFFFFFF, offset >> 32, seed & 0xFFFFFFFF, seed >> 32)
    return u01(np.stack(words, axis=1).reshape(-1)[:n])


def key16(x: np.ndarray, dtype: str = "bf16") -> np.ndarray:
    """Ordered 16-bit key of a logit (lsk_key16): exact at the model dtype's resolution.  bf16: the sign-folded upper half of the
    fp32 pattern; fp16 (the -DLSK_ELEM_F16 library): the sign-folded fp16 pattern itself."""
    if dtype == "fp16":
        k = x.astype(np.float16).view(np.uint16).astype(np.int64)
        return np.where(k & 0x8000, (~k) & 0xFFFF, k | 0x8000)
    b = x.astype(np.float32).view(np.uint32)
    k = (b >> np.uint32(16)).astype(np.int64)
    return np.where(b & np.uint32(0x80000000), (~k) & 0xFFFF, k | 0x8000)


def device_warp(logits: np.ndarray, temperature: float, top_k: int, top_p: float, dtype: str = "bf16"):
    """(kept mask, probabilities) exactly as lsk_sample_kernel computes them (thresholds in the 16-bit key space)."""
    x = logits.astype(np.float32)
    v = x.shape[0]
    keys = key16(x, dtype)
    m = x.max()
    e = np.exp(((x - m) * np.float32(1.0 / temperature)).astype(np.float32)).astype(np.float32)
    K = 0
    if 0 < top_k < v:
        K = int(np.sort(keys)[-top_k])                       # largest key with count{key >= K} >= k
    P = K
    if top_p < 1.0:
        if not top_p > 0.0:
            P = int(keys.max())
        else:
            z = e[keys >= K].sum(dtype=np.float32)
            budget = np.float32(top_p) * z
            cand = np.unique(keys[keys >= K])
            P = int(keys.max())
            for c in cand:                                   # smallest key with mass{key > c} < budget
                if e[keys > c].sum(dtype=np.float32) < budget:
                    P = int(c)
                    break
    keep = keys >= P
    probs = np.where(keep, e, np.float32(0.0)).astype(np.float32)
    return keep, (probs / probs.sum(dtype=np.float32)).astype(np.float32)


def device_warp_histogram(logits: np.ndarray, temperature: float, top_k: int, top_p: float):
    """(kept mask, probabilities, P) as the LARGE-VOCABULARY form of the device sampler computes them (lsk_sample.h, V > 32 768):
    no search -- the probability mass of every key as a 64-bit fixed-point integer (exp x 2^40, truncated), exact sums, and
    for top_k == 0 the two 256-bin levels (by the key's high byte, then the low byte inside the byte where the budget is crossed);
    with top-k the full-resolution histogram.  The SAME definitions of K and P as device_warp: the two agree unless a mass
    comparison sits within rounding of its budget (tests/test_sampling_oracle.py measures how often)."""
    x = logits.astype(np.float32)
    v = x.shape[0]
    keys = key16(x)
    m = x.max()
    e = np.exp(((x - m) * np.float32(1.0 / temperature)).astype(np.float32)).astype(np.float32)
    q = (e.astype(np.float64) * 2.0 ** 40).astype(np.uint64)          # exact: a float32 times a power of two, truncated
    key_max = int(keys.max())
    mass = np.zeros(65536, dtype=np.uint64)
    np.add.at(mass, keys, q)
    K = 0
    if 0 < top_k < v:
        cnt = np.bincount(keys, minlength=65536)
        above = np.cumsum(cnt[::-1])[::-1]                            # count{key >= k}
        K = int(np.nonzero(above >= top_k)[0].max())
    P = K
    if top_p < 1.0:
        if not top_p > 0.0:
            P = key_max
        elif 0 < top_k < v:
            Z = int(mass[K:].sum(dtype=np.uint64))
            budget = float(np.float32(top_p)) * float(Z)
            above = 0                                                 # mass{key > P} walking P downwards
            P = 65535
            for c in range(65535, -1, -1):
                if float(above) < budget:
                    P = c
                else:
                    break
                above += int(mass[c])
            P = max(P, K)
        else:
            coarse = mass.reshape(256, 256).sum(axis=1, dtype=np.uint64)
            Z = int(coarse.sum(dtype=np.uint64))
            budget = float(np.float32(top_p)) * float(Z)
            above, cb = 0, -1
            for b in range(255, -1, -1):
                if float(above) < budget <= float(above + int(coarse[b])):
                    cb = b
                    break
                above += int(coarse[b])
            assert cb >= 0
            fine = mass[cb * 256:(cb + 1) * 256]
            a, P = above, cb * 256 + 255
            for b in range(255, -1, -1):
                if float(a) < budget:
                    P = cb * 256 + b
                else:
                    break
                a += int(fine[b])
    keep = keys >= P
    zk = int(mass[P:].sum(dtype=np.uint64))
    inv_z = np.float32(1.0) / (np.float32(zk) * np.float32(2.0 ** -40))
    probs = np.where(keep, e * inv_z, np.float32(0.0)).astype(np.float32)
    return keep, probs, P


def device_sample_row(logits: np.ndarray, temperature: float, top_k: int, top_p: float, seed: int, offset: int, tag: int):
    """(token, probabilities) of one row: Gumbel-max over the kept set with the device's random stream."""
    keep, probs = device_warp(logits, temperature, top_k, top_p)
    x = logits.astype(np.float32)
    z = ((x - x.max()) * np.float32(1.0 / temperature)).astype(np.float32)
    u = device_uniforms(x.shape[0], tag, seed, offset)
    g = -np.log(-np.log(u.astype(np.float64)))
    score = np.where(keep, z.astype(np.float64) + g, -np.inf)
    return int(np.argmax(score)), probs


def device_accept_test(draft_tokens: Sequence[int], p_draft_of_token: Sequence[float], p_verify: Sequence[np.ndarray], eos: Sequence[int],
                       seed: int, offset: int) -> Tuple[int, int]:
    """(num_matches, num_drafts): the acceptance TEST of lsk_accept_sampled_kernel / lsk_pipeline_accept_sampled_kernel.  It needs only
    the scalars p_i(x_i) of the draft distributions -- what the layer pipeline's header carries to the last rank."""
    td = len(draft_tokens)
    for i, t in enumerate(draft_tokens):
        if t in eos:
            td = i + 1
            break
    u = u01(philox4x32_10(np.arange(max(td, 1)), TAG_ACCEPT, offset & 0xFFFFFFFF, offset >> 32, seed & 0xFFFFFFFF, seed >> 32)[0])
    n = 0
    for i in range(td):
        tok = draft_tokens[i]
        if u[i] < min(np.float32(1.0), np.float32(p_verify[i][tok]) / np.float32(p_draft_of_token[i])):
            n += 1
        else:
            break
    return n, td


def device_residual(q_row: np.ndarray, p_row: np.ndarray, fallback: int, seed: int, offset: int) -> int:
    """The Gumbel-max draw from max(q - p, 0) (lsk_residual_draw): what rank 0 of the layer pipeline does with the q_n it receives."""
    w = q_row.astype(np.float32) - p_row.astype(np.float32)
    if not (w > 0).any():
        return int(fallback)
    uu = device_uniforms(w.shape[0], TAG_RESIDUAL, seed, offset)
    with np.errstate(divide="ignore", invalid="ignore"):
        score = np.where(w > 0, np.log(np.where(w > 0, w, 1.0).astype(np.float64)) - np.log(-np.log(uu.astype(np.float64))), -np.inf)
    return int(np.argmax(score))


def device_accept(draft_tokens: Sequence[int], verified_tokens: Sequence[int], p_draft: Sequence[np.ndarray],
                  p_verify: Sequence[np.ndarray], eos: Sequence[int], seed: int, offset: int) -> Tuple[int, int, int]:
    """(num_matches, num_drafts, emitted token) as lsk_accept_sampled_kernel decides them."""
    n, td = device_accept_test(draft_tokens, [p_draft[i][t] for i, t in enumerate(draft_tokens)], p_verify, eos, seed, offset)
    if n == td:
        return n, td, int(verified_tokens[td])
    return n, td, device_residual(p_verify[n], p_draft[n], draft_tokens[n], seed, offset)
