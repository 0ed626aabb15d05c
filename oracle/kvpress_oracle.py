"""CPU oracle for the kvpress score -> top-k -> gather hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``kvpress_amd``) never imports, calls or falls back to anything
in this directory; it fails loudly when the HIP extension is missing.

It is a numpy restatement of the reference algorithm (NVIDIA/kvpress v0.5.4,
``/root/reference``).  Every function cites the reference file:line it follows.
The arithmetic is done in ``ctype`` (float64 by default: the "exact math" the
fp32 reference and the fp32 HIP kernels both approximate to ~1e-6; float32 for
the large cpu_baseline runs) and results are returned as float32.

Parity pin: ``oracle/gen_golden*.py`` run the *real* reference (imported from
/root/reference, torch CPU, fp32 mode and bf16 mode) on the seeded inputs of
``tests/_inputs.py`` and commit its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` (scorers), ``tests/test_wrappers.py``, ``tests/test_finch.py``
and ``tests/test_think.py`` check this file against those fixtures.
``oracle/gen_golden_fullsize.py`` does the same at BASELINE.json's full sizes (configs 2-4; CPU-seeded inputs of
``tests/_fullsize.py``; compact fixtures ``tests/golden/full_*.npz`` checked by ``tests/test_gpu_fullsize.py``), and
``oracle/torch_path.py`` restates the reference's op sequence in plain PyTorch (pinned bit for bit to the same fixtures by
``tests/test_oracle_golden.py``): it is what ``bench.py``'s ``cpu_baseline`` leg times on the GPU box's host cores.
Caveat stated by SURVEY.md §8(c): the reference's own tests hold no score
values for SnapKV / ExpectedAttention and torch.topk's tie order is
unspecified, so the pin is "reference executed here", not "reference's own
golden vectors" (those only pin shapes / n_kept, which are checked too).
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [
    "n_kept",
    "rotate_half",
    "repeat_kv",
    "knorm_score",
    "snapkv_window_queries",
    "snapkv_window_attention",
    "snapkv_score",
    "snapkv_score_from_attentions",
    "keydiff_score",
    "cur_score",
    "tova_score",
    "pyramidkv_budget",
    "streaming_llm_score",
    "chunk_press_indices",
    "adakv_pruned",
    "rerotate_keys",
    "ea_query_stats",
    "ea_avg_rope",
    "ea_score",
    "topk_select",
    "topk_select_by_score",
    "topk_is_valid",
    "gather_kv",
    "compress",
    "finch_score",
    "finch_indices",
    "observed_attention_score",
    "lagkv_score",
    "think_channel_scores",
    "think_prune",
    "qfilter_score",
    "chunkkv_indices",
]


# --------------------------------------------------------------------------------------
# ScorerPress.compress  (kvpress/presses/scorer_press.py:76-102)
# --------------------------------------------------------------------------------------
def n_kept(k_len: int, compression_ratio: float) -> int:
    """``n_kept = int(k_len * (1 - compression_ratio))`` in Python double arithmetic
    (scorer_press.py:93-94).  e.g. int(131072*(1-0.7)) == 39321, int(23*(1-0.4)) == 13."""
    return int(k_len * (1 - compression_ratio))


def _ordered_key(scores_f32: np.ndarray) -> np.ndarray:
    """Monotone map float32 -> uint32 (larger float <=> larger key); -0.0 == +0.0 as in torch."""
    s = np.ascontiguousarray(scores_f32, dtype=np.float32)
    u = s.view(np.uint32).copy()
    u[u == np.uint32(0x80000000)] = np.uint32(0)  # -0.0 -> +0.0
    neg = (u >> np.uint32(31)).astype(bool)
    u = np.where(neg, ~u, u | np.uint32(0x80000000))
    return u


def topk_select(scores: np.ndarray, k: int) -> np.ndarray:
    """Top-k retained set per row of ``scores[..., S]`` (scorer_press.py:95,
    ``scores.topk(n_kept, dim=-1).indices``).

    torch.topk leaves the order among equal scores unspecified; this framework
    defines it: **among equal scores the lowest position wins**.  Indices are
    returned in **ascending position** order (int32) — the reference returns
    them in descending-score order, which no reference test depends on
    (tests/presses/test_presses.py:143-162 sort both sides); see DESIGN.md.
    """
    s = np.asarray(scores, dtype=np.float32)
    lead = s.shape[:-1]
    S = s.shape[-1]
    assert 0 <= k <= S
    flat = _ordered_key(s.reshape(-1, S))
    out = np.empty((flat.shape[0], k), dtype=np.int32)
    for r in range(flat.shape[0]):
        # stable sort on descending key: ties keep ascending position order
        order = np.argsort(~flat[r], kind="stable")
        out[r] = np.sort(order[:k]).astype(np.int32)
    return out.reshape(*lead, k)


def topk_select_by_score(scores: np.ndarray, k: int) -> np.ndarray:
    """The same retained set as ``topk_select`` in DESCENDING SCORE order, ties by ascending position: the element order
    ``scores.topk(k, sorted=True).indices`` has wherever the scores are distinct (scorer_press.py:95)."""
    idx = topk_select(scores, k).astype(np.int64)
    sc = np.take_along_axis(np.asarray(scores, dtype=np.float32), idx, axis=-1) + np.float32(0.0)  # -0.0 -> +0.0
    order = np.argsort(-sc.astype(np.float64), axis=-1, kind="stable")  # stable: equal scores keep ascending position
    return np.take_along_axis(idx, order, axis=-1).astype(np.int32)


def topk_is_valid(scores: np.ndarray, idx: np.ndarray, k: int, rel_band: float = 0.0):
    """Tie-tolerant check that ``idx[..., k]`` is *a* valid top-k set of ``scores``:
    every element with score > t is present, none with score < t, the rest are
    drawn from == t (t = k-th largest).  ``rel_band`` widens "== t" to
    |s - t| <= rel_band*|t| for comparisons across implementations whose fp32
    scores differ in the last bits.  Returns (ok, message)."""
    s = np.asarray(scores, dtype=np.float64)
    S = s.shape[-1]
    s2 = s.reshape(-1, S)
    i2 = np.asarray(idx).reshape(-1, k)
    for r in range(s2.shape[0]):
        ids = i2[r].astype(np.int64)
        if k == 0:
            continue
        if len(np.unique(ids)) != k or ids.min() < 0 or ids.max() >= S:
            return False, f"row {r}: indices not unique/in range"
        t = np.sort(s2[r])[::-1][k - 1]
        band = rel_band * abs(t)
        kept = np.zeros(S, dtype=bool)
        kept[ids] = True
        must = s2[r] > t + band
        never = s2[r] < t - band
        if (must & ~kept).any():
            j = int(np.nonzero(must & ~kept)[0][0])
            return False, f"row {r}: position {j} (score {s2[r, j]!r} > t={t!r}) missing"
        if (never & kept).any():
            j = int(np.nonzero(never & kept)[0][0])
            return False, f"row {r}: position {j} (score {s2[r, j]!r} < t={t!r}) kept"
    return True, "ok"


def gather_kv(keys: np.ndarray, values: np.ndarray, idx: np.ndarray):
    """``keys.gather(2, idx.expand(..., D)).contiguous()`` for K and V (scorer_press.py:96-100)."""
    B, H, S, D = keys.shape
    ii = np.asarray(idx).astype(np.int64)[..., None]
    ko = np.take_along_axis(keys, np.broadcast_to(ii, ii.shape[:-1] + (D,)), axis=2)
    vo = np.take_along_axis(values, np.broadcast_to(ii, ii.shape[:-1] + (D,)), axis=2)
    return np.ascontiguousarray(ko), np.ascontiguousarray(vo)


def compress(scores: np.ndarray, keys: np.ndarray, values: np.ndarray, compression_ratio: float):
    """ScorerPress.compress after ``score()`` (scorer_press.py:86-102): ratio 0 returns the
    inputs unchanged; otherwise top-k (tie rule above) + gather.  Returns (K', V', idx)."""
    if compression_ratio == 0:
        return keys, values, None
    k = n_kept(keys.shape[2], compression_ratio)
    idx = topk_select(scores, k)
    ko, vo = gather_kv(keys, values, idx)
    return ko, vo, idx


# --------------------------------------------------------------------------------------
# KnormPress.score  (kvpress/presses/knorm_press.py:29-38)
# --------------------------------------------------------------------------------------
def knorm_score(keys: np.ndarray, ctype=np.float64) -> np.ndarray:
    """``-keys.norm(dim=-1)`` (knorm_press.py:38)."""
    k = np.asarray(keys).astype(ctype)
    return (-np.sqrt((k * k).sum(-1))).astype(np.float32)


# --------------------------------------------------------------------------------------
# transformers helpers the reference imports (modeling_llama.py rotate_half / repeat_kv)
# --------------------------------------------------------------------------------------
def rotate_half(x: np.ndarray) -> np.ndarray:
    """transformers ``rotate_half``: cat(-x2, x1) over the last dim (used at snapkv_press.py:58)."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def repeat_kv(x: np.ndarray, n_rep: int) -> np.ndarray:
    """transformers ``repeat_kv``: [B,H,S,D] -> [B,H*n_rep,S,D], q-head j uses kv-head j//n_rep
    (used at snapkv_press.py:61, expected_attention_press.py:148)."""
    return np.repeat(x, n_rep, axis=1)


# --------------------------------------------------------------------------------------
# SnapKVPress  (kvpress/presses/snapkv_press.py:41-105)
# --------------------------------------------------------------------------------------
def snapkv_window_queries(hidden, wq, bq, cos, sin, num_heads, head_dim, window, ctype=np.float64):
    """Last-``window`` queries after RoPE (snapkv_press.py:53-58 + utils.py:43-46):
    ``q = q_proj(hidden[:, -W:])`` viewed [B,Hq,W,D]; ``q*cos + rotate_half(q)*sin`` with the
    last W rows of cos/sin ([1 or B, S, D])."""
    h = np.asarray(hidden)[:, -window:].astype(ctype)
    q = h @ np.asarray(wq).astype(ctype).T
    if bq is not None:
        q = q + np.asarray(bq).astype(ctype)
    B = h.shape[0]
    q = q.reshape(B, window, num_heads, head_dim).transpose(0, 2, 1, 3)
    c = np.asarray(cos)[:, -window:].astype(ctype)[:, None]
    s = np.asarray(sin)[:, -window:].astype(ctype)[:, None]
    return q * c + rotate_half(q) * s


def snapkv_window_attention(q_win, keys, ctype=np.float64) -> np.ndarray:
    """``compute_window_attention`` from the RoPE'd window queries on (snapkv_press.py:60-69):
    QK^T/sqrt(D) over **all** S keys, causal mask on the last W columns
    (``triu(-inf, diagonal=S-W+1)``: window row w sees columns <= S-W+w), softmax over S,
    drop the last W columns.  Returns [B,Hq,W,S-W]."""
    q = np.asarray(q_win).astype(ctype)
    k = np.asarray(keys).astype(ctype)
    B, Hq, W, D = q.shape
    H, S = k.shape[1], k.shape[2]
    G = Hq // H
    out = np.empty((B, Hq, W, S - W), dtype=ctype)
    col = np.arange(S)[None, :]
    row = np.arange(W)[:, None]
    masked = col > (S - W + row)  # strictly above the diagonal offset S-W+1
    for b in range(B):
        for hq in range(Hq):
            logits = (q[b, hq] @ k[b, hq // G].T) / ctype(math.sqrt(D))
            logits = np.where(masked, -np.inf, logits)
            m = logits.max(-1, keepdims=True)
            p = np.exp(logits - m)
            p /= p.sum(-1, keepdims=True)
            out[b, hq] = p[:, : S - W]
    return out


def _snapkv_from_window_attn(attn, H, window, kernel_size, ctype):
    """snapkv_press.py:95-103: mean over the window rows, avg_pool1d(kernel, pad=kernel//2,
    stride 1, zero padding counted in the divisor), mean over the GQA group, pad the window
    with ``scores.max() + 1`` (max is global over B and H)."""
    B, Hq, W, Sm = attn.shape
    G = Hq // H
    s = attn.mean(axis=-2)  # [B,Hq,S-W]
    pad = kernel_size // 2
    sp = np.pad(s, ((0, 0), (0, 0), (pad, pad)))
    L_out = Sm + 2 * pad - kernel_size + 1
    pooled = np.zeros((B, Hq, L_out), dtype=ctype)
    for j in range(kernel_size):
        pooled += sp[..., j : j + L_out]
    pooled /= kernel_size
    assert L_out == Sm, "even kernel_size changes the length; the reference's view() would fail too"
    sc = pooled.reshape(B, H, G, Sm).mean(2)
    # F.pad(value=scores.max().item() + 1): torch evaluates max()+1 in Python double and
    # stores it in the score dtype
    fill = float(np.float32(sc.max())) + 1.0 if sc.size else 1.0
    out = np.concatenate([sc, np.full((B, H, window), fill, dtype=ctype)], axis=-1)
    return out.astype(np.float32)


def snapkv_score(q_win, keys, kernel_size: int = 5, ctype=np.float64) -> np.ndarray:
    """SnapKVPress.score with ``attentions=None`` (snapkv_press.py:71-105) from the RoPE'd
    window queries ``q_win [B,Hq,W,D]`` and ``keys [B,H,S,D]``.  Returns [B,H,S] float32."""
    W = q_win.shape[2]
    assert keys.shape[2] > W, "Query length should be greater than the window size"  # :84-86
    attn = snapkv_window_attention(q_win, keys, ctype)
    return _snapkv_from_window_attn(attn, keys.shape[1], W, kernel_size, ctype)


def snapkv_score_from_attentions(attentions, H, window, kernel_size: int = 5, ctype=np.float64):
    """SnapKVPress.score when the layer returns attention weights (snapkv_press.py:88-89):
    ``attentions[..., -W:, :-W]``."""
    a = np.asarray(attentions).astype(ctype)[..., -window:, :-window]
    return _snapkv_from_window_attn(a, H, window, kernel_size, ctype)


def finch_score(q_win, keys, normalize_scores: bool = True, ctype=np.float64) -> np.ndarray:
    """FinchPress.score with ``attentions=None`` (finch_press.py:56-83) from the RoPE'd window queries [B,Hq,W,D]:
    window attention as SnapKV (:66-69), every window row times its number of non-masked keys ``arange(S-W, S)``
    (:71-74), mean over the window (:77) and the GQA group (:78-79), pad with max + 1 (:82).  [B,H,S] float32."""
    q = np.asarray(q_win)
    B, Hq, W, _ = q.shape
    H, S = keys.shape[1], keys.shape[2]
    attn = snapkv_window_attention(q, keys, ctype)                    # [B,Hq,W,S-W]
    if normalize_scores:
        attn = attn * np.arange(S - W, S, dtype=ctype)[None, None, :, None]
    sc = attn.mean(axis=-2).reshape(B, H, Hq // H, S - W).mean(2)
    fill = float(np.float32(sc.max())) + 1.0 if sc.size else 1.0
    return np.concatenate([sc, np.full((B, H, W), fill, dtype=ctype)], axis=-1).astype(np.float32)


def finch_indices(scores: np.ndarray, compression_ratio: float, chunk_length=None) -> np.ndarray:
    """The kept positions of FinchPress.compress (finch_press.py:99-112), ascending (the order of the re-rotating
    variant, :114): a global top-k, or per chunk ``max(1, int(len * (1 - ratio)))`` of every ``chunk_length`` block."""
    S = scores.shape[-1]
    if chunk_length is None:
        return topk_select(scores, n_kept(S, compression_ratio))
    parts = []
    for i in range(0, S, chunk_length):
        c = scores[..., i : i + chunk_length]
        parts.append(i + topk_select(c, max(1, int(c.shape[-1] * (1 - compression_ratio)))))
    return np.concatenate(parts, axis=-1).astype(np.int32)


# --------------------------------------------------------------------------------------
# ExpectedAttentionPress  (kvpress/presses/expected_attention_press.py:62-165)
# --------------------------------------------------------------------------------------
def ea_query_stats(q, use_covariance: bool = True, ctype=np.float64):
    """Mean and covariance of the (pre-RoPE, sink-stripped) queries ``q [B,Hq,S',D]``
    (expected_attention_press.py:74-81): mu = mean over S'; cov = (q-mu)^T (q-mu) / S'."""
    x = np.asarray(q).astype(ctype)
    mu = x.mean(axis=2)
    cov = None
    if use_covariance:
        c = x - mu[:, :, None, :]
        cov = np.einsum("bnsi,bnsj->bnij", c, c) / x.shape[2]
    return mu, cov


def ea_avg_rope(mu, cov, cos, sin, ctype=np.float64):
    """apply_avg_rope (expected_attention_press.py:110-123) given the cos/sin [P,D] of the
    n_future_positions positions: R_p = diag(cos_p) + P*sin_p with P[D/2:, :D/2] = I,
    P[:D/2, D/2:] = -I (:114-118); R = mean_p R_p (:119-120); mu <- mu R^T, cov <- R cov R^T."""
    cos = np.asarray(cos).astype(ctype)
    sin = np.asarray(sin).astype(ctype)
    D = cos.shape[-1]
    h = D // 2
    Pm = np.zeros((D, D), dtype=ctype)
    Pm[h:, :h] = np.eye(h)
    Pm[:h, h:] = -np.eye(h)
    R = (cos[:, :, None] * np.eye(D, dtype=ctype)[None] + sin[:, :, None] * Pm[None]).mean(0)
    mu2 = np.asarray(mu).astype(ctype) @ R.T
    cov2 = None
    if cov is not None:
        cov2 = R @ (np.asarray(cov).astype(ctype) @ R.T)
    return mu2, cov2


def ea_score(keys, values, mu, cov, n_sink=4, use_vnorm=True, epsilon=0.0, ctype=np.float64):
    """ExpectedAttentionPress.score after the query statistics (expected_attention_press.py:137-163):
    drop n_sink keys/values; per q-head logits = k.mu/sqrt(D) + k^T cov k / D / 2; softmax over
    the S-n_sink keys; mean over the GQA group; (s + eps) * ||v||; left-pad n_sink with max+1."""
    k = np.asarray(keys).astype(ctype)
    v = np.asarray(values).astype(ctype)
    assert k.shape[2] > n_sink, f"Input should contain more tokens than n_sink={n_sink}"  # :137
    k = k[:, :, n_sink:]
    v = v[:, :, n_sink:]
    B, H, Sp, D = k.shape
    mu = np.asarray(mu).astype(ctype)
    Hq = mu.shape[1]
    G = Hq // H
    kr = repeat_kv(k, G)  # [B,Hq,S',D]
    logits = np.einsum("bhd,bhsd->bhs", mu, kr) / ctype(math.sqrt(D))
    if cov is not None:
        cv = np.asarray(cov).astype(ctype)
        y = np.einsum("bhsi,bhij->bhsj", kr, cv)
        logits = logits + (y * kr).sum(-1) / D / 2
    m = logits.max(-1, keepdims=True)
    p = np.exp(logits - m)
    p /= p.sum(-1, keepdims=True)
    sc = p.reshape(B, H, G, Sp).mean(2)
    if use_vnorm:
        sc = (sc + epsilon) * np.sqrt((v * v).sum(-1))
    fill = float(np.float32(sc.max())) + 1.0
    out = np.concatenate([np.full((B, H, n_sink), fill, dtype=ctype), sc], axis=-1)
    return out.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# scorers that reuse the path's kernels
# ----------------------------------------------------------------------------------------------
def keydiff_score(keys: np.ndarray, ctype=np.float64) -> np.ndarray:
    """KeyDiffPress.score (keydiff_press.py:45-46): ``anchor = F.normalize(keys, p=2, dim=-1).mean(dim=2, keepdim=True)``,
    ``-F.cosine_similarity(keys, anchor, dim=-1)``.  F.normalize: x / max(||x||, 1e-12); cosine_similarity (ATen
    Distance.cpp, torch >= 1.12): sum (x / max(||x||, 1e-8)) * (y / max(||y||, 1e-8))."""
    k = keys.astype(ctype)
    nk = np.sqrt((k * k).sum(-1, keepdims=True))
    anchor = (k / np.maximum(nk, 1e-12)).mean(axis=2, keepdims=True)
    na = np.sqrt((anchor * anchor).sum(-1, keepdims=True))
    cos = ((k / np.maximum(nk, 1e-8)) * (anchor / np.maximum(na, 1e-8))).sum(-1)
    return (-cos).astype(np.float32)


def observed_attention_score(attentions: np.ndarray, H: int, ctype=np.float64) -> np.ndarray:
    """ObservedAttentionPress.score (observed_attention_press.py:42-48): ``attentions.sum(2)`` over the queries, divided
    by ``arange(S, 0, -1)`` (how many queries can see each key), mean over the GQA group.  attentions [B,Hq,Sq,S]."""
    a = np.asarray(attentions).astype(ctype)
    B, Hq, _, S = a.shape
    sc = a.sum(2) / np.arange(S, 0, -1, dtype=ctype)
    return sc.reshape(B, H, Hq // H, S).mean(2).astype(np.float32)


def _lag_states_score(x, ctype):
    """``_get_states_score`` (lagkv_press.py:88-97): x [B,H,P,L,D]; partition p against the min / max of partition p+1."""
    ref, v = x[:, :, 1:], x[:, :, :-1]
    mn, mx = ref.min(axis=-2, keepdims=True), ref.max(axis=-2, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        sd = ((v - mn) / (mx - mn)).std(axis=-1, ddof=1)             # torch.std: unbiased
    e = np.exp(sd - sd.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)                              # softmax over the partition's tokens


def lagkv_score(keys, values, n_sink: int = 4, lag_size: int = 128, cross_scoring: bool = False, ctype=np.float64) -> np.ndarray:
    """LagKVPress.score (lagkv_press.py:56-86).  Ranks (``argsort().argsort()``) break ties by position."""
    k, v = np.asarray(keys).astype(ctype), np.asarray(values).astype(ctype)
    B, H, S, D = k.shape
    if S < n_sink + 2 * lag_size:                                    # :57-63
        sc = np.ones((B, H, S), ctype)
        if S > n_sink:
            sc[:, :, n_sink:] = np.arange(S - n_sink, dtype=ctype) / (S - n_sink)
        return sc.astype(np.float32)
    end = n_sink + ((S - n_sink) // lag_size) * lag_size
    ks = _lag_states_score(k[:, :, n_sink:end].reshape(B, H, -1, lag_size, D), ctype)
    vs = _lag_states_score(v[:, :, n_sink:end].reshape(B, H, -1, lag_size, D), ctype)
    sc = (ks + vs) / 2
    if not cross_scoring:
        sc = np.argsort(np.argsort(sc, axis=-1, kind="stable"), axis=-1, kind="stable") / lag_size
    tail = lag_size + S - end
    return np.concatenate([np.ones((B, H, n_sink), ctype), sc.reshape(B, H, -1), np.ones((B, H, tail), ctype)], axis=-1).astype(np.float32)


def think_channel_scores(q_win, keys, ctype=np.float64) -> np.ndarray:
    """ThinKPress's per-channel scores (think_press.py:72-76): ``pow(queries, 2).mean(2)`` over the window, mean over the
    GQA group, times ``pow(keys, 2).mean(2)`` over the tokens.  q_win [B,Hq,W,D] RoPE'd, keys [B,H,S,D] -> [B,H,D]."""
    q, k = np.asarray(q_win).astype(ctype), np.asarray(keys).astype(ctype)
    B, Hq, _, D = q.shape
    H = k.shape[1]
    qn = (q ** 2).mean(2).reshape(B, H, Hq // H, D).mean(2)
    return (qn * (k ** 2).mean(2)).astype(np.float32)


def think_prune(keys, scores, key_channel_compression_ratio: float):
    """The pruned channels (ascending) and the keys with them zeroed (think_press.py:79-82): the
    ``int(D * ratio)`` lowest-scoring channels per (batch, head); equal scores: lowest channel first."""
    k = np.array(keys, copy=True)
    D = k.shape[-1]
    idx = topk_select(-np.asarray(scores, np.float32), int(D * key_channel_compression_ratio))
    np.put_along_axis(k, np.broadcast_to(idx[:, :, None, :].astype(np.int64), k.shape[:3] + (idx.shape[-1],)), 0, axis=-1)
    return idx, k


def qfilter_score(keys: np.ndarray, q_filter: np.ndarray, ctype=np.float64) -> np.ndarray:
    """QFilterPress.score (qfilter_press.py:79-82): ``-(q_filter[None, :, None] * keys).sum(-1)`` with the layer's
    filters ``q_filter [H, D]``.  [B,H,S] float32."""
    k = np.asarray(keys).astype(ctype)
    f = np.asarray(q_filter).astype(ctype)
    return (-(f[None, :, None, :] * k).sum(-1)).astype(np.float32)


def tova_score(q_last, keys, ctype=np.float64) -> np.ndarray:
    """TOVAPress.score with ``attentions=None`` (tova_press.py:45-59) from the RoPE'd query of the LAST token
    ``q_last [B,Hq,1,D]``: window attention with window 1 (SnapKVPress.compute_window_attention, snapkv_press.py:41-69),
    mean over ALL heads (:52), repeated for every kv-head (:53), right-padded with max + 1 (:58)."""
    attn = snapkv_window_attention(q_last, keys, ctype)          # [B,Hq,1,S-1]
    s = attn.mean(axis=1)                                        # [B,1,S-1]
    s = np.repeat(s, keys.shape[1], axis=1)
    pad = np.full(s.shape[:-1] + (1,), s.max() + 1.0, dtype=s.dtype)
    return np.concatenate([s, pad], axis=-1).astype(np.float32)


def pyramidkv_budget(q_len: int, compression_ratio: float, window_size: int, beta: int, num_layers: int, layer_idx: int) -> int:
    """PyramidKVPress.get_layer_budget (pyramidkv_press.py:47-81): a linear ramp of per-layer budgets from max_num
    (layer 0) to min_num (last layer) whose mean keeps q_len * (1 - ratio) tokens; falls back to the SnapKV budget
    ``round(q_len * (1 - ratio))`` when the ramp would leave [window_size, q_len]."""
    assert beta >= 1
    cap = window_size + q_len * (1 - compression_ratio)
    lo = (cap - window_size) / beta
    hi = (cap - window_size) * 2 - lo
    if hi >= q_len - window_size:
        hi = q_len - window_size
        lo = (cap - window_size) * 2 - hi
    if not (q_len >= hi >= lo >= window_size):
        return round(q_len * (1 - compression_ratio))
    step = (hi - lo) / (num_layers - 1)
    return round(hi - layer_idx * step)


def streaming_llm_score(B: int, H: int, k_len: int, compression_ratio: float, n_sink: int = 4) -> np.ndarray:
    """StreamingLLMPress.score (streaming_llm_press.py:47-52): ones, zeros on the n_pruned positions after the sinks."""
    assert k_len > n_sink
    n_pruned = k_len - int(k_len * (1 - compression_ratio))
    s = np.ones((B, H, k_len), dtype=np.float32)
    s[:, :, n_sink:n_sink + n_pruned] = 0
    return s


# ----------------------------------------------------------------------------------------------
# selection wrappers
# ----------------------------------------------------------------------------------------------
def _round_dtype(x: np.ndarray, dtype: str) -> np.ndarray:
    """float32 values rounded (RNE) to ``dtype`` ("f32" | "f16" | "bf16") and returned as float32."""
    x = np.asarray(x, dtype=np.float32)
    if dtype == "f32":
        return x
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def chunk_press_indices(score_chunk, k_len: int, chunk_length: int, compression_ratio: float) -> np.ndarray:
    """ChunkPress.compress's index set (chunk_press.py:67-83): for every chunk [i, i + L) the ``max(1, int(len * (1 - r)))``
    best positions by ``score_chunk(i, j) -> scores[B,H,j-i]`` of that chunk alone, concatenated (each chunk in ascending
    position order here; the reference keeps topk's order inside a chunk)."""
    out = []
    for i in range(0, k_len, chunk_length):
        j = min(k_len, i + chunk_length)
        sc = score_chunk(i, j)
        n = max(1, int((j - i) * (1 - compression_ratio)))
        out.append(i + topk_select(sc, n))
    return np.concatenate(out, axis=-1).astype(np.int32)


def chunkkv_indices(scores: np.ndarray, chunk_length: int, compression_ratio: float) -> np.ndarray:
    """Kept positions of ChunkKVPress.compress (chunkkv_press.py:80-116), ascending, the same for every batch element and
    head: chunk score = mean over the chunk of the head-summed scores, top ``max(1, int(n_chunks * (1 - ratio)))`` chunks
    of batch element 0.  (Fewer tokens than one chunk: the wrapped press's own top-k, :77-78 -- not handled here.)"""
    sc = np.asarray(scores, dtype=np.float64)
    S = sc.shape[-1]
    n_full, tail = divmod(S, chunk_length)
    assert n_full > 0
    per_token = sc.sum(1)
    cs = per_token[:, : n_full * chunk_length].reshape(sc.shape[0], n_full, chunk_length).mean(-1)
    if tail:
        cs = np.concatenate([cs, per_token[:, -tail:].mean(-1, keepdims=True)], -1)
    n_kept = max(1, int(cs.shape[-1] * (1 - compression_ratio)))
    top = topk_select(cs[:1].astype(np.float32), n_kept)[0]
    pos = (top[:, None].astype(np.int64) * chunk_length + np.arange(chunk_length)[None]).reshape(-1)
    return pos[pos < S].astype(np.int32)


def rerotate_keys(keys_kept: np.ndarray, idx: np.ndarray, inv_freq: np.ndarray, dtype: str = "f32") -> np.ndarray:
    """KeyRerotationPress.rerotate_keys after the gather (key_rerotation_press.py:50-128): token j of the kept
    (position-sorted) keys moves from position idx[..., j] to position j: ``k * cos(f) + rotate_half(k) * sin(f)`` with
    ``f = (j - idx) * inv_freq`` in float32, cos/sin cast to the key dtype and every product / sum rounded in it."""
    k = np.asarray(keys_kept, dtype=np.float32)
    n = k.shape[2]
    delta = (np.arange(n, dtype=np.float32)[None, None, :] - idx.astype(np.float32))            # [B,H,n]
    freqs = (delta[..., None] * np.asarray(inv_freq, dtype=np.float32)[None, None, None, :]).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    cos = _round_dtype(np.cos(emb.astype(np.float64)).astype(np.float32), dtype)
    sin = _round_dtype(np.sin(emb.astype(np.float64)).astype(np.float32), dtype)
    a = _round_dtype(k * cos, dtype)
    b = _round_dtype(rotate_half(k) * sin, dtype)
    return _round_dtype(a + b, dtype)


def adakv_pruned(scores: np.ndarray, compression_ratio: float, alpha_safeguard: float = 0.2) -> np.ndarray:
    """AdaKVPress.compress's pruned set (adakv_press.py:56-75) as sorted flat indices ``h * S + s`` per batch element,
    int64 [B, H * (S - n_kept)]: the n_safe = int(n_kept * alpha) best tokens of every head are protected (set to the
    float maximum), then the lowest scores across all heads are pruned."""
    sc = np.array(scores, dtype=np.float32, copy=True)
    B, H, S = sc.shape
    n_kept = int(S * (1 - compression_ratio))
    n_safe = int(n_kept * alpha_safeguard)
    if n_safe:
        top = topk_select(sc, n_safe)
        np.put_along_axis(sc, top.astype(np.int64), np.finfo(np.float32).max, axis=-1)
    n_pruned = H * (S - n_kept)
    return topk_select(-sc.reshape(B, H * S), n_pruned).astype(np.int64)


def cur_score(keys, values, leverage_type="kv_product", use_local_approximation=True, local_window_size=16, num_sinks=4, ctype=np.float64):
    """CURPress.score without the random projection (cur_press.py:40-64)."""
    k2 = (keys.astype(ctype) ** 2).sum(-1)
    v2 = (values.astype(ctype) ** 2).sum(-1)
    if use_local_approximation:
        B, H, n = k2.shape
        w = local_window_size
        pad = (w - n % w) % w

        def local(x):
            xp = np.concatenate([x, np.zeros((B, H, pad), dtype=x.dtype)], axis=-1).reshape(B, H, -1, w)
            return (xp / xp.sum(-1, keepdims=True)).reshape(B, H, -1)[:, :, :n]
        k2, v2 = local(k2), local(v2)
    sc = {"key": k2, "value": v2, "kv_avg": (k2 + v2) / 2, "kv_product": k2 * v2}[leverage_type]
    sc = sc / sc.sum(-1, keepdims=True)
    sc[:, :, :num_sinks] = 1.0
    return sc.astype(np.float32)
