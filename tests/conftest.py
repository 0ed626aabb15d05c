import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import kvpress_amd.contrib  # noqa: F401  (tests reach the out-of-scope presses as kvpress_amd.contrib.X)


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        # the library normally travels with the tree; if it is absent, compile it (hipcc, in-tree) rather than fail every test --
        # building the extension is not a fallback: without it nothing below can run
        from kvpress_amd import _native, build

        if not os.path.exists(_native.LIB_PATH):
            build.build()
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def knobs(monkeypatch):
    """knobs(KVP_X=1, KVP_Y=None, ...): set / unset KVP_* tuning variables and make the library re-read them (it reads the
    environment once and caches the values: kvp_tuning_reload); everything is restored and re-read when the test ends."""
    from kvpress_amd import _native

    def set_(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _native.tuning_reload()

    yield set_
    monkeypatch.undo()
    _native.tuning_reload()


# ---- CPU stand-in for the HIP entry points (tests of the HOST logic only) ---------------------------------------
# The HIP kernels cannot run in the build container, so host-logic tests swap kvpress_amd._native's entry points for
# oracle-backed fakes.  Test infrastructure only: the product has no such switch and fails loudly without the library.
@pytest.fixture
def fake_native(monkeypatch):
    import numpy as np
    import torch

    from kvpress_amd import _native
    from oracle import kvpress_oracle as O

    def rownorm_score(x, scale):
        return torch.from_numpy(-float(scale) * O.knorm_score(x.float().numpy()))  # scale * ||x||

    def topk_select(scores, k, order=0):
        sc = scores.float().numpy()
        sc = -sc if (order & 0x200) else sc                                          # 0x200 = KVP_TOPK_SMALLEST
        return torch.from_numpy(O.topk_select_by_score(sc, k) if (order & 0xFF) == 1 else O.topk_select(sc, k))

    def scores_fill_at_(scores, idx, value):
        idx = idx.to(torch.int64)
        S = scores.shape[-1]
        tmp = torch.cat([scores, torch.zeros_like(scores[..., :1])], dim=-1)   # out-of-range indices (the kernel skips them) land in a spare column
        tmp.scatter_(-1, torch.where((idx >= 0) & (idx < S), idx, torch.full_like(idx, S)), value)
        scores.copy_(tmp[..., :S])
        return scores

    def gather_kv(keys, values, idx):
        ko, vo = O.gather_kv(keys.numpy(), values.numpy(), idx.numpy())
        return torch.from_numpy(ko), torch.from_numpy(vo)

    def snapkv_score(q_win, keys, kernel_size):
        return torch.from_numpy(O.snapkv_score(q_win.float().numpy(), keys.float().numpy(), kernel_size))

    def snapkv_score_rope(q_pre, cos, sin, keys, kernel_size):
        q = q_pre.double().numpy()
        c, s = cos.double().numpy()[:, None], sin.double().numpy()[:, None]
        q_rot = q * c + O.rotate_half(q) * s  # snapkv_press.py:56-58
        return torch.from_numpy(O.snapkv_score(q_rot, keys.float().numpy(), kernel_size))

    def finch_score(q_pre, cos, sin, keys, normalize_scores):
        q = q_pre.double().numpy()
        c, s = cos.double().numpy()[:, None], sin.double().numpy()[:, None]
        return torch.from_numpy(O.finch_score(q * c + O.rotate_half(q) * s, keys.float().numpy(), bool(normalize_scores)))

    def snapkv_score_from_attn(attn_win, num_kv_heads, k_len, kernel_size):
        a = attn_win.double().numpy()
        B, Hq, W, Sm = a.shape
        full = np.zeros((B, Hq, W, k_len))
        full[..., :Sm] = a
        return torch.from_numpy(O.snapkv_score_from_attentions(full, num_kv_heads, W, kernel_size))

    def cur_score(keys, values, leverage_type, local_window_size, num_sinks):
        return torch.from_numpy(O.cur_score(keys.float().numpy(), values.float().numpy(), leverage_type, local_window_size > 0,
                                            max(local_window_size, 1), num_sinks))

    def observed_attention_score(attentions, num_kv_heads):
        return torch.from_numpy(O.observed_attention_score(attentions.float().numpy(), num_kv_heads))

    def lagkv_score(keys, values, n_sink, lag_size, cross_scoring):
        return torch.from_numpy(O.lagkv_score(keys.float().numpy(), values.float().numpy(), n_sink, lag_size, cross_scoring))

    def think_channel_scores(q_win, keys):
        return torch.from_numpy(O.think_channel_scores(q_win.float().numpy(), keys.float().numpy()))

    def zero_channels_(x, idx):
        x.scatter_(-1, idx.long().unsqueeze(2).expand(-1, -1, x.shape[2], -1), 0)
        return x

    def rowdot_score(x, filt, scale):
        return torch.from_numpy(np.float32(-scale) * O.qfilter_score(x.float().numpy(), filt.float().numpy()))

    def keydiff_score(keys):
        return torch.from_numpy(O.keydiff_score(keys.float().numpy()))

    def scores_head_mean_(scores):
        scores.copy_(scores.mean(dim=1, keepdim=True).expand_as(scores).clone())
        return scores

    def knorm_compress(keys, values, n_kept, order=0):
        return gather_kv(keys, values, topk_select(rownorm_score(keys, -1.0), n_kept, order))

    def snapkv_compress_rope(q_pre, cos, sin, keys, values, kernel_size, n_kept, order=0):
        return gather_kv(keys, values, topk_select(snapkv_score_rope(q_pre, cos, sin, keys, kernel_size), n_kept, order))

    def topk_select_segmented(scores, seg_len, k, pos_base=0):
        sc = scores.float().numpy()
        nseg = sc.shape[-1] // seg_len
        parts = [pos_base + c * seg_len + O.topk_select(sc[..., c * seg_len:(c + 1) * seg_len], k) for c in range(nseg)]
        return torch.from_numpy(np.concatenate(parts, axis=-1).astype(np.int32))

    def rerotate_keys_(keys_kept, idx, inv_freq):
        name = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}[keys_kept.dtype]
        out = O.rerotate_keys(keys_kept.float().numpy(), idx.numpy(), inv_freq.float().numpy(), name)
        keys_kept.copy_(torch.from_numpy(out).to(keys_kept.dtype))
        return keys_kept

    def gather_kv_rerotate(keys, values, idx, inv_freq):
        ko, vo = gather_kv(keys, values, idx)
        return rerotate_keys_(ko, idx, inv_freq), vo

    def qproj_rope_supported(module, hidden_states, window):
        return False  # CPU tensors: the model's own q_proj + the oracle-backed rope path

    def ea_qstats(q, use_cov=True):
        mu, cov = O.ea_query_stats(q.float().numpy(), use_cov)
        return torch.from_numpy(mu.astype(np.float32)), (torch.from_numpy(cov.astype(np.float32)) if cov is not None else None)

    def ea_score(keys, values, mu, cov, n_sink, use_vnorm, eps):
        return torch.from_numpy(O.ea_score(keys.float().numpy(), values.float().numpy(), mu.numpy(),
                                           cov.numpy() if cov is not None else None, n_sink, use_vnorm, eps))

    for name, fn in dict(rownorm_score=rownorm_score, topk_select=topk_select, gather_kv=gather_kv,
                         snapkv_score=snapkv_score, snapkv_score_rope=snapkv_score_rope, finch_score=finch_score, snapkv_score_from_attn=snapkv_score_from_attn,
                         keydiff_score=keydiff_score, rowdot_score=rowdot_score, think_channel_scores=think_channel_scores, zero_channels_=zero_channels_, lagkv_score=lagkv_score, observed_attention_score=observed_attention_score, cur_score=cur_score, scores_head_mean_=scores_head_mean_, knorm_compress=knorm_compress, qproj_rope_supported=qproj_rope_supported, scores_fill_at_=scores_fill_at_, topk_select_segmented=topk_select_segmented, rerotate_keys_=rerotate_keys_, gather_kv_rerotate=gather_kv_rerotate,
                         snapkv_compress_rope=snapkv_compress_rope, ea_qstats=ea_qstats, ea_score=ea_score).items():
        monkeypatch.setattr(_native, name, fn)
    return _native


