"""ctypes binding of libkvpress_hip.so (include/kvpress_hip.h) for torch tensors on a HIP device.

There is deliberately NO fallback: if the shared library is missing, or a tensor is not on a
GPU, the call raises.  PyTorch is used for device memory (outputs / workspaces come from the
caching allocator, so they are stream-ordered) and for the current stream only.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch  # must be imported before the .so so that it binds to torch's libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
# KVPRESS_HIP_LIB: an alternative build of the same library (kernel labs under tools/); unset in production
LIB_PATH = os.environ.get("KVPRESS_HIP_LIB") or os.path.join(_HERE, "lib", "libkvpress_hip.so")
CONTRIB_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libkvpress_hip_contrib.so")

KVP_F32, KVP_F16, KVP_BF16 = 0, 1, 2
ORDER_POSITION, ORDER_SCORE = 0, 1
TOPK_WS_CLEAN = 0x100
_TOPK_WS: dict = {}  # (thread, device index, stream, R, S) -> zero-initialised, self-cleaning workspace
_TOPK_WS_LOCK = threading.Lock()
_DTYPES = {torch.float32: KVP_F32, torch.float16: KVP_F16, torch.bfloat16: KVP_BF16}

# name -> (restype, argtypes); mirrors include/kvpress_hip.h line by line
_I64 = c_int64
SIGNATURES = {
    "kvp_version": (c_int, []),
    "kvp_last_error": (c_char_p, []),
    "kvp_async_error_check": (c_int, []),
    "kvp_rownorm_score": (c_int, [c_void_p, c_int, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_float, c_void_p, c_void_p]),
    "kvp_rowdot_score": (c_int, [c_void_p, c_int, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_void_p, _I64, c_float, c_void_p, c_void_p]),
    "kvp_knorm_compress_workspace_bytes": (c_size_t, [_I64] * 4),
    "kvp_knorm_compress": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "kvp_snapkv_compress_workspace_bytes": (c_size_t, [_I64] * 7),
    "kvp_snapkv_compress_rope": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, c_void_p, _I64, _I64, c_void_p, _I64, _I64, _I64,
                                         c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _I64,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "kvp_cur_workspace_bytes": (c_size_t, [_I64] * 3),
    "kvp_cur_score": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, c_int, _I64, _I64,
                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_keydiff_workspace_bytes": (c_size_t, [_I64] * 4),
    "kvp_keydiff_score": (c_int, [c_void_p, c_int, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_scores_head_mean": (c_int, [c_void_p, _I64, _I64, _I64, _I64, _I64, c_void_p]),
    "kvp_snapkv_workspace_bytes": (c_size_t, [_I64] * 6),
    "kvp_snapkv_score": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int,
                                 _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_snapkv_score_rope": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, c_void_p, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int,
                                      _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_finch_score": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, c_void_p, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int,
                                _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_snapkv_qproj_rope": (c_int, [c_void_p, _I64, _I64, c_void_p, c_void_p, c_void_p, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64,
                                      c_void_p, c_void_p]),
    "kvp_snapkv_score_hidden": (c_int, [c_void_p, _I64, _I64, c_void_p, _I64, c_void_p, c_void_p, _I64, _I64, c_void_p, _I64, _I64, _I64,
                                        c_int, _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_snapkv_compress_hidden": (c_int, [c_void_p, _I64, _I64, c_void_p, _I64, c_void_p, c_void_p, _I64, _I64, c_void_p, _I64, _I64, _I64,
                                           c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _I64,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "kvp_snapkv_score_from_attn": (c_int, [c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, c_int,
                                           c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_ea_qstats_workspace_bytes": (c_size_t, [_I64] * 4),
    "kvp_ea_qstats": (c_int, [c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "kvp_ea_score_workspace_bytes": (c_size_t, [_I64] * 5),
    "kvp_ea_score": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, c_void_p, c_void_p,
                             _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_topk_workspace_bytes": (c_size_t, [_I64] * 3),
    "kvp_topk_order_workspace_bytes": (c_size_t, [_I64] * 3),
    "kvp_topk_select": (c_int, [c_void_p, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_topk_segmented_workspace_bytes": (c_size_t, [_I64] * 4),
    "kvp_topk_select_segmented": (c_int, [c_void_p, _I64, _I64, _I64, _I64, _I64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_scores_fill_at": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, c_float, c_void_p]),
    "kvp_rerotate_keys": (c_int, [c_void_p, c_int, _I64, _I64, _I64, _I64, c_void_p, c_void_p, c_void_p]),
    "kvp_gather_kv": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64,
                              c_void_p, _I64, c_void_p, c_void_p, c_void_p]),
    "kvp_gather_kv_rerotate": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64,
                                       c_void_p, _I64, c_void_p, c_void_p, c_void_p, c_void_p]),
}
# include/kvpress_hip_lab.h: measurement / test aids of the same library (bench.py, tools/, the select's residency tests)
LAB_SIGNATURES = {
    "kvp_prof_enable": (c_int, [c_int]),
    "kvp_prof_count": (c_int, []),
    "kvp_prof_get": (c_int, [c_int, ctypes.POINTER(c_char_p), ctypes.POINTER(c_float)]),
    "kvp_clock_probe": (c_int, [c_void_p, c_int, c_void_p]),
    "kvp_occupy_cus": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "kvp_tuning_reload": (c_int, []),
}
# include/kvpress_hip_extra.h: kernels of the presses outside SURVEY section 8 (kvpress_amd.contrib), their own shared library
CONTRIB_SIGNATURES = {
    "kvp_observed_attention_score": (c_int, [c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, c_void_p, c_void_p]),
    "kvp_lagkv_score": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, _I64, c_int,
                                c_void_p, c_void_p]),
    "kvp_think_workspace_bytes": (c_size_t, [_I64] * 4),
    "kvp_think_channel_scores": (c_int, [c_void_p, _I64, _I64, _I64, c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, _I64, _I64,
                                         c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvp_zero_channels": (c_int, [c_void_p, _I64, _I64, _I64, c_int, _I64, _I64, _I64, _I64, c_void_p, _I64, c_void_p]),
}

_lib = None
_contrib_lib = None


class KvpressHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libkvpress_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KvpressHipError(
                f"{LIB_PATH} not found: build it with `python -m kvpress_amd.build` (hipcc, gfx950). "
                "kvpress_amd has no CPU / pure-PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **LAB_SIGNATURES}.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def contrib_lib() -> ctypes.CDLL:
    """Load libkvpress_hip_contrib.so (once): LagKV / ThinK / ObservedAttention kernels (kvpress_amd.contrib).  It resolves the
    error / launch plumbing against libkvpress_hip.so, which is loaded first."""
    global _contrib_lib
    if _contrib_lib is None:
        lib()
        if not os.path.exists(CONTRIB_LIB_PATH):
            raise KvpressHipError(f"{CONTRIB_LIB_PATH} not found: build it with `python -m kvpress_amd.build` (hipcc, gfx950).")
        handle = ctypes.CDLL(CONTRIB_LIB_PATH)
        for name, (res, args) in CONTRIB_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _contrib_lib = handle
    return _contrib_lib


KVP_EASYNC = -5


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().kvp_last_error()
        if rc == KVP_EASYNC:
            # a kernel of an EARLIER call reported at run time that its result is invalid (the cluster select's barrier timed out:
            # include/kvpress_hip.h).  Its workspace -- whichever it was -- is dirty: no cached "clean" workspace survives.
            with _TOPK_WS_LOCK:
                _TOPK_WS.clear()
        raise KvpressHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def async_error_check() -> None:
    """Raise KvpressHipError if a kernel has reported an asynchronous failure since the last check (no synchronisation: call it
    after a stream sync to cover everything enqueued so far).  No-op when the library has not been loaded."""
    if _lib is not None:
        _check(_lib.kvp_async_error_check(), "kvp_async_error_check")


def _dev(t: torch.Tensor):
    if not t.is_cuda:
        raise KvpressHipError(
            f"kvpress_amd kernels need tensors on a HIP device, got {t.device} (no CPU fallback exists)")
    if t.dtype not in _DTYPES:
        raise KvpressHipError(f"unsupported dtype {t.dtype} (float32 / float16 / bfloat16)")
    return t


def _rows_last_contig(t: torch.Tensor) -> torch.Tensor:
    """[..., D] view whose last dim is contiguous (sliced cache views pass through untouched)."""
    return t if (t.shape[-1] == 1 or t.stride(-1) == 1) else t.contiguous()


def _st(t: torch.Tensor, i: int) -> int:
    """Element stride of dim i; 0 for size-1 dims (torch reports arbitrary strides there)."""
    return 0 if t.shape[i] == 1 else t.stride(i)


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream   # (an int: ctypes converts it for a c_void_p parameter)


class _on_device:
    """``with _on_device(dev)`` only when ``dev`` is not already current: the context manager costs several microseconds
    per call, which is a tenth of a whole decode-regime compress (tools/host_overhead_probe.py)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)


def _p(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None   # (ints / None: ctypes converts them for a c_void_p parameter)


# ------------------------------------------------------------------------------------------------
def rownorm_score(x: torch.Tensor, scale: float) -> torch.Tensor:
    """out[b,h,s] = scale * ||x[b,h,s,:]||_2, float32 [B,H,S]."""
    x = _rows_last_contig(_dev(x))
    B, H, S, D = x.shape
    out = torch.empty((B, H, S), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _check(lib().kvp_rownorm_score(_p(x), _DTYPES[x.dtype], B, H, S, D, _st(x, 0), _st(x, 1), _st(x, 2),
                                       float(scale), _p(out), _stream(x)), "kvp_rownorm_score")
    return out


def observed_attention_score(attentions: torch.Tensor, num_kv_heads: int) -> torch.Tensor:
    """Average attention weight every key receives (observed_attention_press.py:42-48): attentions [B,Hq,Sq,S] -> [B,Hkv,S] float32."""
    a = _rows_last_contig(_dev(attentions))
    B, Hq, Sq, S = a.shape
    assert Hq % num_kv_heads == 0, (a.shape, num_kv_heads)
    out = torch.empty((B, num_kv_heads, S), dtype=torch.float32, device=a.device)
    with _on_device(a.device):
        _check(contrib_lib().kvp_observed_attention_score(_p(a), _st(a, 0), _st(a, 1), _st(a, 2), _DTYPES[a.dtype], B, Hq, num_kv_heads, Sq, S,
                                                  _p(out), _stream(a)), "kvp_observed_attention_score")
    return out


def lagkv_score(keys: torch.Tensor, values: torch.Tensor, n_sink: int, lag_size: int, cross_scoring: bool) -> torch.Tensor:
    """LagKV scores [B,H,S] float32 (lagkv_press.py:56-97) for S >= n_sink + 2 * lag_size."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape == values.shape
    B, H, S, D = keys.shape
    out = torch.empty((B, H, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        _check(contrib_lib().kvp_lagkv_score(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _p(values), _st(values, 0), _st(values, 1),
                                     _st(values, 2), _DTYPES[keys.dtype], B, H, S, D, int(n_sink), int(lag_size), int(bool(cross_scoring)),
                                     _p(out), _stream(keys)), "kvp_lagkv_score")
    return out


def think_channel_scores(q_win: torch.Tensor, keys: torch.Tensor) -> torch.Tensor:
    """ThinK's per-channel scores [B,Hkv,D] float32 (think_press.py:72-76) from the RoPE'd window queries [B,Hq,W,D] and keys."""
    keys = _rows_last_contig(_dev(keys))
    q_win = _rows_last_contig(_dev(q_win))
    if q_win.dtype != keys.dtype:
        q_win = q_win.to(keys.dtype)
    B, Hq, W, D = q_win.shape
    Bk, Hkv, S, Dk = keys.shape
    assert B == Bk and D == Dk and Hq % Hkv == 0, (q_win.shape, keys.shape)
    out = torch.empty((B, Hkv, D), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        nws = contrib_lib().kvp_think_workspace_bytes(B, Hkv, S, D)
        ws = _ws(nws, keys)
        _check(contrib_lib().kvp_think_channel_scores(_p(q_win), _st(q_win, 0), _st(q_win, 1), _st(q_win, 2), _p(keys), _st(keys, 0), _st(keys, 1),
                                              _st(keys, 2), _DTYPES[keys.dtype], B, Hq, Hkv, S, W, D, _p(out), _p(ws), ws.numel(), _stream(keys)),
               "kvp_think_channel_scores")
    return out


def zero_channels_(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x[b,h,:,idx[b,h,j]] = 0 in place (think_press.py:81-82); x [B,H,S,D] with a contiguous last dim, idx [B,H,n] int32."""
    assert x.is_cuda and x.stride(-1) == 1, "zero_channels_ works in place on a device tensor with contiguous rows"
    B, H, S, D = x.shape
    assert idx.is_cuda and idx.device == x.device and tuple(idx.shape[:2]) == (B, H), (idx.shape, x.shape)
    idx = idx.to(torch.int32).contiguous()
    with _on_device(x.device):
        _check(contrib_lib().kvp_zero_channels(_p(x), _st(x, 0), _st(x, 1), _st(x, 2), _DTYPES[x.dtype], B, H, S, D, _p(idx), idx.shape[2], _stream(x)),
               "kvp_zero_channels")
    return x


def rowdot_score(x: torch.Tensor, filt: torch.Tensor, scale: float) -> torch.Tensor:
    """out[b,h,s] = scale * <x[b,h,s,:], filt[h,:]>, float32 [B,H,S]; filt [H,D] in x's dtype (qfilter_press.py:79-82)."""
    x = _rows_last_contig(_dev(x))
    B, H, S, D = x.shape
    filt = _dev(filt)
    assert tuple(filt.shape) == (H, D), (filt.shape, x.shape)
    if filt.dtype != x.dtype or filt.device != x.device:
        filt = filt.to(device=x.device, dtype=x.dtype)
    if filt.stride(-1) != 1:
        filt = filt.contiguous()
    out = torch.empty((B, H, S), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _check(lib().kvp_rowdot_score(_p(x), _DTYPES[x.dtype], B, H, S, D, _st(x, 0), _st(x, 1), _st(x, 2), _p(filt), _st(filt, 0),
                                      float(scale), _p(out), _stream(x)), "kvp_rowdot_score")
    return out


CUR_LEVERAGE = {"key": 0, "value": 1, "kv_avg": 2, "kv_product": 3}


def cur_score(keys: torch.Tensor, values: torch.Tensor, leverage_type: str, local_window_size: int, num_sinks: int) -> torch.Tensor:
    """CUR leverage scores [B,H,S] float32 (local_window_size 0 = no local approximation)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape[:3] == values.shape[:3] and keys.shape[3] == values.shape[3]
    if leverage_type not in CUR_LEVERAGE:
        raise ValueError("Unknown leverage type: choose from 'kv_avg', 'key', 'value' or 'kv_product'")
    B, H, S, D = keys.shape
    scores = torch.empty((B, H, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        ws = _ws(lib().kvp_cur_workspace_bytes(B, H, S), keys)
        _check(lib().kvp_cur_score(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _p(values), _st(values, 0), _st(values, 1),
                                   _st(values, 2), _DTYPES[keys.dtype], B, H, S, D, CUR_LEVERAGE[leverage_type], int(local_window_size),
                                   int(num_sinks), _p(scores), _p(ws), ws.numel(), _stream(keys)), "kvp_cur_score")
    return scores


def keydiff_score(keys: torch.Tensor) -> torch.Tensor:
    """KeyDiff scores [B,H,S] float32: -cos(k, mean of the normalised keys of the head)."""
    keys = _rows_last_contig(_dev(keys))
    B, H, S, D = keys.shape
    scores = torch.empty((B, H, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        nws = lib().kvp_keydiff_workspace_bytes(B, H, S, D)
        ws = _ws(nws, keys)
        _check(lib().kvp_keydiff_score(_p(keys), _DTYPES[keys.dtype], B, H, S, D, _st(keys, 0), _st(keys, 1), _st(keys, 2),
                                       _p(scores), _p(ws), ws.numel(), _stream(keys)), "kvp_keydiff_score")
    return scores


def scores_head_mean_(scores: torch.Tensor) -> torch.Tensor:
    """In place: every head's scores <- the mean over heads (scores float32 [B,H,S], last dim contiguous)."""
    assert scores.dtype == torch.float32 and scores.dim() == 3 and scores.stride(2) == 1
    _dev(scores)
    B, H, S = scores.shape
    with _on_device(scores.device):
        _check(lib().kvp_scores_head_mean(_p(scores), B, H, S, _st(scores, 0), _st(scores, 1), _stream(scores)),
               "kvp_scores_head_mean")
    return scores


def snapkv_score(q_win: torch.Tensor, keys: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """SnapKV scores [B,Hkv,S] float32 from RoPE'd window queries [B,Hq,W,D] and keys [B,Hkv,S,D]."""
    keys = _rows_last_contig(_dev(keys))
    q_win = _rows_last_contig(_dev(q_win))
    if q_win.dtype != keys.dtype:
        q_win = q_win.to(keys.dtype)
    B, Hq, W, D = q_win.shape
    Bk, Hkv, S, Dk = keys.shape
    assert B == Bk and D == Dk and Hq % Hkv == 0, (q_win.shape, keys.shape)
    scores = torch.empty((B, Hkv, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        nws = lib().kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D)
        ws = _ws(nws, keys)
        _check(lib().kvp_snapkv_score(_p(q_win), _st(q_win, 0), _st(q_win, 1), _st(q_win, 2),
                                      _p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _DTYPES[keys.dtype],
                                      B, Hq, Hkv, S, W, D, int(kernel_size), _p(scores), _p(ws), ws.numel(), _stream(keys)),
               "kvp_snapkv_score")
    return scores


def _window_tables(cos: torch.Tensor, sin: torch.Tensor, B: int, W: int, D: int):
    """The window's rotary tables as the library takes them: [1 or B, W, D] with equal strides.  A table with ONE row is
    broadcast over the window (stride 0), which is what the reference's ``cos[:, -W:].unsqueeze(1)`` does when a decoding
    step hands over the current position only (DecodingPress with a window scorer)."""
    assert cos.shape == sin.shape and cos.shape[-1] == D and cos.shape[0] in (1, B) and cos.shape[-2] in (1, W), (cos.shape, (B, W, D))
    if sin.stride() != cos.stride():
        sin, cos = sin.contiguous(), cos.contiguous()
    if cos.shape[-2] == 1 and W > 1:
        cos, sin = cos.expand(-1, W, -1), sin.expand(-1, W, -1)
    return cos, sin


def finch_score(q_pre: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, keys: torch.Tensor, normalize_scores: bool) -> torch.Tensor:
    """FINCH scores (finch_press.py:56-83) from the PRE-RoPE window queries [B,Hq,W,D] (any W) and the window's cos/sin."""
    return snapkv_score_rope(q_pre, cos, sin, keys, 1, _finch_normalize=bool(normalize_scores))


def snapkv_score_rope(q_pre: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, keys: torch.Tensor,
                      kernel_size: int, _finch_normalize=None) -> torch.Tensor:
    """SnapKV scores from the PRE-RoPE window queries [B,Hq,W,D] and the window's cos/sin [1 or B, W, D]:
    the RoPE (q*cos + rotate_half(q)*sin, rounded like torch does in the model dtype) runs in the library."""
    keys = _rows_last_contig(_dev(keys))
    q_pre = _rows_last_contig(_dev(q_pre))
    dt = keys.dtype
    if q_pre.dtype != dt:
        q_pre = q_pre.to(dt)
    cos = _rows_last_contig(_dev(cos.to(dt)))
    sin = _rows_last_contig(_dev(sin.to(dt)))
    B, Hq, W, D = q_pre.shape
    Bk, Hkv, S, Dk = keys.shape
    assert B == Bk and D == Dk and Hq % Hkv == 0, (q_pre.shape, keys.shape)
    cos, sin = _window_tables(cos, sin, B, W, D)
    scores = torch.empty((B, Hkv, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        nws = lib().kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D)
        ws = _ws(nws, keys)
        fn, last, what = ((lib().kvp_snapkv_score_rope, int(kernel_size), "kvp_snapkv_score_rope") if _finch_normalize is None
                          else (lib().kvp_finch_score, int(_finch_normalize), "kvp_finch_score"))
        _check(fn(_p(q_pre), _st(q_pre, 0), _st(q_pre, 1), _st(q_pre, 2), _p(cos), _p(sin), _st(cos, 0), _st(cos, 1), _p(keys),
                  _st(keys, 0), _st(keys, 1), _st(keys, 2), _DTYPES[dt], B, Hq, Hkv, S, W, D, last, _p(scores), _p(ws), ws.numel(),
                  _stream(keys)), what)
    return scores


# The window q_proj + RoPE of the SnapKV-type presses run in the library (qproj.hip) for a plain bias-free bf16 / f16 nn.Linear
# (qproj_rope_eligible): inside bench.py's loop the step is 3-4 us shorter than with the model's own GEMM + the RoPE launch
# (round 4, two boxes, alternating runs: profiles/r04_qproj_lab.txt; rounds 1-2 had measured parity in isolation and kept it off).
# The library sums the eight k-step partials in a fixed order where the GEMM library has its own: 0.01 % of the projected values
# round to the neighbouring 16-bit number (tests/test_gpu_parity.py::test_qproj_rope_kernel_vs_torch).  KVP_LIBRARY_QPROJ=0 (or
# this attribute) keeps the model's own q_proj -- bit-identical queries to the reference's on the same GPU; quantised, LoRA-wrapped,
# biased or q_norm'ed projections always do.
USE_LIBRARY_QPROJ = os.environ.get("KVP_LIBRARY_QPROJ", "1") != "0"


def qproj_rope_supported(module, hidden_states: torch.Tensor, window: int) -> bool:
    """Should the press project the window in the library?  USE_LIBRARY_QPROJ, qproj_rope_eligible, and at most TWO batch elements: the
    kernel streams the q_proj weight once per batch element (its grid is (columns, batch); the second element's slices come from the
    L2s / memory-side cache), a GEMM over all B x 64 rows reads it once.  Measured on the cold weight (profiles/r06_qproj_batch_lab.txt):
    B = 1 22.2 us against GEMM 22.7 + RoPE launch ~5; B = 2 26.7 against 24.3 + 5; B = 4 45.2 against 23.6 + 5 -- so up to two.
    (ChunkPress hands the wrapped press its 128 chunks as a batch of 128 windows: 1239 against 195 us.)"""
    return USE_LIBRARY_QPROJ and hidden_states.shape[0] <= 2 and qproj_rope_eligible(module, hidden_states, window)


_QP_ELIGIBLE: dict = {}   # (id(module), weight data_ptr, weight version, dtype, window) -> the module-side half of the answer


def qproj_rope_eligible(module, hidden_states: torch.Tensor, window: int) -> bool:
    """True if the window's q_proj + RoPE can run in the library (qproj.hip): a plain bias-free ``nn.Linear`` q_proj in
    bf16 / f16 (not a quantised or LoRA-wrapped subclass), no per-head q_norm, window 64, head_dim 128, hidden % 256 == 0."""
    lin = module.__dict__.get("_modules", {}).get("q_proj")
    if type(lin) is torch.nn.Linear and hidden_states.is_cuda and hidden_states.stride(-1) == 1:
        w = lin._parameters.get("weight")
        if _linear_is_hooked(lin):   # (never remembered: hooks come and go)
            return False
        if type(w) in (torch.Tensor, torch.nn.Parameter):   # (the module-side checks are remembered per weight tensor: they cost more than the rest of the call's Python)
            key = (id(module), w.data_ptr(), w._version, hidden_states.dtype, window, lin._parameters.get("bias") is None,
                   "q_norm" in module.__dict__["_modules"])
            hit = _QP_ELIGIBLE.get(key)
            if hit is not None:
                return hit
            ok = _qproj_rope_eligible_slow(module, hidden_states, window)
            if len(_QP_ELIGIBLE) > 512:
                _QP_ELIGIBLE.clear()
            _QP_ELIGIBLE[key] = ok
            return ok
    return _qproj_rope_eligible_slow(module, hidden_states, window)


def _linear_is_hooked(lin) -> bool:
    """The library reads ``q_proj.weight`` directly instead of CALLING q_proj: anything that rides on the call -- forward (pre-)hooks of
    steering / tracing tools, accelerate's ``_hf_hook`` (offloading, device alignment) -- would be skipped, so such a module keeps
    the model's own call."""
    return bool(lin._forward_hooks) or bool(lin._forward_pre_hooks) or hasattr(lin, "_hf_hook")


def _qproj_rope_eligible_slow(module, hidden_states: torch.Tensor, window: int) -> bool:
    lin = getattr(module, "q_proj", None)
    if type(lin) is not torch.nn.Linear or lin.bias is not None or hasattr(module, "q_norm") or _linear_is_hooked(lin):
        return False
    w = lin.weight
    if type(w) not in (torch.Tensor, torch.nn.Parameter):   # tensor subclasses (torchao-quantised, DTensor-sharded ...) keep the nn.Linear type and
        return False                                        # report the plain dtype, but their data_ptr() is not the [out, in] matrix
    return (hidden_states.is_cuda and w.is_cuda and w.dtype == hidden_states.dtype and w.dtype in (torch.bfloat16, torch.float16)
            and w.is_contiguous() and window == 64 and getattr(module, "head_dim", 0) == 128 and w.shape[1] % 256 == 0
            and hidden_states.stride(-1) == 1 and w.shape[0] == module.config.num_attention_heads * 128)


def _hidden_args(hidden_win, wq, cos, sin, dt):
    hidden_win = _dev(hidden_win)
    assert hidden_win.dtype == dt and wq.dtype == dt and hidden_win.stride(-1) == 1 and wq.is_contiguous()
    cos = _rows_last_contig(_dev(cos.to(dt)))
    sin = _rows_last_contig(_dev(sin.to(dt)))
    if sin.stride() != cos.stride():
        sin, cos = sin.contiguous(), cos.contiguous()
    return hidden_win, cos, sin


def snapkv_qproj_rope(hidden_win: torch.Tensor, wq: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, head_dim: int = 128) -> torch.Tensor:
    """RoPE'd window queries [B, Hq, W, D] from hidden_states[:, -W:] and q_proj.weight (bf16 / f16, W = 64, D = 128)."""
    dt = hidden_win.dtype
    hidden_win, cos, sin = _hidden_args(hidden_win, wq, cos, sin, dt)
    B, W, K = hidden_win.shape
    Hq = wq.shape[0] // head_dim
    out = torch.empty((B, Hq, W, head_dim), dtype=dt, device=hidden_win.device)
    with _on_device(hidden_win.device):
        _check(lib().kvp_snapkv_qproj_rope(_p(hidden_win), _st(hidden_win, 0), _st(hidden_win, 1), _p(wq), _p(cos), _p(sin), _st(cos, 0),
                                           _st(cos, 1), _DTYPES[dt], B, Hq, W, head_dim, K, _p(out), _stream(hidden_win)),
               "kvp_snapkv_qproj_rope")
    return out


def snapkv_score_hidden(hidden_win: torch.Tensor, wq: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, keys: torch.Tensor,
                        kernel_size: int) -> torch.Tensor:
    """SnapKV scores [B,Hkv,S] float32 from the window's hidden states and the q_proj weight (projection + RoPE in the library)."""
    keys = _rows_last_contig(_dev(keys))
    dt = keys.dtype
    hidden_win, cos, sin = _hidden_args(hidden_win, wq, cos, sin, dt)
    B, W, K = hidden_win.shape
    Bk, Hkv, S, D = keys.shape
    Hq = wq.shape[0] // D
    assert B == Bk and Hq % Hkv == 0
    scores = torch.empty((B, Hkv, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        ws = _ws(lib().kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D), keys)
        _check(lib().kvp_snapkv_score_hidden(_p(hidden_win), _st(hidden_win, 0), _st(hidden_win, 1), _p(wq), K, _p(cos), _p(sin),
                                             _st(cos, 0), _st(cos, 1), _p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _DTYPES[dt],
                                             B, Hq, Hkv, S, W, D, int(kernel_size), _p(scores), _p(ws), ws.numel(), _stream(keys)),
               "kvp_snapkv_score_hidden")
    return scores


def snapkv_compress_hidden(hidden_win: torch.Tensor, wq: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, keys: torch.Tensor,
                           values: torch.Tensor, kernel_size: int, n_kept: int, order: int = ORDER_POSITION):
    """SnapKVPress.compress (attentions=None) in one library call from the window's hidden states (order: see snapkv_compress_rope)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    dt = keys.dtype
    assert values.dtype == dt and keys.shape == values.shape
    hidden_win, cos, sin = _hidden_args(hidden_win, wq, cos, sin, dt)
    B, W, K = hidden_win.shape
    Bk, Hkv, S, D = keys.shape
    Hq = wq.shape[0] // D
    assert B == Bk and Hq % Hkv == 0
    n = int(n_kept)
    ko = torch.empty((B, Hkv, n, D), dtype=dt, device=keys.device)
    vo = torch.empty_like(ko)
    if n:
        with _on_device(keys.device):
            st = _stream(keys)
            ws = _clean_ws("snapkv", (B, Hq, Hkv, S, W, D, n), _ws_bytes("kvp_snapkv_compress_workspace_bytes", B, Hq, Hkv, S, W, D, n), keys, st)
            rc = lib().kvp_snapkv_compress_hidden(_p(hidden_win), _st(hidden_win, 0), _st(hidden_win, 1), _p(wq), K, _p(cos), _p(sin),
                                                  _st(cos, 0), _st(cos, 1), _p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2),
                                                  _p(values), _st(values, 0), _st(values, 1), _st(values, 2), _DTYPES[dt], B, Hq, Hkv, S,
                                                  W, D, int(kernel_size), n, _p(ko), _p(vo), _p(ws), ws.numel(), TOPK_WS_CLEAN | (int(order) & ORDER_SCORE), st)
            if rc != 0:
                _drop_ws(ws)
            _check(rc, "kvp_snapkv_compress_hidden")
    return ko, vo


def snapkv_score_from_attn(attn_win: torch.Tensor, num_kv_heads: int, k_len: int, kernel_size: int) -> torch.Tensor:
    """Same from given attention weights: attn_win = attentions[..., -W:, :-W]  [B,Hq,W,S-W]."""
    attn_win = _rows_last_contig(_dev(attn_win))
    B, Hq, W, Sm = attn_win.shape
    S = int(k_len)
    assert Sm == S - W, (attn_win.shape, S)
    scores = torch.empty((B, num_kv_heads, S), dtype=torch.float32, device=attn_win.device)
    with _on_device(attn_win.device):
        nws = lib().kvp_snapkv_workspace_bytes(B, Hq, num_kv_heads, S, W, 1)
        ws = _ws(nws, attn_win)
        _check(lib().kvp_snapkv_score_from_attn(_p(attn_win), _st(attn_win, 0), _st(attn_win, 1), _st(attn_win, 2),
                                                _DTYPES[attn_win.dtype], B, Hq, num_kv_heads, S, W, int(kernel_size),
                                                _p(scores), _p(ws), ws.numel(), _stream(attn_win)),
               "kvp_snapkv_score_from_attn")
    return scores


def ea_qstats(q: torch.Tensor, use_covariance: bool = True):
    """mu [B,Hq,D], cov [B,Hq,D,D] (float32) of the queries q [B,Hq,Sq,D]."""
    q = _rows_last_contig(_dev(q))
    B, Hq, Sq, D = q.shape
    mu = torch.empty((B, Hq, D), dtype=torch.float32, device=q.device)
    cov = torch.empty((B, Hq, D, D), dtype=torch.float32, device=q.device) if use_covariance else None
    with _on_device(q.device):
        nws = lib().kvp_ea_qstats_workspace_bytes(B, Hq, Sq, D)
        ws = _ws(nws, q)
        _check(lib().kvp_ea_qstats(_p(q), _st(q, 0), _st(q, 1), _st(q, 2), _DTYPES[q.dtype], B, Hq, Sq, D,
                                   _p(mu), _p(cov), _p(ws), ws.numel(), _stream(q)), "kvp_ea_qstats")
    return mu, cov


def ea_score(keys: torch.Tensor, values: torch.Tensor, mu: torch.Tensor, cov: Optional[torch.Tensor], n_sink: int,
             use_vnorm: bool, epsilon: float) -> torch.Tensor:
    """ExpectedAttention scores [B,Hkv,S] float32 (post-RoPE mu/cov float32 contiguous)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape == values.shape
    B, Hkv, S, D = keys.shape
    mu = mu.to(torch.float32).contiguous()
    cov = cov.to(torch.float32).contiguous() if cov is not None else None
    Hq = mu.shape[1]
    assert mu.shape == (B, Hq, D) and (cov is None or cov.shape == (B, Hq, D, D))
    scores = torch.empty((B, Hkv, S), dtype=torch.float32, device=keys.device)
    with _on_device(keys.device):
        nws = lib().kvp_ea_score_workspace_bytes(B, Hq, Hkv, S, D)
        ws = _ws(nws, keys)
        _check(lib().kvp_ea_score(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2),
                                  _p(values), _st(values, 0), _st(values, 1), _st(values, 2), _DTYPES[keys.dtype],
                                  _p(mu), _p(cov), B, Hq, Hkv, S, D, int(n_sink), int(bool(use_vnorm)), float(epsilon),
                                  _p(scores), _p(ws), ws.numel(), _stream(keys)), "kvp_ea_score")
    return scores


TOPK_SMALLEST = 0x200


def scores_fill_at_(scores: torch.Tensor, idx: torch.Tensor, value: float) -> torch.Tensor:
    """In place: scores[..., idx[..., j]] = value (float32 scores [..., S] contiguous, int32 idx [..., n])."""
    assert scores.dtype == torch.float32 and scores.is_contiguous() and scores.is_cuda
    S = scores.shape[-1]
    idx = idx.to(torch.int32).contiguous()
    n = idx.shape[-1]
    R = scores.numel() // S if S else 0
    assert idx.numel() == R * n
    with _on_device(scores.device):
        _check(lib().kvp_scores_fill_at(_p(scores), R, S, S, _p(idx), n, float(value), _stream(scores)), "kvp_scores_fill_at")
    return scores


def topk_select(scores: torch.Tensor, k: int, order: int = ORDER_POSITION) -> torch.Tensor:
    """Indices (int32 [..., k]) of the k largest scores per row of float32 scores[..., S]; ties -> lowest position.
    ``order | TOPK_SMALLEST`` selects the k smallest instead."""
    if not scores.is_cuda:
        raise KvpressHipError(f"kvpress_amd kernels need tensors on a HIP device, got {scores.device}")
    s = scores.to(torch.float32)
    if s.stride(-1) != 1:
        s = s.contiguous()
    S = s.shape[-1]
    lead = s.shape[:-1]
    s2 = s.reshape(-1, S)  # a view whenever possible
    if s2.stride(-1) != 1 or (s2.shape[0] > 1 and s2.stride(0) < S):
        s2 = s2.contiguous()
    R = s2.shape[0]
    idx = torch.empty((R, k), dtype=torch.int32, device=s.device)
    if R and k:
        with _on_device(s.device):
            stream = torch.cuda.current_stream(s.device)
            if (int(order) & 0xFF) == ORDER_SCORE:
                # descending-score order: the select's self-cleaning region first, then the sort's tiles and samples (never read before
                # they are written): one cached workspace per (device, stream, shape), zero-filled once like the plain select's
                nws = lib().kvp_topk_order_workspace_bytes(R, S, k)
                ws = _cached_ws(("order", s.device.index, stream.cuda_stream, R, S, int(k)), nws, s.device)
            else:
                nws = lib().kvp_topk_workspace_bytes(R, S, k)
                # self-cleaning workspace: zeroed once per (device, stream, size), reused with KVP_TOPK_WS_CLEAN
                key = (s.device.index, stream.cuda_stream, R, S)  # the layout (hence the clean region) depends on R and S
                ws = _cached_ws(key, nws, s.device)
            try:
                _check(lib().kvp_topk_select(_p(s2), R, S, s2.stride(0) if R > 1 else S, int(k), int(order) | TOPK_WS_CLEAN, _p(idx),
                                             _p(ws), ws.numel(), _stream(s)), "kvp_topk_select")
            except KvpressHipError:
                _drop_ws(ws)  # a failed call may leave dirty histograms behind: never reuse this workspace as "clean"
                raise
    return idx.reshape(*lead, k)


def topk_select_segmented(scores: torch.Tensor, seg_len: int, k: int, pos_base: int = 0) -> torch.Tensor:
    """Per-chunk top-k: scores float32 [..., nseg * seg_len]; returns int32 [..., nseg * k] row positions (+ pos_base),
    chunk after chunk (ascending)."""
    if not scores.is_cuda:
        raise KvpressHipError(f"kvpress_amd kernels need tensors on a HIP device, got {scores.device}")
    s = scores.to(torch.float32).contiguous()
    L = s.shape[-1]
    assert L % seg_len == 0, (L, seg_len)
    nseg = L // seg_len
    lead = s.shape[:-1]
    s2 = s.reshape(-1, L)
    R = s2.shape[0]
    idx = torch.empty((R, nseg * k), dtype=torch.int32, device=s.device)
    if R and k:
        with _on_device(s.device):
            if seg_len <= 32768:
                # chunks this short are selected by one workgroup each (topk_row_kernel): no workspace at all.  (Round 6: a fresh
                # zero-filled 33 MB workspace per call was 11 us of ChunkPress's 128k step -- profiles/r06_rocprofv3_kernel_stats_chunk_snapkv128k.csv)
                _check(lib().kvp_topk_select_segmented(_p(s2), R, nseg, int(seg_len), int(k), int(pos_base), 0, _p(idx), None, 0, _stream(s)),
                       "kvp_topk_select_segmented")
            else:
                stream = torch.cuda.current_stream(s.device)
                nws = max(int(lib().kvp_topk_segmented_workspace_bytes(R, nseg, seg_len, k)), 256)
                ws = _cached_ws(("seg", s.device.index, stream.cuda_stream, R, nseg, int(seg_len)), nws, s.device)   # zeroed once, self-cleaning
                try:
                    _check(lib().kvp_topk_select_segmented(_p(s2), R, nseg, int(seg_len), int(k), int(pos_base), TOPK_WS_CLEAN, _p(idx), _p(ws),
                                                           ws.numel(), _stream(s)), "kvp_topk_select_segmented")
                except KvpressHipError:
                    _drop_ws(ws)
                    raise
    return idx.reshape(*lead, nseg * k)


def rerotate_keys_(keys_kept: torch.Tensor, idx: torch.Tensor, inv_freq: torch.Tensor) -> torch.Tensor:
    """In place: re-rotate gathered keys [B,H,n,D] (kept at ascending positions idx [B,H,n]) to positions 0..n-1."""
    _dev(keys_kept)
    assert keys_kept.is_contiguous()
    B, H, n, D = keys_kept.shape
    idx = idx.to(torch.int32).contiguous()
    assert idx.shape == (B, H, n)
    inv = inv_freq.to(device=keys_kept.device, dtype=torch.float32).contiguous()
    assert inv.numel() == D // 2
    with _on_device(keys_kept.device):
        _check(lib().kvp_rerotate_keys(_p(keys_kept), _DTYPES[keys_kept.dtype], B, H, n, D, _p(idx), _p(inv), _stream(keys_kept)),
               "kvp_rerotate_keys")
    return keys_kept


_WS_BYTES: dict = {}


def _ws_bytes(fn_name: str, *shape) -> int:
    """kvp_*_workspace_bytes(shape), remembered per shape (a ctypes call per compress is measurable in the decode regime)."""
    key = (fn_name,) + shape
    n = _WS_BYTES.get(key)
    if n is None:
        if len(_WS_BYTES) > 256:
            _WS_BYTES.clear()
        n = _WS_BYTES[key] = int(getattr(lib(), fn_name)(*shape))
    return n


def _clean_ws(kind: str, shape: tuple, nbytes: int, like: torch.Tensor, stream_handle: Optional[int] = None) -> torch.Tensor:
    """Zero-filled-once workspace of a fused compress call, one per (device, stream, call shape): the calls leave its
    histogram region clean, so it is passed with KVP_TOPK_WS_CLEAN from then on."""
    if stream_handle is None:
        stream_handle = torch.cuda.current_stream(like.device).cuda_stream
    key = (kind, like.device.index, stream_handle) + tuple(shape)
    return _cached_ws(key, nbytes, like.device)


def _cached_ws(key: tuple, nbytes: int, device) -> torch.Tensor:
    """One zero-filled workspace per (calling thread, key): two Python threads never share one, even on the same stream."""
    key = (threading.get_ident(),) + tuple(key)
    with _TOPK_WS_LOCK:
        ws = _TOPK_WS.get(key)
        if ws is None:
            if len(_TOPK_WS) > 64:
                _TOPK_WS.clear()
            ws = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            _TOPK_WS[key] = ws
    return ws


def _drop_ws(ws: torch.Tensor):
    with _TOPK_WS_LOCK:
        for key in [k for k, v in _TOPK_WS.items() if v is ws]:
            del _TOPK_WS[key]


def knorm_compress(keys: torch.Tensor, values: torch.Tensor, n_kept: int, order: int = ORDER_POSITION):
    """KnormPress.compress in one library call: (K', V') contiguous [B,H,n_kept,D]; order = ORDER_SCORE stores the rows in descending
    score order (the reference's layout)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape == values.shape
    B, H, S, D = keys.shape
    n = int(n_kept)
    ko = torch.empty((B, H, n, D), dtype=keys.dtype, device=keys.device)
    vo = torch.empty_like(ko)
    if n and B and H:
        with _on_device(keys.device):
            st = _stream(keys)
            ws = _clean_ws("knorm", (B, H, S, n), _ws_bytes("kvp_knorm_compress_workspace_bytes", B, H, S, n), keys, st)
            rc = lib().kvp_knorm_compress(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _p(values), _st(values, 0), _st(values, 1),
                                          _st(values, 2), _DTYPES[keys.dtype], B, H, S, D, n, _p(ko), _p(vo), _p(ws), ws.numel(),
                                          TOPK_WS_CLEAN | (int(order) & ORDER_SCORE), st)
            if rc != 0:
                _drop_ws(ws)
            _check(rc, "kvp_knorm_compress")
    return ko, vo


def snapkv_compress_rope(q_pre: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                         kernel_size: int, n_kept: int, order: int = ORDER_POSITION):
    """SnapKVPress.compress (attentions=None) in one library call from the pre-RoPE window queries; order = ORDER_SCORE stores the rows
    in descending score order (window tokens first: the reference's layout)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    q_pre = _rows_last_contig(_dev(q_pre))
    dt = keys.dtype
    assert values.dtype == dt and keys.shape == values.shape
    if q_pre.dtype != dt:
        q_pre = q_pre.to(dt)
    cos = _rows_last_contig(_dev(cos.to(dt)))
    sin = _rows_last_contig(_dev(sin.to(dt)))
    B, Hq, W, D = q_pre.shape
    Bk, Hkv, S, Dk = keys.shape
    assert B == Bk and D == Dk and Hq % Hkv == 0, (q_pre.shape, keys.shape)
    cos, sin = _window_tables(cos, sin, B, W, D)
    n = int(n_kept)
    ko = torch.empty((B, Hkv, n, D), dtype=dt, device=keys.device)
    vo = torch.empty_like(ko)
    if n:
        with _on_device(keys.device):
            st = _stream(keys)
            ws = _clean_ws("snapkv", (B, Hq, Hkv, S, W, D, n), _ws_bytes("kvp_snapkv_compress_workspace_bytes", B, Hq, Hkv, S, W, D, n), keys, st)
            rc = lib().kvp_snapkv_compress_rope(_p(q_pre), _st(q_pre, 0), _st(q_pre, 1), _st(q_pre, 2), _p(cos), _p(sin), _st(cos, 0),
                                                _st(cos, 1), _p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2), _p(values),
                                                _st(values, 0), _st(values, 1), _st(values, 2), _DTYPES[dt], B, Hq, Hkv, S, W, D,
                                                int(kernel_size), n, _p(ko), _p(vo), _p(ws), ws.numel(), TOPK_WS_CLEAN | (int(order) & ORDER_SCORE), st)
            if rc != 0:
                _drop_ws(ws)
            _check(rc, "kvp_snapkv_compress_rope")
    return ko, vo


def gather_kv(keys: torch.Tensor, values: torch.Tensor, idx: torch.Tensor):
    """K'[b,h,j] = K[b,h,idx[b,h,j]] (and V'), contiguous [B,H,n,D] in the input dtype."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape == values.shape
    B, H, S, D = keys.shape
    idx = idx.to(torch.int32).contiguous()
    n = idx.shape[-1]
    assert idx.shape == (B, H, n)
    ko = torch.empty((B, H, n, D), dtype=keys.dtype, device=keys.device)
    vo = torch.empty((B, H, n, D), dtype=values.dtype, device=values.device)
    if B * H * n:
        with _on_device(keys.device):
            _check(lib().kvp_gather_kv(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2),
                                       _p(values), _st(values, 0), _st(values, 1), _st(values, 2), _DTYPES[keys.dtype],
                                       B, H, S, D, _p(idx), n, _p(ko), _p(vo), _stream(keys)), "kvp_gather_kv")
    return ko, vo


def gather_kv_rerotate(keys: torch.Tensor, values: torch.Tensor, idx: torch.Tensor, inv_freq: torch.Tensor):
    """gather_kv + rerotate_keys_ in one pass: K'[b,h,j] = K[b,h,idx[b,h,j]] re-rotated from position idx[b,h,j] to position j, V' the
    plain gather (same bits as the two calls)."""
    keys = _rows_last_contig(_dev(keys))
    values = _rows_last_contig(_dev(values))
    assert keys.dtype == values.dtype and keys.shape == values.shape
    B, H, S, D = keys.shape
    idx = idx.to(torch.int32).contiguous()
    n = idx.shape[-1]
    assert idx.shape == (B, H, n)
    inv = inv_freq.to(device=keys.device, dtype=torch.float32).contiguous()
    assert inv.numel() == D // 2
    ko = torch.empty((B, H, n, D), dtype=keys.dtype, device=keys.device)
    vo = torch.empty((B, H, n, D), dtype=values.dtype, device=values.device)
    if B * H * n:
        with _on_device(keys.device):
            _check(lib().kvp_gather_kv_rerotate(_p(keys), _st(keys, 0), _st(keys, 1), _st(keys, 2),
                                                _p(values), _st(values, 0), _st(values, 1), _st(values, 2), _DTYPES[keys.dtype],
                                                B, H, S, D, _p(idx), n, _p(inv), _p(ko), _p(vo), _stream(keys)), "kvp_gather_kv_rerotate")
    return ko, vo


# ------------------------------------------------------------------------------------------------
def clock_probe(device=None, spin_us: int = 20) -> torch.Tensor:
    """Enqueue a shader-clock probe on the current stream; returns a 1-element float32 device tensor (MHz) that is valid
    once the stream has run it."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(lib().kvp_clock_probe(_p(out), int(spin_us), _stream(out)), "kvp_clock_probe")
    return out


def occupy_cus(blocks: int, threads: int, lds_bytes: int, spin_us: int, stream=None) -> None:
    """TEST AID: enqueue `blocks` workgroups that hold `threads` threads + `lds_bytes` of LDS for `spin_us` microseconds on
    `stream` (default: the current one)."""
    st = stream if stream is not None else torch.cuda.current_stream()
    _check(lib().kvp_occupy_cus(int(blocks), int(threads), int(lds_bytes), int(spin_us), c_void_p(st.cuda_stream)), "kvp_occupy_cus")


def prof_enable(on: bool) -> None:
    """Turn the library's per-kernel HIP-event timing on/off (clears previous records)."""
    _check(lib().kvp_prof_enable(int(bool(on))), "kvp_prof_enable")


def prof_records():
    """[(kernel name, milliseconds)] recorded since prof_enable(True); synchronises on each event."""
    out = []
    name, ms = c_char_p(), c_float()
    for i in range(lib().kvp_prof_count()):
        _check(lib().kvp_prof_get(i, ctypes.byref(name), ctypes.byref(ms)), "kvp_prof_get")
        out.append((name.value.decode(), float(ms.value)))
    return out


def tuning_reload() -> None:
    """Make the library re-read its KVP_* tuning variables (they are read from the environment once and cached)."""
    _check(lib().kvp_tuning_reload(), "kvp_tuning_reload")
