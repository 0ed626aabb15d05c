"""Seeded synthetic inputs shared by oracle/gen_golden.py (runs the real reference in the build
container) and the tests (run here and on the GPU box, where /root/reference does not exist).

Everything is generated with numpy's frozen legacy ``RandomState`` so that both machines
reproduce the same bits.  Arrays are float32; for the half dtypes every value is rounded
(RNE) to that dtype first, so ``.to(bf16)`` / ``.to(fp16)`` is exact and the fp32-mode
reference ("O32") sees exactly the numbers the kernels see.
"""
from __future__ import annotations

import math

import numpy as np

RATIOS = (0.2, 0.5, 0.7, 0.8)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round float32 values to bf16/f16 (RNE) and return them as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype == "f32":
        return x
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    raise ValueError(dtype)


# name -> spec.  kind: knorm | snapkv | ea.   data: "A" flat N(0,1); "B" structured
# (per-channel key scales, heavy sink rows, log-normal value norms) -- SURVEY.md §8(d).
CASES = {
    # ---- KnormPress -------------------------------------------------------------------
    "kn_tiny_d6": dict(kind="knorm", B=5, H=2, G=1, S=256, D=6, dtype="f32", data="A", seed=11,
                       ratios=(0.1, 0.2, 0.4, 0.6, 0.8)),   # [5,2,230,6] of tests/test_per_layer_compression_press.py:19
    "kn_readme": dict(kind="knorm", B=3, H=8, G=1, S=5, D=128, dtype="f32", data="A", seed=12,
                      ratios=(0.4,)),                        # README.md:253-258  [3,8,5,128]->[3,8,3,128]
    "kn_opt_geom": dict(kind="knorm", B=1, H=12, G=1, S=2048, D=64, dtype="f32", data="A", seed=13,
                        ratios=(0.5,)),                      # BASELINE config 1 geometry
    "kn_bf16_A": dict(kind="knorm", B=2, H=8, G=4, S=4096, D=128, dtype="bf16", data="A", seed=14),
    "kn_bf16_B": dict(kind="knorm", B=1, H=8, G=4, S=3001, D=128, dtype="bf16", data="B", seed=15),
    "kn_f16_ragged": dict(kind="knorm", B=1, H=4, G=1, S=1000, D=128, dtype="f16", data="A", seed=16),
    "kn_d96_bf16": dict(kind="knorm", B=2, H=3, G=1, S=515, D=96, dtype="bf16", data="B", seed=17),
    # ---- SnapKVPress ------------------------------------------------------------------
    "sk_tiny": dict(kind="snapkv", B=2, H=2, G=2, S=100, D=16, dtype="f32", data="A", seed=21, W=8, ks=5),
    "sk_w2_d6": dict(kind="snapkv", B=1, H=2, G=1, S=128, D=6, dtype="f32", data="A", seed=22, W=2, ks=5),
    "sk_s65": dict(kind="snapkv", B=1, H=2, G=4, S=65, D=128, dtype="bf16", data="A", seed=23, W=64, ks=5),
    "sk_257_A": dict(kind="snapkv", B=2, H=2, G=4, S=257, D=128, dtype="bf16", data="A", seed=24, W=64, ks=5),
    "sk_257_B": dict(kind="snapkv", B=1, H=2, G=4, S=257, D=128, dtype="bf16", data="B", seed=25, W=64, ks=5),
    "sk_4096": dict(kind="snapkv", B=1, H=2, G=4, S=4096, D=128, dtype="bf16", data="B", seed=26, W=64, ks=5),
    "sk_f16_d64": dict(kind="snapkv", B=1, H=2, G=2, S=1000, D=64, dtype="f16", data="A", seed=27, W=32, ks=7),
    "sk_ks1": dict(kind="snapkv", B=1, H=1, G=4, S=300, D=128, dtype="bf16", data="A", seed=28, W=64, ks=1),
    "sk_f32_d128": dict(kind="snapkv", B=1, H=2, G=4, S=700, D=128, dtype="f32", data="B", seed=29, W=64, ks=3),
    # wider hidden states: the library's own window q_proj + RoPE kernel applies (hidden % 256 == 0)
    "sk_h512_bf16": dict(kind="snapkv", B=2, H=2, G=4, S=300, D=128, dtype="bf16", data="B", seed=30, W=64, ks=5, hsz=512),
    "sk_h1024_f16": dict(kind="snapkv", B=1, H=1, G=8, S=1000, D=128, dtype="f16", data="A", seed=40, W=64, ks=5, hsz=1024),
    # round 6: windows that are not 64 rows and the other head sizes of the reference's supported models on the MFMA passes
    "sk_w100_d96": dict(kind="snapkv", B=1, H=2, G=4, S=700, D=96, dtype="bf16", data="B", seed=141, W=100, ks=5),
    "sk_w200_d256": dict(kind="snapkv", B=1, H=1, G=2, S=1100, D=256, dtype="bf16", data="A", seed=142, W=200, ks=3),
    "sk_w7_d64_bf16": dict(kind="snapkv", B=2, H=2, G=4, S=900, D=64, dtype="bf16", data="B", seed=143, W=7, ks=5),
    "sk_w130_d128": dict(kind="snapkv", B=1, H=2, G=4, S=2000, D=128, dtype="bf16", data="B", seed=144, W=130, ks=7),
    # ---- ExpectedAttentionPress --------------------------------------------------------
    "ea_tiny": dict(kind="ea", B=2, H=2, G=2, S=100, D=16, dtype="f32", data="A", seed=31),
    "ea_23": dict(kind="ea", B=1, H=2, G=1, S=23, D=16, dtype="f32", data="A", seed=32,
                  ratios=(0.4,)),                            # tests/test_pipeline.py:31-32  23 -> 13
    "ea_257_A": dict(kind="ea", B=2, H=2, G=4, S=257, D=128, dtype="bf16", data="A", seed=33),
    "ea_1500_B": dict(kind="ea", B=1, H=2, G=4, S=1500, D=128, dtype="bf16", data="B", seed=34),
    "ea_nocov": dict(kind="ea", B=1, H=2, G=4, S=400, D=128, dtype="bf16", data="B", seed=35, use_covariance=False),
    "ea_novnorm_eps": dict(kind="ea", B=1, H=2, G=2, S=333, D=64, dtype="f16", data="A", seed=36,
                           use_vnorm=False, epsilon=0.0, n_future=64),
    "ea_eps_sink0": dict(kind="ea", B=1, H=1, G=4, S=512, D=128, dtype="bf16", data="A", seed=37,
                         epsilon=0.01, n_sink=0),
    "ea_f32_d128": dict(kind="ea", B=1, H=2, G=4, S=600, D=128, dtype="f32", data="B", seed=38),
    "ea_6000_B": dict(kind="ea", B=1, H=2, G=4, S=6000, D=128, dtype="bf16", data="B", seed=39),  # MFMA statistics path
    # ---- KeyDiff / TOVA / PyramidKV / StreamingLLM -------------------------
    "kd_tiny_d6": dict(kind="keydiff", B=2, H=2, G=1, S=100, D=6, dtype="f32", data="A", seed=41),
    "kd_bf16_A": dict(kind="keydiff", B=2, H=8, G=1, S=4096, D=128, dtype="bf16", data="A", seed=42),
    "kd_bf16_B": dict(kind="keydiff", B=1, H=8, G=1, S=3001, D=128, dtype="bf16", data="B", seed=43),
    "kd_f16_d64": dict(kind="keydiff", B=1, H=4, G=1, S=1000, D=64, dtype="f16", data="A", seed=44),
    "kd_d96_bf16": dict(kind="keydiff", B=2, H=3, G=1, S=515, D=96, dtype="bf16", data="B", seed=45),
    "cur_tiny_d6": dict(kind="cur", B=2, H=2, G=1, S=100, D=6, dtype="f32", data="A", seed=46, leverage="kv_product"),
    "cur_bf16_B": dict(kind="cur", B=1, H=8, G=1, S=3001, D=128, dtype="bf16", data="B", seed=47, leverage="kv_product"),
    "cur_key_nolocal": dict(kind="cur", B=1, H=4, G=1, S=1000, D=64, dtype="f16", data="A", seed=48, leverage="key", local=False, sinks=0),
    "cur_kvavg_w7": dict(kind="cur", B=2, H=3, G=1, S=515, D=96, dtype="bf16", data="B", seed=49, leverage="kv_avg", window=7),
    "tv_tiny": dict(kind="tova", B=2, H=2, G=2, S=100, D=16, dtype="f32", data="A", seed=51, W=1, ks=1),
    "tv_257": dict(kind="tova", B=2, H=2, G=4, S=257, D=128, dtype="bf16", data="A", seed=52, W=1, ks=1),
    "tv_4096_B": dict(kind="tova", B=1, H=2, G=4, S=4096, D=128, dtype="bf16", data="B", seed=53, W=1, ks=1),
    "py_l0": dict(kind="pyramid", B=1, H=2, G=2, S=400, D=16, dtype="f32", data="A", seed=61, W=8, ks=5,
                  n_layers=8, layer_idx=0, beta=20, ratios=(0.3, 0.5, 0.8)),
    "py_l5": dict(kind="pyramid", B=1, H=2, G=2, S=400, D=16, dtype="f32", data="B", seed=62, W=8, ks=5,
                  n_layers=8, layer_idx=5, beta=20, ratios=(0.3, 0.5, 0.8)),
    "py_bf16_l7": dict(kind="pyramid", B=1, H=2, G=4, S=1000, D=128, dtype="bf16", data="A", seed=63, W=64, ks=5,
                       n_layers=8, layer_idx=7, beta=5, ratios=(0.5, 0.9)),
    "qf_tiny_d6": dict(kind="qfilter", B=2, H=2, G=1, S=100, D=6, dtype="f32", data="A", seed=111),
    "qf_bf16_B": dict(kind="qfilter", B=1, H=8, G=1, S=3001, D=128, dtype="bf16", data="B", seed=112),
    "qf_f16_d64": dict(kind="qfilter", B=2, H=4, G=1, S=1000, D=64, dtype="f16", data="A", seed=113),
    "qf_d96_f32": dict(kind="qfilter", B=1, H=3, G=1, S=515, D=96, dtype="f32", data="B", seed=114),
    "lag_tiny": dict(kind="lagkv", B=2, H=2, G=1, S=100, D=6, dtype="f32", data="A", seed=131, n_sink=4, lag=16),
    "lag_short": dict(kind="lagkv", B=1, H=2, G=1, S=30, D=6, dtype="f32", data="A", seed=132, n_sink=4, lag=16),
    "lag_bf16": dict(kind="lagkv", B=1, H=4, G=1, S=1100, D=128, dtype="bf16", data="B", seed=133, n_sink=4, lag=128),
    "lag_cross_f16": dict(kind="lagkv", B=1, H=2, G=1, S=515, D=64, dtype="f16", data="A", seed=134, n_sink=16, lag=64, cross=True),
    "lag_d96_f32": dict(kind="lagkv", B=2, H=3, G=1, S=400, D=96, dtype="f32", data="B", seed=135, n_sink=0, lag=50),
    "oa_tiny": dict(kind="observed", B=2, H=2, G=2, S=100, D=8, dtype="f32", data="A", seed=121),
    "oa_bf16": dict(kind="observed", B=1, H=2, G=4, S=700, D=8, dtype="bf16", data="A", seed=122),
    "oa_f16_g1": dict(kind="observed", B=1, H=4, G=1, S=257, D=8, dtype="f16", data="A", seed=123),
    "st_tiny": dict(kind="streaming", B=1, H=2, G=1, S=100, D=8, dtype="f32", data="A", seed=71),
    "st_sink0": dict(kind="streaming", B=2, H=2, G=1, S=257, D=8, dtype="f32", data="A", seed=72, n_sink=0),
}

_DEFAULTS = dict(W=64, ks=5, n_future=512, n_sink=4, use_covariance=True, use_vnorm=True, epsilon=0.0,
                 ratios=RATIOS)


def spec(name: str) -> dict:
    s = dict(_DEFAULTS)
    s.update(CASES[name])
    s["name"] = name
    s["Hq"] = s["H"] * s["G"]
    s.setdefault("hsz", 16 * s["Hq"])  # LlamaConfig wants hidden_size % num_heads == 0
    return s


def make_kv(B, H, S, D, dtype, data, seed):
    """K and V [B,H,S,D] float32 (values exactly representable in ``dtype``)."""
    rs = np.random.RandomState(seed)
    k = rs.standard_normal((B, H, S, D)).astype(np.float32)
    v = rs.standard_normal((B, H, S, D)).astype(np.float32)
    if data == "B":
        ch = np.exp(0.5 * rs.standard_normal((1, H, 1, D))).astype(np.float32)
        k = k * ch
        k[:, :, : min(4, S)] *= 8.0  # sink rows
        tok = np.exp(0.7 * rs.standard_normal((B, H, S, 1))).astype(np.float32)
        v = v * tok
    return round_to(k, dtype), round_to(v, dtype)


def make_case(name: str) -> dict:
    """All numpy inputs of a case: keys, values, hidden, wq (q_proj weight [Hq*D, hidden])."""
    s = spec(name)
    k, v = make_kv(s["B"], s["H"], s["S"], s["D"], s["dtype"], s["data"], s["seed"])
    rs = np.random.RandomState(s["seed"] + 1000)
    hid = rs.standard_normal((s["B"], s["S"], s["hsz"])).astype(np.float32)
    wq = (rs.standard_normal((s["Hq"] * s["D"], s["hsz"])) * (1.5 / math.sqrt(s["hsz"]))).astype(np.float32)
    if s["data"] == "B":
        # a non-zero query mean and a few dominant channels (exercises EA's centring)
        hid = hid + 0.5
        wq[:: max(1, s["D"] // 4)] *= 3.0
    s.update(keys=k, values=v, hidden=round_to(hid, s["dtype"]), wq=round_to(wq, s["dtype"]))
    return s


QF_LAYERS, QF_LAYER = 3, 1   # the Q-filter cases: filters for 3 layers, the module under test is layer 1


def make_qfilters(s: dict) -> np.ndarray:
    """Stand-in for the learned Q-filters [num_layers, H, D] (the published ones need the hub), exact in the case dtype."""
    rs = np.random.RandomState(s["seed"] + 2000)
    return round_to(rs.standard_normal((QF_LAYERS, s["H"], s["D"])).astype(np.float32), s["dtype"])


def make_attentions(s: dict) -> np.ndarray:
    """Causal attention weights [B,Hq,S,S] for the ObservedAttention cases (what an eager attention layer returns), exact in
    the case dtype."""
    rs = np.random.RandomState(s["seed"] + 3000)
    S = s["S"]
    logits = 2.0 * rs.standard_normal((s["B"], s["Hq"], S, S)).astype(np.float32)
    logits = np.where(np.triu(np.ones((S, S), bool), 1), -np.inf, logits)
    p = np.exp(logits - logits.max(-1, keepdims=True))
    return round_to((p / p.sum(-1, keepdims=True)).astype(np.float32), s["dtype"])


def assert_lag_scores_close(got, ref, s: dict, what: str = ""):
    """LagKV scores: raw softmax scores (cross_scoring) compare numerically; rank scores are k / lag_size and two
    near-equal tokens may swap neighbouring ranks between float32 and float64 arithmetic -- allowed for < 1 % of the tokens
    and by at most two ranks."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    if s.get("cross", False) or s["S"] < s["n_sink"] + 2 * s["lag"]:
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-7, err_msg=what)
        return
    diff = np.abs(got - ref)
    assert diff.max() <= 2.0 / s["lag"] + 1e-6, f"{what}: rank differs by {diff.max() * s['lag']:.1f}"
    assert np.mean(diff > 1e-6) < 0.01, f"{what}: {np.mean(diff > 1e-6):.3%} of the ranks differ"


def torch_dtype(name: str):
    import torch

    return {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[name]


def build_llama_attention(s: dict, dtype, device="cpu"):
    """A random-geometry ``LlamaAttention`` whose q_proj carries ``s['wq']``, plus the model-level
    rotary embedding and cos/sin for positions 0..S-1 (what transformers hands the hook as
    ``position_embeddings``).  ``dtype`` is the *module* dtype (float32 for the O32 oracle)."""
    import torch
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaAttention, LlamaRotaryEmbedding

    cfg = LlamaConfig(
        hidden_size=s["hsz"], num_attention_heads=s["Hq"], num_key_value_heads=s["H"], head_dim=s["D"],
        num_hidden_layers=1, intermediate_size=32, vocab_size=32, max_position_embeddings=max(4096, 2 * s["S"]),
        attention_bias=False,
    )
    cfg._attn_implementation = "eager"
    att = LlamaAttention(cfg, layer_idx=0)
    with torch.no_grad():
        att.q_proj.weight.copy_(torch.from_numpy(s["wq"]))
        # the other projections are seeded too (CriticalKV reads o_proj): module creation order must not matter
        rs = np.random.RandomState(s["seed"] + 4000)
        wo = (rs.standard_normal(tuple(att.o_proj.weight.shape)) * (1.0 / math.sqrt(att.o_proj.weight.shape[1]))).astype(np.float32)
        att.o_proj.weight.copy_(torch.from_numpy(round_to(wo, s["dtype"])))
    rot = LlamaRotaryEmbedding(cfg)
    att = att.to(device=device, dtype=dtype)
    rot = rot.to(device)
    att.rotary_emb = rot  # BasePress.__call__ attaches this (base_press.py:202)
    hidden = torch.from_numpy(s["hidden"]).to(device=device, dtype=dtype)
    pos = torch.arange(s["S"], device=device)[None]
    cos, sin = rot(hidden, pos)
    if "n_layers" in s:  # PyramidKV's budget reads the layer's place in the stack (pyramidkv_press.py:80-81)
        att.config.num_hidden_layers = s["n_layers"]
        att.layer_idx = s["layer_idx"]
    return att, rot, hidden, (cos, sin)


# ---- tiny end-to-end fixtures (pipeline tests) -----------------------------------------------------
TINY_WORDS = [f"w{i}" for i in range(56)]


def make_tiny_llama(seed: int = 0, dtype=None, device="cpu"):
    """The reference's unit-test geometry (2 layers, 2 KV heads, head_dim 6), random-init, eos = 2."""
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=24, num_attention_heads=4, num_key_value_heads=2, head_dim=6, num_hidden_layers=2,
                      intermediate_size=32, vocab_size=64, max_position_embeddings=512, bos_token_id=1, eos_token_id=2,
                      pad_token_id=0)
    torch.manual_seed(seed)
    model = LlamaForCausalLM(cfg).eval()
    if dtype is not None:
        model = model.to(dtype)
    return model.to(device)


def make_tiny_model(family: str, seed: int = 0, dtype=None, device="cpu"):
    """The same tiny geometry for the other model families the reference lists as supported (base_press.py:27-34):
    ``qwen3`` (per-head q_norm / k_norm), ``phi3`` (fused qkv_proj), ``mistral``, ``qwen2`` (q_proj with bias)."""
    import torch
    import transformers as T

    common = dict(hidden_size=24, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2, intermediate_size=48,
                  vocab_size=64, max_position_embeddings=512, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    if family == "llama":
        return make_tiny_llama(seed, dtype, device)
    if family == "llama_eager":   # the attention layers return their weights (ObservedAttentionPress)
        model = make_tiny_llama(seed, dtype, device)
        model.config._attn_implementation = "eager"
        return model
    if family == "qwen3":
        cfg, cls = T.Qwen3Config(head_dim=6, **common), T.Qwen3ForCausalLM
    elif family == "qwen2":
        cfg, cls = T.Qwen2Config(**common), T.Qwen2ForCausalLM
    elif family == "mistral":
        cfg, cls = T.MistralConfig(head_dim=6, sliding_window=None, **common), T.MistralForCausalLM
    elif family == "phi3":
        cfg, cls = T.Phi3Config(**common), T.Phi3ForCausalLM
    else:
        raise ValueError(family)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(seed)
    model = cls(cfg).eval()
    # random init leaves norm weights at 1 and biases at 0: perturb them so that q_norm / biases actually matter
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name.endswith(("q_norm.weight", "k_norm.weight")):
                prm.mul_(1.0 + 0.5 * torch.rand(prm.shape, generator=g))
            elif name.endswith(("q_proj.bias", "k_proj.bias", "v_proj.bias")):
                prm.add_(0.3 * torch.randn(prm.shape, generator=g))
    if dtype is not None:
        model = model.to(dtype)
    return model.to(device)


# other model families through the pipeline: name -> (family, press spec, context words, questions, max_new_tokens)
FAMILY_PIPELINE_CASES = {
    "pipe_qwen3_snapkv": ("qwen3", ("SnapKVPress", dict(compression_ratio=0.5, window_size=16, kernel_size=5)), 150, ["w4 w5"], 8),
    "pipe_qwen3_ea": ("qwen3", ("ExpectedAttentionPress", dict(compression_ratio=0.4)), 60, ["w9 w10"], 6),
    "pipe_phi3_snapkv": ("phi3", ("SnapKVPress", dict(compression_ratio=0.5, window_size=16, kernel_size=5)), 130, ["w1 w2"], 8),
    "pipe_phi3_ea": ("phi3", ("ExpectedAttentionPress", dict(compression_ratio=0.5)), 70, ["w3"], 6),
    "pipe_eager_observed": ("llama_eager", ("ObservedAttentionPress", dict(compression_ratio=0.5)), 90, ["w2 w3", "w6"], 6),
    "pipe_eager_snapkv_attn": ("llama_eager", ("SnapKVPress", dict(compression_ratio=0.5, window_size=8, kernel_size=5)), 90, ["w2 w3"], 6),
    "pipe_mistral_knorm": ("mistral", ("KnormPress", dict(compression_ratio=0.5)), 100, ["w7 w8", "w1"], 6),
    "pipe_qwen2_tova": ("qwen2", ("TOVAPress", dict(compression_ratio=0.5)), 90, ["w2 w8"], 6),
    "pipe_qwen2_snapkv": ("qwen2", ("SnapKVPress", dict(compression_ratio=0.3, window_size=8, kernel_size=3)), 90, ["w5"], 6),
}


def make_tiny_tokenizer():
    """Word-level tokenizer built in memory (no download): <unk>=0 <s>=1 </s>=2, then w0..w55; no chat template."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in TINY_WORDS:
        vocab[w] = len(vocab)
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                                   model_max_length=512)


def tiny_context(n_words: int, seed: int = 0) -> str:
    rng = np.random.default_rng(seed)
    return " ".join(TINY_WORDS[i] for i in rng.integers(0, len(TINY_WORDS), n_words))


def press_class(ns, name):
    """class `name` of namespace `ns`; for this package also its contrib sub-package (presses outside SURVEY §8)"""
    if hasattr(ns, name):
        return getattr(ns, name)
    import importlib

    return getattr(importlib.import_module(ns.__name__ + ".contrib"), name)


def build_press(ns, spec):
    """Instantiate a press from a nested (class name, kwargs) spec in namespace ``ns`` -- the reference package ``kvpress``
    or this package ``kvpress_amd``: the class names and constructor arguments are the same (that is the drop-in claim)."""
    if spec is None:
        return None
    if isinstance(spec, list):
        return [build_press(ns, x) for x in spec]
    cls, kw = spec
    is_spec = lambda v: (isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], str) and isinstance(v[1], dict)) or \
        (isinstance(v, list) and v and isinstance(v[0], tuple))
    return press_class(ns, cls)(**{k: (build_press(ns, v) if is_spec(v) else v) for k, v in kw.items()})


_KN = lambda r=0.0: ("KnormPress", dict(compression_ratio=r))
PIPELINE_CASES = {
    # name: (press spec, context words, questions, max_new_tokens)
    "pipe_knorm": (_KN(0.5), 120, ["w1 w2 w3", "w7"], 8),
    "pipe_snapkv": (("SnapKVPress", dict(compression_ratio=0.5, window_size=16, kernel_size=5)), 150, ["w4 w5"], 8),
    "pipe_ea": (("ExpectedAttentionPress", dict(compression_ratio=0.4)), 23, ["w9 w10 w11 w12 w13"], 6),
    "pipe_none": (None, 40, ["w3"], 6),
    # SURVEY §8 f-2 .. f-4 through the pipeline
    "pipe_tova": (("TOVAPress", dict(compression_ratio=0.5)), 90, ["w2 w8"], 6),
    "pipe_keydiff": (("KeyDiffPress", dict(compression_ratio=0.3)), 77, ["w5"], 6),
    "pipe_streaming_rerot": (("KeyRerotationPress", dict(press=("StreamingLLMPress", dict(compression_ratio=0.5, n_sink=4)))), 100, ["w6 w7", "w1"], 6),
    "pipe_chunk": (("ChunkPress", dict(press=_KN(0.5), chunk_length=32)), 110, ["w3 w4"], 6),
    "pipe_composed": (("ComposedPress", dict(presses=[_KN(0.25), _KN(0.5)])), 120, ["w9"], 6),
    "pipe_adakv": (("AdaKVPress", dict(press=_KN(0.5), alpha_safeguard=0.2)), 90, ["w4", "w5 w6"], 6),
    "pipe_decoding": (("DecodingPress", dict(base_press=_KN(), compression_interval=3, target_size=40, hidden_states_buffer_size=0)),
                      60, ["w1 w2"], 14),
    "pipe_prefill_decoding": (("PrefillDecodingPress", dict(
        prefilling_press=_KN(0.5),
        decoding_press=("DecodingPress", dict(base_press=_KN(), compression_interval=4, target_size=36, hidden_states_buffer_size=2)))),
        80, ["w5 w6 w7"], 16),
    "pipe_ratio_decoding": (("CompressionRatioDecodingPress", dict(base_press=_KN(), target_compression_ratio=0.5, compression_interval=3,
                                                                    hidden_states_buffer_size=4)), 70, ["w2 w3 w4"], 13),
    # QuantizedCache write-back of the hook (base_press.py:152-157) and the pipeline's answer removal
    "pipe_knorm_quantized": (_KN(0.5), 120, ["w1 w2 w3", "w7"], 8),
    "pipe_snapkv_quantized": (("SnapKVPress", dict(compression_ratio=0.5, window_size=16, kernel_size=5)), 150, ["w4 w5"], 8),
    "pipe_none_quantized": (None, 40, ["w3"], 6),
}
# cases that run on a QuantizedCache instead of a DynamicCache
PIPELINE_QUANTIZED = ("pipe_knorm_quantized", "pipe_snapkv_quantized", "pipe_none_quantized")


def make_pipeline_cache(name: str, config):
    from transformers import DynamicCache

    return make_fixed_point_cache(config) if name in PIPELINE_QUANTIZED else DynamicCache()


def make_fixed_point_cache(config, residual_length: int = 8):
    """A transformers ``QuantizedCache`` whose layers use an in-test backend (optimum-quanto and hqq are not installed):
    int8 fixed point with 4 fractional bits, stored as a plain sliceable tensor.  Everything else -- the residual buffer,
    ``cumulative_length``, the re-quantisation on overflow -- is transformers' own ``QuantizedLayer``."""
    import torch
    from transformers.cache_utils import Cache, QuantizedCache, QuantizedLayer

    class FixedPointLayer(QuantizedLayer):
        def _quantize(self, tensor, axis):
            self._float_dtype = tensor.dtype
            return (tensor.float() * 16.0).round().clamp(-127, 127).to(torch.int8)

        def _dequantize(self, q_tensor):
            return (q_tensor.float() / 16.0).to(self._float_dtype)

    class FixedPointCache(QuantizedCache):
        def __init__(self, n_layers):
            Cache.__init__(self, layers=[FixedPointLayer(8, 0, 0, 64, residual_length) for _ in range(n_layers)])

    return FixedPointCache(config.num_hidden_layers)


# FinchPress through the pipeline: context + delimiter + question in ONE prefill (the press must first register its delimiter
# with the tokenizer and the embedding table, finch_press.py:139-151), then the question is asked again as usual.
FINCH_PIPELINE_CASES = {
    # name: (FinchPress kwargs, context words, question, max_new_tokens)
    "pipe_finch": (dict(compression_ratio=0.5), 80, "w4 w5 w6", 8),
    "pipe_finch_chunk_norerot": (dict(compression_ratio=0.5, chunk_length=20, rerotate_keys=False, normalize_scores=False), 90, "w9 w1", 6),
}


def run_finch_pipeline(ns, pipeline_factory, name, cache, dtype=None, device="cpu"):
    """``ns``: the reference package or this one; ``pipeline_factory(model, tokenizer)`` -> the matching pipeline object."""
    kw, n_words, question, max_new = FINCH_PIPELINE_CASES[name]
    model, tok = make_tiny_llama(dtype=dtype, device=device), make_tiny_tokenizer()
    press = ns.FinchPress(**kw)
    tok = press.update_model_and_tokenizer(model, tok)
    context = tiny_context(n_words) + " " + press.delimiter_token + " " + question
    return pipeline_factory(model, tok)(context, questions=[question], press=press, max_new_tokens=max_new, cache=cache), press


# ---- selection wrappers: ChunkPress / KeyRerotationPress around a scorer ---------------------------
WRAP_CASES = {
    # name: wrapper, inner press kind, geometry (as CASES), wrapper parameters
    "wrap_chunk_knorm": dict(wrapper="chunk", kind="knorm", B=2, H=2, G=1, S=1000, D=16, dtype="f32", data="A", seed=81,
                             chunk_length=256, ratios=(0.25, 0.5, 0.9)),
    "wrap_chunk_snapkv": dict(wrapper="chunk", kind="snapkv", B=1, H=2, G=2, S=700, D=16, dtype="f32", data="B", seed=82,
                              chunk_length=200, W=8, ks=5, ratios=(0.5,)),
    "wrap_chunk_knorm_bf16": dict(wrapper="chunk", kind="knorm", B=1, H=8, G=1, S=4096, D=128, dtype="bf16", data="B", seed=83,
                                  chunk_length=1024, ratios=(0.5,)),
    "wrap_chunk_keydiff_tail": dict(wrapper="chunk", kind="keydiff", B=1, H=2, G=1, S=515, D=64, dtype="f16", data="A", seed=84,
                                    chunk_length=128, ratios=(0.3,)),
    "wrap_rerot_knorm": dict(wrapper="rerot", kind="knorm", B=2, H=2, G=1, S=300, D=16, dtype="f32", data="A", seed=85, ratios=(0.5,)),
    "wrap_rerot_knorm_bf16": dict(wrapper="rerot", kind="knorm", B=1, H=2, G=4, S=515, D=128, dtype="bf16", data="B", seed=86,
                                  ratios=(0.5, 0.8)),
    "wrap_block_knorm": dict(wrapper="block", kind="knorm", B=2, H=2, G=1, S=300, D=16, dtype="f32", data="A", seed=91,
                             block_size=32, ratios=(0.25, 0.5, 0.9)),
    "wrap_block_keydiff": dict(wrapper="block", kind="keydiff", B=1, H=2, G=1, S=515, D=64, dtype="f32", data="B", seed=92,
                               block_size=128, ratios=(0.5,)),
    "wrap_block_snapkv": dict(wrapper="block", kind="snapkv", B=1, H=2, G=2, S=400, D=16, dtype="f32", data="B", seed=93,
                              block_size=64, W=8, ks=5, ratios=(0.5,)),
    "wrap_chunkkv_knorm": dict(wrapper="chunkkv", kind="knorm", B=2, H=2, G=1, S=1000, D=16, dtype="f32", data="B", seed=94,
                               chunk_length=20, ratios=(0.25, 0.5, 0.9)),
    "wrap_chunkkv_snapkv_tail": dict(wrapper="chunkkv", kind="snapkv", B=1, H=2, G=2, S=707, D=16, dtype="f32", data="B", seed=95,
                                     chunk_length=64, W=8, ks=5, ratios=(0.5, 0.97)),
    "wrap_chunkkv_short": dict(wrapper="chunkkv", kind="knorm", B=1, H=2, G=1, S=15, D=16, dtype="f32", data="A", seed=96,
                               chunk_length=20, ratios=(0.5,)),
    "wrap_adakv_knorm": dict(wrapper="adakv", kind="knorm", B=2, H=4, G=1, S=300, D=16, dtype="f32", data="B", seed=88, alpha=0.2,
                             ratios=(0.25, 0.5, 0.9)),
    "wrap_adakv_snapkv": dict(wrapper="adakv", kind="snapkv", B=1, H=2, G=4, S=700, D=128, dtype="f32", data="B", seed=89, alpha=0.5,
                              W=64, ks=5, ratios=(0.5,)),
    "wrap_adakv_knorm_alpha0": dict(wrapper="adakv", kind="knorm", B=1, H=8, G=1, S=1000, D=64, dtype="f32", data="A", seed=90,
                                    alpha=0.0, ratios=(0.7,)),
    "wrap_rerot_streaming_f16": dict(wrapper="rerot", kind="streaming", B=1, H=2, G=1, S=257, D=64, dtype="f16", data="A", seed=87,
                                     ratios=(0.4,)),
}


# ---- FinchPress: question-window scores, optional per-chunk selection and key re-rotation ---------------
FINCH_CASES = {
    # W = the question length (any); chunk_length > W / (1 - ratio) (finch_press.py:103)
    "finch_tiny": dict(B=1, H=2, G=2, S=100, D=16, dtype="f32", data="A", seed=101, W=11, normalize=True, chunk_length=None,
                       rerotate=True, ratios=(0.25, 0.5)),
    "finch_nonorm_chunk": dict(B=2, H=2, G=1, S=300, D=16, dtype="f32", data="B", seed=102, W=7, normalize=False, chunk_length=64,
                               rerotate=False, ratios=(0.5,)),
    "finch_64_bf16": dict(B=1, H=2, G=4, S=1000, D=128, dtype="bf16", data="B", seed=103, W=64, normalize=True, chunk_length=256,
                          rerotate=True, ratios=(0.5,)),
    "finch_37_f16": dict(B=1, H=2, G=2, S=515, D=64, dtype="f16", data="A", seed=104, W=37, normalize=True, chunk_length=None,
                         rerotate=True, ratios=(0.3, 0.8)),
}


def make_finch_case(name: str) -> dict:
    extra = ("normalize", "chunk_length", "rerotate")
    CASES[name] = dict({k: v for k, v in FINCH_CASES[name].items() if k not in extra}, kind="snapkv", ks=1)
    try:
        s = make_case(name)
    finally:
        del CASES[name]
    s.update({k: FINCH_CASES[name][k] for k in extra})
    return s


def make_finch_press(mod, s: dict, ratio: float):
    p = mod.FinchPress(compression_ratio=ratio, chunk_length=s["chunk_length"], normalize_scores=s["normalize"], rerotate_keys=s["rerotate"])
    p.window_size = s["W"]  # what the embedding hook sets from the delimiter position (finch_press.py:133)
    return p


# ---- ThinKPress: key-channel pruning -------------------------------------------------------------------------------
THINK_CASES = {
    "think_tiny": dict(B=2, H=2, G=2, S=100, D=16, dtype="f32", data="B", seed=141, W=8, ratios=(0.25, 0.5)),
    "think_bf16": dict(B=1, H=2, G=4, S=700, D=128, dtype="bf16", data="B", seed=142, W=32, ratios=(0.5,)),
    "think_f16_d64": dict(B=1, H=4, G=1, S=257, D=64, dtype="f16", data="B", seed=143, W=16, ratios=(0.2, 0.8)),
}


def make_think_case(name: str) -> dict:
    CASES[name] = dict(THINK_CASES[name], kind="snapkv", ks=1)
    try:
        return make_case(name)
    finally:
        del CASES[name]


def make_wrap_case(name: str) -> dict:
    CASES[name] = {k: v for k, v in WRAP_CASES[name].items() if k not in ("wrapper", "chunk_length", "alpha", "block_size")}
    try:
        s = make_case(name)
    finally:
        del CASES[name]
    s["wrapper"] = WRAP_CASES[name]["wrapper"]
    s["chunk_length"] = WRAP_CASES[name].get("chunk_length")
    s["alpha"] = WRAP_CASES[name].get("alpha")
    s["block_size"] = WRAP_CASES[name].get("block_size")
    return s
