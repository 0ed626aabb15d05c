/*
 * kvpress_hip.h -- C ABI of libkvpress_hip.so: the MI355X (gfx950) implementation of the
 * kvpress per-layer KV score -> top-k -> gather hot path.
 *
 * The reference (NVIDIA/kvpress v0.5.4) is pure Python and has no FFI; its boundary for this
 * path is the Python protocol BasePress.forward_hook -> ScorerPress.compress -> score()
 * (kvpress/presses/base_press.py:101-162, scorer_press.py:76-102).  This header is what a
 * binding of that path binds: every entry point names the reference lines it replaces.
 * kvpress_amd/_native.py is the ctypes binding; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only (no torch / framework types);
 *   - all pointers are DEVICE pointers on the current HIP device unless stated otherwise;
 *   - 4-D tensors are [B, H, S, D] views with the last dim contiguous; strides are in ELEMENTS
 *     (sb, sh, ss) so sliced cache views (keys[:, :, n_sink:], chunk slices) need no copy;
 *   - outputs and workspaces are caller-allocated; *_workspace_bytes() gives the size;
 *   - every call is asynchronous on `stream` (a hipStream_t), allocates nothing, never
 *     synchronises, keeps no global state and is thread-safe;
 *   - return 0 (KVP_OK) or a negative KVP_E* code; kvp_last_error() returns the calling
 *     thread's last message.
 *   - scores are always float32 (the reference returns the model dtype; float32 is a superset
 *     and makes the top-k well defined -- DESIGN.md "Parity contract").
 */
#ifndef KVPRESS_HIP_H
#define KVPRESS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVP_VERSION 100 /* 0.1.0 */

typedef void* kvp_stream_t; /* hipStream_t */

enum kvp_dtype { KVP_F32 = 0, KVP_F16 = 1, KVP_BF16 = 2 };
enum kvp_status { KVP_OK = 0, KVP_EINVAL = -1, KVP_EUNSUPPORTED = -2, KVP_EWORKSPACE = -3, KVP_EHIP = -4, KVP_EASYNC = -5 };
/* Order of the retained indices.  POSITION: ascending token position (default; makes the gather a
 * monotone stream).  SCORE: descending score, ties by ascending position -- the element order
 * torch.topk(sorted=True) produces at scorer_press.py:95, for tensor-exact comparisons. */
enum kvp_order { KVP_ORDER_POSITION = 0, KVP_ORDER_SCORE = 1 };
/* OR-ed into `order`: the workspace was zero-filled once (its first kvp_topk_workspace_bytes() bytes) and has
 * only been used by kvp_topk_select on ONE stream since.  The call then skips its memset: the histograms in
 * the workspace are self-cleaning (left zero on completion). */
#define KVP_TOPK_WS_CLEAN 0x100
/* OR-ed into `order`: select the k SMALLEST scores instead of the k largest (ties still go to the lowest position);
 * AdaKVPress's bottom-k across heads (kvpress/presses/adakv_press.py:66-67) without negating the scores. */
#define KVP_TOPK_SMALLEST 0x200

int kvp_version(void);
const char* kvp_last_error(void);
/* Failures a kernel can only detect while it RUNS.  The one-launch select of long rows (kvp_topk_select and the fused
 * kvp_*_compress entry points on rows of 16385 .. 262144 scores) synchronises the 32 workgroups of a row with in-kernel barriers;
 * if they never become co-resident (CU masking, a partitioned device) its bounded spin gives up.  torch.topk
 * (kvpress/presses/scorer_press.py:95) cannot return wrong indices silently, so neither does this library: the affected rows'
 * indices are written as -1 (kvp_gather_kv* turn a negative index into a row of all-ones bit patterns = NaN instead of reading
 * anywhere), a process-wide host-pinned status word is set, and the NEXT call of kvp_topk_select*, kvp_*_compress*, kvp_gather_kv*
 * or kvp_async_error_check on any thread returns KVP_EASYNC (once per report; kvp_last_error() has the text).  After KVP_EASYNC
 * every workspace that was being used with KVP_TOPK_WS_CLEAN must be zero-filled again.  kvp_async_error_check() only polls the
 * word (no synchronisation): call it after a stream sync to learn about everything enqueued before. */
int kvp_async_error_check(void);

/* ---- KnormPress.score: -keys.norm(dim=-1)  (kvpress/presses/knorm_press.py:38) -------------
 * out[b,h,s] = scale * ||x[b,h,s,:]||_2   (scale = -1 for Knorm; +1 for ||V|| in ExpectedAttention,
 * expected_attention_press.py:160).  out is contiguous [B,H,S] float32. */
int kvp_rownorm_score(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                      int64_t sb, int64_t sh, int64_t ss, float scale, float* out, kvp_stream_t stream);

/* ---- QFilterPress.score: -(q_filter * keys).sum(dim=-1)  (kvpress/presses/qfilter_press.py:79-82) ----------------
 * out[b,h,s] = scale * sum_d x[b,h,s,d] * filt[h,d]   (scale = -1; filt = the layer's [H,D] slice of the learned
 * Q-filters, same dtype as x, f_sh = element stride between heads, rows contiguous).  out contiguous [B,H,S] float32. */
int kvp_rowdot_score(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                     int64_t sb, int64_t sh, int64_t ss, const void* filt, int64_t f_sh, float scale, float* out,
                     kvp_stream_t stream);

/* ---- SnapKVPress.score (kvpress/presses/snapkv_press.py:60-105) ----------------------------
 * q: RoPE'd queries of the last W tokens [B,Hq,W,D] (the host keeps q_proj + RoPE,
 *    snapkv_press.py:53-58 / utils.py:43-46: q_proj is a model-owned nn.Linear);
 * k: keys [B,Hkv,S,D].  Computes softmax(q k^T / sqrt(D)) over all S keys with the causal mask
 * on the last W columns (:62-66), mean over the W rows (:95), avg_pool1d(kernel_size,
 * pad=kernel_size/2, stride 1, zero padded, divisor kernel_size) (:96), mean over the
 * Hq/Hkv group (:99-100), and fills the last W positions with max(scores)+1, max global over
 * B and Hkv (:103) -- on the device, no host sync.  scores: [B,Hkv,S] float32, contiguous.
 * kernel_size must be odd.  Requires S > W. */
size_t kvp_snapkv_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D);
int kvp_snapkv_score(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                     const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                     int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                     float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* Same, with the RoPE of the window queries (snapkv_press.py:56-58) done inside the library:
 * q is the PRE-RoPE q_proj output viewed [B,Hq,W,D]; cos/sin are the last W rows of the layer's
 * position embeddings, [Bc,W,D] with Bc = B or 1 (cs_sb = 0 broadcasts), same dtype as q.
 * q_rot = q*cos + rotate_half(q)*sin is evaluated exactly as the reference's torch ops do in the
 * model dtype (every product and the sum rounded to that dtype), so q_rot is bit-identical. */
int kvp_snapkv_score_rope(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                          const void* cos, const void* sin, int64_t cs_sb, int64_t cs_sw,
                          const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                          int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                          float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* FINCH scores (kvpress/presses/finch_press.py:56-83): the window attention of kvp_snapkv_score_rope for a window of ANY
 * length W (the question that follows the context), no pooling; with normalize_scores != 0 window row w is weighted by its
 * number of visible keys S - W + w before the mean over the window (:71-74).  Window columns = max + 1 (:82).
 * Arguments and workspace (kvp_snapkv_workspace_bytes) as kvp_snapkv_score_rope. */
int kvp_finch_score(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                    const void* cos, const void* sin, int64_t cs_sb, int64_t cs_sw,
                    const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                    int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int normalize_scores,
                    float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* Same, when the attention layer returned its weights (snapkv_press.py:88-89):
 * attn is the [B,Hq,W,S-W] view attentions[..., -W:, :-W] (last dim contiguous). */
int kvp_snapkv_score_from_attn(const void* attn, int64_t a_sb, int64_t a_sh, int64_t a_sw, int dtype,
                               int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int kernel_size,
                               float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- ExpectedAttentionPress.get_query_statistics (kvpress/presses/expected_attention_press.py:62-86) ----
 * kvp_ea_qstats: mean and covariance of the pre-RoPE queries q [B,Hq,Sq,D] (sinks already
 * stripped by the host, :70-71): mu[b,h,:] = mean_s q; cov = (q-mu)^T (q-mu) / Sq (:74-80).
 * mu [B,Hq,D], cov [B,Hq,D,D] float32 contiguous; cov may be NULL (use_covariance=False).
 * The averaged-RoPE step (:110-123) is 128x128 host-side math on mu/cov and stays in the host. */
size_t kvp_ea_qstats_workspace_bytes(int64_t B, int64_t Hq, int64_t Sq, int64_t D);
int kvp_ea_qstats(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype,
                  int64_t B, int64_t Hq, int64_t Sq, int64_t D,
                  float* mu, float* cov, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- ExpectedAttentionPress.score (kvpress/presses/expected_attention_press.py:126-165) ----
 * kvp_ea_score (:137-163): with the post-RoPE mu [B,Hq,D] / cov [B,Hq,D,D] (float32, cov may be
 * NULL): for keys/values [B,Hkv,S,D] drop the first n_sink positions; logits = k.mu/sqrt(D) +
 * k^T cov k / (2D) per q-head; softmax over the S-n_sink keys; mean over the group;
 * (s + epsilon) * ||v||_2 if use_vnorm; positions < n_sink get max(scores)+1 (global max).
 * scores [B,Hkv,S] float32. */
size_t kvp_ea_score_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t D);
int kvp_ea_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                 const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                 const float* mu, const float* cov,
                 int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t D,
                 int64_t n_sink, int use_vnorm, float epsilon,
                 float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- ScorerPress.compress: top-k + gather (kvpress/presses/scorer_press.py:94-100) ----------
 * kvp_topk_select: for each of the R rows of scores[R, S] (row stride in elements) write the
 * indices of the k largest scores to idx[R, k] (int32).  torch.topk leaves ties unspecified;
 * here the LOWEST POSITION wins among equal scores, and -0.0 == +0.0.  0 <= k <= S. */
size_t kvp_topk_workspace_bytes(int64_t R, int64_t S, int64_t k);
/* workspace for order == KVP_ORDER_SCORE (the select's + the sort's); KVP_ORDER_POSITION needs only the one above */
size_t kvp_topk_order_workspace_bytes(int64_t R, int64_t S, int64_t k);
int kvp_topk_select(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int order,
                    int32_t* idx, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* kvp_gather_kv: k_out[b,h,j,:] = k[b,h,idx[b*H+h, j],:], same for v (scorer_press.py:96-100).
 * Outputs are contiguous [B,H,n,D] in the input dtype; inputs are not modified.  When K + V exceed the memory-side cache
 * (cutoff 192 MiB; the cache itself is 256 MiB) the rows are moved with non-temporal loads / stores: the copy then neither displaces what the next kernels
 * re-read nor leaves its output behind as dirty cache lines. */
int kvp_gather_kv(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                  const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                  int64_t B, int64_t H, int64_t S, int64_t D,
                  const int32_t* idx, int64_t n, void* k_out, void* v_out, kvp_stream_t stream);

/* scores[r, idx[r, j]] = value for every j < n (AdaKVPress's safeguard `scores.scatter_(-1, top_indices, finfo.max)`,
 * kvpress/presses/adakv_press.py:62-63); idx int32 [R, n] contiguous, scores rows row_stride elements apart. */
int kvp_scores_fill_at(float* scores, int64_t R, int64_t S, int64_t row_stride, const int32_t* idx, int64_t n, float value,
                       kvp_stream_t stream);

/* ---- ChunkPress: per-chunk top-k (kvpress/presses/chunk_press.py:67-85) ---------------------------------------------
 * scores[R, nseg * seg_len] contiguous; the k largest of EACH chunk of seg_len columns are selected (same tie rule as
 * kvp_topk_select); idx[R, nseg * k] holds them chunk after chunk as row positions + pos_base, i.e. ascending overall. */
size_t kvp_topk_segmented_workspace_bytes(int64_t R, int64_t nseg, int64_t seg_len, int64_t k);
int kvp_topk_select_segmented(const float* scores, int64_t R, int64_t nseg, int64_t seg_len, int64_t k, int64_t pos_base,
                              int order, int32_t* idx, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- KeyRerotationPress.rerotate_keys after the gather (kvpress/presses/key_rerotation_press.py:50-128) ------------
 * k: contiguous [B,H,n,D] keys gathered with idx[B*H, n] (ascending positions); in place
 * k[b,h,j] <- k * cos(f) + rotate_half(k) * sin(f),  f = (j - idx[b,h,j]) * inv_freq[d mod D/2]  (inv_freq: D/2 floats,
 * module.rotary_emb.inv_freq), cos/sin cast to the key dtype and every op rounded in it, as torch does. */
int kvp_rerotate_keys(void* k, int dtype, int64_t B, int64_t H, int64_t n, int64_t D, const int32_t* idx,
                      const float* inv_freq, kvp_stream_t stream);
/* The gather and the re-rotation above in one pass (key_rerotation_press.py:157-160 + :107-128; finch_press.py:111-119): same
 * arguments as kvp_gather_kv + inv_freq, same bits as kvp_gather_kv followed by kvp_rerotate_keys on k_out; the gathered keys are
 * written once instead of written, read and written again.  2-byte dtypes with D % 16 == 0 and 16-byte aligned rows take the
 * one-pass kernel, everything else runs the two kernels it replaces. */
int kvp_gather_kv_rerotate(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh,
                           int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, const int32_t* idx, int64_t n,
                           const float* inv_freq, void* k_out, void* v_out, kvp_stream_t stream);

/* ---- fused ScorerPress.compress (scorer_press.py:76-102 with the scorer inlined) ---------------------------------
 * One call = score -> select n_kept -> gather into caller-allocated contiguous k_out / v_out [B,H,n_kept,D].  Same
 * kernels and results as the modular sequence kvp_*_score + kvp_topk_select + kvp_gather_kv; the score-writing kernel
 * also accumulates the select's first histogram, and SnapKV's window columns are appended to the selection instead of
 * being scored with max + 1 (snapkv_press.py:103), which removes the global max and the pad fill.
 * flags: KVP_TOPK_WS_CLEAN if the first kvp_topk_workspace_bytes(R, S, n_kept) bytes of ws were zero-filled once and
 * the workspace has only been used by these calls on ONE stream since (they leave it clean), else 0;
 * | KVP_ORDER_SCORE: the rows of k_out / v_out in descending score order, ties by ascending position (SnapKV: the window tokens
 * first) -- the order in which the reference stores them (scorer_press.py:95-100); default: ascending position. */
size_t kvp_knorm_compress_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t n_kept);
int kvp_knorm_compress(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                       const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                       int64_t B, int64_t H, int64_t S, int64_t D, int64_t n_kept,
                       void* k_out, void* v_out, void* ws, size_t ws_bytes, int flags, kvp_stream_t stream);
size_t kvp_snapkv_compress_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int64_t n_kept);
/* arguments as kvp_snapkv_score_rope (pre-RoPE window queries + the window's cos/sin) plus V and the outputs */
int kvp_snapkv_compress_rope(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                             const void* cos, const void* sin, int64_t cs_sb, int64_t cs_sw,
                             const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                             const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                             int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                             int64_t n_kept, void* k_out, void* v_out, void* ws, size_t ws_bytes, int flags,
                             kvp_stream_t stream);

/* ---- KeyDiffPress.score (kvpress/presses/keydiff_press.py:45-46) ----------------------------------
 * anchor[b,h,:] = mean_s k[b,h,s,:] / max(||k[b,h,s,:]||, 1e-12)   (F.normalize(keys).mean(dim=2))
 * scores[b,h,s] = -cosine_similarity(k[b,h,s,:], anchor[b,h,:])     (eps 1e-8), contiguous [B,H,S] float32. */
size_t kvp_keydiff_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t D);
int kvp_keydiff_score(const void* k, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                      int64_t sb, int64_t sh, int64_t ss, float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- CURPress.score (kvpress/presses/cur_press.py:32-66) without the random projection ------------------------------
 * k2 = sum_d k^2, v2 = sum_d v^2; with local_window_size w > 0 each is divided by its sum over the windows of w consecutive
 * tokens (zero-padded tail); combined by leverage type; divided by the row sum; the first num_sinks positions are set to 1.
 * scores: contiguous [B,H,S] float32. */
enum kvp_cur_leverage { KVP_CUR_KEY = 0, KVP_CUR_VALUE = 1, KVP_CUR_KV_AVG = 2, KVP_CUR_KV_PRODUCT = 3 };
size_t kvp_cur_workspace_bytes(int64_t B, int64_t H, int64_t S);
int kvp_cur_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                  int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int leverage_type, int64_t local_window_size,
                  int64_t num_sinks, float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);

/* ---- TOVAPress.score tail (kvpress/presses/tova_press.py:52-53) -----------------------------------
 * scores[b,h,s] <- mean over h' of scores[b,h',s] for every h, in place (`attn_weights.mean(1)` followed by
 * `.repeat(1, num_kv_heads, 1)`, applied to the per-kv-group means kvp_snapkv_score* produce with W = 1, kernel 1).
 * Strides in elements. */
int kvp_scores_head_mean(float* scores, int64_t B, int64_t H, int64_t S, int64_t stride_b, int64_t stride_h,
                         kvp_stream_t stream);

/* ---- window q_proj + RoPE in the library (kvpress/utils.py:43-46 + snapkv_press.py:53-58) ---------------------------
 * hidden_win: hidden_states[:, -W:] [B, W, hidden] (element strides x_sb, x_sw; last dim contiguous); wq: q_proj.weight
 * [Hq * D, hidden] contiguous, no bias; cos/sin: the window's rotary tables [1 or B, W, D].  bf16 / f16, W = 64, D = 128,
 * hidden % 256 == 0, else KVP_EUNSUPPORTED (the caller then runs its own q_proj and kvp_snapkv_*_rope).
 * kvp_snapkv_qproj_rope writes the RoPE'd window queries q_rot [B, Hq, W, D] (contiguous, input dtype): fp32 accumulation
 * in a fixed order, rounded to the dtype like a GEMM output, then rotated with torch's per-op rounding.
 * kvp_snapkv_score_hidden / kvp_snapkv_compress_hidden = kvp_snapkv_score_rope / kvp_snapkv_compress_rope from there. */
int kvp_snapkv_qproj_rope(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, const void* cos, const void* sin,
                          int64_t cs_sb, int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t W, int64_t D, int64_t hidden,
                          void* q_rot, kvp_stream_t stream);
int kvp_snapkv_score_hidden(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden,
                            const void* cos, const void* sin, int64_t cs_sb, int64_t cs_sw,
                            const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                            int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                            float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);
int kvp_snapkv_compress_hidden(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden,
                               const void* cos, const void* sin, int64_t cs_sb, int64_t cs_sw,
                               const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                               const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                               int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                               int64_t n_kept, void* k_out, void* v_out, void* ws, size_t ws_bytes, int flags,
                               kvp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KVPRESS_HIP_H */
