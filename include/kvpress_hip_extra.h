/* kvpress_hip_extra.h -- entry points OUTSIDE the hot path of SURVEY.md section 8.
 *
 * include/kvpress_hip.h is the boundary a maintainer binds for the score -> top-k -> gather path (KnormPress, SnapKVPress,
 * ExpectedAttentionPress and the section-8(f) presses that reuse its kernels).  The functions below serve presses that
 * SURVEY.md section 2 marks out of scope (ObservedAttention, LagKV, ThinK: python package kvpress_amd.contrib); they were built in
 * round 1, are kept and tested, and are built into their OWN shared library (libkvpress_hip_contrib.so, sources
 * kvpress_amd/csrc/contrib/), which resolves its error / launch plumbing against libkvpress_hip.so.
 * Conventions (return codes, dtypes, strides, streams): as in kvpress_hip.h. */
#ifndef KVPRESS_HIP_EXTRA_H
#define KVPRESS_HIP_EXTRA_H
#include "kvpress_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- ObservedAttentionPress.score (kvpress/presses/observed_attention_press.py:42-48) --------------------------------
 * attn: the attention weights the (eager) attention layer returned, [B,Hq,Sq,S] (element strides a_sb, a_sh, a_sq; last
 * dim contiguous).  scores[b,h,s] = mean over the kv-head's G q-heads of sum_q attn[b,hq,q,s] / (S - s): the average
 * weight key s receives from the queries that can see it.  scores contiguous [B,Hkv,S] float32. */
int kvp_observed_attention_score(const void* attn, int64_t a_sb, int64_t a_sh, int64_t a_sq, int dtype,
                                 int64_t B, int64_t Hq, int64_t Hkv, int64_t Sq, int64_t S, float* scores, kvp_stream_t stream);

/* ---- LagKVPress.score (kvpress/presses/lagkv_press.py:56-97), sequences of at least n_sink + 2 * lag_size tokens --------
 * After n_sink sinks the sequence is cut into partitions of lag_size tokens; partition p is scored against partition
 * p + 1: per-channel min / max of p + 1, tokens of p normalised with them, score = std over the channels (unbiased),
 * softmax over the partition; K and V scores averaged; unless cross_scoring the score becomes rank / lag_size inside
 * the partition (equal scores rank by position).  Sinks, the last complete partition and the remainder score 1.
 * head_dim <= 512, lag_size <= 1024.  scores contiguous [B,H,S] float32.  (Shorter sequences: constant scores, host side.) */
int kvp_lagkv_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                    const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                    int64_t B, int64_t H, int64_t S, int64_t D, int64_t n_sink, int64_t lag_size, int cross_scoring,
                    float* scores, kvp_stream_t stream);

/* ---- ThinKPress (kvpress/presses/think_press.py:56-85): key-channel pruning ----------------------------------------------
 * kvp_think_channel_scores: scores[b,h,d] = mean over the kv-head's G q-heads and the W window rows of q[b,hq,w,d]^2
 *   times mean over the S keys of k[b,h,s,d]^2 (:72-76); q = RoPE'd queries of the last W tokens [B,Hq,W,D].  head_dim <= 256.
 *   scores contiguous [B,Hkv,D] float32.  The channels to prune are the n lowest: kvp_topk_select | KVP_TOPK_SMALLEST.
 * kvp_zero_channels: x[b,h,s,idx[b,h,j]] = 0 for all s, j < n, IN PLACE (`keys.scatter_(-1, indices, 0)`, :81-82);
 *   idx contiguous [B,H,n] int32, n <= 1024. */
size_t kvp_think_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t D);
int kvp_think_channel_scores(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                             const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                             int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D,
                             float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream);
int kvp_zero_channels(void* x, int64_t sb, int64_t sh, int64_t ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                      const int32_t* idx, int64_t n, kvp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KVPRESS_HIP_EXTRA_H */
