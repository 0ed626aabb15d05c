// Internal interface between snapkv.hip (host entry, generic kernels) and snapkv_mfma.hip.
#pragma once
#include "kvp_common.h"

struct SnapArgs {
    const void* q;  // [B,Hq,W,D]  RoPE'd window queries
    const void* k;  // [B,Hkv,S,D]
    int64_t q_sb, q_sh, q_sw;  // element strides
    int64_t k_sb, k_sh, k_ss;
    uint32_t B, Hq, Hkv, G, S, W, D;
    float c;  // log2(e) / sqrt(D): logits in log2 units
};

// MFMA fast path (bf16/f16, D = 128, W = 64, G <= 8, 16-byte aligned rows)
bool snapkv_mfma_eligible(const SnapArgs& a, int dtype);
uint32_t snapkv_mfma_nchunk(const SnapArgs& a);
int snapkv_mfma_p1(const SnapArgs& a, int dtype, uint32_t nchunk, float* part_m, float* part_z, hipStream_t stream);
// colsum2: scratch of the size of colsum, needed when G > 4 (the second group-block's sums; merged in a fixed order: deterministic)
int snapkv_mfma_p2(const SnapArgs& a, int dtype, const float* rowstat, float* colsum, float* colsum2, hipStream_t stream);

enum { SNAP_FINISH_FULL = 0,    // pool + scale into `scores`, pad columns = max + 1
       SNAP_FINISH_NO_PAD = 1,  // pool + scale, pad columns left unwritten (fused compress: they are kept by construction)
       SNAP_FINISH_COLSUM = 2 };  // stop after pass 2: the un-pooled column sums stay in the workspace (snapkv_ws_colsum)
float* snapkv_ws_colsum(void* ws, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D);
float snapkv_pool_scale(int64_t Hq, int64_t Hkv, int64_t W, int kernel_size);  // 1 / (G * W * kernel_size)

// Score entry points with an optional fused first top-k histogram (hist1 [B*Hkv][4096] over the S - W non-pad columns;
// the pad columns of `scores` are then left unwritten).  finish: what happens after the two attention passes.  Defined in snapkv.hip, used by the fused compress (compress.hip).
int snapkv_score_rope_impl(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* cosp, const void* sinp,
                           int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                           int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                           float* scores, void* ws, size_t ws_bytes, hipStream_t stream, uint32_t* hist1, bool count_norm = false,
                           int finish = 0);
int snapkv_score_hidden_impl(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden, const void* cosp,
                             const void* sinp, int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                             int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                             float* scores, void* ws, size_t ws_bytes, hipStream_t stream, uint32_t* hist1, int finish = 0);
