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
    // MFMA kernels (round 6: any window size).  They work on blocks of 64 window rows; a window of W rows is padded AT THE FRONT to
    // Wp = 64 * ceil(W / 64) rows (padded row p = real row p - (Wp - W); rows p < Wp - W are padding: they read real row 0, their
    // statistics are never used and their normaliser is +inf, so they add 0 to every column sum).  Padded row p may attend keys
    // <= S - Wp + p: the reference's causal rule (snapkv_press.py:63-65) in padded coordinates.  The partial statistics and the
    // normalisers of this path are indexed [B, Hq, Wp]; `rblk` is the 64-row block a launch works on.
    uint32_t Wp, rblk;
};
__host__ __device__ inline uint32_t snapkv_wp(uint32_t W) { return (W + 63u) / 64u * 64u; }

// MFMA fast path (bf16/f16, D = 64 / 96 / 128 / 256, G <= 16, 16-byte aligned rows; any window size since round 6)
bool snapkv_mfma_eligible(const SnapArgs& a, int dtype);
uint32_t snapkv_mfma_nchunk(const SnapArgs& a);
// p1_ticks (may be null): [planes][nchunk] wall time of every pass-1 workgroup in 10 ns ticks (plane = (b, kv-head, group-block)) --
// the input of pass 2's tile shares, see snapkv_p2_shares
int snapkv_mfma_p1(const SnapArgs& a, int dtype, uint32_t nchunk, float* part_m, float* part_z, uint32_t* p1_ticks, hipStream_t stream);
// colsum2: (group-blocks - 1) scratch slabs of the size of colsum, needed when G > 4 (the further group-blocks' sums; merged in block order: deterministic)
// p2_ranges (may be null): [planes][nchunk][2] = (first tile, tiles) of every pass-2 workgroup; null = the interleaved static walk
// colsumx: a third scratch slab of the size of colsum, needed when W > 64 (the column sums of the row blocks after the first, added in order)
int snapkv_mfma_p2(const SnapArgs& a, int dtype, const float* rowstat, float* colsum, float* colsum2, float* colsumx, const uint32_t* p2_ranges,
                   hipStream_t stream);
// Pass 2 balanced by pass 1's clock (round 6).  Under the package power limit the XCDs run the window-attention passes at clocks 5-6 %
// apart (profiles/r05_clock_power.txt), so with equal tile counts the slowest XCD sets the launch span.  Pass 1 and pass 2 have the same
// shape and run microseconds apart: the wall time of a pass-1 workgroup predicts the speed of the pass-2 workgroup with the same block
// index (same XCD).  snapkv_p2_shares_plan says whether the shapes allow it (hand-scheduled kernels, the same grid in both passes, 2 .. 64
// workgroups per plane, walks of >= 8 tiles); if so pass 1 records its workgroup times and every pass-2 workgroup derives its contiguous
// tile range from its plane's times in its prologue (snapkv_p2_asm).  Pass 2's column sums are computed per 128-key tile by ONE
// workgroup in a fixed order, so its output bits do not depend on who owns which tile: the scores stay run-to-run identical.
// (Pass 1 cannot be treated the same way: its (max, sum) partials are per WALK, i.e. their rounding depends on the tile assignment.)
bool snapkv_p2_shares_plan(const SnapArgs& a, uint32_t nchunk_p1);

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
