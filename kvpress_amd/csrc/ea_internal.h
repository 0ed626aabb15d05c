// Internal interface between ea.hip (host entry, generic kernels) and ea_mfma.hip.
#pragma once
#include "kvp_common.h"

struct EaArgs {
    const void* k;  // [B,Hkv,S,D]; positions < n_sink are skipped
    int64_t k_sb, k_sh, k_ss;
    const float* mu;   // [B,Hq,D]   post-RoPE query mean
    const float* cov;  // [B,Hq,D,D] post-RoPE covariance, or nullptr
    uint32_t B, Hq, Hkv, G, S, Sp, D, n_sink;  // Sp = S - n_sink
    float inv_sqrt_d, inv_2d;
    uint32_t* clear_word;  // nullable: a word the logits kernel sets to 0 (the arrival counter of the finalize pass that follows it in the stream)
};

// MFMA fast paths (bf16/f16; D = 128, the statistics also D = 64 with neighbouring heads, the logits also D = 64 / 96)
bool ea_mfma_qstats_eligible(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype, int64_t Hq, int64_t Sq, int64_t D);
size_t ea_mfma_qstats_ws_bytes(int64_t B, int64_t Hq, int64_t Sq, int64_t D);
int ea_mfma_qstats(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype, int64_t B, int64_t Hq, int64_t Sq,
                   int64_t D, float* mu, float* cov, void* ws, hipStream_t stream);

bool ea_mfma_logits_eligible(const EaArgs& a, int dtype);
size_t ea_mfma_logits_scratch_bytes(int64_t B, int64_t Hq, int64_t D);
uint32_t ea_mfma_logits_nblk(const EaArgs& a);
int ea_mfma_logits(const EaArgs& a, int dtype, float* logits, uint32_t nblk, float* part_m, float* part_z, void* scratch,
                   hipStream_t stream);
