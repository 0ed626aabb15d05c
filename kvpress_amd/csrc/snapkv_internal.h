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
int snapkv_mfma_p2(const SnapArgs& a, int dtype, const float* rowstat, float* colsum, hipStream_t stream);
