// Shared pieces of the two-pass "softmax over all S keys" used by SnapKV and ExpectedAttention.
//
// Work in log2 units: L2 = logit * log2(e) so that exp() is a bare v_exp_f32 (exp2f).
// Pass 1 produces per (row, key-chunk) partials (m = max L2, z = sum 2^(L2-m)); `combine` folds
// them into one number per row, a = M + log2(Z), so that pass 2 evaluates the normalised
// probability as 2^(L2 - a) with a single fma + exp2.
#pragma once
#include "kvp_common.h"

constexpr float KVP_LOG2E = 1.4426950408889634f;
constexpr float KVP_NEG_INF = -__builtin_huge_valf();

// order-preserving key <-> float (for atomicMax over floats of any sign)
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// fold (m2,z2) into (m,z); either side may be the empty partial (m = -inf, z = 0)
__device__ __forceinline__ void softmax_merge(float& m, float& z, float m2, float z2) {
    if (m2 == KVP_NEG_INF) return;
    if (m == KVP_NEG_INF) { m = m2; z = z2; return; }
    const float mn = fmaxf(m, m2);
    z = z * exp2f(m - mn) + z2 * exp2f(m2 - mn);
    m = mn;
}

// a[row] = M + log2(Z) from part_m/part_z [nrows][nchunk]; one wave per row (rows = [.., W] window rows when norm_base is used).
// pad > 0 (SnapKV's MFMA path, snapkv_internal.h): rows are [.., W] PADDED window rows whose first `pad` are padding: their
// normaliser becomes +inf (2^(L2 - inf) = 0: they add nothing to a column sum), the real window row of row p is p - pad.
__device__ __forceinline__ float softmax_row_normaliser(float m, float z, uint32_t row, uint32_t W, uint32_t norm_base, uint32_t pad) {
    const uint32_t wp = W ? row % W : 0u;
    if (wp < pad) return __builtin_huge_valf();
    // norm_base != 0 (FINCH, finch_press.py:71-74): window row w is weighted by its number of visible keys
    // norm_base + w, i.e. its log2-normaliser is lowered by log2 of that count
    return m + log2f(z) - (norm_base ? log2f((float)(norm_base + wp - pad)) : 0.f);
}
static __global__ __launch_bounds__(256) void softmax_combine_kernel(const float* __restrict__ part_m,
                                                              const float* __restrict__ part_z, uint32_t nrows,
                                                              uint32_t nchunk, float* __restrict__ a, uint32_t W = 0,
                                                              uint32_t norm_base = 0, uint32_t pad = 0) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (row >= nrows) return;
    float m = KVP_NEG_INF, z = 0.f;
    for (uint32_t j = lane; j < nchunk; j += 64) softmax_merge(m, z, part_m[(size_t)row * nchunk + j], part_z[(size_t)row * nchunk + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), z2 = __shfl_xor(z, o);
        softmax_merge(m, z, m2, z2);
    }
    if (lane == 0) a[row] = softmax_row_normaliser(m, z, row, W, norm_base, pad);
}

// The same with one THREAD per row, for many rows with few partials each (ChunkPress scores every 1024-token chunk as a batch
// element: 262 144 window rows with one partial each at 128k tokens -- a wave per row spent 55 us on them).  Same merge order for
// nchunk <= 2 (a wave's lanes 0 and 1), hence the same bits there; used for nchunk <= 2 only.
static __global__ __launch_bounds__(256) void softmax_combine_rows_kernel(const float* __restrict__ part_m, const float* __restrict__ part_z,
                                                                           uint32_t nrows, uint32_t nchunk, float* __restrict__ a, uint32_t W = 0,
                                                                           uint32_t norm_base = 0, uint32_t pad = 0) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    float m = KVP_NEG_INF, z = 0.f;
    for (uint32_t j = 0; j < nchunk; ++j) softmax_merge(m, z, part_m[(size_t)row * nchunk + j], part_z[(size_t)row * nchunk + j]);
    a[row] = softmax_row_normaliser(m, z, row, W, norm_base, pad);
}
// launch helper: picks the row-per-thread form for many rows with <= 2 partials
#define KVP_SOFTMAX_COMBINE(stream, part_m, part_z, nrows, nchunk, a, ...)                                                                  \
    do {                                                                                                                                 \
        if ((nchunk) <= 2 && (nrows) >= 4096)                                                                                            \
            KVP_LAUNCH("softmax_combine_kernel", stream, softmax_combine_rows_kernel<<<((nrows) + 255) / 256, 256, 0, stream>>>(part_m, part_z, nrows, nchunk, a, ##__VA_ARGS__)); \
        else                                                                                                                             \
            KVP_LAUNCH("softmax_combine_kernel", stream, softmax_combine_kernel<<<((nrows) + 3) / 4, 256, 0, stream>>>(part_m, part_z, nrows, nchunk, a, ##__VA_ARGS__));       \
    } while (0)

// Global max without same-address atomics (2048 atomicMax on one word cost ~12 ns each = 25 us):
// every workgroup stores its maximum to bmax[its linear id]; the (tiny) pad-fill kernel reduces them.
// lds >= 4 words.
__device__ __forceinline__ void block_store_max(float v, float* lds, float* __restrict__ bmax, uint32_t slot) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = lds[0];
        for (uint32_t i = 1; i < (blockDim.x >> 6); ++i) m = fmaxf(m, lds[i]);
        bmax[slot] = m;
    }
}

// scores[b,h, lo + j] = max + 1 for j < n  (F.pad(value=scores.max().item() + 1):
// snapkv_press.py:103 pads the last W positions, expected_attention_press.py:163 the first n_sink).
// max = max over bmax[0..nb) (per-workgroup maxima of the score-producing kernel).
static __global__ __launch_bounds__(256) void fill_pad_kernel(float* __restrict__ scores, uint32_t BH, uint32_t S, uint32_t lo,
                                                              uint32_t n, const float* __restrict__ bmax, uint32_t nb) {
    __shared__ float red[4];
    float m = KVP_NEG_INF;
    for (uint32_t i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, bmax[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float fill = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + 1.0f;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < BH * n) scores[(size_t)(i / n) * S + lo + (i % n)] = fill;
}
