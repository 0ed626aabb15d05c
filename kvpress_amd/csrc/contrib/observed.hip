// kvp_observed_attention_score: ObservedAttentionPress.score (kvpress/presses/observed_attention_press.py:42-48)
//   scores[b,h,s] = mean over the G q-heads of the kv-head of ( sum_q attn[b,hq,q,s] / (S - s) )
// from the attention weights the (eager) attention layer returned, attn [B,Hq,Sq,S] in the model dtype.
//
// HBM-bound column sums over a B*Hq*Sq*S matrix that the eager attention has just written: one thread per (b, h, column),
// consecutive threads on consecutive columns (coalesced rows), four independent accumulators per thread over the query
// rows; fp32 sums in a fixed order (no float atomics: deterministic).
#include "../kvp_common.h"
#include "../../../include/kvpress_hip_extra.h"

namespace {

constexpr int OA_THREADS = 256;

template <int DT>
__global__ __launch_bounds__(OA_THREADS) void observed_colmean_kernel(const typename Elem<DT>::T* __restrict__ attn, int64_t sb, int64_t sh,
                                                                      int64_t sq, uint32_t Hkv, uint32_t G, uint32_t Sq, uint32_t S,
                                                                      float* __restrict__ scores) {
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / Hkv, h = bh - b * Hkv;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < S; c += gridDim.x * blockDim.x) {
        const float inv_count = 1.0f / (float)(S - c);   // torch.arange(n_tokens, 0, -1): the queries that can see key c
        float total = 0.f;
        for (uint32_t g = 0; g < G; ++g) {
            const typename Elem<DT>::T* p = attn + (int64_t)b * sb + (int64_t)(h * G + g) * sh + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            uint32_t q = 0;
            for (; q + 4 <= Sq; q += 4) {
                a0 += Elem<DT>::ld(p + (int64_t)(q + 0) * sq);
                a1 += Elem<DT>::ld(p + (int64_t)(q + 1) * sq);
                a2 += Elem<DT>::ld(p + (int64_t)(q + 2) * sq);
                a3 += Elem<DT>::ld(p + (int64_t)(q + 3) * sq);
            }
            for (; q < Sq; ++q) a0 += Elem<DT>::ld(p + (int64_t)q * sq);
            total += ((a0 + a1) + (a2 + a3)) * inv_count;
        }
        scores[(size_t)bh * S + c] = total / (float)G;
    }
}

}  // namespace

extern "C" int kvp_observed_attention_score(const void* attn, int64_t a_sb, int64_t a_sh, int64_t a_sq, int dtype, int64_t B, int64_t Hq,
                                            int64_t Hkv, int64_t Sq, int64_t S, float* scores, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "observed_attention: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && Sq >= 0 && S >= 0, "observed_attention: bad shape B=%ld Hq=%ld Hkv=%ld Sq=%ld S=%ld",
                  (long)B, (long)Hq, (long)Hkv, (long)Sq, (long)S);
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && Sq < ((int64_t)1 << 31) && B * Hkv <= 65535, "observed_attention: shape too large");
    if (S == 0) return KVP_OK;
    KVP_CHECK_ARG(attn && scores, "observed_attention: null pointer");
    const uint32_t BH = (uint32_t)(B * Hkv);
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((S + OA_THREADS - 1) / OA_THREADS, 4096));
#define KVP_OA(DT)                                                                                                                       \
    KVP_LAUNCH("observed_colmean_kernel", stream, observed_colmean_kernel<DT><<<dim3(bx, BH), OA_THREADS, 0, stream>>>(                 \
        static_cast<const Elem<DT>::T*>(attn), a_sb, a_sh, a_sq, (uint32_t)Hkv, (uint32_t)(Hq / Hkv), (uint32_t)Sq, (uint32_t)S, scores))
    if (dtype == KVP_F32) KVP_OA(KVP_F32);
    else if (dtype == KVP_F16) KVP_OA(KVP_F16);
    else KVP_OA(KVP_BF16);
#undef KVP_OA
    KVP_CHECK_LAUNCH("observed_attention");
    return KVP_OK;
}
