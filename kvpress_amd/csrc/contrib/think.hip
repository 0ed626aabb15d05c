// ThinKPress (kvpress/presses/think_press.py:56-85): prune key CHANNELS.
//   kvp_think_channel_scores: scores[b,h,d] = mean_g mean_w q[b,hq,w,d]^2  *  mean_s k[b,h,s,d]^2        (:72-76)
//   kvp_zero_channels:        k[b,h,s,idx[b,h,j]] = 0 for every s, in place                              (:81-82)
// (the selection of the lowest-scoring channels in between is kvp_topk_select | KVP_TOPK_SMALLEST on [B*H, D] rows.)
//
// Both are single streaming passes over K (HBM-bound): column sums of squares per (b, h) through per-workgroup partials
// (fixed summation order, no float atomics), and a masked rewrite of the rows that carry a pruned channel.
#include "../kvp_common.h"
#include "../../../include/kvpress_hip_extra.h"

namespace {

constexpr int TH_THREADS = 256;
constexpr int TH_ROWS = 512;  // key rows per workgroup in the column reduction

// partial[bh][chunk][d] = sum over the chunk's rows of x[b,h,s,d]^2 ; thread t: channel t % Dp, row phase t / Dp
template <int DT>
__global__ __launch_bounds__(TH_THREADS) void colsumsq_kernel(const typename Elem<DT>::T* __restrict__ x, int64_t sb, int64_t sh, int64_t ss,
                                                              uint32_t H, uint32_t S, uint32_t D, uint32_t Dp, float* __restrict__ partial) {
    extern __shared__ float th_lds[];  // [phases][D]
    const uint32_t bh = blockIdx.y, chunk = blockIdx.x;
    const uint32_t b = bh / H, h = bh - b * H;
    const typename Elem<DT>::T* base = x + (int64_t)b * sb + (int64_t)h * sh;
    const uint32_t c = threadIdx.x % Dp, ph = threadIdx.x / Dp, nph = TH_THREADS / Dp;
    const uint32_t r0 = chunk * TH_ROWS, r1 = min(r0 + TH_ROWS, S);
    float acc = 0.f;
    if (c < D && ph < nph)
        for (uint32_t r = r0 + ph; r < r1; r += nph) {
            const float v = Elem<DT>::ld(base + (int64_t)r * ss + c);
            acc = fmaf(v, v, acc);
        }
    if (c < D && ph < nph) th_lds[ph * D + c] = acc;
    __syncthreads();
    if (threadIdx.x < D) {
        float s = 0.f;
        for (uint32_t p = 0; p < nph; ++p) s += th_lds[p * D + threadIdx.x];
        partial[((size_t)bh * gridDim.x + chunk) * D + threadIdx.x] = s;
    }
}

// the same partials through 16-byte loads: LPR adjacent lanes own one row (rownorm.hip's layout), a lane keeps the sums of its
// 8 (4 for fp32) channels over the rows it visits, the row groups of the block are combined through LDS.  grid.x blocks stride
// over the rows of (b, h) together; partial[bh][block][d].
template <int DT, int LPR>
__global__ __launch_bounds__(TH_THREADS) void colsumsq_vec_kernel(const typename Elem<DT>::T* __restrict__ x, int64_t sb, int64_t sh, int64_t ss,
                                                                  uint32_t H, uint32_t S, uint32_t chunks, float* __restrict__ partial) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = TH_THREADS / LPR;
    constexpr int UNROLL = 4;
    __shared__ float red[GPB][LPR * PER16 + 1];
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / H, h = bh - b * H;
    const T* __restrict__ base = x + (int64_t)b * sb + (int64_t)h * sh;
    const uint32_t lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const uint32_t g = blockIdx.x * GPB + grp, TG = gridDim.x * GPB;
    float acc[PER16];
#pragma unroll
    for (int j = 0; j < PER16; ++j) acc[j] = 0.f;
    for (uint32_t s0 = g; s0 < S; s0 += TG * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && lir < chunks) v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)s * ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float f[PER16];
            unpack16<DT>(v[u], f);
#pragma unroll
            for (int j = 0; j < PER16; ++j) acc[j] = fmaf(f[j], f[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < PER16; ++j) red[grp][lir * PER16 + j] = acc[j];
    __syncthreads();
    const uint32_t D = chunks * PER16;
    float* __restrict__ out = partial + ((size_t)bh * gridDim.x + blockIdx.x) * D;
    for (uint32_t d = threadIdx.x; d < D; d += TH_THREADS) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < GPB; ++r) t += red[r][d];
        out[d] = t;
    }
}

// scores[bh][d] = (mean over the group's q-heads and window rows of q^2) * (sum of the partials / S)
template <int DT>
__global__ __launch_bounds__(TH_THREADS) void think_finish_kernel(const float* __restrict__ partial, uint32_t nchunk, const typename Elem<DT>::T* __restrict__ q,
                                                                  int64_t q_sb, int64_t q_sh, int64_t q_sw, uint32_t H, uint32_t G, uint32_t W, uint32_t S,
                                                                  uint32_t D, float* __restrict__ scores) {
    const uint32_t bh = blockIdx.x;
    const uint32_t b = bh / H, h = bh - b * H;
    for (uint32_t d = threadIdx.x; d < D; d += TH_THREADS) {
        float ks = 0.f;
        for (uint32_t c = 0; c < nchunk; ++c) ks += partial[((size_t)bh * nchunk + c) * D + d];
        float qs = 0.f;
        for (uint32_t g = 0; g < G; ++g) {
            const typename Elem<DT>::T* qp = q + (int64_t)b * q_sb + (int64_t)(h * G + g) * q_sh + d;
            float a = 0.f;
            for (uint32_t w = 0; w < W; ++w) {
                const float v = Elem<DT>::ld(qp + (int64_t)w * q_sw);
                a = fmaf(v, v, a);
            }
            qs += a / (float)W;
        }
        scores[(size_t)bh * D + d] = (qs / (float)G) * (ks / (float)S);
    }
}

// rows of k get the channels listed in idx[bh][0..n) zeroed; one thread per (row, listed channel)
template <int DT>
__global__ __launch_bounds__(TH_THREADS) void zero_channels_kernel(typename Elem<DT>::T* __restrict__ x, int64_t sb, int64_t sh, int64_t ss, uint32_t H,
                                                                   uint32_t S, uint32_t D, const int32_t* __restrict__ idx, uint32_t n) {
    __shared__ int32_t ch[1024];
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / H, h = bh - b * H;
    for (uint32_t j = threadIdx.x; j < n; j += TH_THREADS) ch[j] = idx[(size_t)bh * n + j];
    __syncthreads();
    typename Elem<DT>::T* base = x + (int64_t)b * sb + (int64_t)h * sh;
    const uint64_t total = (uint64_t)S * n;
    for (uint64_t i = (uint64_t)blockIdx.x * TH_THREADS + threadIdx.x; i < total; i += (uint64_t)gridDim.x * TH_THREADS) {
        const uint32_t r = (uint32_t)(i / n), j = (uint32_t)(i - (uint64_t)r * n);
        const int32_t c = ch[j];
        if (c >= 0 && (uint32_t)c < D) base[(int64_t)r * ss + c] = (typename Elem<DT>::T)0;
    }
}

}  // namespace

extern "C" size_t kvp_think_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t D) {
    if (B < 1 || H < 1 || S < 1 || D < 1) return 256;
    return kvp_align_up((size_t)B * H * ((S + TH_ROWS - 1) / TH_ROWS) * D * 4, 256);
}

extern "C" int kvp_think_channel_scores(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* k, int64_t k_sb, int64_t k_sh,
                                        int64_t k_ss, int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D,
                                        float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "think: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && S >= 1 && W >= 1 && D >= 1 && D <= TH_THREADS,
                  "think: bad shape B=%ld Hq=%ld Hkv=%ld S=%ld W=%ld D=%ld (head_dim <= %d)", (long)B, (long)Hq, (long)Hkv, (long)S, (long)W, (long)D, TH_THREADS);
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B * Hkv <= 65535, "think: shape too large");
    KVP_CHECK_ARG(q && k && scores, "think: null pointer");
    const size_t need = kvp_think_workspace_bytes(B, Hkv, S, D);
    if (!ws || ws_bytes < need) {
        kvp_set_error("think: workspace too small (%zu < %zu)", ws_bytes, need);
        return KVP_EWORKSPACE;
    }
    uint32_t nchunk = (uint32_t)((S + TH_ROWS - 1) / TH_ROWS);
    const uint32_t BH = (uint32_t)(B * Hkv);
    const size_t es = (size_t)kvp_elem_size(dtype), rowbytes = (size_t)D * es;
    const bool vec = rowbytes % 16 == 0 && rowbytes <= 1024 && ((uintptr_t)k % 16 == 0) && (k_sb * es) % 16 == 0 && (k_sh * es) % 16 == 0 &&
                     (k_ss * es) % 16 == 0;
    if (vec) nchunk = std::max<uint32_t>(1, std::min<uint32_t>(nchunk, 2048 / std::max<uint32_t>(1, BH)));  // blocks striding over the rows
    uint32_t Dp = 1;
    while (Dp < (uint32_t)D) Dp <<= 1;   // power of two: TH_THREADS / Dp row phases
    const size_t lds = (size_t)(TH_THREADS / Dp) * D * 4;
    float* partial = static_cast<float*>(ws);
    const uint32_t chunks = (uint32_t)(rowbytes / 16);
    int lpr = 1;
    while (lpr < 64 && (uint32_t)lpr < chunks) lpr <<= 1;
#define KVP_TH_VEC(DT, L)                                                                                                                    \
    case L:                                                                                                                                  \
        KVP_LAUNCH("colsumsq_vec_kernel", stream, (colsumsq_vec_kernel<DT, L><<<dim3(nchunk, BH), TH_THREADS, 0, stream>>>(static_cast<const Elem<DT>::T*>(k), k_sb, k_sh, \
                                                                                                                         k_ss, (uint32_t)Hkv, (uint32_t)S, chunks, partial))); \
        break;
#define KVP_TH(DT)                                                                                                                           \
    if (vec) {                                                                                                                               \
        switch (lpr) { KVP_TH_VEC(DT, 1) KVP_TH_VEC(DT, 2) KVP_TH_VEC(DT, 4) KVP_TH_VEC(DT, 8) KVP_TH_VEC(DT, 16) KVP_TH_VEC(DT, 32) KVP_TH_VEC(DT, 64) } \
    } else                                                                                                                                   \
        KVP_LAUNCH("colsumsq_kernel", stream, colsumsq_kernel<DT><<<dim3(nchunk, BH), TH_THREADS, lds, stream>>>(static_cast<const Elem<DT>::T*>(k), k_sb, k_sh, k_ss, \
                                                                                                                 (uint32_t)Hkv, (uint32_t)S, (uint32_t)D, Dp, partial)); \
    KVP_LAUNCH("think_finish_kernel", stream, think_finish_kernel<DT><<<BH, TH_THREADS, 0, stream>>>(partial, nchunk, static_cast<const Elem<DT>::T*>(q), q_sb, q_sh, \
                                                                                                     q_sw, (uint32_t)Hkv, (uint32_t)(Hq / Hkv), (uint32_t)W, (uint32_t)S, (uint32_t)D, scores))
    if (dtype == KVP_F32) { KVP_TH(KVP_F32); }
    else if (dtype == KVP_F16) { KVP_TH(KVP_F16); }
    else { KVP_TH(KVP_BF16); }
#undef KVP_TH
#undef KVP_TH_VEC
    KVP_CHECK_LAUNCH("think(channel scores)");
    return KVP_OK;
}

extern "C" int kvp_zero_channels(void* x, int64_t sb, int64_t sh, int64_t ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                                 const int32_t* idx, int64_t n, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "zero_channels: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 1 && n >= 0 && n <= 1024 && B * H <= 65535, "zero_channels: bad shape B=%ld H=%ld S=%ld D=%ld n=%ld",
                  (long)B, (long)H, (long)S, (long)D, (long)n);
    if (B * H * S * n == 0) return KVP_OK;
    KVP_CHECK_ARG(x && idx, "zero_channels: null pointer");
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((S * n + TH_THREADS - 1) / TH_THREADS, 2048));
#define KVP_ZC(DT)                                                                                                                             \
    KVP_LAUNCH("zero_channels_kernel", stream, zero_channels_kernel<DT><<<dim3(bx, (uint32_t)(B * H)), TH_THREADS, 0, stream>>>(static_cast<Elem<DT>::T*>(x), sb, sh, ss, \
                                                                                                                                  (uint32_t)H, (uint32_t)S, (uint32_t)D, idx, (uint32_t)n))
    if (dtype == KVP_F32) KVP_ZC(KVP_F32);
    else if (dtype == KVP_F16) KVP_ZC(KVP_F16);
    else KVP_ZC(KVP_BF16);
#undef KVP_ZC
    KVP_CHECK_LAUNCH("zero_channels");
    return KVP_OK;
}
