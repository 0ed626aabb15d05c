// kvp_lagkv_score: LagKVPress.score for a sequence of at least n_sink + 2 * lag_size tokens
// (kvpress/presses/lagkv_press.py:56-97).
//
// After n_sink sink tokens the sequence is cut into partitions of lag_size tokens.  Partition p is scored against the
// NEXT one: per channel, min and max over partition p+1; every token of partition p is normalised with them,
// (x - min) / (max - min), and its score is the standard deviation of that vector over the channels (unbiased, torch's
// default), soft-maxed over the partition's tokens; the K and V scores are averaged.  Unless cross_scoring, the score is
// replaced by its rank inside the partition divided by lag_size (`argsort().argsort() / lag_size`).  Sinks, the last
// complete partition and the remainder score 1.
//
// One workgroup (4 waves) per (b, h, partition).  Fast path (rows of whole 16-byte vectors, <= 1 KiB): LPR adjacent lanes own a
// row, one dwordx4 load each (2.8 TB/s of algorithmic bytes at 128k keys); otherwise a wave per token row, lanes across the channels; channel
// min / max of the reference partition in registers -> LDS, two-pass standard deviation in registers, softmax and ranking
// over the <= 1024 tokens of the partition in LDS.  K and V are each read twice (as reference and as data): HBM / L2 bound.
// Ranks are exact integers; equal scores rank by position (torch.argsort leaves their order open).
#include "../kvp_common.h"
#include "../../../include/kvpress_hip_extra.h"

namespace {

constexpr int LG_THREADS = 256;
constexpr int LG_WAVES = LG_THREADS / 64;
constexpr int LG_MAXCH = 8;      // channels per lane: head_dim <= 512
constexpr int LG_MAXLAG = 1024;  // tokens per partition

struct LagArgs {
    const void* k;
    const void* v;
    int64_t k_sb, k_sh, k_ss, v_sb, v_sh, v_ss;  // element strides
    uint32_t H, S, D, n_sink, lag, n_scored;      // n_scored partitions get a real score
    int cross;
};

__device__ void lag_softmax(float* __restrict__ out, uint32_t lag, float* __restrict__ scr);

// softmax(std over channels of the normalised rows) of partition `part` of x, into out[0..lag)
template <int DT>
__device__ void lag_states_score(const typename Elem<DT>::T* __restrict__ base, int64_t ss, uint32_t D, uint32_t lag, uint32_t row0,
                                 float* __restrict__ cmin, float* __restrict__ cmax, float* __restrict__ out, float* __restrict__ scr) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nch = (D + 63) / 64;
    // ---- channel min / max over the reference partition (rows row0 + lag ..) ----
    float mn[LG_MAXCH], mx[LG_MAXCH];
#pragma unroll
    for (int j = 0; j < LG_MAXCH; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
    for (uint32_t r = wv; r < lag; r += LG_WAVES) {
        const typename Elem<DT>::T* p = base + (int64_t)(row0 + lag + r) * ss;
#pragma unroll
        for (int j = 0; j < LG_MAXCH; ++j) {
            const uint32_t c = lane + 64 * j;
            if (j < (int)nch && c < D) {
                const float x = Elem<DT>::ld(p + c);
                mn[j] = fminf(mn[j], x);
                mx[j] = fmaxf(mx[j], x);
            }
        }
    }
    // cross-wave combine through LDS: cmin / cmax [LG_WAVES][D] laid out wave-major in scr-adjacent arrays
#pragma unroll
    for (int j = 0; j < LG_MAXCH; ++j) {
        const uint32_t c = lane + 64 * j;
        if (j < (int)nch && c < D) {
            cmin[wv * D + c] = mn[j];
            cmax[wv * D + c] = mx[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LG_MAXCH; ++j) {
        const uint32_t c = lane + 64 * j;
        if (j < (int)nch && c < D) {
            float a = cmin[c], b = cmax[c];
            for (int w = 1; w < LG_WAVES; ++w) {
                a = fminf(a, cmin[w * D + c]);
                b = fmaxf(b, cmax[w * D + c]);
            }
            mn[j] = a;
            mx[j] = b;
        }
    }
    // ---- std over the channels of (x - min) / (max - min), one wave per row of the scored partition ----
    for (uint32_t r = wv; r < lag; r += LG_WAVES) {
        const typename Elem<DT>::T* p = base + (int64_t)(row0 + r) * ss;
        float y[LG_MAXCH], sum = 0.f;
#pragma unroll
        for (int j = 0; j < LG_MAXCH; ++j) {
            const uint32_t c = lane + 64 * j;
            y[j] = 0.f;
            if (j < (int)nch && c < D) {
                y[j] = (Elem<DT>::ld(p + c) - mn[j]) / (mx[j] - mn[j]);
                sum += y[j];
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < LG_MAXCH; ++j) {
            const uint32_t c = lane + 64 * j;
            if (j < (int)nch && c < D) sq += (y[j] - mean) * (y[j] - mean);
        }
        sq = wave_sum(sq);
        if (lane == 0) out[r] = sqrtf(sq / (float)(D - 1));
    }
    __syncthreads();
    lag_softmax(out, lag, scr);
}

// softmax over out[0..lag), in place (all threads; ends with a barrier)
__device__ void lag_softmax(float* __restrict__ out, uint32_t lag, float* __restrict__ scr) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float m = -INFINITY;
    for (uint32_t r = threadIdx.x; r < lag; r += LG_THREADS) m = fmaxf(m, out[r]);
    m = wave_max(m);
    if (lane == 0) scr[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
    __syncthreads();
    float z = 0.f;
    for (uint32_t r = threadIdx.x; r < lag; r += LG_THREADS) {
        const float e = __expf(out[r] - m);
        out[r] = e;
        z += e;
    }
    z = wave_sum(z);
    if (lane == 0) scr[wv] = z;
    __syncthreads();
    z = (scr[0] + scr[1]) + (scr[2] + scr[3]);
    for (uint32_t r = threadIdx.x; r < lag; r += LG_THREADS) out[r] = out[r] / z;
    __syncthreads();
}

// average of the K and V scores, then the raw score (cross_scoring) or its rank inside the partition / lag_size
__device__ void lag_finish(float* __restrict__ sk, const float* __restrict__ sv, const LagArgs& a, float* __restrict__ out) {
    for (uint32_t i = threadIdx.x; i < a.lag; i += LG_THREADS) sk[i] = (sk[i] + sv[i]) / 2.f;
    __syncthreads();
    if (a.cross) {
        for (uint32_t i = threadIdx.x; i < a.lag; i += LG_THREADS) out[i] = sk[i];
        return;
    }
    // rank inside the partition: #(smaller scores) + #(equal scores at lower positions)
    for (uint32_t i = threadIdx.x; i < a.lag; i += LG_THREADS) {
        const float s = sk[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < a.lag; ++j) {
            const float t = sk[j];
            rank += (t < s || (t == s && j < i)) ? 1u : 0u;
        }
        out[i] = (float)rank / (float)a.lag;
    }
}

template <int DT>
__global__ __launch_bounds__(LG_THREADS) void lagkv_score_kernel(LagArgs a, float* __restrict__ scores) {
    extern __shared__ float lg_lds[];
    float* sk = lg_lds;                 // [lag]
    float* sv = sk + a.lag;             // [lag]
    float* cmin = sv + a.lag;           // [LG_WAVES][D]
    float* cmax = cmin + LG_WAVES * a.D;
    __shared__ float scr[LG_WAVES];
    using T = typename Elem<DT>::T;
    const uint32_t part = blockIdx.x, bh = blockIdx.y;
    const uint32_t b = bh / a.H, h = bh - b * a.H;
    const uint32_t row0 = a.n_sink + part * a.lag;
    lag_states_score<DT>(static_cast<const T*>(a.k) + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh, a.k_ss, a.D, a.lag, row0, cmin, cmax, sk, scr);
    lag_states_score<DT>(static_cast<const T*>(a.v) + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh, a.v_ss, a.D, a.lag, row0, cmin, cmax, sv, scr);
    lag_finish(sk, sv, a, scores + (size_t)bh * a.S + row0);
}

// ---- the same through 16-byte loads: LPR adjacent lanes own one row (rownorm.hip's layout) -------------------------------
// a lane keeps min / max of ITS 8 (4 for fp32) channels over the reference rows it visits; the block's row groups are
// combined through LDS; the per-row mean and variance are xor-shuffle reductions over the LPR lanes of the row.
template <int DT, int LPR>
__device__ void lag_states_score_vec(const typename Elem<DT>::T* __restrict__ base, int64_t ss, uint32_t chunks, uint32_t lag, uint32_t row0,
                                     float* __restrict__ cmin, float* __restrict__ cmax, float* __restrict__ out, float* __restrict__ scr) {
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = LG_THREADS / LPR;
    const uint32_t lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const uint32_t D = chunks * PER16;
    const bool live = lir < chunks;
    float mn[PER16], mx[PER16];
#pragma unroll
    for (int j = 0; j < PER16; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
    if (live)
        for (uint32_t r = grp; r < lag; r += GPB) {
            float f[PER16];
            unpack16<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(row0 + lag + r) * ss + (size_t)lir * PER16), f);
#pragma unroll
            for (int j = 0; j < PER16; ++j) { mn[j] = fminf(mn[j], f[j]); mx[j] = fmaxf(mx[j], f[j]); }
        }
    if (live) {
#pragma unroll
        for (int j = 0; j < PER16; ++j) {
            cmin[grp * D + lir * PER16 + j] = mn[j];
            cmax[grp * D + lir * PER16 + j] = mx[j];
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int j = 0; j < PER16; ++j) {
            float lo = INFINITY, hi = -INFINITY;
            for (int g = 0; g < GPB; ++g) {
                lo = fminf(lo, cmin[g * D + lir * PER16 + j]);
                hi = fmaxf(hi, cmax[g * D + lir * PER16 + j]);
            }
            mn[j] = lo;
            mx[j] = hi;
        }
    }
    // rows of the scored partition: every row group walks the same number of rounds (the shuffles below need whole waves)
    for (uint32_t r0 = 0; r0 < lag; r0 += GPB) {
        const uint32_t r = r0 + grp;
        float y[PER16], sum = 0.f;
#pragma unroll
        for (int j = 0; j < PER16; ++j) y[j] = 0.f;
        if (live && r < lag) {
            float f[PER16];
            unpack16<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(row0 + r) * ss + (size_t)lir * PER16), f);
#pragma unroll
            for (int j = 0; j < PER16; ++j) {
                y[j] = (f[j] - mn[j]) / (mx[j] - mn[j]);
                sum += y[j];
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)D;
        float sq = 0.f;
        if (live && r < lag) {
#pragma unroll
            for (int j = 0; j < PER16; ++j) sq += (y[j] - mean) * (y[j] - mean);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (lir == 0 && r < lag) out[r] = sqrtf(sq / (float)(D - 1));
    }
    __syncthreads();
    lag_softmax(out, lag, scr);
}

template <int DT, int LPR>
__global__ __launch_bounds__(LG_THREADS) void lagkv_score_vec_kernel(LagArgs a, uint32_t chunks, float* __restrict__ scores) {
    extern __shared__ float lg_lds[];
    constexpr int GPB = LG_THREADS / LPR;
    float* sk = lg_lds;
    float* sv = sk + a.lag;
    float* cmin = sv + a.lag;            // [GPB][D]
    float* cmax = cmin + GPB * a.D;
    __shared__ float scr[LG_WAVES];
    using T = typename Elem<DT>::T;
    const uint32_t part = blockIdx.x, bh = blockIdx.y;
    const uint32_t b = bh / a.H, h = bh - b * a.H;
    const uint32_t row0 = a.n_sink + part * a.lag;
    lag_states_score_vec<DT, LPR>(static_cast<const T*>(a.k) + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh, a.k_ss, chunks, a.lag, row0, cmin, cmax, sk, scr);
    lag_states_score_vec<DT, LPR>(static_cast<const T*>(a.v) + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh, a.v_ss, chunks, a.lag, row0, cmin, cmax, sv, scr);
    lag_finish(sk, sv, a, scores + (size_t)bh * a.S + row0);
}

// sinks, the last complete partition and the remainder: 1
__global__ __launch_bounds__(256) void lagkv_ones_kernel(float* __restrict__ scores, uint32_t S, uint32_t n_sink, uint32_t scored_end) {
    float* row = scores + (size_t)blockIdx.y * S;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x)
        if (i < n_sink || i >= scored_end) row[i] = 1.0f;
}

}  // namespace

extern "C" int kvp_lagkv_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh,
                               int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t n_sink, int64_t lag_size,
                               int cross_scoring, float* scores, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "lagkv: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 1 && H >= 1 && D >= 2 && n_sink >= 0 && lag_size >= 1, "lagkv: bad shape B=%ld H=%ld D=%ld n_sink=%ld lag=%ld", (long)B,
                  (long)H, (long)D, (long)n_sink, (long)lag_size);
    KVP_CHECK_ARG(S >= n_sink + 2 * lag_size, "lagkv: needs at least n_sink + 2 * lag_size = %ld tokens (got %ld)", (long)(n_sink + 2 * lag_size), (long)S);
    KVP_CHECK_ARG(D <= 64 * LG_MAXCH && lag_size <= LG_MAXLAG && S < ((int64_t)1 << 31) && B * H <= 65535, "lagkv: head_dim > %d, lag_size > %d or shape too large",
                  64 * LG_MAXCH, LG_MAXLAG);
    KVP_CHECK_ARG(k && v && scores, "lagkv: null pointer");
    LagArgs a;
    a.k = k; a.v = v;
    a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss; a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss;
    a.H = (uint32_t)H; a.S = (uint32_t)S; a.D = (uint32_t)D; a.n_sink = (uint32_t)n_sink; a.lag = (uint32_t)lag_size;
    const int64_t n_part = (S - n_sink) / lag_size;   // complete partitions; the last one only serves as a reference (:71-73)
    a.n_scored = (uint32_t)(n_part - 1);
    a.cross = cross_scoring ? 1 : 0;
    const uint32_t BH = (uint32_t)(B * H);
    const uint32_t scored_end = (uint32_t)(n_sink + (n_part - 1) * lag_size);
    KVP_LAUNCH("lagkv_ones_kernel", stream, lagkv_ones_kernel<<<dim3((uint32_t)std::min<int64_t>((S + 255) / 256, 1024), BH), 256, 0, stream>>>(scores, (uint32_t)S, (uint32_t)n_sink, scored_end));
    const dim3 grid(a.n_scored, BH);
    const size_t es = (size_t)kvp_elem_size(dtype), rowbytes = (size_t)D * es;
    auto al16 = [&](int64_t st) { return ((size_t)st * es) % 16 == 0; };
    const bool vec = rowbytes % 16 == 0 && rowbytes <= 1024 && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && al16(k_sb) && al16(k_sh) && al16(k_ss) &&
                     al16(v_sb) && al16(v_sh) && al16(v_ss);
    if (a.n_scored && vec) {
        const uint32_t chunks = (uint32_t)(rowbytes / 16);
        int lpr = 1;
        while (lpr < 64 && (uint32_t)lpr < chunks) lpr <<= 1;
        const size_t ldsv = ((size_t)2 * lag_size + (size_t)2 * (LG_THREADS / lpr) * D) * 4;
#define KVP_LG_VEC(DT, L)                                                                                                                  \
    case L:                                                                                                                                \
        KVP_LAUNCH("lagkv_score_vec_kernel", stream, (lagkv_score_vec_kernel<DT, L><<<grid, LG_THREADS, ldsv, stream>>>(a, chunks, scores))); \
        break;
#define KVP_LG_DT(DT) switch (lpr) { KVP_LG_VEC(DT, 1) KVP_LG_VEC(DT, 2) KVP_LG_VEC(DT, 4) KVP_LG_VEC(DT, 8) KVP_LG_VEC(DT, 16) KVP_LG_VEC(DT, 32) KVP_LG_VEC(DT, 64) }
        if (dtype == KVP_F32) { KVP_LG_DT(KVP_F32) }
        else if (dtype == KVP_F16) { KVP_LG_DT(KVP_F16) }
        else { KVP_LG_DT(KVP_BF16) }
#undef KVP_LG_DT
#undef KVP_LG_VEC
        KVP_CHECK_LAUNCH("lagkv");
        return KVP_OK;
    }
    const size_t lds = ((size_t)2 * lag_size + (size_t)2 * LG_WAVES * D) * 4;
    if (a.n_scored) {
        if (dtype == KVP_F32) KVP_LAUNCH("lagkv_score_kernel", stream, lagkv_score_kernel<KVP_F32><<<grid, LG_THREADS, lds, stream>>>(a, scores));
        else if (dtype == KVP_F16) KVP_LAUNCH("lagkv_score_kernel", stream, lagkv_score_kernel<KVP_F16><<<grid, LG_THREADS, lds, stream>>>(a, scores));
        else KVP_LAUNCH("lagkv_score_kernel", stream, lagkv_score_kernel<KVP_BF16><<<grid, LG_THREADS, lds, stream>>>(a, scores));
    }
    KVP_CHECK_LAUNCH("lagkv");
    return KVP_OK;
}
