// kvp_keydiff_score: out[b,h,s] = -cos(k[b,h,s,:], anchor[b,h,:]),  anchor = mean_s k / max(||k||, 1e-12)
// Replaces `anchor = F.normalize(keys, p=2, dim=-1).mean(dim=2, keepdim=True); -F.cosine_similarity(keys, anchor, dim=-1)`
// (kvpress/presses/keydiff_press.py:45-46).  cosine_similarity(x, y) = sum (x / max(||x||, 1e-8)) * (y / max(||y||, 1e-8)).
//
// Two HBM-bound streaming passes over K (algorithmic bytes = 2 * B*H*S*D*esize):
//   pass A  per workgroup: column sums of the normalised rows of its row slice -> partial[bh][wg][D]   (deterministic:
//           no float atomics), then a small reduce kernel -> anchor[bh][D] already divided by max(||anchor||, 1e-8)
//   pass B  per row: dot(k, anchor_unit) / max(||k||, 1e-8)
// Fast path as in rownorm.hip: LPR adjacent lanes own one row (16-byte vectors, rows of <= 1 KiB); any other shape
// takes the scalar kernels (tiny test geometries such as head_dim 6).
#include "kvp_common.h"

namespace {

constexpr int KD_THREADS = 256;
constexpr int KD_UNROLL = 4;
constexpr float KD_EPS_NORMALIZE = 1e-12f;  // F.normalize default eps
constexpr float KD_EPS_COS = 1e-8f;         // F.cosine_similarity default eps

struct KdMap {
    uint32_t H, S;
    int64_t sb, sh, ss;  // element strides
};

// ---- pass A (vector path) ---------------------------------------------------------------------
// rows_per_wg == 0: the workgroups interleave (row group g, then g + all groups, ...); > 0: every workgroup streams ONE contiguous
// range of rows (the walk of topk_cluster.hip's Knorm mode).
template <int DT, int LPR, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void keydiff_anchor_vec_kernel(const typename Elem<DT>::T* __restrict__ x, KdMap map,
                                                                     uint32_t chunks, float* __restrict__ partial, uint32_t rows_per_wg) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = THREADS / LPR;
    __shared__ float red[GPB][LPR * PER16 + 1];
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const T* __restrict__ base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    const uint32_t lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const uint32_t g = rows_per_wg ? blockIdx.x * rows_per_wg + grp : blockIdx.x * GPB + grp;
    const uint32_t TG = rows_per_wg ? GPB : gridDim.x * GPB;
    const uint32_t S = rows_per_wg ? min(map.S, (blockIdx.x + 1) * rows_per_wg) : map.S;

    float acc[PER16];
#pragma unroll
    for (int j = 0; j < PER16; ++j) acc[j] = 0.f;
    for (uint32_t s0 = g; s0 < S; s0 += TG * KD_UNROLL) {
        uint4 v[KD_UNROLL];
#pragma unroll
        for (int u = 0; u < KD_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && lir < chunks) v[u] = ld16<NT>(base + (int64_t)s * map.ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < KD_UNROLL; ++u) {
            float f[PER16];
            unpack16<DT>(v[u], f);
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < PER16; ++j) ss = fmaf(f[j], f[j], ss);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            const float inv = 1.f / fmaxf(sqrtf(ss), KD_EPS_NORMALIZE);  // rows past S are all-zero: contribute 0
#pragma unroll
            for (int j = 0; j < PER16; ++j) acc[j] = fmaf(f[j], inv, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < PER16; ++j) red[grp][lir * PER16 + j] = acc[j];
    __syncthreads();
    const uint32_t D = chunks * PER16;
    float* __restrict__ out = partial + ((size_t)bh * gridDim.x + blockIdx.x) * D;
    for (uint32_t d = threadIdx.x; d < D; d += THREADS) {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < GPB; ++r) s += red[r][d];
        out[d] = s;
    }
}

// ---- pass A (scalar path): one workgroup per (row slice, bh); thread d walks column d --------------------------
template <int DT>
__global__ __launch_bounds__(KD_THREADS) void keydiff_anchor_scalar_kernel(const typename Elem<DT>::T* __restrict__ x, KdMap map,
                                                                           uint32_t D, uint32_t rows_per_wg,
                                                                           float* __restrict__ partial) {
    __shared__ float inv[KD_THREADS];
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const typename Elem<DT>::T* base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    const uint32_t s_beg = blockIdx.x * rows_per_wg, s_end = min(map.S, s_beg + rows_per_wg);
    float* __restrict__ out = partial + ((size_t)bh * gridDim.x + blockIdx.x) * D;
    for (uint32_t d0 = 0; d0 < D; d0 += KD_THREADS) {
        const uint32_t d = d0 + threadIdx.x;
        float acc = 0.f;
        for (uint32_t s0 = s_beg; s0 < s_end; s0 += KD_THREADS) {
            const uint32_t s = s0 + threadIdx.x;  // thread t normalises row s0 + t
            float ss = 0.f;
            if (s < s_end)
                for (uint32_t e = 0; e < D; ++e) {
                    const float f = Elem<DT>::ld(base + (int64_t)s * map.ss + e);
                    ss = fmaf(f, f, ss);
                }
            __syncthreads();
            inv[threadIdx.x] = 1.f / fmaxf(sqrtf(ss), KD_EPS_NORMALIZE);
            __syncthreads();
            if (d < D)
                for (uint32_t r = 0; r < KD_THREADS && s0 + r < s_end; ++r)
                    acc = fmaf(Elem<DT>::ld(base + (int64_t)(s0 + r) * map.ss + d), inv[r], acc);
        }
        if (d < D) out[d] = acc;
    }
}

// anchor[bh][d] = (sum over workgroups of partial) / S, then scaled to unit length (cosine_similarity's y / max(||y||, eps)).
// 1024 threads per (b, h): `stripes` threads share a dimension and add interleaved subsets of the nwg partial rows
// (independent loads in flight), the stripes are then added in a fixed order -- deterministic.  (One thread per dimension
// walking all 256 partial rows one dependent load after the other took 64 us at 8 x 131072: more than a pass over K.)
constexpr int KD_RED_THREADS = 1024;
__global__ __launch_bounds__(KD_RED_THREADS) void keydiff_anchor_reduce_kernel(const float* __restrict__ partial, uint32_t nwg, uint32_t D,
                                                                               uint32_t S, float* __restrict__ anchor) {
    __shared__ float red[KD_RED_THREADS];
    __shared__ float wsum[KD_RED_THREADS / 64];
    const uint32_t bh = blockIdx.x;
    const float* p = partial + (size_t)bh * nwg * D;
    float* a = anchor + (size_t)bh * D;
    uint32_t DB = 1;                                   // dimensions per sweep: the power of two >= min(D, 1024)
    while (DB < D && DB < KD_RED_THREADS) DB <<= 1;
    const uint32_t stripes = KD_RED_THREADS / DB;
    const uint32_t dl = threadIdx.x % DB, st = threadIdx.x / DB;
    float ss = 0.f;
    for (uint32_t db = 0; db < D; db += DB) {
        const uint32_t d = db + dl;
        float s = 0.f;
        if (d < D) {
#pragma unroll 8
            for (uint32_t w = st; w < nwg; w += stripes) s += p[(size_t)w * D + d];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (st == 0 && d < D) {
            float tot = 0.f;
            for (uint32_t i = 0; i < stripes; ++i) tot += red[i * DB + dl];
            tot /= (float)S;
            a[d] = tot;
            ss = fmaf(tot, tot, ss);
        }
        __syncthreads();
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < KD_RED_THREADS / 64; ++i) tot += wsum[i];
    const float inv = 1.f / fmaxf(sqrtf(tot), KD_EPS_COS);
    if (st == 0)
        for (uint32_t d = dl; d < D; d += DB) a[d] *= inv;  // the same thread wrote a[d]
}

// ---- pass B -------------------------------------------------------------------------------------
template <int DT, int LPR, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void keydiff_score_vec_kernel(const typename Elem<DT>::T* __restrict__ x, KdMap map,
                                                                    uint32_t chunks, const float* __restrict__ anchor,
                                                                    float* __restrict__ out, uint32_t rows_per_wg) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = THREADS / LPR;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const T* __restrict__ base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    float* __restrict__ ob = out + (size_t)bh * map.S;
    const uint32_t lir = threadIdx.x % LPR;
    const uint32_t g = (rows_per_wg ? blockIdx.x * rows_per_wg : blockIdx.x * GPB) + threadIdx.x / LPR;
    const uint32_t TG = rows_per_wg ? GPB : gridDim.x * GPB;
    const uint32_t S = rows_per_wg ? min(map.S, (blockIdx.x + 1) * rows_per_wg) : map.S;
    float an[PER16];
#pragma unroll
    for (int j = 0; j < PER16; ++j) an[j] = lir < chunks ? anchor[(size_t)bh * chunks * PER16 + lir * PER16 + j] : 0.f;

    for (uint32_t s0 = g; s0 < S; s0 += TG * KD_UNROLL) {
        uint4 v[KD_UNROLL];
#pragma unroll
        for (int u = 0; u < KD_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && lir < chunks) v[u] = ld16<NT>(base + (int64_t)s * map.ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < KD_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            float f[PER16];
            unpack16<DT>(v[u], f);
            float ss = 0.f, dot = 0.f;
#pragma unroll
            for (int j = 0; j < PER16; ++j) {
                ss = fmaf(f[j], f[j], ss);
                dot = fmaf(f[j], an[j], dot);
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) {
                ss += __shfl_xor(ss, o);
                dot += __shfl_xor(dot, o);
            }
            if (lir == 0 && s < S) ob[s] = -dot / fmaxf(sqrtf(ss), KD_EPS_COS);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(KD_THREADS) void keydiff_score_scalar_kernel(const typename Elem<DT>::T* __restrict__ x, KdMap map,
                                                                          uint32_t D, const float* __restrict__ anchor,
                                                                          float* __restrict__ out) {
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const typename Elem<DT>::T* base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    const float* a = anchor + (size_t)bh * D;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < map.S; s += gridDim.x * blockDim.x) {
        const typename Elem<DT>::T* p = base + (int64_t)s * map.ss;
        float ss = 0.f, dot = 0.f;
        for (uint32_t d = 0; d < D; ++d) {
            const float f = Elem<DT>::ld(p + d);
            ss = fmaf(f, f, ss);
            dot = fmaf(f, a[d], dot);
        }
        out[(size_t)bh * map.S + s] = -dot / fmaxf(sqrtf(ss), KD_EPS_COS);
    }
}

struct KdPlan {
    bool vec;
    uint32_t chunks, nwg;
    int lpr;
    uint32_t threads, rows_per_wg;   // rows_per_wg > 0: slot walk with `threads`-wide workgroups
};

template <int DT>
KdPlan plan_for(const void* x, const KdMap& map, uint32_t BH, uint32_t D) {
    const size_t es = sizeof(typename Elem<DT>::T);
    const size_t rowbytes = (size_t)D * es;
    KdPlan p{};
    p.vec = rowbytes % 16 == 0 && rowbytes <= 1024 && ((uintptr_t)x % 16 == 0) && (map.sb * es) % 16 == 0 &&
            (map.sh * es) % 16 == 0 && (map.ss * es) % 16 == 0;
    if (p.vec) {
        p.chunks = (uint32_t)(rowbytes / 16);
        p.lpr = 1;
        while (p.lpr < 64 && (uint32_t)p.lpr < p.chunks) p.lpr <<= 1;
        const uint32_t gpb = KD_THREADS / p.lpr;
        const uint64_t groups_needed = ((uint64_t)map.S + KD_UNROLL - 1) / KD_UNROLL;
        const uint64_t full = (groups_needed + gpb - 1) / gpb;
        const uint64_t cap = std::max<uint64_t>(1, (256 * 8 + BH - 1) / BH);  // ~8 workgroups per CU in total
        p.nwg = (uint32_t)std::max<uint64_t>(1, std::min(full, cap));
        p.threads = KD_THREADS;
        if (map.S >= 4096) {   // long rows: the slot walk, one 1024-thread workgroup per CU (rownorm.hip)
            p.threads = 1024;
            const uint64_t want = std::max<uint64_t>(1, ((uint64_t)256 + BH - 1) / BH);
            const uint32_t step = p.threads / p.lpr * KD_UNROLL;
            uint64_t rows = ((uint64_t)map.S + want - 1) / want;
            rows = (rows + step - 1) / step * step;
            p.rows_per_wg = (uint32_t)rows;
            p.nwg = (uint32_t)(((uint64_t)map.S + rows - 1) / rows);
        }
    } else {
        p.nwg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)map.S + 1023) / 1024, 256));
    }
    return p;
}

template <int DT>
int launch_keydiff(const void* x, KdMap map, uint32_t BH, uint32_t D, float* scores, float* anchor, float* partial,
                   hipStream_t stream) {
    using T = typename Elem<DT>::T;
    const T* xp = static_cast<const T*>(x);
    const KdPlan p = plan_for<DT>(x, map, BH, D);
    const dim3 grid(p.nwg, BH);
    // Cached loads in both passes: the gather that follows re-reads the kept K rows, and a pass that leaves nothing of K in the memory-side
    // cache costs it 10 us (streaming loads in the anchor / score / both passes: 203 / 208 / 210 us per step against 195 cached,
    // profiles/r03_ab_bench.txt).
    if (p.vec) {
#define KVP_KD_CASE(L)                                                                                                                 \
    case L:                                                                                                                            \
        if (p.threads == 1024) KVP_LAUNCH("keydiff_anchor_kernel", stream, (keydiff_anchor_vec_kernel<DT, L, 1024, false><<<grid, 1024, 0, stream>>>(xp, map, p.chunks, partial, p.rows_per_wg))); \
        else KVP_LAUNCH("keydiff_anchor_kernel", stream, (keydiff_anchor_vec_kernel<DT, L, KD_THREADS, false><<<grid, KD_THREADS, 0, stream>>>(xp, map, p.chunks, partial, p.rows_per_wg))); \
        break;
        switch (p.lpr) { KVP_KD_CASE(1) KVP_KD_CASE(2) KVP_KD_CASE(4) KVP_KD_CASE(8) KVP_KD_CASE(16) KVP_KD_CASE(32) KVP_KD_CASE(64) }
#undef KVP_KD_CASE
    } else {
        const uint32_t rows_per_wg = (map.S + p.nwg - 1) / p.nwg;
        KVP_LAUNCH("keydiff_anchor_kernel", stream, keydiff_anchor_scalar_kernel<DT><<<grid, KD_THREADS, 0, stream>>>(xp, map, D, rows_per_wg, partial));
    }
    KVP_LAUNCH("keydiff_anchor_reduce_kernel", stream, keydiff_anchor_reduce_kernel<<<BH, KD_RED_THREADS, 0, stream>>>(partial, p.nwg, D, map.S, anchor));
    if (p.vec) {
#define KVP_KD_CASE(L)                                                                                                                      \
    case L:                                                                                                                                 \
        if (p.threads == 1024) KVP_LAUNCH("keydiff_score_kernel", stream, (keydiff_score_vec_kernel<DT, L, 1024, false><<<grid, 1024, 0, stream>>>(xp, map, p.chunks, anchor, scores, p.rows_per_wg))); \
        else KVP_LAUNCH("keydiff_score_kernel", stream, (keydiff_score_vec_kernel<DT, L, KD_THREADS, false><<<grid, KD_THREADS, 0, stream>>>(xp, map, p.chunks, anchor, scores, p.rows_per_wg))); \
        break;
        switch (p.lpr) { KVP_KD_CASE(1) KVP_KD_CASE(2) KVP_KD_CASE(4) KVP_KD_CASE(8) KVP_KD_CASE(16) KVP_KD_CASE(32) KVP_KD_CASE(64) }
#undef KVP_KD_CASE
    } else {
        const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)map.S + KD_THREADS - 1) / KD_THREADS, 1024));
        KVP_LAUNCH("keydiff_score_kernel", stream, keydiff_score_scalar_kernel<DT><<<dim3(bx, BH), KD_THREADS, 0, stream>>>(xp, map, D, anchor, scores));
    }
    return 0;
}

size_t keydiff_ws(int64_t BH, int64_t S, int64_t D, size_t* partial_off) {
    const size_t anchor_bytes = kvp_align_up((size_t)BH * D * 4, 256);
    if (partial_off) *partial_off = anchor_bytes;
    // partial rows = BH * nwg: vector plan BH * ceil(2048 / BH) <= 2048 + BH, scalar plan BH * min(256, ceil(S / 1024))
    const int64_t rows = std::max<int64_t>(2048 + BH, BH * std::min<int64_t>(256, (S + 1023) / 1024));
    return anchor_bytes + kvp_align_up((size_t)rows * D * 4, 256);
}

}  // namespace

extern "C" size_t kvp_keydiff_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t D) {
    if (B <= 0 || H <= 0 || D <= 0 || S <= 0) return 256;
    return keydiff_ws(B * H, S, D, nullptr);
}

extern "C" int kvp_keydiff_score(const void* k, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh,
                                 int64_t ss, float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "keydiff: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 1 && D <= 8192, "keydiff: bad shape B=%ld H=%ld S=%ld D=%ld", (long)B, (long)H,
                  (long)S, (long)D);
    if (B * H * S == 0) return KVP_OK;
    KVP_CHECK_ARG(k && scores, "keydiff: null pointer");
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B * H <= 65535, "keydiff: shape too large (S=%ld, B*H=%ld)", (long)S, (long)(B * H));
    size_t poff = 0;
    const size_t need = keydiff_ws(B * H, S, D, &poff);
    if (!ws || ws_bytes < need) {
        kvp_set_error("keydiff: workspace too small (%zu < %zu)", ws_bytes, need);
        return KVP_EWORKSPACE;
    }
    float* anchor = static_cast<float*>(ws);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + poff);
    KdMap map{(uint32_t)H, (uint32_t)S, sb, sh, ss};
    const uint32_t BH = (uint32_t)(B * H);
    switch (dtype) {
        case KVP_F32: launch_keydiff<KVP_F32>(k, map, BH, (uint32_t)D, scores, anchor, partial, stream); break;
        case KVP_F16: launch_keydiff<KVP_F16>(k, map, BH, (uint32_t)D, scores, anchor, partial, stream); break;
        default: launch_keydiff<KVP_BF16>(k, map, BH, (uint32_t)D, scores, anchor, partial, stream); break;
    }
    KVP_CHECK_LAUNCH("keydiff");
    return KVP_OK;
}

// ---- scores[b,h,s] <- mean over h of scores[b,:,s], for every h ------------------------------------------------
// TOVA averages the last token's attention over ALL heads and repeats it for every kv-head
// (kvpress/presses/tova_press.py:52-53: `attn_weights.mean(1)`, `.repeat(1, num_kv_heads, 1)`); the SnapKV kernels
// deliver the per-kv-group means, groups are equally large, so the mean of the H group means is the all-head mean.
namespace {
__global__ __launch_bounds__(256) void head_mean_kernel(float* __restrict__ scores, uint32_t H, uint32_t S, int64_t stride_b,
                                                        int64_t stride_h) {
    float* base = scores + (int64_t)blockIdx.y * stride_b;
    const float inv = 1.f / (float)H;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += gridDim.x * blockDim.x) {
        float a = 0.f;
        for (uint32_t h = 0; h < H; ++h) a += base[(int64_t)h * stride_h + s];
        a *= inv;
        for (uint32_t h = 0; h < H; ++h) base[(int64_t)h * stride_h + s] = a;
    }
}
}  // namespace

extern "C" int kvp_scores_head_mean(float* scores, int64_t B, int64_t H, int64_t S, int64_t stride_b, int64_t stride_h,
                                    kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && B <= 65535, "head_mean: bad shape B=%ld H=%ld S=%ld", (long)B, (long)H, (long)S);
    if (B * H * S == 0) return KVP_OK;
    KVP_CHECK_ARG(scores, "head_mean: null pointer");
    KVP_CHECK_ARG(S < ((int64_t)1 << 31), "head_mean: S too large");
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((S + 255) / 256, 2048));
    KVP_LAUNCH("head_mean_kernel", stream, head_mean_kernel<<<dim3(bx, (uint32_t)B), 256, 0, stream>>>(scores, (uint32_t)H, (uint32_t)S, stride_b, stride_h));
    KVP_CHECK_LAUNCH("head_mean");
    return KVP_OK;
}
