// kvp_gather_kv: K'[b,h,j,:] = K[b,h,idx[b*H+h,j],:], V' likewise.
// Replaces `keys.gather(2, indices).contiguous()` / `values.gather(...)` (scorer_press.py:96-100),
// where `indices` is an int64 [B,H,n,D] expanded view that torch re-reads for K and again for V.
//
// HBM-bound row copy: algorithmic bytes = 2 * n*B*H*D*esize read + the same written.
// One output row = `chunks` 16-byte vectors handled by LPR adjacent lanes; K and V rows for the
// same index are moved by the same lanes (one index load, two dwordx4 loads, two dwordx4 stores).
// An index outside [0, S) -- in particular the -1 a failed select writes -- yields a row of all-ones bytes (NaN), never a read.
// With position-ordered indices (the default of kvp_topk_select) the source addresses of a
// (b,h) row are monotone, so consecutive lane groups stream forward through HBM pages.
// The kernel is dtype-agnostic (bytes); four rows per lane group are in flight.
#include "kvp_common.h"

namespace {

constexpr int GA_THREADS = 256;
constexpr int GA_UNROLL = 4;

struct GatherArgs {
    const char* k;
    const char* v;
    char* ko;
    char* vo;
    const int32_t* idx;
    int64_t k_sb, k_sh, k_ss;  // BYTE strides
    int64_t v_sb, v_sh, v_ss;
    uint32_t H, S, n;
    uint32_t rowbytes;  // D*esize
};

// grid = (row blocks, B*H): blockIdx.y selects the (b, h) plane -> no division per row.
template <int LPR, bool NT>
__global__ __launch_bounds__(GA_THREADS) void gather_vec_kernel(GatherArgs a) {
    constexpr int GPB = GA_THREADS / LPR;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / a.H, h = bh - b * a.H;
    const char* __restrict__ kb = a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const char* __restrict__ vb = a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int32_t* __restrict__ ib = a.idx + (size_t)bh * a.n;
    char* __restrict__ kob = a.ko + (size_t)bh * a.n * a.rowbytes;
    char* __restrict__ vob = a.vo + (size_t)bh * a.n * a.rowbytes;
    const uint32_t lir = threadIdx.x % LPR;
    const uint32_t g = blockIdx.x * GPB + threadIdx.x / LPR;
    const uint32_t TG = gridDim.x * GPB;
    const uint32_t chunks = a.rowbytes / 16;
    const uint32_t n = a.n;

    for (uint32_t j0 = g; j0 < n; j0 += TG * GA_UNROLL) {
        int32_t src[GA_UNROLL];
#pragma unroll
        for (int u = 0; u < GA_UNROLL; ++u) {
            const uint32_t j = j0 + u * TG;
            const int32_t s = j < n ? ib[j] : 0;
            // an index outside [0, S) is never dereferenced AND never papered over: -1 marks it (a select that reported a failure
            // writes -1, topk_cluster.hip) and the output row becomes all-ones bit patterns -- NaN in every supported dtype
            src[u] = (uint32_t)s < a.S ? s : -1;
        }
        uint4 kv[GA_UNROLL], vv[GA_UNROLL];
#pragma unroll
        for (int u = 0; u < GA_UNROLL; ++u) {
            const uint32_t j = j0 + u * TG;
            if (j < n && lir < chunks) {
                if (src[u] >= 0) {
                    kv[u] = ld16<NT>(kb + (int64_t)src[u] * a.k_ss + (size_t)lir * 16);
                    vv[u] = ld16<NT>(vb + (int64_t)src[u] * a.v_ss + (size_t)lir * 16);
                } else {
                    kv[u] = vv[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GA_UNROLL; ++u) {
            const uint32_t j = j0 + u * TG;
            if (j < n) {
                char* kd = kob + (size_t)j * a.rowbytes;
                char* vd = vob + (size_t)j * a.rowbytes;
                if (lir < chunks) {
                    st16<NT>(kd + (size_t)lir * 16, kv[u]);
                    st16<NT>(vd + (size_t)lir * 16, vv[u]);
                }
                if (LPR == 64) {  // rows longer than 1 KiB
                    const int64_t sr = src[u] >= 0 ? src[u] : 0;
                    const char* ks = kb + sr * a.k_ss;
                    const char* vs = vb + sr * a.v_ss;
                    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
                    for (uint32_t c = lir + LPR; c < chunks; c += LPR) {
                        st16<NT>(kd + (size_t)c * 16, src[u] >= 0 ? ld16<NT>(ks + (size_t)c * 16) : ones);
                        st16<NT>(vd + (size_t)c * 16, src[u] >= 0 ? ld16<NT>(vs + (size_t)c * 16) : ones);
                    }
                }
            }
        }
    }
}

// Any row size / alignment: element-granular copy (2- or 4-byte elements), one thread per element.
template <typename E>
__global__ __launch_bounds__(GA_THREADS) void gather_scalar_kernel(GatherArgs a, uint32_t D) {
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / a.H, h = bh - b * a.H;
    const uint32_t total = a.n * D;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t j = i / D, d = i - j * D;
        const int32_t raw = a.idx[(size_t)bh * a.n + j];
        const bool ok = (uint32_t)raw < a.S;   // otherwise: a NaN row (see gather_vec_kernel)
        const int64_t s = ok ? raw : 0;
        const E* ks = reinterpret_cast<const E*>(a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh + s * a.k_ss);
        const E* vs = reinterpret_cast<const E*>(a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh + s * a.v_ss);
        reinterpret_cast<E*>(a.ko)[(size_t)bh * total + i] = ok ? ks[d] : (E)~(E)0;
        reinterpret_cast<E*>(a.vo)[(size_t)bh * total + i] = ok ? vs[d] : (E)~(E)0;
    }
}

}  // namespace

extern "C" int kvp_gather_kv(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb,
                             int64_t v_sh, int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                             const int32_t* idx, int64_t n, void* k_out, void* v_out, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = kvp_async_check("kvp_gather_kv")) return rc;
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "gather: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 1 && n >= 0 && n <= S, "gather: bad shape B=%ld H=%ld S=%ld D=%ld n=%ld",
                  (long)B, (long)H, (long)S, (long)D, (long)n);
    const int64_t nrows64 = B * H * n;
    if (nrows64 == 0) return KVP_OK;
    KVP_CHECK_ARG(n * D < ((int64_t)1 << 31) && S < ((int64_t)1 << 31) && B * H <= 65535, "gather: shape too large");
    KVP_CHECK_ARG(k && v && idx && k_out && v_out, "gather: null pointer");
    const int64_t es = kvp_elem_size(dtype);
    GatherArgs a;
    a.k = static_cast<const char*>(k); a.v = static_cast<const char*>(v);
    a.ko = static_cast<char*>(k_out); a.vo = static_cast<char*>(v_out);
    a.idx = idx;
    a.k_sb = k_sb * es; a.k_sh = k_sh * es; a.k_ss = k_ss * es;
    a.v_sb = v_sb * es; a.v_sh = v_sh * es; a.v_ss = v_ss * es;
    a.H = (uint32_t)H; a.S = (uint32_t)S; a.n = (uint32_t)n;
    a.rowbytes = (uint32_t)(D * es);
    const uint32_t BH = (uint32_t)(B * H);
    auto al16 = [](int64_t x) { return x % 16 == 0; };
    const bool vec_ok = a.rowbytes % 16 == 0 && al16((int64_t)(uintptr_t)k) && al16((int64_t)(uintptr_t)v) &&
                        al16((int64_t)(uintptr_t)k_out) && al16((int64_t)(uintptr_t)v_out) && al16(a.k_sb) &&
                        al16(a.k_sh) && al16(a.k_ss) && al16(a.v_sb) && al16(a.v_sh) && al16(a.v_ss);
    if (!vec_ok) {
        const uint64_t total = (uint64_t)n * (uint64_t)D;
        const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + GA_THREADS - 1) / GA_THREADS, 1024));
        if (es == 4) KVP_LAUNCH("gather_scalar_kernel", stream, gather_scalar_kernel<uint32_t><<<dim3(bx, BH), GA_THREADS, 0, stream>>>(a, (uint32_t)D));
        else KVP_LAUNCH("gather_scalar_kernel", stream, gather_scalar_kernel<uint16_t><<<dim3(bx, BH), GA_THREADS, 0, stream>>>(a, (uint32_t)D));
        KVP_CHECK_LAUNCH("gather(scalar)");
        return KVP_OK;
    }
    const uint32_t chunks = a.rowbytes / 16;
    int lpr = 1;
    while (lpr < 64 && (uint32_t)lpr < chunks) lpr <<= 1;
    const uint32_t gpb = GA_THREADS / lpr;
    const uint64_t groups_needed = ((uint64_t)n + GA_UNROLL - 1) / GA_UNROLL;
    const uint64_t bx_full = (groups_needed + gpb - 1) / gpb;
    // 8 workgroups of 256 threads per CU (4 / 16 measured slower) -- but never fewer than 256 per (b, h) plane: the grid is dispatched
    // plane after plane, so with 256 workgroups per plane at most 8 planes (32 K / V / K' / V' streams) are in flight whatever the
    // batch; 2048 / BH per plane put all 32 planes of a batch of four in flight at once and the copy fell from 6.0 to 4.2 TB/s
    // (round 6, knorm128k_b4: 513 us for 4 x 90)
    const uint64_t bx_cap = std::max<uint64_t>(256, ((uint64_t)256 * 8 + BH - 1) / BH);
    const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min(bx_full, bx_cap));
    // Streaming (non-temporal) loads / stores when K + V do not fit the memory-side cache anyway (the cache is 256 MiB; the cutoff
    // chosen by measurement is 192 MiB of K + V): the copy must not
    // displace what the next kernels re-read nor leave its output behind as dirty lines -- at 128k tokens the following
    // window-attention pass pays for both (measured: its cold K stream 79 -> 50 us).  Smaller caches stay cached: there the
    // rows just scored are still resident (32k tokens: gather 23 us cached vs 26 us streaming).
    const bool nt = (uint64_t)B * H * S * a.rowbytes * 2 > (192ull << 20);
#define KVP_GA_CASE(L)                                                                                                   \
    case L:                                                                                                              \
        if (nt) KVP_LAUNCH("gather_vec_kernel", stream, gather_vec_kernel<L, true><<<dim3(bx, BH), GA_THREADS, 0, stream>>>(a)); \
        else KVP_LAUNCH("gather_vec_kernel", stream, gather_vec_kernel<L, false><<<dim3(bx, BH), GA_THREADS, 0, stream>>>(a));   \
        break;
    switch (lpr) { KVP_GA_CASE(1) KVP_GA_CASE(2) KVP_GA_CASE(4) KVP_GA_CASE(8) KVP_GA_CASE(16) KVP_GA_CASE(32) KVP_GA_CASE(64) }
#undef KVP_GA_CASE
    KVP_CHECK_LAUNCH("gather");
    return KVP_OK;
}
