// kvp_rownorm_score: out[b,h,s] = scale * ||x[b,h,s,:]||_2
// Replaces `-keys.norm(dim=-1)` (kvpress/presses/knorm_press.py:38) and `values.norm(dim=-1)`
// (expected_attention_press.py:160).
//
// HBM-bound streaming reduction (algorithmic bytes = B*H*S*D*esize read + 4*B*H*S written).
// Fast path: a row is `chunks` 16-byte vectors; LPR (= next pow2 >= chunks, <= 64) adjacent lanes
// own one row, so one wave-wide dwordx4 load covers 64/LPR consecutive rows = 1 KiB of contiguous
// HBM when the tensor is contiguous (D=128 bf16: 16 lanes per 256-B row, 4 rows per instruction).
// Four independent rows per lane are in flight before any reduction starts; the sum of squares
// is accumulated in fp32 and reduced with xor-shuffles inside the LPR-lane group.
#include "kvp_common.h"
#include "topk_internal.h"

namespace {

template <int DT>
__device__ __forceinline__ float sumsq16(const uint4& v) {
    float f[Elem<DT>::PER16];
    unpack16<DT>(v, f);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < Elem<DT>::PER16; ++i) a = fmaf(f[i], f[i], a);
    return a;
}

constexpr int RN_THREADS = 256;
constexpr int RN_UNROLL = 4;

// grid = (row blocks, B*H): blockIdx.y selects the (b, h) plane, so the per-row address is one
// multiply-add (no integer division in the streaming loop).
struct PlaneMap {
    uint32_t H, S;
    int64_t sb, sh, ss;  // element strides
    bool squared;        // out = scale * sum of squares (CURPress) instead of scale * sqrt(sum of squares)
};

// HIST: the kernel also accumulates the top-k's first radix histogram of the scores it writes (hist1[bh][4096]).
template <int DT, int LPR, bool NT, bool HIST>
__global__ __launch_bounds__(RN_THREADS) void rownorm_vec_kernel(
    const typename Elem<DT>::T* __restrict__ x, PlaneMap map, uint32_t chunks, float scale, float* __restrict__ out,
    uint32_t* __restrict__ hist1) {
    using T = typename Elem<DT>::T;
    __shared__ uint32_t lh[HIST ? 4096 : 1];
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < 4096; i += RN_THREADS) lh[i] = 0;
        __syncthreads();
    }
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = RN_THREADS / LPR;  // row groups per block
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const T* __restrict__ base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    float* __restrict__ ob = out + (size_t)bh * map.S;
    const uint32_t lir = threadIdx.x % LPR;
    const uint32_t g = blockIdx.x * GPB + threadIdx.x / LPR;
    const uint32_t TG = gridDim.x * GPB;
    const uint32_t S = map.S;

    for (uint32_t s0 = g; s0 < S; s0 += TG * RN_UNROLL) {
        uint4 v[RN_UNROLL];
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && lir < chunks)
                v[u] = ld16<NT>(base + (int64_t)s * map.ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            float acc = sumsq16<DT>(v[u]);
            if (LPR == 64 && s < S) {  // rows longer than 1 KiB: keep striding
                const T* rowp = base + (int64_t)s * map.ss;
                for (uint32_t c = lir + LPR; c < chunks; c += LPR)
                    acc += sumsq16<DT>(*reinterpret_cast<const uint4*>(rowp + (size_t)c * PER16));
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            const float sc = scale * (map.squared ? acc : sqrtf(acc));
            if (lir == 0 && s < S) ob[s] = sc;
            // 64 / LPR scores per wave: plain LDS atomics (the wave-aggregated topk_hist1_add costs more than it saves here)
            if (HIST && lir == 0 && s < S) atomicAdd(&lh[float_to_key(sc) >> 20], 1u);
        }
    }
    if (HIST) {
        __syncthreads();
        topk_hist1_flush(lh, hist1 + (size_t)bh * 4096);
    }
}

// Two tensors of the same shape in ONE launch (CURPress: sum of squares of the keys' and of the values' rows, cur_press.py:40-41):
// blockIdx.z selects the tensor, so both streams are in flight together (two launches ran 118 us for 537 MB at 8 x 131072; one
// boundary and one ramp less, and the second stream starts while the first one drains).
template <int DT, int LPR, bool NT>
__global__ __launch_bounds__(RN_THREADS) void rowsumsq2_vec_kernel(const typename Elem<DT>::T* __restrict__ x0, const typename Elem<DT>::T* __restrict__ x1,
                                                                   PlaneMap map0, PlaneMap map1, uint32_t chunks, float* __restrict__ out0,
                                                                   float* __restrict__ out1) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = RN_THREADS / LPR;
    const bool second = blockIdx.z != 0;
    const PlaneMap map = second ? map1 : map0;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const T* __restrict__ base = (second ? x1 : x0) + (int64_t)b * map.sb + (int64_t)h * map.sh;
    float* __restrict__ ob = (second ? out1 : out0) + (size_t)bh * map.S;
    const uint32_t lir = threadIdx.x % LPR;
    const uint32_t g = blockIdx.x * GPB + threadIdx.x / LPR;
    const uint32_t TG = gridDim.x * GPB;
    const uint32_t S = map.S;
    for (uint32_t s0 = g; s0 < S; s0 += TG * RN_UNROLL) {
        uint4 v[RN_UNROLL];
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && lir < chunks) v[u] = ld16<NT>(base + (int64_t)s * map.ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            float acc = sumsq16<DT>(v[u]);   // (same lanes, fma chain and shuffle order as rownorm_vec_kernel: same bits)
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lir == 0 && s < S) ob[s] = acc;
        }
    }
}

// Slot shape: every workgroup streams ONE contiguous range of rows (the walk of topk_cluster.hip's Knorm mode, which moves 268 MB in
// ~38 us where the strided walk above takes 53-65), RN_UNROLL steps of THREADS / LPR rows in flight.  blockIdx.z selects the tensor
// (CURPress: K and V energies in one launch).  Same lanes per row, same fma chain and shuffle order: same bits as rownorm_vec_kernel.
template <int DT, int LPR, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void rownorm_slot_kernel(const typename Elem<DT>::T* __restrict__ x0, const typename Elem<DT>::T* __restrict__ x1,
                                                               PlaneMap map0, PlaneMap map1, uint32_t chunks, float scale, float* __restrict__ out0,
                                                               float* __restrict__ out1, uint32_t rows_per_wg) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = THREADS / LPR;
    const bool second = blockIdx.z != 0;
    const PlaneMap map = second ? map1 : map0;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const T* __restrict__ base = (second ? x1 : x0) + (int64_t)b * map.sb + (int64_t)h * map.sh;
    float* __restrict__ ob = (second ? out1 : out0) + (size_t)bh * map.S;
    const uint32_t lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const uint32_t r0 = blockIdx.x * rows_per_wg;
    const uint32_t r1 = min(map.S, r0 + rows_per_wg);
    for (uint32_t it = r0; it < r1; it += GPB * RN_UNROLL) {
        uint4 v[RN_UNROLL];
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = it + u * GPB + grp;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < r1 && lir < chunks) v[u] = ld16<NT>(base + (int64_t)s * map.ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < RN_UNROLL; ++u) {
            const uint32_t s = it + u * GPB + grp;
            float acc = sumsq16<DT>(v[u]);
            if (LPR == 64 && s < r1) {
                const T* rowp = base + (int64_t)s * map.ss;
                for (uint32_t c = lir + LPR; c < chunks; c += LPR)
                    acc += sumsq16<DT>(*reinterpret_cast<const uint4*>(rowp + (size_t)c * PER16));
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lir == 0 && s < r1) ob[s] = scale * (map.squared ? acc : sqrtf(acc));
        }
    }
}

// Any D / alignment: one thread per row, scalar loads (tiny test shapes such as head_dim 6).
template <int DT>
__global__ __launch_bounds__(RN_THREADS) void rownorm_scalar_kernel(
    const typename Elem<DT>::T* __restrict__ x, PlaneMap map, uint32_t D, float scale, float* __restrict__ out) {
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / map.H, h = bh - b * map.H;
    const typename Elem<DT>::T* base = x + (int64_t)b * map.sb + (int64_t)h * map.sh;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < map.S; s += gridDim.x * blockDim.x) {
        const typename Elem<DT>::T* p = base + (int64_t)s * map.ss;
        float acc = 0.f;
        for (uint32_t d = 0; d < D; ++d) {
            const float f = Elem<DT>::ld(p + d);
            acc = fmaf(f, f, acc);
        }
        out[(size_t)bh * map.S + s] = scale * (map.squared ? acc : sqrtf(acc));
    }
}

// Streaming (non-temporal) loads for a tensor that is read ONCE on this path and cannot stay in the 256 MiB memory-side cache anyway (the
// cutoff of kvp_gather_kv).  Whether that pays is decided by what runs next, so the caller says so (`read_once`), measured inside the
// bench loops (round 3 A/B of the bench loops, record: profiles/r03_ab_bench.txt; the script that made it was retired with the knobs in round 5): ExpectedAttention's ||V|| 58 -> 47 us, CUR's two energies 105 -> 91 us
// (its gather + 7); but the stand-alone K norm of a press whose gather re-reads the kept K rows right after (Knorm through
// KeyRerotationPress) LOSES 14 us when the stream leaves nothing of K behind -- kvp_rownorm_score stays cached.
bool rn_streaming(uint64_t bytes, bool read_once) { return read_once && bytes > (192ull << 20); }

template <int DT, int THREADS>
void launch_rownorm_slot_t(const typename Elem<DT>::T* x0, const typename Elem<DT>::T* x1, PlaneMap m0, PlaneMap m1, uint32_t BH, uint32_t ntens,
                           uint32_t chunks, int lpr, float scale, float* o0, float* o1, hipStream_t stream, bool nt) {
    const uint32_t gpb = THREADS / lpr;
    const uint32_t wgs_per_cu = THREADS >= 1024 ? 1 : 2048 / THREADS;
    const uint64_t want = std::max<uint64_t>(1, ((uint64_t)256 * wgs_per_cu + (uint64_t)BH * ntens - 1) / ((uint64_t)BH * ntens));
    const uint32_t step = gpb * RN_UNROLL;
    uint64_t rows = ((uint64_t)m0.S + want - 1) / want;
    rows = (rows + step - 1) / step * step;
    const uint32_t bx = (uint32_t)(((uint64_t)m0.S + rows - 1) / rows);
    const dim3 grid(bx, BH, ntens);
#define KVP_RS_CASE(L)                                                                                                                    \
    case L:                                                                                                                               \
        if (nt) KVP_LAUNCH("rownorm_vec_kernel", stream, (rownorm_slot_kernel<DT, L, THREADS, true><<<grid, THREADS, 0, stream>>>(x0, x1, m0, m1, chunks, scale, o0, o1, (uint32_t)rows))); \
        else KVP_LAUNCH("rownorm_vec_kernel", stream, (rownorm_slot_kernel<DT, L, THREADS, false><<<grid, THREADS, 0, stream>>>(x0, x1, m0, m1, chunks, scale, o0, o1, (uint32_t)rows))); \
        break;
    switch (lpr) {
        KVP_RS_CASE(1) KVP_RS_CASE(2) KVP_RS_CASE(4) KVP_RS_CASE(8) KVP_RS_CASE(16) KVP_RS_CASE(32) KVP_RS_CASE(64)
    }
#undef KVP_RS_CASE
}
template <int DT>
void launch_rownorm_slot(const typename Elem<DT>::T* x0, const typename Elem<DT>::T* x1, PlaneMap m0, PlaneMap m1, uint32_t BH, uint32_t ntens,
                         uint32_t chunks, int lpr, float scale, float* o0, float* o1, hipStream_t stream, bool nt) {
    // one 1024-thread workgroup per CU (256 / 512 threads with 8 / 4 workgroups per CU measured slower: profiles/r03_stream_lab.txt)
    launch_rownorm_slot_t<DT, 1024>(x0, x1, m0, m1, BH, ntens, chunks, lpr, scale, o0, o1, stream, nt);
}

// returns 1 if hist1 was requested and produced (vector path only), else 0
template <int DT>
int launch_rownorm(const void* x, PlaneMap map, uint32_t BH, uint32_t D, float scale, float* out, uint32_t* hist1, hipStream_t stream, bool read_once) {
    using T = typename Elem<DT>::T;
    const T* xp = static_cast<const T*>(x);
    const size_t es = sizeof(T);
    const size_t rowbytes = (size_t)D * es;
    const bool vec_ok = rowbytes % 16 == 0 && ((uintptr_t)x % 16 == 0) && (map.sb * es) % 16 == 0 &&
                        (map.sh * es) % 16 == 0 && (map.ss * es) % 16 == 0;
    if (!vec_ok) {
        const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)map.S + RN_THREADS - 1) / RN_THREADS, 1024));
        KVP_LAUNCH("rownorm_scalar_kernel", stream, rownorm_scalar_kernel<DT><<<dim3(bx, BH), RN_THREADS, 0, stream>>>(xp, map, D, scale, out));
        return 0;
    }
    const uint32_t chunks = (uint32_t)(rowbytes / 16);
    int lpr = 1;
    while (lpr < 64 && (uint32_t)lpr < chunks) lpr <<= 1;
    const uint32_t gpb = RN_THREADS / lpr;
    const uint64_t groups_needed = ((uint64_t)map.S + RN_UNROLL - 1) / RN_UNROLL;
    const uint64_t bx_full = (groups_needed + gpb - 1) / gpb;
    const uint64_t bx_cap = std::max<uint64_t>(1, (256 * 8 + BH - 1) / BH);  // ~8 workgroups per CU in total
    const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min(bx_full, bx_cap));
    const bool nt = rn_streaming((uint64_t)BH * map.S * rowbytes, read_once);
    if (!hist1 && map.S >= 4096) {   // long rows: the slot walk; short rows (and the <HIST> variant): the interleaved walk below
        launch_rownorm_slot<DT>(xp, xp, map, map, BH, 1, chunks, lpr, scale, out, out, stream, nt);
        return 0;
    }
#define KVP_RN_CASE(L)                                                                                     \
    case L:                                                                                                \
        if (hist1) KVP_LAUNCH("rownorm_vec_kernel", stream, rownorm_vec_kernel<DT, L, false, true><<<dim3(bx, BH), RN_THREADS, 0, stream>>>(xp, map, chunks, scale, out, hist1)); \
        else if (nt) KVP_LAUNCH("rownorm_vec_kernel", stream, rownorm_vec_kernel<DT, L, true, false><<<dim3(bx, BH), RN_THREADS, 0, stream>>>(xp, map, chunks, scale, out, nullptr)); \
        else KVP_LAUNCH("rownorm_vec_kernel", stream, rownorm_vec_kernel<DT, L, false, false><<<dim3(bx, BH), RN_THREADS, 0, stream>>>(xp, map, chunks, scale, out, nullptr)); \
        break;
    switch (lpr) {
        KVP_RN_CASE(1) KVP_RN_CASE(2) KVP_RN_CASE(4) KVP_RN_CASE(8) KVP_RN_CASE(16) KVP_RN_CASE(32) KVP_RN_CASE(64)
    }
#undef KVP_RN_CASE
    return hist1 ? 1 : 0;
}

}  // namespace

// Internal entry shared with the ExpectedAttention path (||V||) and the fused Knorm compress.
// hist1 (nullable): [B*H][4096] first-pass radix histogram of the top-k over the rows (b, h); *hist1_done tells whether it
// was produced (only the vector path does; the caller runs the separate pass otherwise).
static int rownorm_launch_impl(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh,
                               int64_t ss, float scale, float* out, hipStream_t stream, uint32_t* hist1, bool* hist1_done, bool squared, bool read_once) {
    if (hist1_done) *hist1_done = false;
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "rownorm: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 1, "rownorm: bad shape B=%ld H=%ld S=%ld D=%ld", (long)B, (long)H,
                  (long)S, (long)D);
    const int64_t nrows64 = B * H * S;
    if (nrows64 == 0) return KVP_OK;
    KVP_CHECK_ARG(x && out, "rownorm: null pointer");
    // collapse (h, s) and (b, h) when the view is contiguous across them: fewer, longer planes
    // (not with a fused histogram: its rows are the (b, h) planes)
    if (!hist1 && H > 1 && sh == S * ss) { S *= H; H = 1; sh = 0; }
    if (!hist1 && H == 1 && B > 1 && sb == S * ss) { S *= B; B = 1; sb = 0; }
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B * H <= 65535, "rownorm: shape too large (S=%ld, B*H=%ld)", (long)S, (long)(B * H));
    PlaneMap map{(uint32_t)H, (uint32_t)S, sb, sh, ss, squared};
    const uint32_t BH = (uint32_t)(B * H);
    int done = 0;
    switch (dtype) {
        case KVP_F32: done = launch_rownorm<KVP_F32>(x, map, BH, (uint32_t)D, scale, out, hist1, stream, read_once); break;
        case KVP_F16: done = launch_rownorm<KVP_F16>(x, map, BH, (uint32_t)D, scale, out, hist1, stream, read_once); break;
        default: done = launch_rownorm<KVP_BF16>(x, map, BH, (uint32_t)D, scale, out, hist1, stream, read_once); break;
    }
    if (hist1_done) *hist1_done = done != 0;
    KVP_CHECK_LAUNCH("rownorm");
    return KVP_OK;
}

int kvp_rownorm_launch(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh,
                       int64_t ss, float scale, float* out, hipStream_t stream, uint32_t* hist1, bool* hist1_done) {
    return rownorm_launch_impl(x, dtype, B, H, S, D, sb, sh, ss, scale, out, stream, hist1, hist1_done, false, false);
}
// the same for a tensor this path reads once (ExpectedAttention's ||V||): streaming loads when it is larger than the memory-side cache can hold
int kvp_rownorm_launch_read_once(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh, int64_t ss,
                                 float scale, float* out, hipStream_t stream) {
    return rownorm_launch_impl(x, dtype, B, H, S, D, sb, sh, ss, scale, out, stream, nullptr, nullptr, false, true);
}
// out[b,h,s] = sum_d x^2  (the row "energy" of CURPress, kvpress/presses/cur_press.py:40-41)
int kvp_rowsumsq_launch(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh, int64_t ss,
                        float* out, hipStream_t stream) {
    return rownorm_launch_impl(x, dtype, B, H, S, D, sb, sh, ss, 1.0f, out, stream, nullptr, nullptr, true, false);
}

// out_k[b,h,s] = sum_d k^2 and out_v likewise in one launch when both tensors take the 16-lanes-per-256-byte-row vector path (else two
// launches of the general kernel).  Same values as kvp_rowsumsq_launch.
int kvp_rowsumsq2_launch(const void* k, const void* v, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                         int64_t v_sb, int64_t v_sh, int64_t v_ss, float* out_k, float* out_v, hipStream_t stream) {
    const int64_t es = kvp_elem_size(dtype);
    auto al = [&](const void* p, int64_t sb, int64_t sh, int64_t ss) {
        return ((uintptr_t)p % 16) == 0 && (sb * es) % 16 == 0 && (sh * es) % 16 == 0 && (ss * es) % 16 == 0;
    };
    const bool fast = (dtype == KVP_BF16 || dtype == KVP_F16) && D * es == 256 && B * H >= 1 && B * H <= 65535 && S >= 1 && S < ((int64_t)1 << 31) &&
                      al(k, k_sb, k_sh, k_ss) && al(v, v_sb, v_sh, v_ss) && k && v && out_k && out_v;
    if (!fast) {
        if (int rc = kvp_rowsumsq_launch(k, dtype, B, H, S, D, k_sb, k_sh, k_ss, out_k, stream)) return rc;
        return kvp_rowsumsq_launch(v, dtype, B, H, S, D, v_sb, v_sh, v_ss, out_v, stream);
    }
    PlaneMap mk{(uint32_t)H, (uint32_t)S, k_sb, k_sh, k_ss, true}, mv{(uint32_t)H, (uint32_t)S, v_sb, v_sh, v_ss, true};
    const uint32_t BH = (uint32_t)(B * H);
    if (S >= 4096) {
        const bool nts = rn_streaming((uint64_t)BH * 2 * S * 256, true);
        if (dtype == KVP_BF16) launch_rownorm_slot<KVP_BF16>(static_cast<const uint16_t*>(k), static_cast<const uint16_t*>(v), mk, mv, BH, 2, 16, 16, 1.0f, out_k, out_v, stream, nts);
        else launch_rownorm_slot<KVP_F16>(static_cast<const _Float16*>(k), static_cast<const _Float16*>(v), mk, mv, BH, 2, 16, 16, 1.0f, out_k, out_v, stream, nts);
        KVP_CHECK_LAUNCH("rowsumsq2");
        return KVP_OK;
    }
    constexpr uint32_t gpb = RN_THREADS / 16;
    const uint64_t bx_full = (((uint64_t)S + RN_UNROLL - 1) / RN_UNROLL + gpb - 1) / gpb;
    const uint64_t bx_cap = std::max<uint64_t>(1, (256 * 4 + BH - 1) / BH);   // ~8 workgroups per CU over the two tensors
    const dim3 grid((uint32_t)std::max<uint64_t>(1, std::min(bx_full, bx_cap)), BH, 2);
    const bool nt = rn_streaming((uint64_t)BH * 2 * S * 256, true);
#define KVP_RS2(DTV, TT, NTV) KVP_LAUNCH("rownorm_vec_kernel", stream, (rowsumsq2_vec_kernel<DTV, 16, NTV><<<grid, RN_THREADS, 0, stream>>>(static_cast<const TT*>(k), static_cast<const TT*>(v), mk, mv, 16, out_k, out_v)))
    if (dtype == KVP_BF16) { if (nt) KVP_RS2(KVP_BF16, uint16_t, true); else KVP_RS2(KVP_BF16, uint16_t, false); }
    else { if (nt) KVP_RS2(KVP_F16, _Float16, true); else KVP_RS2(KVP_F16, _Float16, false); }
#undef KVP_RS2
    KVP_CHECK_LAUNCH("rowsumsq2");
    return KVP_OK;
}

extern "C" int kvp_rownorm_score(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb,
                                 int64_t sh, int64_t ss, float scale, float* out, kvp_stream_t stream) {
    return kvp_rownorm_launch(x, dtype, B, H, S, D, sb, sh, ss, scale, out, static_cast<hipStream_t>(stream), nullptr, nullptr);
}
