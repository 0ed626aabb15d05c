// kvp_rowdot_score: out[b,h,s] = scale * sum_d x[b,h,s,d] * f[h,d]
// Replaces `-(q_filter * keys).sum(dim=-1)` (kvpress/presses/qfilter_press.py:79-82): one learned filter vector per
// (layer, kv-head); the caller passes the layer's [H, D] slice.
//
// HBM-bound streaming reduction, laid out like rownorm.hip: LPR adjacent lanes own one row (one 16-byte vector each), so a
// wave-wide dwordx4 load covers 64 / LPR consecutive rows; four rows per lane are in flight.  Each lane keeps ITS piece of
// the head's filter in registers for the whole kernel (its column range never changes).  fp32 products and sums of the
// stored (bf16 / f16 / f32) inputs.
#include "kvp_common.h"

namespace {

constexpr int RD_THREADS = 256;
constexpr int RD_UNROLL = 4;

template <int DT>
__device__ __forceinline__ float dot16(const uint4& v, const float* f) {
    float x[Elem<DT>::PER16];
    unpack16<DT>(v, x);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < Elem<DT>::PER16; ++i) a = fmaf(x[i], f[i], a);
    return a;
}

// grid = (row blocks, B*H); rows of at most 64 vectors (D <= 512 bf16 / 256 f32)
template <int DT, int LPR>
__global__ __launch_bounds__(RD_THREADS) void rowdot_vec_kernel(const typename Elem<DT>::T* __restrict__ x, uint32_t H, uint32_t S,
                                                                int64_t sb, int64_t sh, int64_t ss,
                                                                const typename Elem<DT>::T* __restrict__ filt, int64_t f_sh,
                                                                uint32_t chunks, float scale, float* __restrict__ out) {
    using T = typename Elem<DT>::T;
    constexpr int PER16 = Elem<DT>::PER16;
    constexpr int GPB = RD_THREADS / LPR;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / H, h = bh - b * H;
    const T* __restrict__ base = x + (int64_t)b * sb + (int64_t)h * sh;
    float* __restrict__ ob = out + (size_t)bh * S;
    const uint32_t lir = threadIdx.x % LPR;
    const uint32_t g = blockIdx.x * GPB + threadIdx.x / LPR;
    const uint32_t TG = gridDim.x * GPB;
    const bool live = lir < chunks;

    float f[PER16];
    {
        uint4 fv = make_uint4(0, 0, 0, 0);
        if (live) fv = *reinterpret_cast<const uint4*>(filt + (int64_t)h * f_sh + (size_t)lir * PER16);
        unpack16<DT>(fv, f);
    }
    for (uint32_t s0 = g; s0 < S; s0 += TG * RD_UNROLL) {
        uint4 v[RD_UNROLL];
#pragma unroll
        for (int u = 0; u < RD_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            v[u] = make_uint4(0, 0, 0, 0);
            if (s < S && live) v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)s * ss + (size_t)lir * PER16);
        }
#pragma unroll
        for (int u = 0; u < RD_UNROLL; ++u) {
            const uint32_t s = s0 + u * TG;
            float acc = dot16<DT>(v[u], f);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lir == 0 && s < S) ob[s] = scale * acc;
        }
    }
}

// any D / alignment: one thread per row
template <int DT>
__global__ __launch_bounds__(RD_THREADS) void rowdot_scalar_kernel(const typename Elem<DT>::T* __restrict__ x, uint32_t H, uint32_t S,
                                                                   int64_t sb, int64_t sh, int64_t ss,
                                                                   const typename Elem<DT>::T* __restrict__ filt, int64_t f_sh,
                                                                   uint32_t D, float scale, float* __restrict__ out) {
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / H, h = bh - b * H;
    const typename Elem<DT>::T* base = x + (int64_t)b * sb + (int64_t)h * sh;
    const typename Elem<DT>::T* fp = filt + (int64_t)h * f_sh;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += gridDim.x * blockDim.x) {
        const typename Elem<DT>::T* p = base + (int64_t)s * ss;
        float acc = 0.f;
        for (uint32_t d = 0; d < D; ++d) acc = fmaf(Elem<DT>::ld(p + d), Elem<DT>::ld(fp + d), acc);
        out[(size_t)bh * S + s] = scale * acc;
    }
}

template <int DT>
void launch_rowdot(const void* x, uint32_t BH, uint32_t H, uint32_t S, uint32_t D, int64_t sb, int64_t sh, int64_t ss, const void* filt,
                   int64_t f_sh, float scale, float* out, hipStream_t stream) {
    using T = typename Elem<DT>::T;
    const T* xp = static_cast<const T*>(x);
    const T* fp = static_cast<const T*>(filt);
    const size_t es = sizeof(T);
    const size_t rowbytes = (size_t)D * es;
    const bool vec_ok = rowbytes % 16 == 0 && rowbytes <= 1024 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)filt % 16 == 0) &&
                        (sb * es) % 16 == 0 && (sh * es) % 16 == 0 && (ss * es) % 16 == 0 && (f_sh * es) % 16 == 0;
    if (!vec_ok) {
        const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)S + RD_THREADS - 1) / RD_THREADS, 1024));
        KVP_LAUNCH("rowdot_scalar_kernel", stream, rowdot_scalar_kernel<DT><<<dim3(bx, BH), RD_THREADS, 0, stream>>>(xp, H, S, sb, sh, ss, fp, f_sh, D, scale, out));
        return;
    }
    const uint32_t chunks = (uint32_t)(rowbytes / 16);
    int lpr = 1;
    while (lpr < 64 && (uint32_t)lpr < chunks) lpr <<= 1;
    const uint32_t gpb = RD_THREADS / lpr;
    const uint64_t groups_needed = ((uint64_t)S + RD_UNROLL - 1) / RD_UNROLL;
    const uint64_t bx_full = (groups_needed + gpb - 1) / gpb;
    const uint64_t bx_cap = std::max<uint64_t>(1, (256 * 8 + BH - 1) / BH);  // ~8 workgroups per CU in total
    const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min(bx_full, bx_cap));
#define KVP_RD_CASE(L)                                                                                                                  \
    case L:                                                                                                                             \
        KVP_LAUNCH("rowdot_vec_kernel", stream, rowdot_vec_kernel<DT, L><<<dim3(bx, BH), RD_THREADS, 0, stream>>>(xp, H, S, sb, sh, ss, fp, f_sh, chunks, scale, out)); \
        break;
    switch (lpr) {
        KVP_RD_CASE(1) KVP_RD_CASE(2) KVP_RD_CASE(4) KVP_RD_CASE(8) KVP_RD_CASE(16) KVP_RD_CASE(32) KVP_RD_CASE(64)
    }
#undef KVP_RD_CASE
}

}  // namespace

extern "C" int kvp_rowdot_score(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh,
                                int64_t ss, const void* filt, int64_t f_sh, float scale, float* out, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "rowdot: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 1, "rowdot: bad shape B=%ld H=%ld S=%ld D=%ld", (long)B, (long)H, (long)S, (long)D);
    if (B * H * S == 0) return KVP_OK;
    KVP_CHECK_ARG(x && filt && out, "rowdot: null pointer");
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B * H <= 65535, "rowdot: shape too large (S=%ld, B*H=%ld)", (long)S, (long)(B * H));
    const uint32_t BH = (uint32_t)(B * H);
    switch (dtype) {
        case KVP_F32: launch_rowdot<KVP_F32>(x, BH, (uint32_t)H, (uint32_t)S, (uint32_t)D, sb, sh, ss, filt, f_sh, scale, out, stream); break;
        case KVP_F16: launch_rowdot<KVP_F16>(x, BH, (uint32_t)H, (uint32_t)S, (uint32_t)D, sb, sh, ss, filt, f_sh, scale, out, stream); break;
        default: launch_rowdot<KVP_BF16>(x, BH, (uint32_t)H, (uint32_t)S, (uint32_t)D, sb, sh, ss, filt, f_sh, scale, out, stream); break;
    }
    KVP_CHECK_LAUNCH("rowdot");
    return KVP_OK;
}
