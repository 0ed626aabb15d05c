// kvp_rowl1_score: out[r] = scale * sum_c |x[r, c]|  over the rows of a 2-D view [R, N]
// Replaces `torch.norm(head_WoV, p=1, dim=-1)` (kvpress/presses/criticalkv_press.py:72): the L1 norm of every token's
// value vector projected through the head's slice of the output projection (N = hidden size, thousands of columns).
//
// HBM-bound streaming reduction over long rows: one wave per row, lanes stride over the row's 16-byte vectors (a wave
// instruction covers 1 KiB of contiguous HBM), 4 independent loads in flight, fp32 accumulation, xor-shuffle reduce.
#include "kvp_common.h"

namespace {

constexpr int L1_THREADS = 256;

template <int DT>
__device__ __forceinline__ float abssum16(const uint4& v) {
    float f[Elem<DT>::PER16];
    unpack16<DT>(v, f);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < Elem<DT>::PER16; ++i) a += fabsf(f[i]);
    return a;
}

template <int DT, bool VEC>
__global__ __launch_bounds__(L1_THREADS) void rowl1_kernel(const typename Elem<DT>::T* __restrict__ x, int64_t row_stride, uint64_t R, uint32_t N,
                                                           float scale, float* __restrict__ out) {
    constexpr int PER16 = Elem<DT>::PER16;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (L1_THREADS / 64) + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * (L1_THREADS / 64);
    for (uint64_t r = wave0; r < R; r += nwaves) {
        const typename Elem<DT>::T* p = x + (int64_t)r * row_stride;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (VEC) {
            const uint32_t nvec = N / PER16;
            uint32_t c = lane;
            for (; c + 192 < nvec; c += 256) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(p + (size_t)c * PER16);
                const uint4 v1 = *reinterpret_cast<const uint4*>(p + (size_t)(c + 64) * PER16);
                const uint4 v2 = *reinterpret_cast<const uint4*>(p + (size_t)(c + 128) * PER16);
                const uint4 v3 = *reinterpret_cast<const uint4*>(p + (size_t)(c + 192) * PER16);
                a0 += abssum16<DT>(v0); a1 += abssum16<DT>(v1); a2 += abssum16<DT>(v2); a3 += abssum16<DT>(v3);
            }
            for (; c < nvec; c += 64) a0 += abssum16<DT>(*reinterpret_cast<const uint4*>(p + (size_t)c * PER16));
            for (uint32_t e = nvec * PER16 + lane; e < N; e += 64) a1 += fabsf(Elem<DT>::ld(p + e));
        } else {
            for (uint32_t e = lane; e < N; e += 64) a0 += fabsf(Elem<DT>::ld(p + e));
        }
        const float s = wave_sum((a0 + a1) + (a2 + a3));
        if (lane == 0) out[r] = scale * s;
    }
}

}  // namespace

extern "C" int kvp_rowl1_score(const void* x, int dtype, int64_t R, int64_t N, int64_t row_stride, float scale, float* out, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "rowl1: bad dtype %d", dtype);
    KVP_CHECK_ARG(R >= 0 && N >= 1 && N < ((int64_t)1 << 31) && row_stride >= N, "rowl1: bad shape R=%ld N=%ld stride=%ld", (long)R, (long)N, (long)row_stride);
    if (R == 0) return KVP_OK;
    KVP_CHECK_ARG(x && out, "rowl1: null pointer");
    const size_t es = (size_t)kvp_elem_size(dtype);
    const bool vec = ((uintptr_t)x % 16 == 0) && ((size_t)row_stride * es) % 16 == 0;
    const uint32_t blocks = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((R + 3) / 4, 256 * 16));
#define KVP_L1(DT)                                                                                                                     \
    do {                                                                                                                               \
        if (vec) KVP_LAUNCH("rowl1_kernel", stream, (rowl1_kernel<DT, true><<<blocks, L1_THREADS, 0, stream>>>(static_cast<const Elem<DT>::T*>(x), row_stride, (uint64_t)R, (uint32_t)N, scale, out))); \
        else KVP_LAUNCH("rowl1_kernel", stream, (rowl1_kernel<DT, false><<<blocks, L1_THREADS, 0, stream>>>(static_cast<const Elem<DT>::T*>(x), row_stride, (uint64_t)R, (uint32_t)N, scale, out))); \
    } while (0)
    if (dtype == KVP_F32) KVP_L1(KVP_F32);
    else if (dtype == KVP_F16) KVP_L1(KVP_F16);
    else KVP_L1(KVP_BF16);
#undef KVP_L1
    KVP_CHECK_LAUNCH("rowl1");
    return KVP_OK;
}
