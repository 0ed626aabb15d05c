// kvp_rerotate_keys: re-rotate already gathered keys so that the kept tokens sit at positions 0..n-1.
// Replaces KeyRerotationPress.rerotate_keys after its gather (kvpress/presses/key_rerotation_press.py:50-128):
//   delta_j = j - idx[b,h,j];  freq = delta_j * inv_freq[d mod D/2] (fp32);  cos/sin = cos(freq), sin(freq) cast to the
//   key dtype;  k' = k * cos + rotate_half(k) * sin  with torch's per-op rounding in the key dtype.
// In place on the contiguous [B,H,n,D] output of kvp_gather_kv (one extra read + write of K'; the rotation angle
// differs per (head, kept token), so there is nothing to share between rows).  One thread per (row, d < D/2).
#include "kvp_common.h"

namespace {

template <int DT> __device__ __forceinline__ void st_elem(typename Elem<DT>::T* p, float x);
template <> __device__ __forceinline__ void st_elem<KVP_F32>(float* p, float x) { *p = x; }
template <> __device__ __forceinline__ void st_elem<KVP_F16>(_Float16* p, float x) { *p = (_Float16)x; }
template <> __device__ __forceinline__ void st_elem<KVP_BF16>(uint16_t* p, float x) { *p = (uint16_t)(__float_as_uint(round_dt<KVP_BF16>(x)) >> 16); }

template <int DT>
__global__ __launch_bounds__(256) void rerotate_kernel(typename Elem<DT>::T* __restrict__ k, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ inv_freq, uint32_t n, uint32_t D) {
    const uint32_t half = D / 2;
    const uint32_t bh = blockIdx.y;
    typename Elem<DT>::T* kb = k + (size_t)bh * n * D;
    const int32_t* ib = idx + (size_t)bh * n;
    const uint32_t total = n * half;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t j = i / half, d = i - j * half;
        const float delta = (float)((int32_t)j - ib[j]);
        const float freq = __fmul_rn(delta, inv_freq[d]);
        const float c = round_dt<DT>(cosf(freq)), s = round_dt<DT>(sinf(freq));  // same angle for d and d + half
        typename Elem<DT>::T* row = kb + (size_t)j * D;
        const float k0 = Elem<DT>::ld(row + d), k1 = Elem<DT>::ld(row + d + half);
        st_elem<DT>(row + d, rope_elem<DT>(k0, c, -k1, s));
        st_elem<DT>(row + d + half, rope_elem<DT>(k1, c, k0, s));
    }
}

}  // namespace

extern "C" int kvp_rerotate_keys(void* k, int dtype, int64_t B, int64_t H, int64_t n, int64_t D, const int32_t* idx,
                                 const float* inv_freq, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "rerotate: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && n >= 0 && D >= 2 && D % 2 == 0, "rerotate: bad shape B=%ld H=%ld n=%ld D=%ld", (long)B, (long)H,
                  (long)n, (long)D);
    if (B * H * n == 0) return KVP_OK;
    KVP_CHECK_ARG(k && idx && inv_freq, "rerotate: null pointer");
    KVP_CHECK_ARG(B * H <= 65535 && n * D < ((int64_t)1 << 32), "rerotate: shape too large");
    const uint32_t total = (uint32_t)(n * (D / 2));
    const uint32_t BH = (uint32_t)(B * H);
    const uint32_t bx = std::max<uint32_t>(1, std::min<uint32_t>((total + 255) / 256, std::max<uint32_t>(1, 4096 / BH)));
    switch (dtype) {
        case KVP_F32: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_F32><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<float*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
        case KVP_F16: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_F16><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<_Float16*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
        default: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_BF16><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<uint16_t*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
    }
    KVP_CHECK_LAUNCH("rerotate");
    return KVP_OK;
}
