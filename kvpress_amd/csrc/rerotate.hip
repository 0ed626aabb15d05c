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

// cos / sin of a float32 angle (torch: `freqs.cos()`, `.sin()` of the float32 product delta * inv_freq).  The angles reach ~1e5 rad,
// where cosf and sinf each run their slow argument reduction (140 of the kernel's 150 us at 8 x 65536 rows).  Instead: ONE reduction
// in double -- k = rint(angle * 2/pi), r = (angle * 2/pi - k) * pi/2 in [-pi/4, pi/4], exact to ~1e-16 of a quarter turn -- then the
// two minimax polynomials of that interval in float32 (Cephes sinf / cosf coefficients, |error| < 1.2e-7 = the accuracy class of
// cosf / sinf themselves) and the quadrant's swap / signs.
// Error bound against the reference's `emb.cos()` / `emb.sin()` (torch, float32, then cast to the key dtype): the float32 values differ by
// <= 2.4e-7 absolute, so after the cast < 0.2 % of the cos / sin values land on the neighbouring 16-bit number (never further);
// an output element then differs by <= 2.1 ulp of its (d, d + D/2) PAIR norm -- i.e. <= 4.2 ulp of the element itself wherever the two
// products do not cancel (|out| >= half the pair norm) -- and is bit-identical otherwise (tests/test_wrappers.py: both bounds, bf16 / f16;
// float32: rtol 1e-5).  Non-finite or absurd angles (|angle * 2/pi| >= 2^31) give NaN / garbage as cosf does for inf; the conversion
// stays defined.
__device__ __forceinline__ void sincos_f32_angle(float angle, float& s, float& c) {
    const double t = (double)angle * 0.6366197723675814;       // 2 / pi
    const double kq = rint(t);
    const float r = (float)((t - kq) * 1.5707963267948966);     // pi / 2
    const int q = fabs(kq) < 2147483520.0 ? (int)kq : 0;   // non-finite / absurd angles: r is NaN or garbage anyway, the conversion must stay defined
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

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
        float sv, cv;
        sincos_f32_angle(freq, sv, cv);
        const float c = round_dt<DT>(cv), s = round_dt<DT>(sv);  // same angle for d and d + half
        typename Elem<DT>::T* row = kb + (size_t)j * D;
        const float k0 = Elem<DT>::ld(row + d), k1 = Elem<DT>::ld(row + d + half);
        st_elem<DT>(row + d, rope_elem<DT>(k0, c, -k1, s));
        st_elem<DT>(row + d + half, rope_elem<DT>(k1, c, k0, s));
    }
}

// 2-byte dtypes, D % 16 == 0, 16-byte aligned rows: a thread owns 8 consecutive dims d0 .. d0+7 of the first half and the matching 8 of
// the second half: two 16-byte loads, 8 angles, two 16-byte stores (the scalar kernel above moves 2 bytes per lane and instruction).
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<KVP_BF16>(float lo, float hi) {
    return (__float_as_uint(round_dt<KVP_BF16>(lo)) >> 16) | (__float_as_uint(round_dt<KVP_BF16>(hi)) & 0xFFFF0000u);
}
template <> __device__ __forceinline__ uint32_t pack2<KVP_F16>(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// dims d0 .. d0+7 of the first half (a) and of the second half (b) of one row, rotated by delta * inv_freq[d0 .. d0+7]
template <int DT>
__device__ __forceinline__ void rotate8(const uint4& a, const uint4& b, float delta, const float* __restrict__ invf, uint4& out0, uint4& out1) {
    float k0[8], k1[8], o0[8], o1[8];
    unpack16<DT>(a, k0);
    unpack16<DT>(b, k1);
    const float4 f0 = *reinterpret_cast<const float4*>(invf), f1 = *reinterpret_cast<const float4*>(invf + 4);
    const float fr[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float freq = __fmul_rn(delta, fr[e]);
        float sv, cv;
        sincos_f32_angle(freq, sv, cv);
        const float c = round_dt<DT>(cv), s = round_dt<DT>(sv);
        o0[e] = rope_elem<DT>(k0[e], c, -k1[e], s);
        o1[e] = rope_elem<DT>(k1[e], c, k0[e], s);
    }
    out0 = make_uint4(pack2<DT>(o0[0], o0[1]), pack2<DT>(o0[2], o0[3]), pack2<DT>(o0[4], o0[5]), pack2<DT>(o0[6], o0[7]));
    out1 = make_uint4(pack2<DT>(o1[0], o1[1]), pack2<DT>(o1[2], o1[3]), pack2<DT>(o1[4], o1[5]), pack2<DT>(o1[6], o1[7]));
}

template <int DT>
__global__ __launch_bounds__(256) void rerotate_vec_kernel(typename Elem<DT>::T* __restrict__ k, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ inv_freq, uint32_t n, uint32_t D) {
    const uint32_t half = D / 2, tpr = half / 8;   // threads per row
    const uint32_t bh = blockIdx.y;
    typename Elem<DT>::T* kb = k + (size_t)bh * n * D;
    const int32_t* ib = idx + (size_t)bh * n;
    const uint32_t total = n * tpr;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t j = i / tpr, d0 = (i - j * tpr) * 8;
        const float delta = (float)((int32_t)j - ib[j]);
        typename Elem<DT>::T* row = kb + (size_t)j * D;
        const uint4 a = *reinterpret_cast<const uint4*>(row + d0), b = *reinterpret_cast<const uint4*>(row + d0 + half);
        uint4 o0, o1;
        rotate8<DT>(a, b, delta, inv_freq + d0, o0, o1);
        *reinterpret_cast<uint4*>(row + d0) = o0;
        *reinterpret_cast<uint4*>(row + d0 + half) = o1;
    }
}

// ---- gather + re-rotation in ONE pass (KeyRerotationPress / FinchPress: key_rerotation_press.py:157-160 then :107-128) ------------
// K'[b,h,j] = rotate(K[b,h,idx[j]], (j - idx[j]) * inv_freq), V'[b,h,j] = V[b,h,idx[j]]: the gathered keys never make the round trip
// through HBM that kvp_gather_kv + kvp_rerotate_keys costs (n * D * esize written, read and written again).  A thread owns 8 dims of
// both halves of a row (rotate8: the same arithmetic as rerotate_vec_kernel, bit for bit) and copies the matching 2 x 16 bytes of V;
// GR_UNROLL rows per thread are in flight.
constexpr int GR_UNROLL = 2;
struct GrArgs {
    const char* k;
    const char* v;
    char* ko;
    char* vo;
    const int32_t* idx;
    const float* inv_freq;
    int64_t k_sb, k_sh, k_ss;  // BYTE strides
    int64_t v_sb, v_sh, v_ss;
    uint32_t H, S, n, D;
};
template <int DT, bool NT>
__global__ __launch_bounds__(256) void gather_rerotate_kernel(GrArgs a) {
    const uint32_t half = a.D / 2, tpr = half / 8, rowbytes = a.D * 2;
    const uint32_t bh = blockIdx.y;
    const uint32_t b = bh / a.H, h = bh - b * a.H;
    const char* __restrict__ kb = a.k + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const char* __restrict__ vb = a.v + (int64_t)b * a.v_sb + (int64_t)h * a.v_sh;
    const int32_t* __restrict__ ib = a.idx + (size_t)bh * a.n;
    char* __restrict__ kob = a.ko + (size_t)bh * a.n * rowbytes;
    char* __restrict__ vob = a.vo + (size_t)bh * a.n * rowbytes;
    const uint32_t total = a.n * tpr, stride = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += stride * GR_UNROLL) {
        uint4 ka[GR_UNROLL], kc[GR_UNROLL], va[GR_UNROLL], vc[GR_UNROLL];
        uint32_t j[GR_UNROLL], d0[GR_UNROLL];
        float delta[GR_UNROLL];
        bool ok[GR_UNROLL];
#pragma unroll
        for (int u = 0; u < GR_UNROLL; ++u) {
            const uint32_t i = i0 + u * stride;
            j[u] = i / tpr;
            d0[u] = (i - j[u] * tpr) * 8;
            if (i < total) {
                const int32_t raw = ib[j[u]];
                ok[u] = (uint32_t)raw < a.S;   // else (the -1 of a select that reported a failure): a NaN row, as in gather_vec_kernel
                const int32_t src = ok[u] ? raw : 0;
                delta[u] = (float)((int32_t)j[u] - raw);
                const char* ks = kb + (int64_t)src * a.k_ss + (size_t)d0[u] * 2;
                const char* vs = vb + (int64_t)src * a.v_ss + (size_t)d0[u] * 2;
                ka[u] = ld16<NT>(ks);
                kc[u] = ld16<NT>(ks + half * 2);
                va[u] = ld16<NT>(vs);
                vc[u] = ld16<NT>(vs + half * 2);
            }
        }
#pragma unroll
        for (int u = 0; u < GR_UNROLL; ++u) {
            const uint32_t i = i0 + u * stride;
            if (i < total) {
                uint4 o0, o1;
                rotate8<DT>(ka[u], kc[u], delta[u], a.inv_freq + d0[u], o0, o1);
                if (!ok[u]) o0 = o1 = va[u] = vc[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
                char* kd = kob + (size_t)j[u] * rowbytes + (size_t)d0[u] * 2;
                char* vd = vob + (size_t)j[u] * rowbytes + (size_t)d0[u] * 2;
                st16<NT>(kd, o0);
                st16<NT>(kd + half * 2, o1);
                st16<NT>(vd, va[u]);
                st16<NT>(vd + half * 2, vc[u]);
            }
        }
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
    if (dtype != KVP_F32 && D % 16 == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)inv_freq % 16) == 0) {
        const uint32_t tv = (uint32_t)(n * (D / 16));
        const uint32_t bv = std::max<uint32_t>(1, std::min<uint32_t>((tv + 255) / 256, std::max<uint32_t>(1, 4096 / BH)));
        if (dtype == KVP_F16) KVP_LAUNCH("rerotate_kernel", stream, rerotate_vec_kernel<KVP_F16><<<dim3(bv, BH), 256, 0, stream>>>(static_cast<_Float16*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D));
        else KVP_LAUNCH("rerotate_kernel", stream, rerotate_vec_kernel<KVP_BF16><<<dim3(bv, BH), 256, 0, stream>>>(static_cast<uint16_t*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D));
        KVP_CHECK_LAUNCH("rerotate");
        return KVP_OK;
    }
    switch (dtype) {
        case KVP_F32: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_F32><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<float*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
        case KVP_F16: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_F16><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<_Float16*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
        default: KVP_LAUNCH("rerotate_kernel", stream, rerotate_kernel<KVP_BF16><<<dim3(bx, BH), 256, 0, stream>>>(static_cast<uint16_t*>(k), idx, inv_freq, (uint32_t)n, (uint32_t)D)); break;
    }
    KVP_CHECK_LAUNCH("rerotate");
    return KVP_OK;
}

extern "C" int kvp_gather_kv(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                             int dtype, int64_t B, int64_t H, int64_t S, int64_t D, const int32_t* idx, int64_t n, void* k_out, void* v_out,
                             kvp_stream_t stream);

extern "C" int kvp_gather_kv_rerotate(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh,
                                      int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, const int32_t* idx, int64_t n,
                                      const float* inv_freq, void* k_out, void* v_out, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = kvp_async_check("kvp_gather_kv_rerotate")) return rc;
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "gather_rerotate: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 0 && H >= 0 && S >= 0 && D >= 2 && D % 2 == 0 && n >= 0 && n <= S, "gather_rerotate: bad shape B=%ld H=%ld S=%ld D=%ld n=%ld",
                  (long)B, (long)H, (long)S, (long)D, (long)n);
    if (B * H * n == 0) return KVP_OK;
    KVP_CHECK_ARG(k && v && idx && inv_freq && k_out && v_out, "gather_rerotate: null pointer");
    auto al16 = [](int64_t x) { return x % 16 == 0; };
    const bool fused = dtype != KVP_F32 && D % 16 == 0 && B * H <= 65535 && n * D < ((int64_t)1 << 31) && S < ((int64_t)1 << 31) &&
                       al16((int64_t)(uintptr_t)k) && al16((int64_t)(uintptr_t)v) && al16((int64_t)(uintptr_t)k_out) &&
                       al16((int64_t)(uintptr_t)v_out) && al16((int64_t)(uintptr_t)inv_freq) && al16(k_sb * 2) && al16(k_sh * 2) && al16(k_ss * 2) &&
                       al16(v_sb * 2) && al16(v_sh * 2) && al16(v_ss * 2);
    if (!fused) {   // any other shape / dtype: the two kernels it replaces
        if (int rc = kvp_gather_kv(k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, H, S, D, idx, n, k_out, v_out, stream_)) return rc;
        return kvp_rerotate_keys(k_out, dtype, B, H, n, D, idx, inv_freq, stream_);
    }
    GrArgs a;
    a.k = static_cast<const char*>(k); a.v = static_cast<const char*>(v);
    a.ko = static_cast<char*>(k_out); a.vo = static_cast<char*>(v_out);
    a.idx = idx; a.inv_freq = inv_freq;
    a.k_sb = k_sb * 2; a.k_sh = k_sh * 2; a.k_ss = k_ss * 2;
    a.v_sb = v_sb * 2; a.v_sh = v_sh * 2; a.v_ss = v_ss * 2;
    a.H = (uint32_t)H; a.S = (uint32_t)S; a.n = (uint32_t)n; a.D = (uint32_t)D;
    const uint32_t BH = (uint32_t)(B * H);
    const uint64_t total = (uint64_t)n * (uint64_t)(D / 16);
    const uint64_t bx_full = (total + 256 * GR_UNROLL - 1) / (256 * GR_UNROLL);
    const uint64_t bx_cap = std::max<uint64_t>(1, ((uint64_t)256 * 8 + BH - 1) / BH);
    const dim3 grid((uint32_t)std::max<uint64_t>(1, std::min(bx_full, bx_cap)), BH);
    const bool nt = (uint64_t)B * H * S * D * 2 * 2 > (192ull << 20);   // the same rule as kvp_gather_kv
#define KVP_GR(DTV, NTV) KVP_LAUNCH("gather_rerotate_kernel", stream, (gather_rerotate_kernel<DTV, NTV><<<grid, 256, 0, stream>>>(a)))
    if (dtype == KVP_F16) { if (nt) KVP_GR(KVP_F16, true); else KVP_GR(KVP_F16, false); }
    else { if (nt) KVP_GR(KVP_BF16, true); else KVP_GR(KVP_BF16, false); }
#undef KVP_GR
    KVP_CHECK_LAUNCH("gather_rerotate");
    return KVP_OK;
}
