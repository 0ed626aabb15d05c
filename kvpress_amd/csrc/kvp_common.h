// Shared device/host helpers for libkvpress_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/kvpress_hip.h"
#include "../../include/kvpress_hip_lab.h"

// ---- error reporting (thread-local last message; defined in capi.hip) ------------------------
void kvp_set_error(const char* fmt, ...);

#define KVP_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            kvp_set_error(__VA_ARGS__);          \
            return KVP_EINVAL;                   \
        }                                        \
    } while (0)

#define KVP_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            kvp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return KVP_EHIP;                                                       \
        }                                                                          \
    } while (0)

// ---- asynchronous failure reports (capi.hip) -------------------------------------------------------------------------------
// A kernel that detects at RUN time that its result is invalid (the cluster select when its workgroups never become co-resident,
// topk_cluster.hip) poisons its output AND stores a code into one process-wide, host-pinned status word; every entry point that
// produces or consumes a selection calls kvp_async_check() first and turns a pending report into KVP_EASYNC + kvp_last_error().
// The word holds (launch sequence number << 8) | code: all the stores of ONE failed launch make one report, however late the last of
// them lands (a repeat of a sequence number that was already reported is dropped).
uint32_t* kvp_async_flag();             // device-visible address of the status word (nullptr: pinned allocation failed -> poison only)
uint32_t kvp_async_next_seq();          // sequence number (1 .. 2^24 - 1, wrapping) for the next launch that may report
int kvp_async_check(const char* who);   // KVP_OK, or KVP_EASYNC exactly once per report

// ---- opt-in per-kernel timing (kvp_prof_* in include/kvpress_hip.h; implemented in capi.hip) ----
bool kvp_prof_enabled();
void kvp_prof_begin(const char* name, hipStream_t stream);
void kvp_prof_end(hipStream_t stream);
// every kernel launch of the library goes through this macro
#define KVP_LAUNCH(name, stream, ...)                         \
    do {                                                      \
        const bool prof__ = kvp_prof_enabled();               \
        if (prof__) kvp_prof_begin(name, stream);             \
        __VA_ARGS__;                                          \
        if (prof__) kvp_prof_end(stream);                     \
    } while (0)

static inline size_t kvp_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int kvp_elem_size(int dtype) { return dtype == KVP_F32 ? 4 : 2; }

// ---- element access ---------------------------------------------------------------------------
template <int DT> struct Elem;
template <> struct Elem<KVP_F32> {
    using T = float;
    static constexpr int PER16 = 4;  // elements per 16-byte vector
    static __device__ __forceinline__ float ld(const T* p) { return *p; }
};
template <> struct Elem<KVP_F16> {
    using T = _Float16;
    static constexpr int PER16 = 8;
    static __device__ __forceinline__ float ld(const T* p) { return (float)*p; }
};
template <> struct Elem<KVP_BF16> {
    using T = uint16_t;
    static constexpr int PER16 = 8;
    static __device__ __forceinline__ float ld(const T* p) { return __uint_as_float(((uint32_t)*p) << 16); }
};

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

// unpack a 16-byte vector into floats (4 for f32, 8 for f16/bf16)
template <int DT> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<KVP_F32>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<KVP_BF16>(const uint4& v, float* f) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
template <> __device__ __forceinline__ void unpack16<KVP_F16>(const uint4& v, float* f) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h2 p = __builtin_bit_cast(h2, w[i]);
        f[2 * i] = (float)p.x; f[2 * i + 1] = (float)p.y;
    }
}

// value rounded to the storage dtype and back (torch's eager ops round after every mul / add in the model dtype)
template <int DT> __device__ __forceinline__ float round_dt(float x);
template <> __device__ __forceinline__ float round_dt<KVP_F32>(float x) { return x; }
template <> __device__ __forceinline__ float round_dt<KVP_F16>(float x) {
    // opaque to the optimiser: written as (float)(_Float16)x, hipcc demotes the surrounding mul / add to f16 and then
    // contracts them into v_fma_f16 -- one rounding where torch's separate half ops have two
    float r;
    asm("v_cvt_f16_f32 %0, %1\n\tv_cvt_f32_f16 %0, %0" : "=v"(r) : "v"(x));
    return r;
}
template <> __device__ __forceinline__ float round_dt<KVP_BF16>(float x) {  // round-to-nearest-even to bf16
    // the hardware conversion (gfx950): one instruction + a shift instead of the 7-instruction integer emulation with its
    // inf / nan branch -- the RoPE / re-rotation kernels round six times per output pair and were bound by exactly that
#if defined(__gfx950__)
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return __uint_as_float(r << 16);
#else   // any other target of a stray --offload-arch: the integer round-to-nearest-even (NaN stays a quiet NaN)
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return __uint_as_float((u | 0x00400000u) & 0xFFFF0000u);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
#endif
}
// one RoPE output element with torch's rounding: round(round(a * ca) + round(b * sb)); __fmul_rn / __fadd_rn keep the
// ops separately rounded as in torch's eager mul, mul, add (never contracted to an fma)
template <int DT> __device__ __forceinline__ float rope_elem(float a, float ca, float b, float sb) {
    return round_dt<DT>(__fadd_rn(round_dt<DT>(__fmul_rn(a, ca)), round_dt<DT>(__fmul_rn(b, sb))));
}

// 16-byte loads/stores with an optional non-temporal (streaming, read/write-once) hint
typedef uint32_t kvp_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const void* p) {
    if (NT) {
        const kvp_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const kvp_u32x4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}
template <bool NT> __device__ __forceinline__ void st16(void* p, const uint4& v) {
    if (NT) {
        kvp_u32x4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<kvp_u32x4*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}

// tuning knobs read from the environment once (KVP_* variables; used for A/B runs on hardware)
int kvp_env_int(const char* name, int dflt);

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// float <-> order-preserving uint32 (larger float <=> larger key); -0.0 is canonicalised to +0.0
__device__ __forceinline__ uint32_t float_to_key(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
