// kvp_snapkv_score / kvp_snapkv_score_from_attn -- SnapKVPress.score
// (kvpress/presses/snapkv_press.py:60-105) from the RoPE'd window queries on.
//
// Reference dataflow at S=128k: repeat_kv (1 GiB) -> QK^T logits [32,64,S]
// (512 MiB) -> 3 mask temporaries -> fp32 softmax (1 GiB) -> mean / avg_pool1d / group mean
// -> F.pad with a host-synchronising .item().  Here nothing of size [Hq, W, S] ever exists:
//
//   pass 1  (p1)      per (kv-head, key chunk): logits tile -> per-row partial (max, sum-exp)
//   combine           per row: a = M + log2 Z                      (softmax_stats.h)
//   pass 2  (p2)      per (kv-head, key chunk): recompute the logits tile, P = 2^(L2 - a),
//                     column sums over all G*W rows of the kv-head -> colsum[B,Hkv,S-W]
//   pool              avg_pool1d(kernel) of colsum, scaled by 1/(G*W*kernel); device-side max
//   fill              last W positions <- max + 1                   (no host sync)
//
// mean-over-window, pool and mean-over-group are all linear, so summing the G*W rows first
// and pooling once is the same arithmetic up to fp32 rounding.  The causal mask
// (triu(-inf, diagonal=S-W+1), snapkv_press.py:63-65) only touches the last W columns, which
// are dropped from the result (:67): it matters for the normaliser (pass 1) only.
//
// Two implementations of p1/p2: the MFMA kernels in snapkv_mfma.hip (bf16 / f16, head size 64 / 96 / 128 / 256, G <= 16, ANY window size
// since round 6: blocks of 64 padded window rows, snapkv_internal.h -- hand-scheduled loops for D = 128 with G % 4 == 0, the
// Llama-3.1-8B hot path, compiler-scheduled kernels otherwise) and the generic VALU kernels in this file (any D, float32, any
// stride: correctness fallbacks, ~40x slower -- profiles/r06_shape_sweep*.txt).
#include "kvp_common.h"
#include "softmax_stats.h"
#include "snapkv_internal.h"
#include "topk_internal.h"

namespace {

constexpr int SK_THREADS = 256;
constexpr int SK_SUB = 64;           // keys per sub-tile (= one wave wide)
constexpr int SK_CHUNK_GENERIC = 512;  // keys per workgroup in the generic kernels

template <int DT>
__device__ __forceinline__ void load_k_subtile(const typename Elem<DT>::T* __restrict__ kbase, int64_t k_ss,
                                               uint32_t key0, uint32_t S, uint32_t D, float* kt) {
    for (uint32_t e = threadIdx.x; e < SK_SUB * D; e += SK_THREADS) {
        const uint32_t r = e / D, d = e - r * D;
        const uint32_t kk = key0 + r;
        kt[r * (D + 1) + d] = kk < S ? Elem<DT>::ld(kbase + (int64_t)kk * k_ss + d) : 0.f;
    }
}

// ---- generic pass 1: partial (max, sum-exp) per (b, hq, w, chunk) ------------------------------
template <int DT>
__global__ __launch_bounds__(SK_THREADS) void snapkv_p1_generic(SnapArgs a, uint32_t nchunk, float* __restrict__ part_m,
                                                                float* __restrict__ part_z) {
    using T = typename Elem<DT>::T;
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    float* kt = sk_lds;                            // [64][D+1]
    float* m_run = kt + SK_SUB * (a.D + 1);        // [W]
    float* z_run = m_run + a.W;                    // [W]
    const uint32_t chunk = blockIdx.x, hq = blockIdx.y, b = blockIdx.z;
    const uint32_t h = hq / a.G;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const T* qb = static_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)hq * a.q_sh;
    const T* kb = static_cast<const T*>(a.k) + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    for (uint32_t r = threadIdx.x; r < a.W; r += SK_THREADS) { m_run[r] = KVP_NEG_INF; z_run[r] = 0.f; }
    const uint32_t kbeg = chunk * SK_CHUNK_GENERIC;
    const uint32_t kend = min(kbeg + SK_CHUNK_GENERIC, a.S);
    for (uint32_t key0 = kbeg; key0 < kend; key0 += SK_SUB) {
        __syncthreads();
        load_k_subtile<DT>(kb, a.k_ss, key0, a.S, a.D, kt);
        __syncthreads();
        const uint32_t kk = key0 + lane;
        const float* krow = kt + lane * (a.D + 1);
        for (uint32_t r = wv; r < a.W; r += SK_THREADS / 64) {
            const T* qr = qb + (int64_t)r * a.q_sw;
            float dot = 0.f;
            for (uint32_t d = 0; d < a.D; ++d) dot = fmaf(Elem<DT>::ld(qr + d), krow[d], dot);
            // window row r is token S-W+r and may attend keys <= S-W+r
            const bool masked = kk >= a.S || kk > a.S - a.W + r;
            const float l2 = masked ? KVP_NEG_INF : dot * a.c;
            const float m = wave_max(l2);
            if (m != KVP_NEG_INF) {
                const float z = wave_sum(masked ? 0.f : exp2f(l2 - m));
                if (lane == 0) {
                    float mr = m_run[r], zr = z_run[r];
                    softmax_merge(mr, zr, m, z);
                    m_run[r] = mr; z_run[r] = zr;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < a.W; r += SK_THREADS) {
        const size_t o = ((size_t)(b * a.Hq + hq) * a.W + r) * nchunk + chunk;
        part_m[o] = m_run[r];
        part_z[o] = z_run[r];
    }
}

// ---- generic pass 2: colsum[b,h,c] = sum over the G*W rows of 2^(L2 - a_row) --------------------
template <int DT>
__global__ __launch_bounds__(SK_THREADS) void snapkv_p2_generic(SnapArgs a, const float* __restrict__ rowstat,
                                                                float* __restrict__ colsum) {
    using T = typename Elem<DT>::T;
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    float* kt = sk_lds;                      // [64][D+1]
    float* red = kt + SK_SUB * (a.D + 1);    // [4][64]
    const uint32_t chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t Sm = a.S - a.W;
    const T* kb = static_cast<const T*>(a.k) + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh;
    const uint32_t kbeg = chunk * SK_CHUNK_GENERIC;
    const uint32_t kend = min(kbeg + SK_CHUNK_GENERIC, Sm);
    for (uint32_t key0 = kbeg; key0 < kend; key0 += SK_SUB) {
        __syncthreads();
        load_k_subtile<DT>(kb, a.k_ss, key0, a.S, a.D, kt);
        __syncthreads();
        const float* krow = kt + lane * (a.D + 1);
        float acc = 0.f;
        for (uint32_t g = 0; g < a.G; ++g) {
            const uint32_t hq = h * a.G + g;
            const T* qb = static_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)hq * a.q_sh;
            const float* ar = rowstat + (size_t)(b * a.Hq + hq) * a.W;
            for (uint32_t r = wv; r < a.W; r += SK_THREADS / 64) {
                const T* qr = qb + (int64_t)r * a.q_sw;
                float dot = 0.f;
                for (uint32_t d = 0; d < a.D; ++d) dot = fmaf(Elem<DT>::ld(qr + d), krow[d], dot);
                acc += exp2f(fmaf(dot, a.c, -ar[r]));
            }
        }
        red[wv * 64 + lane] = acc;
        __syncthreads();
        if (wv == 0) {
            const uint32_t kk = key0 + lane;
            if (kk < Sm) colsum[(size_t)(b * a.Hkv + h) * Sm + kk] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        }
    }
}

// ---- colsum from given attention weights: attn[..., -W:, :-W] (snapkv_press.py:88-89) ----------
template <int DT>
__global__ __launch_bounds__(SK_THREADS) void snapkv_colsum_from_attn(const typename Elem<DT>::T* __restrict__ attn,
                                                                      int64_t sb, int64_t sh, int64_t sw, uint32_t B,
                                                                      uint32_t Hq, uint32_t Hkv, uint32_t Sm, uint32_t W,
                                                                      float* __restrict__ colsum) {
    const uint32_t G = Hq / Hkv;
    const uint64_t total = (uint64_t)B * Hkv * Sm;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(i % Sm);
        const uint32_t bh = (uint32_t)(i / Sm);
        const uint32_t b = bh / Hkv, h = bh - b * Hkv;
        float acc = 0.f;
        for (uint32_t g = 0; g < G; ++g) {
            const typename Elem<DT>::T* p = attn + (int64_t)b * sb + (int64_t)(h * G + g) * sh + c;
            for (uint32_t w = 0; w < W; ++w) acc += Elem<DT>::ld(p + (int64_t)w * sw);
        }
        colsum[i] = acc;
    }
}

// ---- RoPE of the window queries: q*cos + rotate_half(q)*sin, with torch's per-op rounding --------
template <int DT> __device__ __forceinline__ void st_dt(typename Elem<DT>::T* p, float x);
template <> __device__ __forceinline__ void st_dt<KVP_F32>(float* p, float x) { *p = x; }
template <> __device__ __forceinline__ void st_dt<KVP_F16>(_Float16* p, float x) { *p = (_Float16)x; }
template <> __device__ __forceinline__ void st_dt<KVP_BF16>(uint16_t* p, float x) { *p = (uint16_t)(__float_as_uint(round_dt<KVP_BF16>(x)) >> 16); }

// one thread per (row, d < D/2): out[d] = q[d]*cos[d] - q[d+h]*sin[d];  out[d+h] = q[d+h]*cos[d+h] + q[d]*sin[d+h]
template <int DT>
__global__ __launch_bounds__(SK_THREADS) void snapkv_rope_kernel(const typename Elem<DT>::T* __restrict__ q, int64_t q_sb,
                                                                 int64_t q_sh, int64_t q_sw,
                                                                 const typename Elem<DT>::T* __restrict__ cosp,
                                                                 const typename Elem<DT>::T* __restrict__ sinp, int64_t cs_sb,
                                                                 int64_t cs_sw, uint32_t B, uint32_t Hq, uint32_t W, uint32_t D,
                                                                 typename Elem<DT>::T* __restrict__ out) {
    const uint32_t half = D / 2;
    const uint32_t total = B * Hq * W * half;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t d = i % half, row = i / half;
        const uint32_t w = row % W, bh = row / W;
        const uint32_t hq = bh % Hq, b = bh / Hq;
        const typename Elem<DT>::T* qr = q + (int64_t)b * q_sb + (int64_t)hq * q_sh + (int64_t)w * q_sw;
        const typename Elem<DT>::T* cr = cosp + (int64_t)b * cs_sb + (int64_t)w * cs_sw;
        const typename Elem<DT>::T* sr = sinp + (int64_t)b * cs_sb + (int64_t)w * cs_sw;
        const float q0 = Elem<DT>::ld(qr + d), q1 = Elem<DT>::ld(qr + d + half);
        const float lo = rope_elem<DT>(q0, Elem<DT>::ld(cr + d), -q1, Elem<DT>::ld(sr + d));
        const float hi = rope_elem<DT>(q1, Elem<DT>::ld(cr + d + half), q0, Elem<DT>::ld(sr + d + half));
        typename Elem<DT>::T* o = out + (size_t)row * D;
        st_dt<DT>(o + d, lo);
        st_dt<DT>(o + d + half, hi);
    }
}

// 2-byte dtypes, D % 16 == 0, 16-byte aligned rows: a thread owns 8 consecutive dims of the first half and the matching 8 of the second
// half (six 16-byte loads, two 16-byte stores; the same rope_elem arithmetic: bit-identical).  The scalar kernel above moves 2 bytes per
// lane and instruction and divides three times per element: 81 us for the 33.5 M elements of a 128-chunk ChunkPress window batch.
template <int DT> __device__ __forceinline__ uint32_t rope_pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t rope_pack2<KVP_BF16>(float lo, float hi) {
    return (__float_as_uint(round_dt<KVP_BF16>(lo)) >> 16) | (__float_as_uint(round_dt<KVP_BF16>(hi)) & 0xFFFF0000u);
}
template <> __device__ __forceinline__ uint32_t rope_pack2<KVP_F16>(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
template <int DT>
__global__ __launch_bounds__(SK_THREADS) void snapkv_rope_vec_kernel(const typename Elem<DT>::T* __restrict__ q, int64_t q_sb, int64_t q_sh, int64_t q_sw,
                                                                     const typename Elem<DT>::T* __restrict__ cosp,
                                                                     const typename Elem<DT>::T* __restrict__ sinp, int64_t cs_sb, int64_t cs_sw, uint32_t B,
                                                                     uint32_t Hq, uint32_t W, uint32_t D, typename Elem<DT>::T* __restrict__ out) {
    const uint32_t half = D / 2, tpr = half / 8;   // threads per row
    const uint32_t total = B * Hq * W * tpr;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t row = i / tpr, d0 = (i - row * tpr) * 8;
        const uint32_t w = row % W, bh = row / W;
        const uint32_t hq = bh % Hq, b = bh / Hq;
        const typename Elem<DT>::T* qr = q + (int64_t)b * q_sb + (int64_t)hq * q_sh + (int64_t)w * q_sw;
        const typename Elem<DT>::T* cr = cosp + (int64_t)b * cs_sb + (int64_t)w * cs_sw;
        const typename Elem<DT>::T* sr = sinp + (int64_t)b * cs_sb + (int64_t)w * cs_sw;
        float q0[8], q1[8], c0[8], c1[8], s0[8], s1[8], lo[8], hi[8];
        unpack16<DT>(*reinterpret_cast<const uint4*>(qr + d0), q0);
        unpack16<DT>(*reinterpret_cast<const uint4*>(qr + d0 + half), q1);
        unpack16<DT>(*reinterpret_cast<const uint4*>(cr + d0), c0);
        unpack16<DT>(*reinterpret_cast<const uint4*>(cr + d0 + half), c1);
        unpack16<DT>(*reinterpret_cast<const uint4*>(sr + d0), s0);
        unpack16<DT>(*reinterpret_cast<const uint4*>(sr + d0 + half), s1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lo[e] = rope_elem<DT>(q0[e], c0[e], -q1[e], s0[e]);
            hi[e] = rope_elem<DT>(q1[e], c1[e], q0[e], s1[e]);
        }
        typename Elem<DT>::T* o = out + (size_t)row * D;
        *reinterpret_cast<uint4*>(o + d0) = make_uint4(rope_pack2<DT>(lo[0], lo[1]), rope_pack2<DT>(lo[2], lo[3]), rope_pack2<DT>(lo[4], lo[5]), rope_pack2<DT>(lo[6], lo[7]));
        *reinterpret_cast<uint4*>(o + d0 + half) = make_uint4(rope_pack2<DT>(hi[0], hi[1]), rope_pack2<DT>(hi[2], hi[3]), rope_pack2<DT>(hi[4], hi[5]), rope_pack2<DT>(hi[6], hi[7]));
    }
}

// ---- avg_pool1d(kernel, pad=kernel/2, zero padded, divisor = kernel) + scaling + global max ----
// HIST (fused compress): instead of the block maximum for the pad value, the kernel accumulates the top-k's first radix
// histogram of the S - W scores it writes; the pad columns are appended to the selection by construction.
template <bool HIST>
__global__ __launch_bounds__(SK_THREADS) void snapkv_pool_kernel(const float* __restrict__ colsum, uint32_t S, uint32_t W,
                                                                 int pad, float inv, float* __restrict__ scores,
                                                                 float* __restrict__ bmax, uint32_t* __restrict__ hist1) {
    __shared__ float scr[4];
    __shared__ uint32_t lh[HIST ? 4096 : 1];
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < 4096; i += SK_THREADS) lh[i] = 0;
        __syncthreads();
    }
    const uint32_t Sm = S - W, bh = blockIdx.y;
    const float* __restrict__ row = colsum + (size_t)bh * Sm;
    float* __restrict__ out = scores + (size_t)bh * S;
    float vmax = KVP_NEG_INF;
    // (loop bound rounded up to whole waves: the histogram's wave-level aggregation wants converged lanes)
    const uint32_t Smw = (Sm + 63) & ~63u;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < Smw; c += gridDim.x * blockDim.x) {
        float s = 0.f;
        if (c < Sm) {
            for (int j = -pad; j <= pad; ++j) {
                const int cc = (int)c + j;
                if (cc >= 0 && cc < (int)Sm) s += row[cc];
            }
            s *= inv;
            out[c] = s;
            vmax = fmaxf(vmax, s);
        }
        if (HIST) topk_hist1_add(lh, s, c < Sm);
    }
    if (HIST) {
        __syncthreads();
        topk_hist1_flush(lh, hist1 + (size_t)bh * 4096);
    } else {
        block_store_max(vmax, scr, bmax, blockIdx.y * gridDim.x + blockIdx.x);
    }
}

// The same for kernel_size 5 with four consecutive scores per thread and iteration: 8-byte loads of the 8 inputs they share, one
// 16-byte store, term-for-term the additions of the kernel above (same scores, same histogram).  Needs S - W even and S % 4 == 0
// (alignment of the row starts); the row's first and last group take the guarded loads.
template <bool HIST>
__global__ __launch_bounds__(SK_THREADS) void snapkv_pool5_vec_kernel(const float* __restrict__ colsum, uint32_t S, uint32_t W, float inv,
                                                                      float* __restrict__ scores, float* __restrict__ bmax,
                                                                      uint32_t* __restrict__ hist1) {
    __shared__ float scr[4];
    __shared__ uint32_t lh[HIST ? 4096 : 1];
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < 4096; i += SK_THREADS) lh[i] = 0;
        __syncthreads();
    }
    const uint32_t Sm = S - W, bh = blockIdx.y;
    const float* __restrict__ row = colsum + (size_t)bh * Sm;
    float* __restrict__ out = scores + (size_t)bh * S;
    float vmax = KVP_NEG_INF;
    const uint32_t ngrp = (Sm + 3) / 4, ngrpw = (ngrp + 63) & ~63u;   // whole waves: the histogram's aggregation wants converged lanes
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngrpw; g += gridDim.x * blockDim.x) {
        const uint32_t c0 = 4 * g;
        float x[8], s[4];
        if (c0 >= 2 && c0 + 5 < Sm) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 v = *reinterpret_cast<const float2*>(row + c0 - 2 + 2 * i);
                x[2 * i] = v.x; x[2 * i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cc = (int)c0 - 2 + i;
                x[i] = (cc >= 0 && cc < (int)Sm) ? row[cc] : 0.f;   // (adding the zero of a position outside the row changes nothing: sums are >= +0)
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) t += x[i + j];
            s[i] = t * inv;
        }
        if (c0 + 3 < Sm) {
            *reinterpret_cast<float4*>(out + c0) = make_float4(s[0], s[1], s[2], s[3]);
            vmax = fmaxf(fmaxf(vmax, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + i < Sm) {
                    out[c0 + i] = s[i];
                    vmax = fmaxf(vmax, s[i]);
                }
        }
        if (HIST) {
#pragma unroll
            for (int i = 0; i < 4; ++i) topk_hist1_add(lh, s[i], c0 + i < Sm);
        }
    }
    if (HIST) {
        __syncthreads();
        topk_hist1_flush(lh, hist1 + (size_t)bh * 4096);
    } else {
        block_store_max(vmax, scr, bmax, blockIdx.y * gridDim.x + blockIdx.x);
    }
}

struct SnapWs {
    float* part_m;
    float* part_z;
    float* rowstat;
    float* colsum;
    float* colsum2;  // G > 4 (several group-blocks of four q-heads per kv-head in the MFMA pass 2): [group-blocks - 1] slabs of column sums, added in block order
    float* colsumx;  // W > 64 (several 64-row blocks of the window in the MFMA passes): the later blocks' column sums, added in block order
    float* bmax;  // per-workgroup maxima of the pool kernel (<= 4096)
    void* qrot;   // [B,Hq,W,D] RoPE'd window queries (kvp_snapkv_score_rope)
    uint32_t* p1_ticks;   // [planes][nchunk] pass-1 workgroup times -> pass 2's tile ranges (snapkv_internal.h: snapkv_p2_shares_plan)
    size_t total_bytes;
};

SnapWs carve_snap_ws(void* ws, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D) {
    SnapWs w;
    size_t off = 0;
    char* base = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += kvp_align_up(bytes, 256);
        return p;
    };
    // generic kernels: one workgroup per 512 keys; MFMA kernels: up to one per 128-key tile, never more than 256 (snapkv_mfma_nchunk)
    const int64_t nchunk_max = std::max<int64_t>((S + SK_CHUNK_GENERIC - 1) / SK_CHUNK_GENERIC, std::min<int64_t>((S + 127) / 128, 256));
    const size_t rows = (size_t)B * Hq * snapkv_wp((uint32_t)W);   // the MFMA path keeps its statistics per PADDED window row (snapkv_internal.h)
    w.bmax = (float*)take((size_t)std::max<int64_t>(4096, B * Hkv) * 4);
    w.part_m = (float*)take(rows * nchunk_max * 4);
    w.part_z = (float*)take(rows * nchunk_max * 4);
    w.rowstat = (float*)take(rows * 4);
    w.colsum = (float*)take((size_t)B * Hkv * (S > W ? S - W : 0) * 4);
    const int64_t ngb = (Hq / std::max<int64_t>(1, Hkv) + 3) / 4;   // group-blocks of four q-heads per kv-head (MFMA pass 2: one column-sum slab each)
    w.colsum2 = ngb > 1 ? (float*)take((size_t)(ngb - 1) * B * Hkv * (S > W ? S - W : 0) * 4) : nullptr;
    w.colsumx = W > 64 ? (float*)take((size_t)B * Hkv * (S > W ? S - W : 0) * 4) : nullptr;
    w.qrot = take((size_t)B * Hq * W * D * 4);
    // (planes * nchunk <= max(planes, 256): one resident round of workgroups; planes = B * Hkv * group-blocks)
    const size_t nwg = (size_t)std::max<int64_t>(B * Hkv * ngb, 256);
    w.p1_ticks = (uint32_t*)take(nwg * 4);
    w.total_bytes = off;
    return w;
}

// hist1 != nullptr (fused compress): scores[.., :S-W] are written and histogrammed, the pad columns are NOT filled
// finish = SNAP_FINISH_NO_PAD without hist1: the same, for a select that needs no histogram (short rows, topk_row_kernel);
// SNAP_FINISH_COLSUM: nothing is launched, the caller selects straight from w.colsum (topk_select_pooled_rows)
int finish_scores(const SnapWs& w, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int kernel_size,
                  float* scores, hipStream_t stream, uint32_t* hist1 = nullptr, int finish = SNAP_FINISH_FULL) {
    if (finish == SNAP_FINISH_COLSUM) return KVP_OK;
    const bool skip_pad = finish != SNAP_FINISH_FULL;
    const uint32_t BH = (uint32_t)(B * Hkv);
    const float inv = snapkv_pool_scale(Hq, Hkv, W, kernel_size);
    // four scores per thread: kernel_size 5, aligned row starts, long rows (short ones are launch-bound either way)
    const bool vec = kernel_size == 5 && (S - W) % 2 == 0 && S % 4 == 0 && ((uintptr_t)scores % 16) == 0 && S - W >= 8192;
    const uint64_t per_wg = (uint64_t)SK_THREADS * (vec ? 4 : 1);
    const uint64_t wg_cap = vec ? 1024 : 2048;   // <= 4096: the size of w.bmax
    const uint32_t bx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)(S - W) + per_wg - 1) / per_wg, std::max<uint64_t>(1, std::min<uint64_t>(wg_cap, 4096) / BH)));
    const dim3 grid(bx, BH);
    const int pad = kernel_size / 2;
    if (hist1) {
        if (vec) KVP_LAUNCH("snapkv_pool_kernel", stream, snapkv_pool5_vec_kernel<true><<<grid, SK_THREADS, 0, stream>>>(w.colsum, (uint32_t)S, (uint32_t)W, inv, scores, w.bmax, hist1));
        else KVP_LAUNCH("snapkv_pool_kernel", stream, snapkv_pool_kernel<true><<<grid, SK_THREADS, 0, stream>>>(w.colsum, (uint32_t)S, (uint32_t)W, pad, inv, scores, w.bmax, hist1));
        KVP_CHECK_LAUNCH("snapkv(pool+hist)");
        return KVP_OK;
    }
    if (vec) KVP_LAUNCH("snapkv_pool_kernel", stream, snapkv_pool5_vec_kernel<false><<<grid, SK_THREADS, 0, stream>>>(w.colsum, (uint32_t)S, (uint32_t)W, inv, scores, w.bmax, nullptr));
    else KVP_LAUNCH("snapkv_pool_kernel", stream, snapkv_pool_kernel<false><<<grid, SK_THREADS, 0, stream>>>(w.colsum, (uint32_t)S, (uint32_t)W, pad, inv, scores, w.bmax, nullptr));
    if (skip_pad) {
        KVP_CHECK_LAUNCH("snapkv(pool)");
        return KVP_OK;
    }
    const uint32_t nfill = BH * (uint32_t)W;
    KVP_LAUNCH("fill_pad_kernel", stream, fill_pad_kernel<<<(nfill + 255) / 256, 256, 0, stream>>>(scores, BH, (uint32_t)S, (uint32_t)(S - W), (uint32_t)W, w.bmax, bx * BH));
    KVP_CHECK_LAUNCH("snapkv(pool/fill)");
    return KVP_OK;
}

int check_common(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int kernel_size) {
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0, "snapkv: bad heads B=%ld Hq=%ld Hkv=%ld", (long)B, (long)Hq, (long)Hkv);
    KVP_CHECK_ARG(W >= 1 && S > W, "snapkv: query length %ld should be greater than the window size %ld", (long)S, (long)W);
    KVP_CHECK_ARG(kernel_size >= 1 && (kernel_size & 1), "snapkv: kernel_size must be odd (got %d)", kernel_size);
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B * Hq <= 65535 && B <= 65535, "snapkv: shape too large");
    return KVP_OK;
}

}  // namespace

extern "C" size_t kvp_snapkv_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D) {
    if (B < 1 || Hq < 1 || Hkv < 1 || S < 1 || W < 1 || D < 1) return 256;
    return carve_snap_ws(nullptr, B, Hq, Hkv, S, W, D).total_bytes;
}

float* snapkv_ws_colsum(void* ws, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D) {
    return carve_snap_ws(ws, B, Hq, Hkv, S, W, D).colsum;
}
float snapkv_pool_scale(int64_t Hq, int64_t Hkv, int64_t W, int kernel_size) {
    return (float)(1.0 / ((double)(Hq / Hkv) * (double)W * (double)kernel_size));
}

int snapkv_score_impl(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* k, int64_t k_sb,
                      int64_t k_sh, int64_t k_ss, int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S,
                      int64_t W, int64_t D, int kernel_size, float* scores, void* ws, size_t ws_bytes,
                      hipStream_t stream, uint32_t* hist1, bool count_norm = false, int finish = SNAP_FINISH_FULL) {
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "snapkv: bad dtype %d", dtype);
    if (int rc = check_common(B, Hq, Hkv, S, W, kernel_size)) return rc;
    KVP_CHECK_ARG(D >= 1 && D <= 1024, "snapkv: unsupported head_dim %ld", (long)D);
    const uint32_t norm_base = count_norm ? (uint32_t)(S - W) : 0;  // window row w sees S - W + w keys
    KVP_CHECK_ARG(q && k && scores, "snapkv: null pointer");
    SnapWs w = carve_snap_ws(ws, B, Hq, Hkv, S, W, D);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("snapkv: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }

    SnapArgs a;
    a.q = q; a.k = k;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_sw = q_sw;
    a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.B = (uint32_t)B; a.Hq = (uint32_t)Hq; a.Hkv = (uint32_t)Hkv; a.G = (uint32_t)(Hq / Hkv);
    a.S = (uint32_t)S; a.W = (uint32_t)W; a.D = (uint32_t)D;
    a.Wp = snapkv_wp(a.W); a.rblk = 0;
    a.c = (float)(1.4426950408889634 / sqrt((double)D));
    const uint32_t nrows = (uint32_t)(B * Hq * W);

    if (snapkv_mfma_eligible(a, dtype)) {
        const uint32_t nchunk = snapkv_mfma_nchunk(a);
        const bool shares = snapkv_p2_shares_plan(a, nchunk);   // pass 2's tile ranges from pass 1's workgroup times (snapkv_internal.h)
        if (int rc = snapkv_mfma_p1(a, dtype, nchunk, w.part_m, w.part_z, shares ? w.p1_ticks : nullptr, stream)) return rc;
        // (this path's statistics are per PADDED window row: Wp rows per head, the first Wp - W of them padding with normaliser +inf)
        const uint32_t nrows_p = (uint32_t)(B * Hq) * a.Wp, pad = a.Wp - a.W;
        KVP_SOFTMAX_COMBINE(stream, w.part_m, w.part_z, nrows_p, nchunk, w.rowstat, a.Wp, norm_base, pad);
        if (int rc = snapkv_mfma_p2(a, dtype, w.rowstat, w.colsum, w.colsum2, w.colsumx, shares ? w.p1_ticks : nullptr, stream)) return rc;
    } else {
        const uint32_t nchunk = (uint32_t)((S + SK_CHUNK_GENERIC - 1) / SK_CHUNK_GENERIC);
        const size_t lds1 = ((size_t)SK_SUB * (D + 1) + 2 * (size_t)W) * 4;
        const size_t lds2 = ((size_t)SK_SUB * (D + 1) + 256) * 4;
        KVP_CHECK_ARG(lds1 <= 160 * 1024 && lds2 <= 160 * 1024, "snapkv: W=%ld, D=%ld exceed the generic kernel's LDS budget", (long)W, (long)D);
        if (lds1 > 64 * 1024 || lds2 > 64 * 1024) {
            // head sizes beyond ~250 (Gemma's 256): raise the kernels' dynamic-LDS limit (a CU has 160 KiB) -- round 6: D = 256 used to be
            // refused although the header promises D <= 1024 (the 64-key sub-tile of float32 K rows is what grows with D)
            const void* fns[2] = {nullptr, nullptr};
            if (dtype == KVP_F32) { fns[0] = (const void*)snapkv_p1_generic<KVP_F32>; fns[1] = (const void*)snapkv_p2_generic<KVP_F32>; }
            else if (dtype == KVP_F16) { fns[0] = (const void*)snapkv_p1_generic<KVP_F16>; fns[1] = (const void*)snapkv_p2_generic<KVP_F16>; }
            else { fns[0] = (const void*)snapkv_p1_generic<KVP_BF16>; fns[1] = (const void*)snapkv_p2_generic<KVP_BF16>; }
            for (int i = 0; i < 2; ++i)
                if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds1, lds2)) != hipSuccess) {
                    kvp_set_error("snapkv: cannot raise the dynamic LDS limit to %zu bytes (D=%ld, W=%ld)", std::max(lds1, lds2), (long)D, (long)W);
                    return KVP_EHIP;
                }
        }
        const dim3 g1(nchunk, (uint32_t)Hq, (uint32_t)B);
        const uint32_t nchunk2 = (uint32_t)((S - W + SK_CHUNK_GENERIC - 1) / SK_CHUNK_GENERIC);
        const dim3 g2(nchunk2, (uint32_t)Hkv, (uint32_t)B);
#define KVP_SK_GENERIC(DT)                                                                                \
    KVP_LAUNCH("snapkv_p1_generic", stream, snapkv_p1_generic<DT><<<g1, SK_THREADS, lds1, stream>>>(a, nchunk, w.part_m, w.part_z));               \
    KVP_LAUNCH("softmax_combine_kernel", stream, softmax_combine_kernel<<<(nrows + 3) / 4, 256, 0, stream>>>(w.part_m, w.part_z, nrows, nchunk, w.rowstat, (uint32_t)W, norm_base)); \
    KVP_LAUNCH("snapkv_p2_generic", stream, snapkv_p2_generic<DT><<<g2, SK_THREADS, lds2, stream>>>(a, w.rowstat, w.colsum));
        if (dtype == KVP_F32) { KVP_SK_GENERIC(KVP_F32) }
        else if (dtype == KVP_F16) { KVP_SK_GENERIC(KVP_F16) }
        else { KVP_SK_GENERIC(KVP_BF16) }
#undef KVP_SK_GENERIC
    }
    KVP_CHECK_LAUNCH("snapkv(p1/p2)");
    return finish_scores(w, B, Hq, Hkv, S, W, kernel_size, scores, stream, hist1, finish);
}

extern "C" int kvp_snapkv_score(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* k, int64_t k_sb,
                                int64_t k_sh, int64_t k_ss, int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S,
                                int64_t W, int64_t D, int kernel_size, float* scores, void* ws, size_t ws_bytes,
                                kvp_stream_t stream_) {
    return snapkv_score_impl(q, q_sb, q_sh, q_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D, kernel_size, scores, ws, ws_bytes,
                             static_cast<hipStream_t>(stream_), nullptr);
}

int snapkv_score_rope_impl(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* cosp, const void* sinp,
                           int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                           int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                           float* scores, void* ws, size_t ws_bytes, hipStream_t stream, uint32_t* hist1, bool count_norm, int finish) {
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "snapkv: bad dtype %d", dtype);
    if (int rc = check_common(B, Hq, Hkv, S, W, kernel_size)) return rc;
    KVP_CHECK_ARG(D >= 2 && D <= 1024 && D % 2 == 0, "snapkv: RoPE needs an even head_dim (got %ld)", (long)D);
    KVP_CHECK_ARG(q && cosp && sinp && k && scores, "snapkv: null pointer");
    KVP_CHECK_ARG(B * Hq * W * D < ((int64_t)1 << 31), "snapkv: window too large");
    SnapWs w = carve_snap_ws(ws, B, Hq, Hkv, S, W, D);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("snapkv: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    const uint32_t total = (uint32_t)(B * Hq * W * (D / 2));
    const uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((total + SK_THREADS - 1) / SK_THREADS, 2048));
#define KVP_SK_ROPE(DT)                                                                                                      \
    KVP_LAUNCH("snapkv_rope_kernel", stream, snapkv_rope_kernel<DT><<<blocks, SK_THREADS, 0, stream>>>(                        \
        static_cast<const Elem<DT>::T*>(q), q_sb, q_sh, q_sw, static_cast<const Elem<DT>::T*>(cosp),                          \
        static_cast<const Elem<DT>::T*>(sinp), cs_sb, cs_sw, (uint32_t)B, (uint32_t)Hq, (uint32_t)W, (uint32_t)D,              \
        static_cast<Elem<DT>::T*>(w.qrot)));
    const bool rvec = dtype != KVP_F32 && D % 16 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)cosp % 16) == 0 && ((uintptr_t)sinp % 16) == 0 &&
                      ((uintptr_t)w.qrot % 16) == 0 && (q_sb * 2) % 16 == 0 && (q_sh * 2) % 16 == 0 && (q_sw * 2) % 16 == 0 && (cs_sb * 2) % 16 == 0 && (cs_sw * 2) % 16 == 0;
    const uint32_t vblocks = std::max<uint32_t>(1, std::min<uint32_t>((total / 8 + SK_THREADS - 1) / SK_THREADS, 2048));
#define KVP_SK_ROPE_VEC(DT)                                                                                                  \
    KVP_LAUNCH("snapkv_rope_kernel", stream, snapkv_rope_vec_kernel<DT><<<vblocks, SK_THREADS, 0, stream>>>(                   \
        static_cast<const Elem<DT>::T*>(q), q_sb, q_sh, q_sw, static_cast<const Elem<DT>::T*>(cosp),                          \
        static_cast<const Elem<DT>::T*>(sinp), cs_sb, cs_sw, (uint32_t)B, (uint32_t)Hq, (uint32_t)W, (uint32_t)D,              \
        static_cast<Elem<DT>::T*>(w.qrot)));
    if (rvec && dtype == KVP_F16) { KVP_SK_ROPE_VEC(KVP_F16) }
    else if (rvec) { KVP_SK_ROPE_VEC(KVP_BF16) }
    else if (dtype == KVP_F32) { KVP_SK_ROPE(KVP_F32) }
    else if (dtype == KVP_F16) { KVP_SK_ROPE(KVP_F16) }
    else { KVP_SK_ROPE(KVP_BF16) }
#undef KVP_SK_ROPE_VEC
#undef KVP_SK_ROPE
    KVP_CHECK_LAUNCH("snapkv(rope)");
    return snapkv_score_impl(w.qrot, Hq * W * D, W * D, D, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D, kernel_size, scores, ws,
                             ws_bytes, stream, hist1, count_norm, finish);
}

// window queries from the hidden states (fused q_proj + RoPE, qproj.hip), then as above
bool kvp_qproj_rope_eligible(int dtype, int64_t W, int64_t D, int64_t K, const void* x, int64_t x_sb, int64_t x_sw, const void* w,
                             const void* cosp, const void* sinp, int64_t cs_sb, int64_t cs_sw);
int kvp_qproj_rope_launch(const void* x, int64_t x_sb, int64_t x_sw, const void* w, const void* cosp, const void* sinp, int64_t cs_sb,
                          int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t K, void* out, hipStream_t stream);

int snapkv_score_hidden_impl(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden, const void* cosp,
                             const void* sinp, int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                             int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                             float* scores, void* ws, size_t ws_bytes, hipStream_t stream, uint32_t* hist1, int finish) {
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "snapkv: bad dtype %d", dtype);
    if (int rc = check_common(B, Hq, Hkv, S, W, kernel_size)) return rc;
    KVP_CHECK_ARG(hidden_win && wq && cosp && sinp && k && scores, "snapkv: null pointer");
    if (!kvp_qproj_rope_eligible(dtype, W, D, hidden, hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw)) {
        kvp_set_error("snapkv: fused q_proj needs bf16/f16, W = 64, D = 128, hidden %% 256 == 0, 16-byte aligned rows");
        return KVP_EUNSUPPORTED;
    }
    SnapWs w = carve_snap_ws(ws, B, Hq, Hkv, S, W, D);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("snapkv: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    if (int rc = kvp_qproj_rope_launch(hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw, dtype, B, Hq, hidden, w.qrot, stream)) return rc;
    return snapkv_score_impl(w.qrot, Hq * W * D, W * D, D, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D, kernel_size, scores, ws,
                             ws_bytes, stream, hist1, false, finish);
}

extern "C" int kvp_snapkv_score_hidden(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden,
                                       const void* cosp, const void* sinp, int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb,
                                       int64_t k_sh, int64_t k_ss, int dtype, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W,
                                       int64_t D, int kernel_size, float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    return snapkv_score_hidden_impl(hidden_win, x_sb, x_sw, wq, hidden, cosp, sinp, cs_sb, cs_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv,
                                    S, W, D, kernel_size, scores, ws, ws_bytes, static_cast<hipStream_t>(stream_), nullptr, SNAP_FINISH_FULL);
}

extern "C" int kvp_snapkv_score_rope(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* cosp, const void* sinp,
                                     int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                                     int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                                     float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    return snapkv_score_rope_impl(q, q_sb, q_sh, q_sw, cosp, sinp, cs_sb, cs_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D,
                                  kernel_size, scores, ws, ws_bytes, static_cast<hipStream_t>(stream_), nullptr, false);
}

extern "C" int kvp_snapkv_score_from_attn(const void* attn, int64_t a_sb, int64_t a_sh, int64_t a_sw, int dtype, int64_t B,
                                          int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int kernel_size, float* scores,
                                          void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "snapkv: bad dtype %d", dtype);
    if (int rc = check_common(B, Hq, Hkv, S, W, kernel_size)) return rc;
    KVP_CHECK_ARG(attn && scores, "snapkv: null pointer");
    SnapWs w = carve_snap_ws(ws, B, Hq, Hkv, S, W, 1);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("snapkv: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    const uint64_t total = (uint64_t)B * Hkv * (S - W);
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + SK_THREADS - 1) / SK_THREADS, 4096));
#define KVP_SK_ATTN(DT) \
    KVP_LAUNCH("snapkv_colsum_from_attn", stream, snapkv_colsum_from_attn<DT><<<blocks, SK_THREADS, 0, stream>>>(static_cast<const Elem<DT>::T*>(attn), a_sb, a_sh, a_sw, (uint32_t)B, (uint32_t)Hq, (uint32_t)Hkv, (uint32_t)(S - W), (uint32_t)W, w.colsum));
    if (dtype == KVP_F32) { KVP_SK_ATTN(KVP_F32) }
    else if (dtype == KVP_F16) { KVP_SK_ATTN(KVP_F16) }
    else { KVP_SK_ATTN(KVP_BF16) }
#undef KVP_SK_ATTN
    KVP_CHECK_LAUNCH("snapkv(from_attn)");
    return finish_scores(w, B, Hq, Hkv, S, W, kernel_size, scores, stream);
}

// FINCH scores (finch_press.py:56-83): the window attention of kvp_snapkv_score_rope for an arbitrary window (the
// question), each window row optionally weighted by its number of visible keys, no pooling.
extern "C" int kvp_finch_score(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* cosp, const void* sinp,
                               int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, int dtype,
                               int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int normalize_scores,
                               float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    return snapkv_score_rope_impl(q, q_sb, q_sh, q_sw, cosp, sinp, cs_sb, cs_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D,
                                  1, scores, ws, ws_bytes, static_cast<hipStream_t>(stream_), nullptr, normalize_scores != 0);
}
