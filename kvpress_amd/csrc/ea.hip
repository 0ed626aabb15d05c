// kvp_ea_qstats / kvp_ea_score -- ExpectedAttentionPress (kvpress/presses/expected_attention_press.py).
//
//   kvp_ea_qstats  (:74-80)   mu = mean_s q ; cov = (q-mu)^T (q-mu) / Sq      per (b, q-head)
//   [host]         (:110-123) averaged RoPE applied to mu / cov (D x D math, stays in the host)
//   kvp_ea_score   (:137-163) logits = k.mu/sqrt(D) + k^T cov k/(2D) per q-head  -> softmax over
//                             the S-n_sink keys -> group mean -> (s+eps)*||v|| -> sinks = max+1
//
// Reference dataflow at S=128k: repeat_kv (1 GiB), a [B,32,128,S] einsum intermediate (>= 1 GiB),
// softmax and norm temporaries.  Here: one pass over K writes the [B,Hq,S'] log2-logits (16 MiB)
// together with per-block softmax partials; ||V|| comes from the rownorm kernel; the finalize
// pass fuses softmax normalisation, group mean, the value-norm rescale and the device-side max.
//
// This file holds the host entry points and the generic VALU kernels (any D, dtype, stride).
// The MFMA kernels for bf16/f16 D=128 live in ea_mfma.hip.
#include "kvp_common.h"
#include "softmax_stats.h"
#include "ea_internal.h"

int kvp_rownorm_launch_read_once(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh, int64_t ss,
                                 float scale, float* out, hipStream_t stream);

namespace {

constexpr int EA_THREADS = 256;
constexpr int EA_SUB = 64;

// ------------------------------------------------------------------------------------------------
// query statistics (generic)
// ------------------------------------------------------------------------------------------------
// partial column sums: psum[(b*Hq+h)][chunk][d] = sum over the chunk's rows of q[b,h,s,d]
template <int DT>
__global__ __launch_bounds__(EA_THREADS) void ea_colsum_partial(const typename Elem<DT>::T* __restrict__ q, int64_t sb,
                                                                int64_t sh, int64_t ss, uint32_t Sq, uint32_t D,
                                                                uint32_t rows_per_chunk, float* __restrict__ psum) {
    const uint32_t chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const uint32_t Hq = gridDim.y, nchunk = gridDim.x;
    const typename Elem<DT>::T* base = q + (int64_t)b * sb + (int64_t)h * sh;
    const uint32_t s0 = chunk * rows_per_chunk, s1 = min(s0 + rows_per_chunk, Sq);
    for (uint32_t d = threadIdx.x; d < D; d += EA_THREADS) {
        float acc = 0.f;
        for (uint32_t s = s0; s < s1; ++s) acc += Elem<DT>::ld(base + (int64_t)s * ss + d);
        psum[((size_t)(b * Hq + h) * nchunk + chunk) * D + d] = acc;
    }
}

// mu[bh][d] = sum_chunks psum / Sq
__global__ void ea_mean_finalize(const float* __restrict__ psum, uint32_t nchunk, uint32_t D, float inv_n, uint32_t total,
                                 float* __restrict__ mu) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t bh = i / D, d = i - bh * D;
    float acc = 0.f;
    for (uint32_t c = 0; c < nchunk; ++c) acc += psum[((size_t)bh * nchunk + c) * D + d];
    mu[i] = acc * inv_n;
}

// partial centred outer products on 64x64 tiles of the D x D matrix:
// pcov[bh][chunk][i][j] = sum over the chunk's rows of (q_i - mu_i)(q_j - mu_j)
template <int DT>
__global__ __launch_bounds__(EA_THREADS) void ea_cov_partial(const typename Elem<DT>::T* __restrict__ q, int64_t sb,
                                                             int64_t sh, int64_t ss, uint32_t Sq, uint32_t D,
                                                             uint32_t rows_per_chunk, uint32_t nchunk, uint32_t ntile,
                                                             const float* __restrict__ mu, float* __restrict__ pcov) {
    __shared__ float xi[32][64];
    __shared__ float xj[32][64];
    const uint32_t tile = blockIdx.x % (ntile * ntile), chunk = blockIdx.x / (ntile * ntile);
    const uint32_t it = tile / ntile, jt = tile - it * ntile;
    const uint32_t h = blockIdx.y, b = blockIdx.z, Hq = gridDim.y;
    const uint32_t bh = b * Hq + h;
    const typename Elem<DT>::T* base = q + (int64_t)b * sb + (int64_t)h * sh;
    const float* mub = mu + (size_t)bh * D;
    const uint32_t ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    const uint32_t s0 = chunk * rows_per_chunk, s1 = min(s0 + rows_per_chunk, Sq);
    for (uint32_t sbeg = s0; sbeg < s1; sbeg += 32) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < 32 * 64; e += EA_THREADS) {
            const uint32_t r = e >> 6, cc = e & 63;
            const uint32_t s = sbeg + r;
            const uint32_t di = it * 64 + cc, dj = jt * 64 + cc;
            xi[r][cc] = (s < s1 && di < D) ? Elem<DT>::ld(base + (int64_t)s * ss + di) - mub[di] : 0.f;
            xj[r][cc] = (s < s1 && dj < D) ? Elem<DT>::ld(base + (int64_t)s * ss + dj) - mub[dj] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            float vi[4], vj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { vi[a] = xi[r][ti + 16 * a]; vj[a] = xj[r][tj + 16 * a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = fmaf(vi[a], vj[c], acc[a][c]);
        }
    }
    float* out = pcov + ((size_t)bh * nchunk + chunk) * D * D;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = it * 64 + ti + 16 * a, j = jt * 64 + tj + 16 * c;
            if (i < D && j < D) out[(size_t)i * D + j] = acc[a][c];
        }
}

// cov[bh][i][j] = sum_chunks pcov / Sq
__global__ void ea_cov_finalize(const float* __restrict__ pcov, uint32_t nchunk, uint32_t DD, float inv_n, uint64_t total,
                                float* __restrict__ cov) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t bh = i / DD, e = i - bh * DD;
    float acc = 0.f;
    for (uint32_t c = 0; c < nchunk; ++c) acc += pcov[(bh * nchunk + c) * DD + e];
    cov[i] = acc * inv_n;
}

// ------------------------------------------------------------------------------------------------
// score (generic logits)
// ------------------------------------------------------------------------------------------------
// log2-logits of 64 keys for one q-head + the block's softmax partial.
template <int DT>
__global__ __launch_bounds__(EA_THREADS) void ea_logits_generic(EaArgs a, float* __restrict__ logits, uint32_t nblk,
                                                                float* __restrict__ part_m, float* __restrict__ part_z) {
    using T = typename Elem<DT>::T;
    extern __shared__ __attribute__((aligned(16))) float ea_lds[];
    float* kt = ea_lds;                       // [64][D+1]
    float* red = kt + EA_SUB * (a.D + 1);     // [2][4][64]
    const uint32_t blk = blockIdx.x, hq = blockIdx.y, b = blockIdx.z;
    if (a.clear_word && blk == 0 && hq == 0 && b == 0 && threadIdx.x == 0) *a.clear_word = 0;
    const uint32_t h = hq / a.G;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const T* kb = static_cast<const T*>(a.k) + (int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)a.n_sink * a.k_ss;
    const uint32_t key0 = blk * EA_SUB;
    for (uint32_t e = threadIdx.x; e < EA_SUB * a.D; e += EA_THREADS) {
        const uint32_t r = e / a.D, d = e - r * a.D;
        const uint32_t kk = key0 + r;
        kt[r * (a.D + 1) + d] = kk < a.Sp ? Elem<DT>::ld(kb + (int64_t)kk * a.k_ss + d) : 0.f;
    }
    __syncthreads();
    const float* krow = kt + lane * (a.D + 1);
    const float* mu = a.mu + (size_t)(b * a.Hq + hq) * a.D;
    const float* cov = a.cov ? a.cov + (size_t)(b * a.Hq + hq) * a.D * a.D : nullptr;
    float lin = 0.f, quad = 0.f;
    for (uint32_t i = wv; i < a.D; i += EA_THREADS / 64) {
        const float ki = krow[i];
        lin = fmaf(mu[i], ki, lin);
        if (cov) {
            const float* ci = cov + (size_t)i * a.D;
            float inner = 0.f;
            for (uint32_t j = 0; j < a.D; ++j) inner = fmaf(ci[j], krow[j], inner);
            quad = fmaf(ki, inner, quad);
        }
    }
    red[wv * 64 + lane] = lin;
    red[256 + wv * 64 + lane] = quad;
    __syncthreads();
    if (wv == 0) {
        const float l = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        const float qd = red[256 + lane] + red[320 + lane] + red[384 + lane] + red[448 + lane];
        const uint32_t kk = key0 + lane;
        const bool valid = kk < a.Sp;
        // scores = k.mu / sqrt(d) + k^T cov k / d / 2   (:149-151), in log2 units
        const float l2 = valid ? (l * a.inv_sqrt_d + qd * a.inv_2d) * KVP_LOG2E : KVP_NEG_INF;
        if (valid) logits[(size_t)(b * a.Hq + hq) * a.Sp + kk] = l2;
        const float m = wave_max(l2);
        const float z = wave_sum(valid ? exp2f(l2 - m) : 0.f);
        if (lane == 0) {
            part_m[(size_t)(b * a.Hq + hq) * nblk + blk] = m;
            part_z[(size_t)(b * a.Hq + hq) * nblk + blk] = z;
        }
    }
}

// scores[b,h,n_sink+s] = (mean_g 2^(l2 - a_g) [+eps]) [* ||v||]  ; device-side global max
__global__ __launch_bounds__(EA_THREADS) void ea_finalize_kernel(const float* __restrict__ logits, const float* __restrict__ rowstat,
                                                                 const float* __restrict__ vnorm, uint32_t B, uint32_t Hq,
                                                                 uint32_t Hkv, uint32_t S, uint32_t n_sink, int use_vnorm,
                                                                 float epsilon, float* __restrict__ scores,
                                                                 float* __restrict__ bmax) {
    __shared__ float scr[4];
    const uint32_t Sp = S - n_sink, G = Hq / Hkv;
    const uint64_t total = (uint64_t)B * Hkv * Sp;
    const float invG = 1.0f / (float)G;
    float vmax = KVP_NEG_INF;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bh = (uint32_t)(i / Sp);
        const uint32_t s = (uint32_t)(i - (uint64_t)bh * Sp);
        const uint32_t b = bh / Hkv, h = bh - b * Hkv;
        float p = 0.f;
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t row = b * Hq + h * G + g;
            p += exp2f(logits[(size_t)row * Sp + s] - rowstat[row]);
        }
        p *= invG;
        if (use_vnorm) p = (p + epsilon) * vnorm[i];
        scores[(size_t)bh * S + n_sink + s] = p;
        vmax = fmaxf(vmax, p);
    }
    block_store_max(vmax, scr, bmax, blockIdx.x);
}

// The value norms, the row normalisers and the finalize above in ONE pass over V (bf16 / f16, 256-byte rows, G <= 16): what
// rownorm_slot_kernel + softmax_combine_kernel + ea_finalize_kernel do in three launches (47 + 6 + 12 us at 8 x 131072), term for term
// -- a_g = M + log2 Z with the combine kernel's merge order (one wave per query head of the group), ||v|| with the row-norm kernel's
// lanes, fma chain and shuffle order, the group mean in ea_finalize_kernel's order: the same bits -- while V streams through once (slot
// walk: a workgroup takes one contiguous range of keys; streaming loads when V is read once and cannot stay cached, rownorm.hip).
// The 16 lanes of a row group share the score arithmetic of the group's four keys in flight: lane (u, g) = (lir / G, lir % G) loads the
// logit of key u and query head g TOGETHER with the V rows (the first version left the whole finalize to lane 0 after the reduction: its
// G dependent loads per key sat exposed behind every step, 85 us instead of the three kernels' 65), exponentiates it, and a G - 1 step
// DPP scan (row_shr:1) adds the G terms in ea_finalize_kernel's order; the lane of the last head writes the score.  G = 1, 2, 4; round 6: 8 and 16 (in rounds).
constexpr int EVF_THREADS = 1024;
template <int DT, bool NT, int G>
__global__ __launch_bounds__(EVF_THREADS) void ea_vnorm_finalize_kernel(const typename Elem<DT>::T* __restrict__ v, int64_t v_sb, int64_t v_sh, int64_t v_ss,
                                                                        const float* __restrict__ logits, const float* __restrict__ part_m,
                                                                        const float* __restrict__ part_z, uint32_t nblk, uint32_t Hq, uint32_t Hkv,
                                                                        uint32_t S, uint32_t n_sink, float epsilon, float* __restrict__ scores,
                                                                        float* __restrict__ bmax, uint32_t rows_per_wg, uint32_t* __restrict__ arrivals) {
    using T = typename Elem<DT>::T;
    __shared__ float ag[16];
    __shared__ float scr[EVF_THREADS / 64];
    const uint32_t Sp = S - n_sink;
    const uint32_t bh = blockIdx.y, b = bh / Hkv, h = bh - b * Hkv;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t row0 = b * Hq + h * G;   // first logits row of this kv-head's group
    if (wv < G) {   // softmax_combine_kernel's merge for row row0 + wv
        const uint32_t row = row0 + wv;
        float m = KVP_NEG_INF, z = 0.f;
        for (uint32_t j = lane; j < nblk; j += 64) softmax_merge(m, z, part_m[(size_t)row * nblk + j], part_z[(size_t)row * nblk + j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o), z2 = __shfl_xor(z, o);
            softmax_merge(m, z, m2, z2);
        }
        if (lane == 0) ag[wv] = m + log2f(z);
    }
    __syncthreads();
    const T* __restrict__ base = v + (int64_t)b * v_sb + (int64_t)h * v_sh;
    float* __restrict__ out = scores + (size_t)bh * S + n_sink;
    const float invG = 1.0f / (float)G;
    const uint32_t lir = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // G <= 4: lanes 0 .. 4 G - 1 of the group score the four keys in flight at once (key lir / G, query head lir % G).  G = 8 / 16 (round 6: the
    // 70B- / 405B-class groups): the 16 lanes cover 2 / 1 keys x G heads per ROUND, NR = 2 / 4 rounds per step -- the same scan, the same order.
    constexpr int KPR = G <= 4 ? 4 : 16 / G;   // keys scored per round
    constexpr int NR = 4 / KPR;                // rounds per step of four keys
    const uint32_t my_u = lir / G, my_g = lir % G;
    const bool scorer = lir < KPR * G;
    const float my_a = scorer ? ag[my_g] : 0.f;
    const float* __restrict__ lrow = logits + (size_t)(row0 + my_g) * Sp;
    const uint32_t r0 = blockIdx.x * rows_per_wg, r1 = min(Sp, r0 + rows_per_wg);
    float vmax = KVP_NEG_INF;
    for (uint32_t it = r0; it < r1; it += 64 * 4) {
        uint4 vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = it + u * 64 + grp;
            vv[u] = make_uint4(0, 0, 0, 0);
            if (s < r1) vv[u] = ld16<NT>(base + (int64_t)s * v_ss + (size_t)lir * 8);
        }
        float lg[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t s_r = it + (r * KPR + my_u) * 64 + grp;
            lg[r] = (scorer && s_r < r1) ? lrow[s_r] : 0.f;
        }
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[8];
            unpack16<DT>(vv[u], f);
            acc[u] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[u] = fmaf(f[i], f[i], acc[u]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o);   // every lane of the group ends with the row's sum
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float e = exp2f(lg[r] - my_a);
            float p = e;   // ea_finalize_kernel: p = 0; p += e_0; p += e_1; ...  (0 + e_0 = e_0)
#pragma unroll
            for (int k = 1; k < G; ++k) {
                const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
                if (my_g == (uint32_t)k) p = prev + e;
            }
            const uint32_t u_r = r * KPR + my_u, s_r = it + u_r * 64 + grp;
            if (scorer && my_g == G - 1 && s_r < r1) {
                const float ss = u_r == 0 ? acc[0] : u_r == 1 ? acc[1] : u_r == 2 ? acc[2] : acc[3];
                p *= invG;
                p = (p + epsilon) * (1.0f * sqrtf(ss));
                out[s_r] = p;
                vmax = fmaxf(vmax, p);
            }
        }
    }
    // Global maximum and the sink pad (max + 1, expected_attention_press.py:163) without a second launch: every workgroup publishes its
    // maximum and takes a ticket; the LAST one to arrive folds them all and writes the pads.  No workgroup waits for another (nothing can
    // hang); the words other workgroups read are written and read with agent-scope atomics (the XCDs' L2s are not coherent for plain
    // accesses within a launch), the store is drained (vmcnt) before the relaxed ticket.  `arrivals` was zeroed by the logits kernel earlier in the stream.
    __shared__ uint32_t ticket;
    vmax = wave_max(vmax);
    if (lane == 0) scr[wv] = vmax;
    __syncthreads();
    const uint32_t nwg = gridDim.x * gridDim.y, slot = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
        float m = scr[0];
        for (uint32_t i = 1; i < EVF_THREADS / 64; ++i) m = fmaxf(m, scr[i]);
        __hip_atomic_store(&bmax[slot], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the maximum has reached the agent-coherent level before the ticket is taken (an
                                              // acquire / release ticket made every workgroup write back its XCD's L2: + 10 us)
        ticket = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (ticket != nwg - 1 || n_sink == 0) return;
    float m = KVP_NEG_INF;
    for (uint32_t i = threadIdx.x; i < nwg; i += EVF_THREADS) m = fmaxf(m, __hip_atomic_load(&bmax[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    m = wave_max(m);
    __syncthreads();
    if (lane == 0) scr[wv] = m;
    __syncthreads();
    m = scr[0];
    for (uint32_t i = 1; i < EVF_THREADS / 64; ++i) m = fmaxf(m, scr[i]);
    const float fill = m + 1.0f;
    const uint32_t BH = gridDim.y;
    for (uint32_t i = threadIdx.x; i < BH * n_sink; i += EVF_THREADS) scores[(size_t)(i / n_sink) * S + (i % n_sink)] = fill;
}

struct EaScoreWs {
    float* bmax;  // per-workgroup maxima of the finalize kernel (<= 2048)
    float* logits;
    float* part_m;
    float* part_z;
    float* rowstat;
    float* vnorm;
    void* scratch;
    size_t total_bytes;
};

EaScoreWs carve_score_ws(void* ws, int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t D) {
    EaScoreWs w;
    size_t off = 0;
    char* base = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += kvp_align_up(bytes, 256);
        return p;
    };
    const size_t nblk = (size_t)(S + EA_SUB - 1) / EA_SUB;
    w.bmax = (float*)take(2048 * 4);
    w.logits = (float*)take((size_t)B * Hq * S * 4);
    w.part_m = (float*)take((size_t)B * Hq * nblk * 4);
    w.part_z = (float*)take((size_t)B * Hq * nblk * 4);
    w.rowstat = (float*)take((size_t)B * Hq * 4);
    w.vnorm = (float*)take((size_t)B * Hkv * S * 4);
    w.scratch = take(ea_mfma_logits_scratch_bytes(B, Hq, D) + 256);
    w.total_bytes = off;
    return w;
}

struct EaStatWs {
    float* psum;
    float* pcov;
    uint32_t rows_mean, nchunk_mean, rows_cov, nchunk_cov;
    size_t total_bytes;
};

EaStatWs carve_stat_ws(void* ws, int64_t B, int64_t Hq, int64_t Sq, int64_t D) {
    EaStatWs w;
    size_t off = 0;
    char* base = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += kvp_align_up(bytes, 256);
        return p;
    };
    auto pick = [&](int64_t max_chunks, uint32_t& rows, uint32_t& nchunk) {
        int64_t r = std::max<int64_t>(256, (Sq + max_chunks - 1) / max_chunks);
        r = (r + 31) / 32 * 32;
        rows = (uint32_t)r;
        nchunk = (uint32_t)std::max<int64_t>(1, (Sq + r - 1) / r);
    };
    pick(64, w.rows_mean, w.nchunk_mean);
    pick(16, w.rows_cov, w.nchunk_cov);
    w.psum = (float*)take((size_t)B * Hq * w.nchunk_mean * D * 4);
    w.pcov = (float*)take((size_t)B * Hq * w.nchunk_cov * D * D * 4);
    w.total_bytes = off;
    return w;
}

}  // namespace

// ================================================================================================
extern "C" size_t kvp_ea_qstats_workspace_bytes(int64_t B, int64_t Hq, int64_t Sq, int64_t D) {
    if (B < 1 || Hq < 1 || Sq < 1 || D < 1) return 256;
    return std::max(carve_stat_ws(nullptr, B, Hq, Sq, D).total_bytes, ea_mfma_qstats_ws_bytes(B, Hq, Sq, D));
}

extern "C" int kvp_ea_qstats(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype, int64_t B, int64_t Hq,
                             int64_t Sq, int64_t D, float* mu, float* cov, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "ea_qstats: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Sq >= 1 && D >= 1 && D <= 1024, "ea_qstats: bad shape B=%ld Hq=%ld Sq=%ld D=%ld",
                  (long)B, (long)Hq, (long)Sq, (long)D);
    KVP_CHECK_ARG(B <= 65535 && Hq <= 65535 && Sq < ((int64_t)1 << 31), "ea_qstats: shape too large");
    KVP_CHECK_ARG(q && mu, "ea_qstats: null pointer");
    const size_t need = kvp_ea_qstats_workspace_bytes(B, Hq, Sq, D);
    if (!ws || ws_bytes < need) {
        kvp_set_error("ea_qstats: workspace too small (%zu < %zu)", ws_bytes, need);
        return KVP_EWORKSPACE;
    }
    if (ea_mfma_qstats_eligible(q, q_sb, q_sh, q_ss, dtype, Hq, Sq, D))
        return ea_mfma_qstats(q, q_sb, q_sh, q_ss, dtype, B, Hq, Sq, D, mu, cov, ws, stream);

    EaStatWs w = carve_stat_ws(ws, B, Hq, Sq, D);
    const float inv_n = (float)(1.0 / (double)Sq);
    const uint32_t ntile = (uint32_t)((D + 63) / 64);
    const uint32_t tot_mu = (uint32_t)(B * Hq * D);
    const uint64_t tot_cov = (uint64_t)B * Hq * D * D;
#define KVP_EA_STATS(DT)                                                                                              \
    {                                                                                                                 \
        const Elem<DT>::T* qp = static_cast<const Elem<DT>::T*>(q);                                                   \
        KVP_LAUNCH("ea_colsum_partial", stream, ea_colsum_partial<DT><<<dim3(w.nchunk_mean, (uint32_t)Hq, (uint32_t)B), EA_THREADS, 0, stream>>>(             \
            qp, q_sb, q_sh, q_ss, (uint32_t)Sq, (uint32_t)D, w.rows_mean, w.psum));                                    \
        KVP_LAUNCH("ea_mean_finalize", stream, ea_mean_finalize<<<(tot_mu + 255) / 256, 256, 0, stream>>>(w.psum, w.nchunk_mean, (uint32_t)D, inv_n, tot_mu, mu)); \
        if (cov) {                                                                                                    \
            KVP_LAUNCH("ea_cov_partial", stream, ea_cov_partial<DT><<<dim3(ntile * ntile * w.nchunk_cov, (uint32_t)Hq, (uint32_t)B), EA_THREADS, 0, stream>>>( \
                qp, q_sb, q_sh, q_ss, (uint32_t)Sq, (uint32_t)D, w.rows_cov, w.nchunk_cov, ntile, mu, w.pcov));        \
            KVP_LAUNCH("ea_cov_finalize", stream, ea_cov_finalize<<<(uint32_t)((tot_cov + 255) / 256), 256, 0, stream>>>(w.pcov, w.nchunk_cov, (uint32_t)(D * D), inv_n, tot_cov, cov)); \
        }                                                                                                             \
    }
    if (dtype == KVP_F32) KVP_EA_STATS(KVP_F32)
    else if (dtype == KVP_F16) KVP_EA_STATS(KVP_F16)
    else KVP_EA_STATS(KVP_BF16)
#undef KVP_EA_STATS
    KVP_CHECK_LAUNCH("ea_qstats");
    return KVP_OK;
}

extern "C" size_t kvp_ea_score_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t D) {
    if (B < 1 || Hq < 1 || Hkv < 1 || S < 1 || D < 1) return 256;
    return carve_score_ws(nullptr, B, Hq, Hkv, S, D).total_bytes;
}

extern "C" int kvp_ea_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh,
                            int64_t v_ss, int dtype, const float* mu, const float* cov, int64_t B, int64_t Hq, int64_t Hkv,
                            int64_t S, int64_t D, int64_t n_sink, int use_vnorm, float epsilon, float* scores, void* ws,
                            size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(dtype == KVP_F32 || dtype == KVP_F16 || dtype == KVP_BF16, "ea_score: bad dtype %d", dtype);
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && D >= 1 && D <= 1024, "ea_score: bad shape");
    KVP_CHECK_ARG(n_sink >= 0 && S > n_sink, "ea_score: Input should contain more tokens than n_sink=%ld", (long)n_sink);
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && B <= 65535 && Hq <= 65535, "ea_score: shape too large");
    KVP_CHECK_ARG(k && mu && scores && (v || !use_vnorm), "ea_score: null pointer");
    EaScoreWs w = carve_score_ws(ws, B, Hq, Hkv, S, D);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("ea_score: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    const int64_t Sp = S - n_sink;
    EaArgs a;
    a.k = k; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
    a.mu = mu; a.cov = cov;
    a.B = (uint32_t)B; a.Hq = (uint32_t)Hq; a.Hkv = (uint32_t)Hkv; a.G = (uint32_t)(Hq / Hkv);
    a.S = (uint32_t)S; a.Sp = (uint32_t)Sp; a.D = (uint32_t)D; a.n_sink = (uint32_t)n_sink;
    a.inv_sqrt_d = (float)(1.0 / sqrt((double)D));
    a.inv_2d = (float)(1.0 / (2.0 * (double)D));
    a.clear_word = reinterpret_cast<uint32_t*>(w.bmax + 2047);   // the last word of the maxima buffer (<= 1280 of its 2048 are maxima): arrival counter

    uint32_t nblk;
    if (ea_mfma_logits_eligible(a, dtype)) {
        nblk = ea_mfma_logits_nblk(a);
        if (int rc = ea_mfma_logits(a, dtype, w.logits, nblk, w.part_m, w.part_z, w.scratch, stream)) return rc;
    } else {
        nblk = (uint32_t)((Sp + EA_SUB - 1) / EA_SUB);
        const size_t lds = ((size_t)EA_SUB * (D + 1) + 512) * 4;
        KVP_CHECK_ARG(lds <= 160 * 1024, "ea_score: D=%ld exceeds the generic kernel's LDS budget", (long)D);
        if (lds > 64 * 1024) {   // head sizes of 256 and more (Gemma: 66 KiB for the 64-key tile of float32 rows): above the default limit of a launch, inside the CU's 160 KiB -- was refused until round 6
            const void* fn = dtype == KVP_F32 ? (const void*)ea_logits_generic<KVP_F32> : dtype == KVP_F16 ? (const void*)ea_logits_generic<KVP_F16> : (const void*)ea_logits_generic<KVP_BF16>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                kvp_set_error("ea_score: cannot raise the dynamic LDS limit to %zu bytes (D=%ld)", lds, (long)D);
                return KVP_EHIP;
            }
        }
        const dim3 grid(nblk, (uint32_t)Hq, (uint32_t)B);
        if (dtype == KVP_F32) KVP_LAUNCH("ea_logits_generic", stream, ea_logits_generic<KVP_F32><<<grid, EA_THREADS, lds, stream>>>(a, w.logits, nblk, w.part_m, w.part_z));
        else if (dtype == KVP_F16) KVP_LAUNCH("ea_logits_generic", stream, ea_logits_generic<KVP_F16><<<grid, EA_THREADS, lds, stream>>>(a, w.logits, nblk, w.part_m, w.part_z));
        else KVP_LAUNCH("ea_logits_generic", stream, ea_logits_generic<KVP_BF16><<<grid, EA_THREADS, lds, stream>>>(a, w.logits, nblk, w.part_m, w.part_z));
    }
    const uint32_t nrows = (uint32_t)(B * Hq);
    const int64_t es = kvp_elem_size(dtype);
    auto al16 = [&](int64_t elems) { return (elems * es) % 16 == 0; };
    if (use_vnorm && dtype != KVP_F32 && D * es == 256 && (Hq / Hkv == 1 || Hq / Hkv == 2 || Hq / Hkv == 4 || Hq / Hkv == 8 || Hq / Hkv == 16) && Sp >= 4096 && B * Hkv <= 1024 && ((uintptr_t)v % 16) == 0 && al16(v_sb) &&
        al16(v_sh) && al16(v_ss) && kvp_env_int("KVP_EA_FUSED_FINALIZE", 1) != 0) {
        const uint32_t BH = (uint32_t)(B * Hkv);
        const uint64_t want = std::max<uint64_t>(1, (256 + BH - 1) / BH);   // about one 1024-thread workgroup per CU
        uint64_t rows = ((uint64_t)Sp + want - 1) / want;
        rows = (rows + 255) / 256 * 256;
        const uint32_t nslot = (uint32_t)(((uint64_t)Sp + rows - 1) / rows);
        const dim3 grid(nslot, BH);
        const bool nt = (uint64_t)BH * Sp * 256 > (192ull << 20);   // read-once V (rownorm.hip: rn_streaming)
        const char* vp = static_cast<const char*>(v) + n_sink * v_ss * es;
#define KVP_EVF(DTV, TT, NTV, GV) KVP_LAUNCH("ea_vnorm_finalize_kernel", stream, (ea_vnorm_finalize_kernel<DTV, NTV, GV><<<grid, EVF_THREADS, 0, stream>>>(reinterpret_cast<const TT*>(vp), v_sb, v_sh, v_ss, w.logits, w.part_m, w.part_z, nblk, (uint32_t)Hq, (uint32_t)Hkv, (uint32_t)S, (uint32_t)n_sink, epsilon, scores, w.bmax, (uint32_t)rows, a.clear_word)))
#define KVP_EVF_G(DTV, TT, NTV) do { switch (Hq / Hkv) { case 1: KVP_EVF(DTV, TT, NTV, 1); break; case 2: KVP_EVF(DTV, TT, NTV, 2); break; case 4: KVP_EVF(DTV, TT, NTV, 4); break; case 8: KVP_EVF(DTV, TT, NTV, 8); break; default: KVP_EVF(DTV, TT, NTV, 16); break; } } while (0)
        if (dtype == KVP_BF16) { if (nt) KVP_EVF_G(KVP_BF16, uint16_t, true); else KVP_EVF_G(KVP_BF16, uint16_t, false); }
        else { if (nt) KVP_EVF_G(KVP_F16, _Float16, true); else KVP_EVF_G(KVP_F16, _Float16, false); }
#undef KVP_EVF_G
#undef KVP_EVF
        KVP_CHECK_LAUNCH("ea_score(vnorm + finalize)");
        return KVP_OK;
    }
    KVP_LAUNCH("softmax_combine_kernel", stream, softmax_combine_kernel<<<(nrows + 3) / 4, 256, 0, stream>>>(w.part_m, w.part_z, nrows, nblk, w.rowstat));
    KVP_CHECK_LAUNCH("ea_score(logits)");
    if (use_vnorm) {
        const char* vp = static_cast<const char*>(v) + n_sink * v_ss * kvp_elem_size(dtype);
        if (int rc = kvp_rownorm_launch_read_once(vp, dtype, B, Hkv, Sp, D, v_sb, v_sh, v_ss, 1.0f, w.vnorm, stream)) return rc;
    }
    const uint64_t total = (uint64_t)B * Hkv * Sp;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + EA_THREADS - 1) / EA_THREADS, 2048));
    KVP_LAUNCH("ea_finalize_kernel", stream, ea_finalize_kernel<<<blocks, EA_THREADS, 0, stream>>>(w.logits, w.rowstat, w.vnorm, (uint32_t)B, (uint32_t)Hq, (uint32_t)Hkv,
                                                          (uint32_t)S, (uint32_t)n_sink, use_vnorm, epsilon, scores, w.bmax));
    if (n_sink > 0) {
        const uint32_t BH = (uint32_t)(B * Hkv), nfill = BH * (uint32_t)n_sink;
        KVP_LAUNCH("fill_pad_kernel", stream, fill_pad_kernel<<<(nfill + 255) / 256, 256, 0, stream>>>(scores, BH, (uint32_t)S, 0, (uint32_t)n_sink, w.bmax, blocks));
    }
    KVP_CHECK_LAUNCH("ea_score(finalize)");
    return KVP_OK;
}
