// ExpectedAttention on the matrix cores (bf16 / f16; D = 128, round 6: also 64 and -- the logits -- 96): query statistics and quadratic-form logits.
//
// (1) ea_qstats_mfma  -- mu, cov of the pre-RoPE queries (expected_attention_press.py:74-80), one pass over Q.
//     cov = X^T X needs, for both MFMA operands, "column fragments" (a lane holds several ROWS s of one dimension d), while X is
//     stored row-major: the transpose happens in the LDS read (ds_read_b64_tr_b16, see (1b) below).  Each workgroup writes a
//     partial (raw second moments, column sums, row count); ea_qstats_combine merges them with the pairwise (Chan) update.
//     Accuracy of the raw moments: see (1b); end-to-end scores stay within 1e-3 (tests/test_gpu_parity.py::
//     test_ea_full_chain_mfma_vs_oracle).  The path is taken for Sq >= 4096; shorter sequences use the exact fp32 generic kernels.
//
// (2) ea_logits_mfma  -- log2-logits k.mu/sqrt(D) + k^T cov k/(2D) per q-head (:148-151) and per-chunk
//     softmax partials.  C = cov_strip . K_tile^T with cov split into hi + lo 16-bit parts (two MFMA chains,
//     ~2^-17 relative), then a row-dot with K read back in the C layout (ds_read_b64 from the swizzled tile).
//     (k^T A k depends only on the symmetric part of A, so feeding rows of cov as the "columns" is exact.)
#include "ea_internal.h"
#include "softmax_stats.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EM_THREADS = 256;
constexpr int EM_TILE = 64;    // rows per LDS tile
constexpr int EM_ROWB = 256;   // D = 128, 2-byte elements
constexpr int EM_TILEB = EM_TILE * EM_ROWB;

template <int DT> __device__ __forceinline__ f32x16 mma32(const uint4& a, const uint4& b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mma32<KVP_BF16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma32<KVP_F16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two floats -> one dword of 16-bit values (round to nearest even), low half = a
template <int DT> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<KVP_BF16>(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <> __device__ __forceinline__ uint32_t pack2<KVP_F16>(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 p = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, p);
}
template <int DT> __device__ __forceinline__ float lo16(uint32_t w);
template <int DT> __device__ __forceinline__ float hi16(uint32_t w);
template <> __device__ __forceinline__ float lo16<KVP_BF16>(uint32_t w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float hi16<KVP_BF16>(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
template <> __device__ __forceinline__ float lo16<KVP_F16>(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); }
template <> __device__ __forceinline__ float hi16<KVP_F16>(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
template <int DT> __device__ __forceinline__ uint32_t one16();
template <> __device__ __forceinline__ uint32_t one16<KVP_BF16>() { return 0x3F80u; }
template <> __device__ __forceinline__ uint32_t one16<KVP_F16>() { return 0x3C00u; }

// 16-byte fragment: row sub*32 + n, 16-byte column c16
__device__ __forceinline__ uint4 frag16(const unsigned char* buf, uint32_t sub, uint32_t c16, uint32_t n) {
    const uint32_t row = sub * 32 + n;
    return *reinterpret_cast<const uint4*>(buf + row * EM_ROWB + ((c16 ^ (row & 15)) << 4));
}

// =================================================================================================
// (1) query statistics
// =================================================================================================
struct QstatArgs {
    const void* q;
    int64_t q_sb, q_sh, q_ss;  // element strides
    uint32_t B, Hq, Sq, nchunk, rows_per_chunk;
    float* s2;    // [B*Hq][nchunk][128][128]
    float* dsum;  // [B*Hq][nchunk][128]   sum over the chunk's rows of (x - m0)
    float* m0;    // [B*Hq][nchunk][128]
    uint32_t nt;  // non-temporal Q stream (read once)
    uint32_t nch; // 16-byte chunks a row of one head really has: 16 (D = 128); 12 / 8 (round 6: D = 96 / 64 as heads of 128 dimensions whose upper
                  // dimensions are zero: the LDS tiles are zeroed once and the lanes of the missing chunks never request anything)
    uint32_t quarters;  // D = 256 (round 6): a "head" of this kernel is one of the SIX pairs of 64-dimension quarters of a real head (EQ_QA / EQ_QB):
                        // every pair of dimensions meets in one of them, the combine scatters the 128 x 128 results into the 256 x 256 covariance
};
// the six pairs of quarters: pairs 0 and 1 also own the diagonal blocks (quarters 0, 1 and 2, 3) and the means
__device__ __constant__ const uint32_t EQ_QA[6] = {0, 2, 0, 1, 0, 1};
__device__ __constant__ const uint32_t EQ_QB[6] = {1, 3, 2, 3, 3, 2};

// ---- (1b) query statistics with transposed LDS reads (gfx950 ds_read_b64_tr_b16) -------------------------------------------------
// Per (head, chunk of rows) partials: raw second moments sum x x^T (m0 = 0) and column sums; the row -> column transpose a syrk needs
// happens in the LDS read (round 1 transposed BY an MFMA against a selection matrix: twice the matrix-core work, removed in round 5).  Within a 16-lane group, lane i supplies the address of an
// 8-byte piece and lane c receives element c % 4 of pieces c / 4 + {0, 4, 8, 12} (tools/probe_tr.hip): with lane i pointing
// at X[token t0 + i / 4][dim d0 + 4 (i % 4) ..], lane c gets X[t0 .. t0 + 3][d0 + c] -- four tokens of ONE dimension, which is
// what both operands of the syrk MFMA want (a lane = one dimension, 8 consecutive tokens of the 16-token k-step: two reads).
// Q rows (256 B of one head, 8 KiB apart) travel HBM -> LDS by LDS-DMA into a ring of 64-token tiles (ET_NBUF - 1 in flight per
// workgroup, ET_NBUF - 1 tiles in flight); the 16-byte slots of a row are XOR-swizzled by (token & 3) << 2 on the global side so that the four token rows a
// transposed read touches sit in different banks.  Per 16-token k-step and wave: 8 transposed reads (fragments of all four
// 32-dim strips), 4 syrk MFMAs (own strip x every strip) + 1 MFMA against a ones fragment for the column sums.
// Raw moments cancel when |mean| >> sigma: sum x x^T is accumulated in fp32 over the rows of one partial (4096 .. 8192 at 128k tokens,
// qstats_plan), so the relative error of a covariance entry is ~2^-24 (mean / sigma)^2 sqrt(rows / 16): 1e-4 .. 2e-4 at |mean| = 10 sigma
// (tests/test_gpu_fullsize.py::test_ea_qstats_128k_large_mean_adversarial, bound 1e-3).  The partials are merged by the same pairwise update.
template <int DT, int ET_NBUF, int ET_OCC>
__global__ __launch_bounds__(EM_THREADS, ET_OCC) void ea_qstats_tr_kernel(QstatArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[ET_NBUF * EM_TILEB];
    typedef short v4s __attribute__((ext_vector_type(4)));
    const uint32_t hq = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t g = lane >> 4, i16 = lane & 15, t2 = i16 >> 2;
    const uint32_t bh = b * a.Hq + hq;
    const uint32_t hreal = a.quarters ? hq / 6 : hq, qpair = a.quarters ? hq % 6 : 0;
    const char* base = static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hreal * a.q_sh) * 2;
    const int64_t row_bytes = a.q_ss * 2;
    const uint32_t rbeg = chunk * a.rows_per_chunk;
    const uint32_t rend = min(rbeg + a.rows_per_chunk, a.Sq);
    if (rbeg >= rend) return;
    const uint32_t ntiles = (rend - rbeg + EM_TILE - 1) / EM_TILE;

    // ---- LDS-DMA: request j (0..3) of a tile moves rows 16 j + 4 wv + g; lane slot i16 fetches chunk i16 ^ (g << 2) (row & 3 == g)
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t gchunk = i16 ^ (g << 2);
    // byte offset of this lane's 16-byte chunk inside the head's row (quarters: chunks 0 .. 7 lie in quarter QA, 8 .. 15 in quarter QB of a 512-byte row)
    const uint32_t gch = a.quarters ? (gchunk < 8 ? EQ_QA[qpair] * 128 + gchunk * 16 : EQ_QB[qpair] * 128 + (gchunk - 8) * 16) : gchunk << 4;
    const bool has_chunk = gchunk < a.nch;
    if (a.nch < 16) {   // zero dimensions of a narrow head: written once, never overwritten (only the lanes with a chunk request)
        for (uint32_t e = threadIdx.x; e < (uint32_t)(ET_NBUF * EM_TILEB / 16); e += EM_THREADS) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    auto request_tile = [&](uint32_t t, uint32_t buf) {
        const uint32_t row0 = rbeg + min(t, ntiles - 1) * EM_TILE;   // past the end: re-fetch the last tile (never read)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t row = min(row0 + 16 * j + 4 * wv + g, a.Sq - 1);   // rows past the end are zeroed in LDS below
            const char* gp = base + (int64_t)row * row_bytes + gch;
            const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf * EM_TILEB + (16 * j + 4 * wv) * EM_ROWB);
            if (a.nch == 16) {
                if (a.nt) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(la), "v"(gp) : "memory");
                else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
            } else {   // (every wave has lanes with a chunk: all waves issue all requests, the vmcnt bookkeeping below holds)
                asm volatile("s_mov_b32 m0, %0" ::"s"(la) : "memory");
                if (has_chunk) {
                    if (a.nt) asm volatile("global_load_lds_dwordx4 %0, off nt" ::"v"(gp) : "memory");
                    else asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp) : "memory");
                }
            }
        }
    };
#pragma unroll
    for (int p = 0; p < ET_NBUF - 1; ++p) request_tile(p, p);

    // ---- transposed fragment reads: strip s, k-step ks, half hf -> lds + buf + off0 ^ (s << 6) + ks * 4096 + hf * 1024
    const uint32_t off0 = (8 * (g >> 1) + t2) * EM_ROWB + (((((g & 1) * 2 + ((i16 & 3) >> 1)) ^ (t2 << 2))) << 4) + (i16 & 1) * 8;
    uint32_t ones[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ones[e] = one16<DT>() | (one16<DT>() << 16);
    const uint4 vone = make_uint4(ones[0], ones[1], ones[2], ones[3]);

    f32x16 acc[4], accs;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[r] = 0.f;

    __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (ET_NBUF - 2)));   // tile 0 landed (this wave's part)
    __syncthreads();
    uint32_t bc = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        unsigned char* buf = lds + bc * EM_TILEB;
        request_tile(t + ET_NBUF - 1, (bc + ET_NBUF - 1) % ET_NBUF);   // into the buffer tile t-1 just left
        const uint32_t valid = rend - (rbeg + t * EM_TILE);            // rows of this tile that exist (>= 1)
        if (valid < (uint32_t)EM_TILE) {                               // last, ragged tile: zero the rows past the end
            for (uint32_t e = threadIdx.x; e < (EM_TILE - valid) * 16; e += EM_THREADS)
                *reinterpret_cast<uint4*>(buf + (valid + (e >> 4)) * EM_ROWB + ((e & 15) << 4)) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 F[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t st = (wv + s) & 3;   // F[0] = this wave's own strip
                const unsigned char* p = buf + (off0 ^ (st << 6)) + ks * 4096;
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(uintptr_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)p);
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(uintptr_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(p + 1024));
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                F[s] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = mma32<DT>(F[0], F[s], acc[s]);   // C[own dim][dim of strip (wv + s) & 3]
            accs = mma32<DT>(F[0], vone, accs);                                   // C[own dim][*] = sum over the 16 tokens
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070 | (4 * (ET_NBUF - 2)));   // lgkmcnt(0) + tile t+1 landed
        __syncthreads();
        bc = (bc + 1) % ET_NBUF;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    float* s2 = a.s2 + ((size_t)bh * a.nchunk + chunk) * 128 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t jt = (wv + i) & 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t di = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            s2[(size_t)di * 128 + jt * 32 + n] = acc[i][r];
        }
    }
    if (n == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t di = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            a.dsum[((size_t)bh * a.nchunk + chunk) * 128 + di] = accs[r];
            a.m0[((size_t)bh * a.nchunk + chunk) * 128 + di] = 0.f;
        }
    }
}

// merge the per-chunk partials: mu[d], cov[i][j]  (Chan et al. pairwise update, all in fp32).
// grid = (16, B*Hq): workgroup x handles cov elements [x*1024, (x+1)*1024) of one head (and x == 0 also writes mu).
__global__ __launch_bounds__(256) void ea_qstats_combine(const float* __restrict__ s2, const float* __restrict__ dsum,
                                                         const float* __restrict__ m0, uint32_t Sq, uint32_t nchunk,
                                                         uint32_t rows_per_chunk, float* __restrict__ mu, float* __restrict__ cov, uint32_t pair, uint32_t dout, uint32_t quarters) {
    // pair != 0 (D = 64, ea_mfma_qstats): a "head" of this kernel is a PAIR of 64-dimensional heads that sit next to each other in a token row;
    // mu [.., 2 p + {0, 1}, 64] is the pair's 128 means as they are, cov gets the two diagonal 64 x 64 blocks (the cross-head blocks are dropped)
    // dout < 128 (D = 96, D = 64 in other layouts): the head's statistics are the leading dout x dout block of a 128-wide head padded with zeros
    extern __shared__ float sm[];  // muc[nchunk][128], del[nchunk][128], mug[128]
    float* muc = sm;
    float* del = sm + nchunk * 128;
    float* mug = del + nchunk * 128;
    const uint32_t bh = blockIdx.y;
    const float invN = 1.0f / (float)Sq;
    for (uint32_t e = threadIdx.x; e < nchunk * 128; e += 256) {
        const uint32_t c = e >> 7;
        const float nc = (float)(min((c + 1) * rows_per_chunk, Sq) - c * rows_per_chunk);
        const float d = dsum[(size_t)bh * nchunk * 128 + e] / nc;
        del[e] = d;
        muc[e] = m0[(size_t)bh * nchunk * 128 + e] + d;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
        for (uint32_t c = 0; c < nchunk; ++c) {
            const float nc = (float)(min((c + 1) * rows_per_chunk, Sq) - c * rows_per_chunk);
            s += nc * muc[c * 128 + threadIdx.x];
        }
        mug[threadIdx.x] = s * invN;
        if (quarters) {   // (bh = real head * 6 + pair of quarters: pairs 0 and 1 hold dimensions 0 .. 127 and 128 .. 255 in order)
            if (blockIdx.x == 0 && bh % 6 < 2) mu[(size_t)(bh / 6) * 256 + (bh % 6) * 128 + threadIdx.x] = s * invN;
        } else if (blockIdx.x == 0 && threadIdx.x < dout) mu[(size_t)bh * dout + threadIdx.x] = s * invN;
    }
    __syncthreads();
    if (!cov) return;
    // The 67 MB of raw second moments (32 heads x <= 32 chunks x 64 KiB) are this kernel's whole cost: every thread keeps its four
    // elements' loads of several chunks in flight (4 x 8 independent 4-byte loads; one dependent load per iteration ran at 2 TB/s).
    const uint32_t e0 = blockIdx.x * 1024 + threadIdx.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* sp = s2 + (size_t)bh * nchunk * 16384 + e0;
    for (uint32_t c0 = 0; c0 < nchunk; c0 += 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = min(c0 + u, nchunk - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = sp[(size_t)c * 16384 + q * 256];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = c0 + u;
            if (c < nchunk) {
                const float nc = (float)(min((c + 1) * rows_per_chunk, Sq) - c * rows_per_chunk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t e = e0 + q * 256, i = e >> 7, j = e & 127;
                    const float ai = muc[c * 128 + i] - mug[i], aj = muc[c * 128 + j] - mug[j];
                    s[q] += v[u][q] + nc * (ai * aj - del[c * 128 + i] * del[c * 128 + j]);   // (chunk order as before: same sums)
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t e = e0 + q * 256, i = e >> 7, j = e & 127;
        if (quarters) {
            const uint32_t p = bh % 6, Qi = (i >> 6) ? EQ_QB[p] : EQ_QA[p], Qj = (j >> 6) ? EQ_QB[p] : EQ_QA[p];
            if (Qi != Qj || p < 2) cov[((size_t)(bh / 6) * 256 + Qi * 64 + (i & 63)) * 256 + Qj * 64 + (j & 63)] = s[q] * invN;
        } else if (!pair) { if (i < dout && j < dout) cov[((size_t)bh * dout + i) * dout + j] = s[q] * invN; }
        else if (((i ^ j) & 64) == 0) cov[(((size_t)bh * 2 + (i >> 6)) * 64 + (i & 63)) * 64 + (j & 63)] = s[q] * invN;
    }
}

// Partials per head: about two workgroups per CU over all heads (32 heads: 16 chunks of 8192 rows), at least 4096 rows each.  Fewer
// partials are less to write, re-read and merge (32 per head: 67 MB), and a workgroup that walks more consecutive rows keeps its ring full
// for longer; below one workgroup per CU the stream starves.  Inside the ExpectedAttention bench loop (round 3 A/B, record
// profiles/r03_ab_bench.txt; statistics + combine, streaming loads): 32 per head 207 + 23 us, 16: 175 + 13, 8: 207 + 10.
void qstats_plan(int64_t Sq, int64_t nbh, uint32_t& nchunk, uint32_t& rows) {
    const int64_t target = std::max<int64_t>(1, 512 / std::max<int64_t>(1, nbh));
    const int64_t cap = std::min<int64_t>(32, target);
    int64_t nc = std::min<int64_t>(cap, std::max<int64_t>(1, (Sq + 4095) / 4096));
    nc = std::max<int64_t>(nc, std::min<int64_t>(32, (Sq + 16383) / 16384));   // many heads: still <= 16384 rows per fp32 partial (the raw moments' accuracy, see (1b))
    int64_t r = ((Sq + nc - 1) / nc + EM_TILE - 1) / EM_TILE * EM_TILE;
    nchunk = (uint32_t)((Sq + r - 1) / r);
    rows = (uint32_t)r;
}

// =================================================================================================
// (2) logits
// =================================================================================================
constexpr int EL_CHUNK = 4096;  // keys per workgroup, at least (ea_mfma_logits_chunk doubles it until the grid is one resident round)
constexpr int EL_TILE = 128;    // keys per LDS tile: ONE workgroup barrier per 128 keys (64-key tiles: a third of the wave cycles parked at it)
constexpr int EL_SUBS = EL_TILE / 32;
constexpr int EL_TILEB = EL_TILE * EM_ROWB;

template <int DT, bool HAS_COV>
__global__ __launch_bounds__(EM_THREADS, 2) void ea_logits_mfma_kernel(EaArgs a, float* __restrict__ logits, uint32_t nblk, uint32_t chunk_keys,
                                                                        float* __restrict__ part_m, float* __restrict__ part_z) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * EL_TILEB];
    __shared__ float red[3][8][EL_TILE];   // [tile % 3][wave strip x lane half][key]: the eight partial row-dots of a key.  Three: a tile's
                                           // last row-dot is written during the NEXT tile, its fold runs after that tile's barrier
    // XCD-aware order: workgroups go round-robin over the 8 XCDs (linear id % 8), each with its own L2.  The G query heads of a
    // (kv-head, key chunk) unit read the same K chunk, so they take CONSECUTIVE slots of ONE XCD: the chunk enters that L2 once
    // instead of G times through G different XCDs (measured at 128k: L2 fetch traffic 1.09 GB -> see DESIGN section 5).
    if (a.clear_word && blockIdx.x == 0 && threadIdx.x == 0) *a.clear_word = 0;
    const uint32_t slot = blockIdx.x >> 3, g = slot % a.G;
    const uint32_t unit = (slot / a.G) * 8 + (blockIdx.x & 7);
    if (unit >= nblk * a.B * a.Hkv) return;   // padding of the last round of 8 units
    const uint32_t chunk = unit % nblk, bh = unit / nblk;
    const uint32_t b = bh / a.Hkv, h = bh - b * a.Hkv;
    const uint32_t hq = h * a.G + g, bhq = b * a.Hq + hq;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)a.n_sink * a.k_ss) * 2;
    const int64_t row_bytes = a.k_ss * 2;

    // this wave's 32-row strip of cov (rows j = 32 wv + n), split hi + lo: A fragments for all 8 k-steps
    uint4 chi[8], clo[8];
    constexpr bool has_cov = HAS_COV;
    {
        const float* crow = has_cov ? a.cov + ((size_t)bhq * 128 + wv * 32 + n) * 128 : nullptr;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float x[8];
            if (has_cov) {
                const float4 u = *reinterpret_cast<const float4*>(crow + ks * 16 + kg * 8);
                const float4 w = *reinterpret_cast<const float4*>(crow + ks * 16 + kg * 8 + 4);
                x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = w.x; x[5] = w.y; x[6] = w.z; x[7] = w.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= a.inv_2d;   // 1 / 2d = 2^-8 (D = 128 on this path): exact, and the chains deliver (cov k)_r / 2d directly
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                hw[p] = pack2<DT>(x[2 * p], x[2 * p + 1]);
                lw[p] = pack2<DT>(x[2 * p] - lo16<DT>(hw[p]), x[2 * p + 1] - hi16<DT>(hw[p]));
            }
            chi[ks] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            clo[ks] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    }
    // mu of the 16 cov rows this lane sees in the C layout (row = 32 wv + (r&3) + 8 (r>>2) + 4 kg), pre-scaled
    // -- it is the accumulator every chain STARTS from: C = mu_r / sqrt(d) for every key, so the chain ends with mu_r / sqrt(d) + (cov k)_r / 2d
    f32x16 muv;
#pragma unroll
    for (int r = 0; r < 16; ++r) muv[r] = a.mu[(size_t)bhq * 128 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] * a.inv_sqrt_d;

    const uint32_t kbeg = chunk * chunk_keys;
    const uint32_t kend = min(kbeg + chunk_keys, a.Sp);
    const uint32_t ntiles = (kend - kbeg + EL_TILE - 1) / EL_TILE;
    float* lrow = logits + (size_t)bhq * a.Sp;
    float m_run = KVP_NEG_INF, z_run = 0.f;  // threads 0..127: running softmax partial of the keys they own

    // K fragments of 32-key sub-tile `sub`: all eight k-steps requested together, one sub-tile AHEAD of their use (the compiler's
    // own schedule requested two fragments at a time right in front of the MFMAs that consume them: one LDS round trip per four
    // MFMAs, which -- not the matrix pipe, the barrier or the K stream -- is what the round-2 kernel's 290 us were made of)
    auto frags = [&](const unsigned char* buf, int sub, uint4 (&kf)[8]) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = frag16(buf, sub, ks * 2 + kg, n);
    };
    // the MFMA chain of a sub-tile: cov strip (hi part, then lo part) x K sub-tile on ONE accumulator that starts at mu / sqrt(d).
    // (Round 2 ran two chains from zero and left "(hi + lo) * 1/2d + mu" to the row-dot: 48 VALU per sub-tile where 16 suffice; the lo
    // products are ~2^-9 of the hi ones and land in an fp32 accumulator of the sum's own magnitude: ~1e-6 relative, as before.)
    auto chains = [&](const uint4 (&kf)[8], f32x16& acc) {
        acc = muv;
        if (has_cov) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mma32<DT>(chi[ks], kf[ks], acc);  // C[cov row][key]
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mma32<DT>(clo[ks], kf[ks], acc);
        }
    };
    // row-dot of sub-tile `sub`: K in the C layout: key = sub*32 + n, dims 32 wv + 8 q + 4 kg + {0..3}: 8 bytes of 16-byte column 4 wv + q.
    // The four reads (krows) are issued one pipeline group ahead of the arithmetic (rowdot).
    auto krows = [&](const unsigned char* buf, int sub, uint2 (&kk)[4]) {
        const uint32_t row = sub * 32 + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) kk[q] = *reinterpret_cast<const uint2*>(buf + row * EM_ROWB + (((wv * 4 + q) ^ (row & 15)) << 4) + kg * 8);
    };
    auto rowdot = [&](const uint2 (&kk)[4], int sub, const f32x16& acc, float* redrow) {
        float v0 = 0.f, v1 = 0.f;   // two chains: sixteen dependent fma in a row would expose their latency
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v0 = fmaf(lo16<DT>(kk[q].x), acc[4 * q + 0], v0);
            v1 = fmaf(hi16<DT>(kk[q].x), acc[4 * q + 1], v1);
            v0 = fmaf(lo16<DT>(kk[q].y), acc[4 * q + 2], v0);
            v1 = fmaf(hi16<DT>(kk[q].y), acc[4 * q + 3], v1);
        }
        redrow[kg * EL_TILE + sub * 32 + n] = v0 + v1;   // both lane halves store their half of the strip's dims: the fold across them (a
                                                         // ds_bpermute round trip + a full LDS drain in the MFMA stream per sub-tile) happens in the tile's epilogue
    };
    // One pipeline group = the 16 MFMAs of a chain with the LDS reads of the NEXT group (nrd of them) and the row-dot arithmetic of the
    // PREVIOUS sub-tile spread between them: hipcc's own order clusters the row-dot between MFMA bursts and waits for each of its reads in turn.
#define EL_GROUP(nrd)                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
        if (i_ < (nrd)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                              \
    }                                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);

    // K tiles travel HBM / L2 -> LDS by LDS-DMA (no staging registers, no ds_write): request j of a tile moves rows 16 j + 4 wv + g (g = lane
    // / 16), lane slot i16 fetches the 16-byte chunk i16 ^ (row & 15) -- the XOR swizzle of frag16 / krows applied on the global side.
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t dg = lane >> 4, di16 = lane & 15;
    auto request_tile = [&](uint32_t row0, uint32_t buf_off) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t trow = 16 * j + 4 * wv + dg;
            const uint32_t r = min(row0 + trow, a.Sp - 1);   // rows past the end: any valid row (masked in the fold)
            const char* gp = kb + (int64_t)r * row_bytes + ((di16 ^ (trow & 15)) << 4);
            const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf_off + (16 * j + 4 * wv) * EM_ROWB);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
        }
    };
    unsigned char* bufc = lds;
    unsigned char* bufn = lds + EL_TILEB;
    request_tile(kbeg, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of tile 0 has landed
    __syncthreads();
    uint4 kfa[8], kfb[8];
    // Software pipeline over the 32-key sub-tiles, across tile boundaries: the chain of sub-tile s runs beside the row-dot of sub-tile
    // s - 1 (two accumulators) -- also the first chain of a tile, beside the LAST row-dot of the previous tile (its K values and its
    // accumulator are still in registers) -- and a tile's partial row-dots are folded after the NEXT tile's barrier.
    auto fold = [&](uint32_t tile, uint32_t rb) {   // logit of key tile*128 + (tid & 127) from the eight partials, running softmax partial;
#ifndef EL_FOLD_ALT
#define EL_FOLD_ALT 1
#endif
        if ((threadIdx.x >> 7) == (EL_FOLD_ALT ? (tile & 1) : 0u)) {     // waves 0-1 take the even tiles, waves 2-3 the odd ones (every thread keeps its own partial)
            const uint32_t kk = kbeg + tile * EL_TILE + (threadIdx.x & 127);
            if (kk < kend) {
                const float* rr = &red[rb][0][threadIdx.x & 127];
                const float l2 = (((rr[0] + rr[EL_TILE]) + (rr[2 * EL_TILE] + rr[3 * EL_TILE])) + ((rr[4 * EL_TILE] + rr[5 * EL_TILE]) + (rr[6 * EL_TILE] + rr[7 * EL_TILE]))) * KVP_LOG2E;
                lrow[kk] = l2;
                softmax_merge(m_run, z_run, l2, 1.0f);
            }
        }
    };
    f32x16 acc0, acc1 = muv;
    uint2 kr0[4], kr1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kr1[q] = make_uint2(0, 0);   // tile 0 has no predecessor: its deferred row-dot writes zeros into a buffer nobody reads
    uint32_t rcur = 0, rprev = 2;                            // red buffers of tile t and of tile t - 1 (t % 3, (t - 1) % 3)
    static_assert(EL_SUBS == 4, "the pipeline below is written out for four sub-tiles");
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t key0 = kbeg + t * EL_TILE;
        if (t + 1 < ntiles) request_tile(key0 + EL_TILE, (uint32_t)(bufn - lds));   // into the buffer tile t - 1 left at the last barrier
        float* redw = red[rcur][2 * wv];
        __builtin_amdgcn_sched_barrier(0);
        if (has_cov) frags(bufc, 0, kfa);
        __builtin_amdgcn_sched_barrier(0);
        if (has_cov) frags(bufc, 1, kfb);
        krows(bufc, 0, kr0);
        chains(kfa, acc0);                                 // sub-tile 0 || row-dot 3 of the previous tile
        rowdot(kr1, 3, acc1, red[rprev][2 * wv]);
        EL_GROUP(12)
        __builtin_amdgcn_sched_barrier(0);
        krows(bufc, 1, kr1);
        if (has_cov) frags(bufc, 2, kfa);
        chains(kfb, acc1);                                 // sub-tile 1 || row-dot 0
        rowdot(kr0, 0, acc0, redw);
        EL_GROUP(12)
        __builtin_amdgcn_sched_barrier(0);
        krows(bufc, 2, kr0);
        if (has_cov) frags(bufc, 3, kfb);
        chains(kfa, acc0);                                 // sub-tile 2 || row-dot 1
        rowdot(kr1, 1, acc1, redw);
        EL_GROUP(12)
        __builtin_amdgcn_sched_barrier(0);
        krows(bufc, 3, kr1);
        chains(kfb, acc1);                                 // sub-tile 3 || row-dot 2
        rowdot(kr0, 2, acc0, redw);
        EL_GROUP(4)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): tile t + 1 has landed (this wave's part; the barrier covers the others)
        __syncthreads();
        unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
        if (t > 0) fold(t - 1, rprev);
        rprev = rcur;
        rcur = rcur == 2 ? 0 : rcur + 1;
    }
    rowdot(kr1, 3, acc1, red[rprev][2 * wv]);   // the last tile's last row-dot
    __syncthreads();
    fold(ntiles - 1, rprev);
    // every thread owns keys: merge inside each wave, then across the four through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m_run, o), z2 = __shfl_xor(z_run, o);
        softmax_merge(m_run, z_run, m2, z2);
    }
    __syncthreads();
    if (lane == 0 && wv > 0) { red[0][0][2 * wv] = m_run; red[0][0][2 * wv + 1] = z_run; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) softmax_merge(m_run, z_run, red[0][0][2 * w], red[0][0][2 * w + 1]);
        part_m[(size_t)bhq * nblk + chunk] = m_run;
        part_z[(size_t)bhq * nblk + chunk] = z_run;
    }
}

// ---- (2b) the quadratic form on the UPPER TRIANGLE of the covariance ----------------------------------------------------------------
// k^T C k = k^T U k with U_jj = C_jj, U_jc = C_jc + C_cj for c > j, 0 below the diagonal (exact for any C; for a symmetric one the doubled
// upper triangle).  Strip s (rows 32 s .. 32 s + 31) of U only has columns >= 32 s, i.e. the 16-wide k-steps 2 s .. 7: 8 + 6 + 4 + 2 = 20
// (strip, k-step) products per 32-key sub-tile instead of 32 -- 40 MFMAs with the hi / lo split instead of 64.  Balance: waves 0 / 1 hold
// strips 0 and 3 (8 + 2 k-steps), waves 2 / 3 strips 1 and 2 (6 + 4); within a pair the even wave takes sub-tiles 0, 1 of every 128-key
// tile, the odd one sub-tiles 2, 3: 2 x 10 x 2 = 40 MFMAs per wave and tile each, and a wave reads the K fragments of only two sub-tiles.
// A first version (VERDICT r2 #4b) measured slower than the full form: the round-2 kernel was bound by LDS round trips, not by the matrix
// pipe.  With the pipeline of (2) -- LDS-DMA tiles, chain of unit u || row-dot of unit u - 1 || reads of unit u + 1, pinned by
// sched_group_barrier, across tile boundaries -- the matrix pipe and its power are what is left, and fewer MFMAs are fewer microseconds.
// mu / sqrt(d) lives in LDS: a chain's accumulator is initialised by four broadcast reads (no registers held for it).
template <int DT, int S_> struct ElTriFrag { uint4 hi[8 - 2 * S_], lo[8 - 2 * S_]; };

template <int DT, int S_>
__device__ __forceinline__ void el_tri_build(const float* __restrict__ cov_head, uint32_t n, uint32_t kg, float inv_2d, ElTriFrag<DT, S_>& f) {
    const uint32_t j = 32 * S_ + n;   // this lane's row of U
#pragma unroll
    for (int i = 0; i < 8 - 2 * S_; ++i) {
        const int ks = 2 * S_ + i;
        const float* crow = cov_head + (size_t)j * 128 + ks * 16 + kg * 8;
        const float4 u = *reinterpret_cast<const float4*>(crow), w = *reinterpret_cast<const float4*>(crow + 4);
        float x[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t c = ks * 16 + kg * 8 + e;
            const float t = cov_head[(size_t)c * 128 + j];   // C_cj: lanes n -> consecutive addresses
            x[e] = (c > j ? x[e] + t : (c == j ? x[e] : 0.f)) * inv_2d;   // 1 / 2d = 2^-8: exact
        }
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            hw[p] = pack2<DT>(x[2 * p], x[2 * p + 1]);
            lw[p] = pack2<DT>(x[2 * p] - lo16<DT>(hw[p]), x[2 * p + 1] - hi16<DT>(hw[p]));
        }
        f.hi[i] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        f.lo[i] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

// One pipeline group: the chain's MFMAs with, in its FIRST half, the LDS reads of the next unit and the row-dot of the previous one (so that
// the accumulator it consumed is free early), and in its second half the four reads that re-initialise that accumulator for the next group.
#define EL_TRI_GROUP(nmfma, nrd, nvalu)                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < (nmfma) / 2; ++i_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, ((nrd) + (nmfma) / 2 - 1) / ((nmfma) / 2), 0);              \
        __builtin_amdgcn_sched_group_barrier(0x002, ((nvalu) + (nmfma) / 2 - 1) / ((nmfma) / 2), 0);            \
    }                                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < (nmfma) - (nmfma) / 2; ++i_) {                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, (4 + (nmfma) - (nmfma) / 2 - 1) / ((nmfma) - (nmfma) / 2), 0); \
    }

template <int DT>
__global__ __launch_bounds__(EM_THREADS, 2) void ea_logits_mfma_tri_kernel(EaArgs a, float* __restrict__ logits, uint32_t nblk, uint32_t chunk_keys,
                                                                       float* __restrict__ part_m, float* __restrict__ part_z) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * EL_TILEB];
    __shared__ float red[3][8][EL_TILE];
    __shared__ __attribute__((aligned(16))) float mus[128];
    if (a.clear_word && blockIdx.x == 0 && threadIdx.x == 0) *a.clear_word = 0;
    const uint32_t slot = blockIdx.x >> 3, g = slot % a.G;   // XCD-aware order: see ea_logits_mfma_kernel
    const uint32_t unit = (slot / a.G) * 8 + (blockIdx.x & 7);
    if (unit >= nblk * a.B * a.Hkv) return;
    const uint32_t chunk = unit % nblk, bh = unit / nblk;
    const uint32_t b = bh / a.Hkv, h = bh - b * a.Hkv;
    const uint32_t hq = h * a.G + g, bhq = b * a.Hq + hq;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)a.n_sink * a.k_ss) * 2;
    const int64_t row_bytes = a.k_ss * 2;
    if (threadIdx.x < 128) mus[threadIdx.x] = a.mu[(size_t)bhq * 128 + threadIdx.x] * a.inv_sqrt_d;

    const uint32_t kbeg = chunk * chunk_keys;
    const uint32_t kend = min(kbeg + chunk_keys, a.Sp);
    const uint32_t ntiles = (kend - kbeg + EL_TILE - 1) / EL_TILE;
    float* lrow = logits + (size_t)bhq * a.Sp;
    float m_run = KVP_NEG_INF, z_run = 0.f;

    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t dg = lane >> 4, di16 = lane & 15;
    // Round 6: a tile that lies inside the sequence is requested with ONE per-lane offset (row 4 wv + dg of a 16-row group, swizzled chunk:
    // the same for all eight groups, 16 j does not touch the row's low bits) on top of a SCALAR base that advances by 16 rows per request --
    // no vector address arithmetic in the tile loop (it was ~48 of its ~200 VALU instructions per tile, on a SIMD where VALU and matrix
    // instructions do not overlap).  The last tile of a sequence (rows clamped to Sp - 1) and rows further than 2 GiB apart keep the per-lane form.
    const uint32_t grow = 4 * wv + dg;
    const bool fast_rows = row_bytes > 0 && row_bytes * EL_TILE < ((int64_t)1 << 31);
    const uint32_t goff = (uint32_t)(grow * (uint32_t)row_bytes) + ((di16 ^ (grow & 15)) << 4);
    auto request_tile = [&](uint32_t row0, uint32_t buf_off) {
        if (fast_rows && row0 + EL_TILE <= a.Sp) {
            const char* sb = kb + (int64_t)row0 * row_bytes;   // uniform: scalar registers
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf_off + (16 * j + 4 * wv) * EM_ROWB);
                const uint64_t sj = (uint64_t)(uintptr_t)(sb + (int64_t)(16 * j) * row_bytes);
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sj), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sj >> 32));
                const uint64_t sbase = ((uint64_t)hi << 32) | lo;
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(goff), "s"(sbase) : "memory");
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t trow = 16 * j + 4 * wv + dg;
            const uint32_t r = min(row0 + trow, a.Sp - 1);
            const char* gp = kb + (int64_t)r * row_bytes + ((di16 ^ (trow & 15)) << 4);
            const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf_off + (16 * j + 4 * wv) * EM_ROWB);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
        }
    };
    auto fold = [&](uint32_t tile, uint32_t rb) {
        if ((threadIdx.x >> 7) == (tile & 1)) {
            const uint32_t kk = kbeg + tile * EL_TILE + (threadIdx.x & 127);
            if (kk < kend) {
                const float* rr = &red[rb][0][threadIdx.x & 127];
                const float l2 = (((rr[0] + rr[EL_TILE]) + (rr[2 * EL_TILE] + rr[3 * EL_TILE])) + ((rr[4 * EL_TILE] + rr[5 * EL_TILE]) + (rr[6 * EL_TILE] + rr[7 * EL_TILE]))) * KVP_LOG2E;
                lrow[kk] = l2;
                softmax_merge(m_run, z_run, l2, 1.0f);
            }
        }
    };
    unsigned char* bufc = lds;
    unsigned char* bufn = lds + EL_TILEB;
    request_tile(kbeg, 0);

    // the walk of one wave: strips SA (the longer chain) and SB, sub-tiles sa = 2 (wv & 1) and sa + 1 of every tile
    auto walk = [&](auto sa_tag, auto sb_tag) {
        constexpr int SA = decltype(sa_tag)::value, SB = decltype(sb_tag)::value;
        constexpr int NA = 8 - 2 * SA, NB = 8 - 2 * SB, KS0 = 2 * (SA < SB ? SA : SB);
        ElTriFrag<DT, SA> fa;
        ElTriFrag<DT, SB> fb;
        const float* cov_head = a.cov + (size_t)bhq * 128 * 128;
        el_tri_build<DT, SA>(cov_head, n, kg, a.inv_2d, fa);
        el_tri_build<DT, SB>(cov_head, n, kg, a.inv_2d, fb);
        const uint32_t sa = 2 * (wv & 1), sb = sa + 1;
        auto frags = [&](const unsigned char* buf, uint32_t sub, uint4 (&kf)[8]) {
#pragma unroll
            for (int ks = KS0; ks < 8; ++ks) kf[ks] = frag16(buf, sub, ks * 2 + kg, n);
        };
        auto krows = [&](const unsigned char* buf, uint32_t sub, int strip, uint2 (&kk)[4]) {
            const uint32_t row = sub * 32 + n;
#pragma unroll
            for (int q = 0; q < 4; ++q) kk[q] = *reinterpret_cast<const uint2*>(buf + row * EM_ROWB + (((strip * 4 + q) ^ (row & 15)) << 4) + kg * 8);
        };
        // an accumulator is (re)initialised with mu / sqrt(d) by four broadcast LDS reads issued at the END of the group in which its
        // previous contents were consumed -- one whole group before its next chain starts, so that chain never waits for them
        auto acc_init = [&](int strip, f32x16& acc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 m4 = *reinterpret_cast<const float4*>(&mus[32 * strip + 8 * q + 4 * kg]);
                acc[4 * q] = m4.x; acc[4 * q + 1] = m4.y; acc[4 * q + 2] = m4.z; acc[4 * q + 3] = m4.w;
            }
        };
        auto chain_a = [&](const uint4 (&kf)[8], f32x16& acc) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc = mma32<DT>(fa.hi[i], kf[2 * SA + i], acc);
#pragma unroll
            for (int i = 0; i < NA; ++i) acc = mma32<DT>(fa.lo[i], kf[2 * SA + i], acc);
        };
        auto chain_b = [&](const uint4 (&kf)[8], f32x16& acc) {
#pragma unroll
            for (int i = 0; i < NB; ++i) acc = mma32<DT>(fb.hi[i], kf[2 * SB + i], acc);
#pragma unroll
            for (int i = 0; i < NB; ++i) acc = mma32<DT>(fb.lo[i], kf[2 * SB + i], acc);
        };
        auto rowdot = [&](const uint2 (&kk)[4], uint32_t sub, int strip, const f32x16& acc, float (*redb)[EL_TILE]) {
            // two interleaved fma chains (even / odd accumulator rows) as ONE packed chain (v_pk_fma_f32: 8 instead of 16 issue slots;
            // component for component the arithmetic of the scalar form: the same bits)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 v = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 k0 = {lo16<DT>(kk[q].x), hi16<DT>(kk[q].x)}, a0 = {acc[4 * q + 0], acc[4 * q + 1]};
                v = __builtin_elementwise_fma(k0, a0, v);
                const f32x2 k1 = {lo16<DT>(kk[q].y), hi16<DT>(kk[q].y)}, a1 = {acc[4 * q + 2], acc[4 * q + 3]};
                v = __builtin_elementwise_fma(k1, a1, v);
            }
            redb[2 * strip + kg][sub * 32 + n] = v.x + v.y;
        };
        f32x16 acc0, acc1;
        uint4 kfa[8], kfb[8];
        uint2 krAa[4], krBa[4], krAb[4], krBb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) krBb[q] = make_uint2(0, 0);   // tile 0 has no predecessor: its deferred row-dot writes zeros nobody reads
        uint32_t rcur = 0, rprev = 2;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of tile 0 has landed
        __syncthreads();                      // (also publishes mus)
        acc_init(SA, acc0);
        acc_init(SB, acc1);
        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint32_t key0 = kbeg + t * EL_TILE;
            if (t + 1 < ntiles) request_tile(key0 + EL_TILE, (uint32_t)(bufn - lds));
            __builtin_amdgcn_sched_barrier(0);
            frags(bufc, sa, kfa);
            krows(bufc, sa, SA, krAa);
            __builtin_amdgcn_sched_barrier(0);
            krows(bufc, sa, SB, krBa);
            {   // unit (sa, SA) || row-dot (sb, SB) of the previous tile, then acc1 <- mu
                const f32x16 prev = acc1;
                chain_a(kfa, acc0);
                rowdot(krBb, sb, SB, prev, red[rprev]);
                acc_init(SB, acc1);
            }
            EL_TRI_GROUP(2 * NA, 4, 26)
            __builtin_amdgcn_sched_barrier(0);
            frags(bufc, sb, kfb);       // (here, not a unit earlier: sixteen fragment registers live at once were the register budget)
            krows(bufc, sb, SA, krAb);
            {   // unit (sa, SB) || row-dot (sa, SA), then acc0 <- mu
                const f32x16 prev = acc0;
                chain_b(kfa, acc1);
                rowdot(krAa, sa, SA, prev, red[rcur]);
                acc_init(SA, acc0);
            }
            EL_TRI_GROUP(2 * NB, 8 - KS0 + 4, 26)
            __builtin_amdgcn_sched_barrier(0);
            krows(bufc, sb, SB, krBb);
            {   // unit (sb, SA) || row-dot (sa, SB), then acc1 <- mu
                const f32x16 prev = acc1;
                chain_a(kfb, acc0);
                rowdot(krBa, sa, SB, prev, red[rcur]);
                acc_init(SB, acc1);
            }
            EL_TRI_GROUP(2 * NA, 4, 26)
            __builtin_amdgcn_sched_barrier(0);
            {   // unit (sb, SB) || row-dot (sb, SA), then acc0 <- mu
                const f32x16 prev = acc0;
                chain_b(kfb, acc1);
                rowdot(krAb, sb, SA, prev, red[rcur]);
                acc_init(SA, acc0);
            }
            EL_TRI_GROUP(2 * NB, 0, 26)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): tile t + 1 has landed
            __syncthreads();
            unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
            if (t > 0) fold(t - 1, rprev);
            rprev = rcur;
            rcur = rcur == 2 ? 0 : rcur + 1;
        }
        rowdot(krBb, sb, SB, acc1, red[rprev]);   // the last tile's last row-dot
        __syncthreads();
        fold(ntiles - 1, rprev);
    };
    if (wv < 2) walk(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    else walk(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m_run, o), z2 = __shfl_xor(z_run, o);
        softmax_merge(m_run, z_run, m2, z2);
    }
    __syncthreads();
    if (lane == 0 && wv > 0) { red[0][0][2 * wv] = m_run; red[0][0][2 * wv + 1] = z_run; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) softmax_merge(m_run, z_run, red[0][0][2 * w], red[0][0][2 * w + 1]);
        part_m[(size_t)bhq * nblk + chunk] = m_run;
        part_z[(size_t)bhq * nblk + chunk] = z_run;
    }
}

// ---- (2c) head sizes 64 and 96 (round 6) -----------------------------------------------------------------------------------------------
// The same quadratic form on the doubled upper triangle, hi + lo split, for heads of DK = 4 / 6 k-steps of 16 dimensions (Llama-3.2-1B,
// Qwen2-0.5B: D = 64; Phi-3-mini: 96) -- they took the scalar generic kernel (32k tokens x 32 heads: 1.06 ms against 0.17 for D = 128).
// With D / 32 = 2 or 3 strips of U there is nothing to balance across waves: wave w takes the 32-key sub-tile w of every 128-key tile with ALL
// strips (6 / 12 (strip, k-step) products: 12 / 24 MFMAs per wave and tile), adds the strips' row-dots in registers, folds its two lane halves
// with one cross-lane read and owns its 32 logits (store + running softmax partial): no partial-sum buffers, one barrier per tile (the K ring).
// K rows of 128 bytes sit two per LDS bank line, so their 16-byte slots rotate with bits 1 .. 3 of the row; rows of 192 bytes keep 256-byte LDS
// rows and the lanes whose chunk does not exist request nothing (as snapkv_mfma.hip's KGeo).  Compiler-scheduled: these shapes are not the benchmark's.
template <int DK> struct ElGeo {
    static constexpr int D = DK * 16;
    static constexpr int NS = DK / 2;                     // 32-row strips of U
    static constexpr int ROWB = DK == 4 ? 128 : 256;      // bytes per key row IN LDS
    static constexpr int NCH = DK * 2;                    // 16-byte chunks a key row has
    static constexpr int CPR = ROWB / 16;
    static constexpr int RPW = 1024 / ROWB;               // rows one wave's request moves
    static constexpr int RPR = (EM_THREADS / 64) * RPW;   // rows one request of the workgroup moves
    static constexpr int NREQ = EL_TILE / RPR;
    static constexpr int TILEB = EL_TILE * ROWB;
    static constexpr int OCC = DK == 4 ? 3 : 2;           // workgroups per CU (registers: 6 / 12 hi + lo fragment pairs -- four per CU spilled five; LDS: 2 x 16 / 32 KiB)
    static_assert(RPR % 16 == 0 && EL_TILE % RPR == 0, "the swizzle of a row depends on its index inside a request only");
    static __device__ __forceinline__ uint32_t sw(uint32_t row) { return DK == 4 ? ((row >> 1) & 7u) : (row & 15u); }
};
template <int DT, int DK, int S_> struct ElSmallFrag { uint4 hi[DK - 2 * S_], lo[DK - 2 * S_]; };

// BATCH > 0: a scheduling barrier after every BATCH fragments -- the compiler otherwise hoists ALL their loads to the top (16 values per fragment:
// with the 18 fragments of the D = 256 kernel that is more than the register file, and the f16 build spilled 312 registers at start-up)
template <int DT, int DK, int S_, int BATCH = 0>
__device__ __forceinline__ void el_small_build(const float* __restrict__ cov_head, uint32_t n, uint32_t kg, float inv_2d, ElSmallFrag<DT, DK, S_>& f) {
    constexpr int D = DK * 16;
    const uint32_t j = 32 * S_ + n;   // this lane's row of U
#pragma unroll
    for (int i = 0; i < DK - 2 * S_; ++i) {
        const int ks = 2 * S_ + i;
        const float* crow = cov_head + (size_t)j * D + ks * 16 + kg * 8;
        const float4 u = *reinterpret_cast<const float4*>(crow), w = *reinterpret_cast<const float4*>(crow + 4);
        float x[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t c = ks * 16 + kg * 8 + e;
            const float t = cov_head[(size_t)c * D + j];
            x[e] = (c > j ? x[e] + t : (c == j ? x[e] : 0.f)) * inv_2d;   // (1 / 2d is a power of two for D = 64 only; the hi + lo split carries whatever the product rounds to)
        }
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            hw[p] = pack2<DT>(x[2 * p], x[2 * p + 1]);
            lw[p] = pack2<DT>(x[2 * p] - lo16<DT>(hw[p]), x[2 * p + 1] - hi16<DT>(hw[p]));
        }
        f.hi[i] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        f.lo[i] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        if (BATCH > 0 && i % (BATCH > 0 ? BATCH : 1) == (BATCH > 0 ? BATCH : 1) - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int DT, int DK, bool HAS_COV>
__global__ __launch_bounds__(EM_THREADS, ElGeo<DK>::OCC) void ea_logits_mfma_small_kernel(EaArgs a, float* __restrict__ logits, uint32_t nblk, uint32_t chunk_keys,
                                                                                          float* __restrict__ part_m, float* __restrict__ part_z) {
    using Geo = ElGeo<DK>;
    constexpr int D = Geo::D;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * Geo::TILEB];
    __shared__ __attribute__((aligned(16))) float mus[D];
    __shared__ float wred[8];
    if (a.clear_word && blockIdx.x == 0 && threadIdx.x == 0) *a.clear_word = 0;
    const uint32_t slot = blockIdx.x >> 3, g = slot % a.G;   // XCD-aware order: see ea_logits_mfma_kernel
    const uint32_t unit = (slot / a.G) * 8 + (blockIdx.x & 7);
    if (unit >= nblk * a.B * a.Hkv) return;
    const uint32_t chunk = unit % nblk, bh = unit / nblk;
    const uint32_t b = bh / a.Hkv, h = bh - b * a.Hkv;
    const uint32_t hq = h * a.G + g, bhq = b * a.Hq + hq;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)a.n_sink * a.k_ss) * 2;
    const int64_t row_bytes = a.k_ss * 2;
    if (threadIdx.x < D) mus[threadIdx.x] = a.mu[(size_t)bhq * D + threadIdx.x] * a.inv_sqrt_d;

    const uint32_t kbeg = chunk * chunk_keys;
    const uint32_t kend = min(kbeg + chunk_keys, a.Sp);
    const uint32_t ntiles = (kend - kbeg + EL_TILE - 1) / EL_TILE;
    float* lrow = logits + (size_t)bhq * a.Sp;
    float m_run = KVP_NEG_INF, z_run = 0.f;   // lanes 0 .. 31 of every wave: the keys they own

    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t rrow = Geo::RPW * wv + lane / Geo::CPR;                 // this lane's row inside a request
    const uint32_t rchunk = (lane % Geo::CPR) ^ Geo::sw(rrow);             // (a request's first row is a multiple of the swizzle period)
    auto request_tile = [&](uint32_t row0, uint32_t buf_off) {
#pragma unroll
        for (int j = 0; j < Geo::NREQ; ++j) {
            const uint32_t r = min(row0 + Geo::RPR * j + rrow, a.Sp - 1);   // rows past the end: any valid row (never stored)
            const char* gp = kb + (int64_t)r * row_bytes + (rchunk << 4);
            const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf_off + (Geo::RPR * j + Geo::RPW * wv) * Geo::ROWB);
            if (Geo::NCH == Geo::CPR) {
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
            } else {
                asm volatile("s_mov_b32 m0, %0" ::"s"(la) : "memory");
                if (rchunk < (uint32_t)Geo::NCH) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gp) : "memory");
            }
        }
    };
    request_tile(kbeg, 0);

    ElSmallFrag<DT, DK, 0> f0;
    ElSmallFrag<DT, DK, 1> f1;
    ElSmallFrag<DT, DK, (DK > 4 ? 2 : 1)> f2;   // (D = 64: unused)
    if (HAS_COV) {
        const float* cov_head = a.cov + (size_t)bhq * D * D;
        el_small_build<DT, DK, 0>(cov_head, n, kg, a.inv_2d, f0);
        el_small_build<DT, DK, 1>(cov_head, n, kg, a.inv_2d, f1);
        if (DK > 4) el_small_build<DT, DK, (DK > 4 ? 2 : 1)>(cov_head, n, kg, a.inv_2d, f2);
    }
    const uint32_t row = wv * 32 + n;                       // this lane's key row of every tile
    const uint32_t rsw = Geo::sw(row);
    // one strip: C[row of U][key] = mu_r / sqrt(d) + (U k)_r / 2d on an accumulator that starts at mu / sqrt(d), then its dot with k (read back in
    // the C layout: dims 32 s + 8 q + 4 kg + {0 .. 3} of key n)
    auto strip = [&](auto s_tag, const auto& f, const unsigned char* buf, const uint4 (&kf)[DK]) -> float {
        constexpr int S = decltype(s_tag)::value;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 m4 = *reinterpret_cast<const float4*>(&mus[32 * S + 8 * q + 4 * kg]);
            acc[4 * q] = m4.x; acc[4 * q + 1] = m4.y; acc[4 * q + 2] = m4.z; acc[4 * q + 3] = m4.w;
        }
        uint2 kk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kk[q] = *reinterpret_cast<const uint2*>(buf + row * Geo::ROWB + (((S * 4 + q) ^ rsw) << 4) + kg * 8);
        if (HAS_COV) {
#pragma unroll
            for (int i = 0; i < DK - 2 * S; ++i) acc = mma32<DT>(f.hi[i], kf[2 * S + i], acc);
#pragma unroll
            for (int i = 0; i < DK - 2 * S; ++i) acc = mma32<DT>(f.lo[i], kf[2 * S + i], acc);
        }
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v0 = fmaf(lo16<DT>(kk[q].x), acc[4 * q + 0], v0);
            v1 = fmaf(hi16<DT>(kk[q].x), acc[4 * q + 1], v1);
            v0 = fmaf(lo16<DT>(kk[q].y), acc[4 * q + 2], v0);
            v1 = fmaf(hi16<DT>(kk[q].y), acc[4 * q + 3], v1);
        }
        return v0 + v1;
    };
    unsigned char* bufc = lds;
    unsigned char* bufn = lds + Geo::TILEB;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of tile 0 has landed
    __syncthreads();                      // (also publishes mus)
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t key0 = kbeg + t * EL_TILE;
        if (t + 1 < ntiles) request_tile(key0 + EL_TILE, (uint32_t)(bufn - lds));   // into the buffer tile t - 1 left at the last barrier
        uint4 kf[DK];
        if (HAS_COV) {
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(bufc + row * Geo::ROWB + (((ks * 2 + kg) ^ rsw) << 4));
        }
        float v = strip(std::integral_constant<int, 0>{}, f0, bufc, kf);
        v += strip(std::integral_constant<int, 1>{}, f1, bufc, kf);
        if (DK > 4) v += strip(std::integral_constant<int, (DK > 4 ? 2 : 1)>{}, f2, bufc, kf);
        v += __shfl_xor(v, 32);   // the two lane halves hold the two halves of every strip's dims
        const uint32_t kk = key0 + row;
        if (kg == 0 && kk < kend) {
            const float l2 = v * KVP_LOG2E;
            lrow[kk] = l2;
            softmax_merge(m_run, z_run, l2, 1.0f);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): tile t + 1 has landed (this wave's part; the barrier covers the others)
        __syncthreads();
        unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m_run, o), z2 = __shfl_xor(z_run, o);
        softmax_merge(m_run, z_run, m2, z2);
    }
    if (lane == 0) { wred[2 * wv] = m_run; wred[2 * wv + 1] = z_run; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) softmax_merge(m_run, z_run, wred[2 * w], wred[2 * w + 1]);
        part_m[(size_t)bhq * nblk + chunk] = m_run;
        part_z[(size_t)bhq * nblk + chunk] = z_run;
    }
}

// ---- (2d) head size 256 (Gemma; round 6) -------------------------------------------------------------------------------------------
// Eight strips of U, 16 k-steps: strip s has k-steps 2 s .. 15, i.e. 16 + 14 + ... + 2 = 72 (strip, k-step) products per 32-key sub-tile.
// Four waves: wave w holds strips w and 7 - w (18 products: the same for every wave; 144 hi / lo fragment registers) and walks all four
// sub-tiles of every 128-key tile -- ONE workgroup per CU with the whole register file (launch_bounds(256, 1)): with eight waves of 256
// registers (two per SIMD, measured the same: 215 against 226 us) the f16 build sat ON the register cliff and spilled 308 of them at any
// change of its start-up code.  A sub-tile's K fragments are read eight k-steps at a time into ONE register set (k-steps
// 0 .. 7: only the longer strip has them; 8 .. 15: both strips, one after the other on one accumulator): 144 + 32 + 16 registers + addresses.
// Every LDS address is a per-lane base computed once + an immediate (sub-tile: 16 KiB, upper half of a row: 256 bytes) -- the first version
// recomputed them per sub-tile and kept accumulators in AGPRs: 234 VALU instructions per (wave, sub-tile) beside 36 matrix instructions,
// 226 us for 32k tokens x 32 heads with the matrix pipe 34 % busy (PMC).  The strips' row-dots of a key are added in registers, the lane halves
// folded by one cross-lane read, the four strip pairs' partials meet in LDS (double-buffered: the fold of tile t runs after the barrier that
// also publishes tile t + 1).  K rows are 512 bytes: a wave's LDS-DMA request moves two rows, the slots of a row rotate with its low four
// bits inside each 256-byte half.  (The generic kernel this replaces took 13.6 ms and was refused outright before round 6.)
constexpr int EB_THREADS = 256;
constexpr int EB_ROWB = 512;
constexpr int EB_TILEB = EL_TILE * EB_ROWB;   // 64 KiB
template <int DT, bool HAS_COV>
__global__ __launch_bounds__(EB_THREADS, 1) void ea_logits_mfma_big_kernel(EaArgs a, float* __restrict__ logits, uint32_t nblk, uint32_t chunk_keys,
                                                                           float* __restrict__ part_m, float* __restrict__ part_z) {
    constexpr int D = 256, DK = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_lds[];   // 2 x 64 KiB of K tiles | red[2][4][128] | mus[256] | wred[4]
    unsigned char* lds = big_lds;
    float (*red)[4][EL_TILE] = reinterpret_cast<float (*)[4][EL_TILE]>(big_lds + 2 * EB_TILEB);
    float* mus = reinterpret_cast<float*>(big_lds + 2 * EB_TILEB + 2 * 4 * EL_TILE * 4);
    float* wred = mus + D;
    if (a.clear_word && blockIdx.x == 0 && threadIdx.x == 0) *a.clear_word = 0;
    const uint32_t slot = blockIdx.x >> 3, g = slot % a.G;   // XCD-aware order: see ea_logits_mfma_kernel
    const uint32_t unit = (slot / a.G) * 8 + (blockIdx.x & 7);
    if (unit >= nblk * a.B * a.Hkv) return;
    const uint32_t chunk = unit % nblk, bh = unit / nblk;
    const uint32_t b = bh / a.Hkv, h = bh - b * a.Hkv;
    const uint32_t hq = h * a.G + g, bhq = b * a.Hq + hq;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)a.n_sink * a.k_ss) * 2;
    const int64_t row_bytes = a.k_ss * 2;
    if (threadIdx.x < D) mus[threadIdx.x] = a.mu[(size_t)bhq * D + threadIdx.x] * a.inv_sqrt_d;

    const uint32_t kbeg = chunk * chunk_keys;
    const uint32_t kend = min(kbeg + chunk_keys, a.Sp);
    const uint32_t ntiles = (kend - kbeg + EL_TILE - 1) / EL_TILE;
    float* lrow = logits + (size_t)bhq * a.Sp;
    float m_run = KVP_NEG_INF, z_run = 0.f;   // threads 0 .. 127: the keys they fold

    // LDS-DMA: request j (0 .. 15) of a tile moves rows 8 j + 2 wv + lane / 32; lane slot p = lane % 32 fetches chunk p ^ (row & 15) (bit 4 of the
    // chunk -- the 256-byte half -- stays): the rows of a request alternate between two swizzle phases (8 j is 0 or 8 mod 16)
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t rrow = 2 * wv + (lane >> 5);
    const uint32_t rch0 = ((lane & 31) ^ (rrow & 15)) << 4, rch1 = ((lane & 31) ^ ((rrow + 8) & 15)) << 4;
    auto request_tile = [&](uint32_t row0, uint32_t buf_off) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t r = min(row0 + 8 * j + rrow, a.Sp - 1);   // rows past the end: any valid row (never stored)
            const char* gp = kb + (int64_t)r * row_bytes + ((j & 1) ? rch1 : rch0);
            const uint32_t la = __builtin_amdgcn_readfirstlane(ldsbase + buf_off + (8 * j + 2 * wv) * EB_ROWB);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
        }
    };
    request_tile(kbeg, 0);

    auto fold = [&](uint32_t tile) {   // logit of key tile * 128 + tid from the four strip pairs' partials, running softmax partial
        if (threadIdx.x < EL_TILE) {
            const uint32_t kk = kbeg + tile * EL_TILE + threadIdx.x;
            if (kk < kend) {
                const float (*rr)[EL_TILE] = red[tile & 1];
                const float l2 = ((rr[0][threadIdx.x] + rr[1][threadIdx.x]) + (rr[2][threadIdx.x] + rr[3][threadIdx.x])) * KVP_LOG2E;
                lrow[kk] = l2;
                softmax_merge(m_run, z_run, l2, 1.0f);
            }
        }
    };
    auto walk = [&](auto sa_tag) {
        constexpr int SA = decltype(sa_tag)::value, SB = 7 - SA;   // SA < 4 <= SB: strip SB only has k-steps 8 .. 15
        ElSmallFrag<DT, DK, SA> fa;
        ElSmallFrag<DT, DK, SB> fb;
        if (HAS_COV) {
            const float* cov_head = a.cov + (size_t)bhq * D * D;
            el_small_build<DT, DK, SA, 3>(cov_head, n, kg, a.inv_2d, fa);
            __builtin_amdgcn_sched_barrier(0);
            el_small_build<DT, DK, SB, 3>(cov_head, n, kg, a.inv_2d, fb);
        }
        const uint32_t row = n;                         // sub-tile 0 (row & 15 == n & 15 for every sub-tile)
        // per-lane byte offsets inside a tile buffer, sub-tile 0 (sub-tile u is + u * 16 KiB): K fragments; K in the C layout for strip S,
        // quarter q (dims 32 S + 8 q + 4 kg + {0 .. 3}: 8 bytes of chunk 4 S + q)
        // (the swizzle is an XOR on bits 4 .. 7 of the offset, which the row, the lane half and the 256-byte half do not touch: chunk c of this lane's
        //  row is at base ^ (c << 4) -- ONE register per address family instead of a table, the kernel is at the edge of its 256 registers)
        const uint32_t fbase = row * EB_ROWB + ((kg ^ (n & 15)) << 4);      // fragment of k-step ks: (fbase ^ ((ks & 7) << 5)) + (ks >> 3) * 256
        const uint32_t kbase = row * EB_ROWB + ((n & 15) << 4) + kg * 8;    // C-layout piece of chunk c: (kbase ^ ((c & 15) << 4)) + (c >> 4) * 256
        auto fo = [&](int i) { return fbase ^ (uint32_t)(i << 5); };
        auto ko = [&](int c) { return (kbase ^ (uint32_t)((c & 15) << 4)) + (uint32_t)((c >> 4) * 256); };
        auto acc_init = [&](int strip, f32x16& acc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 m4 = *reinterpret_cast<const float4*>(&mus[32 * strip + 8 * q + 4 * kg]);
                acc[4 * q] = m4.x; acc[4 * q + 1] = m4.y; acc[4 * q + 2] = m4.z; acc[4 * q + 3] = m4.w;
            }
        };
        auto rowdot = [&](const uint2 (&kk)[4], const f32x16& acc) -> float {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v0 = fmaf(lo16<DT>(kk[q].x), acc[4 * q + 0], v0);
                v1 = fmaf(hi16<DT>(kk[q].x), acc[4 * q + 1], v1);
                v0 = fmaf(lo16<DT>(kk[q].y), acc[4 * q + 2], v0);
                v1 = fmaf(hi16<DT>(kk[q].y), acc[4 * q + 3], v1);
            }
            return v0 + v1;
        };
        unsigned char* bufc = lds;
        unsigned char* bufn = lds + EB_TILEB;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's part of tile 0 has landed
        __syncthreads();                      // (also publishes mus)
        for (uint32_t t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) request_tile(kbeg + (t + 1) * EL_TILE, (uint32_t)(bufn - lds));   // into the buffer tile t - 1 left at the last barrier
#pragma unroll 1
            for (int u = 0; u < EL_SUBS; ++u) {   // the tile's four sub-tiles (not unrolled)
                const unsigned char* tb = bufc + u * 16384;
                uint4 kf[8];
                uint2 kka[4], kkb[4];
                f32x16 acc;
                acc_init(SA, acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) kka[q] = *reinterpret_cast<const uint2*>(tb + ko(SA * 4 + q));
                if (HAS_COV) {
#pragma unroll
                    for (int ks = 2 * SA; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(tb + fo(ks));
#pragma unroll
                    for (int ks = 2 * SA; ks < 8; ++ks) acc = mma32<DT>(fa.hi[ks - 2 * SA], kf[ks], acc);
#pragma unroll
                    for (int ks = 2 * SA; ks < 8; ++ks) acc = mma32<DT>(fa.lo[ks - 2 * SA], kf[ks], acc);
#pragma unroll
                    for (int ks = 8; ks < 16; ++ks) kf[ks - 8] = *reinterpret_cast<const uint4*>(tb + fo(ks - 8) + 256);
#pragma unroll
                    for (int ks = 8; ks < 16; ++ks) acc = mma32<DT>(fa.hi[ks - 2 * SA], kf[ks - 8], acc);
#pragma unroll
                    for (int ks = 8; ks < 16; ++ks) acc = mma32<DT>(fa.lo[ks - 2 * SA], kf[ks - 8], acc);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) kkb[q] = *reinterpret_cast<const uint2*>(tb + ko(SB * 4 + q));
                float v = rowdot(kka, acc);
                acc_init(SB, acc);
                if (HAS_COV) {
#pragma unroll
                    for (int ks = 2 * SB; ks < 16; ++ks) acc = mma32<DT>(fb.hi[ks - 2 * SB], kf[ks - 8], acc);
#pragma unroll
                    for (int ks = 2 * SB; ks < 16; ++ks) acc = mma32<DT>(fb.lo[ks - 2 * SB], kf[ks - 8], acc);
                }
                v += rowdot(kkb, acc);
                v += __shfl_xor(v, 32);   // the two lane halves hold the two halves of a strip's dims
                if (kg == 0) red[t & 1][wv][row + u * 32] = v;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): tile t + 1 has landed (this wave's part; the barrier covers the others)
            __syncthreads();
            unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
            fold(t);
        }
    };
    switch (wv) {
        case 0: walk(std::integral_constant<int, 0>{}); break;
        case 1: walk(std::integral_constant<int, 1>{}); break;
        case 2: walk(std::integral_constant<int, 2>{}); break;
        default: walk(std::integral_constant<int, 3>{}); break;
    }

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m_run, o), z2 = __shfl_xor(z_run, o);
        softmax_merge(m_run, z_run, m2, z2);
    }
    if (lane == 0 && wv < 2) { wred[2 * wv] = m_run; wred[2 * wv + 1] = z_run; }
    __syncthreads();
    if (threadIdx.x == 0) {
        softmax_merge(m_run, z_run, wred[2], wred[3]);   // (waves 0 and 1 own the keys)
        part_m[(size_t)bhq * nblk + chunk] = m_run;
        part_z[(size_t)bhq * nblk + chunk] = z_run;
    }
}
constexpr size_t EB_LDS_BYTES = 2 * EB_TILEB + 2 * 4 * EL_TILE * 4 + 256 * 4 + 4 * 4;

bool aligned8(int64_t x) { return x % 8 == 0; }

}  // namespace

// ---- host ---------------------------------------------------------------------------------------
bool ea_mfma_qstats_eligible(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype, int64_t Hq, int64_t Sq, int64_t D) {
    // 16-bit rounding of the shifted products: relative covariance error ~ 4.6e-3 / sqrt(Sq) (1 sigma); shorter
    // sequences take the exact fp32 generic kernels
    const bool ok = (dtype == KVP_BF16 || dtype == KVP_F16) && Sq >= 4096 && (uintptr_t)q % 16 == 0 && aligned8(q_sb) && aligned8(q_sh) && aligned8(q_ss);
    (void)q_sh; (void)Hq;
    // Round 6, D = 64: the queries as the projection leaves them -- [B, S, Hq, 64] seen as [B, Hq, S, 64], heads 64 elements apart -- are rows
    // of Hq / 2 "heads" of 128 dimensions; the syrk of such a pair holds the two heads' second moments in its diagonal blocks (twice the matrix
    // work a 64-wide syrk needs, on a kernel that waits for HBM).  D = 96 and the other layouts of D = 64: a head of 128 dimensions whose upper
    // ones are zero (QstatArgs::nch).
    return ok && (D == 128 || D == 64 || D == 96 || (D == 256 && Hq * 6 <= 65535));
}
static bool qstats_pairs(int64_t q_sh, int64_t Hq, int64_t D) { return D == 64 && q_sh == 64 && Hq % 2 == 0; }

size_t ea_mfma_qstats_ws_bytes(int64_t B, int64_t Hq, int64_t Sq, int64_t D) {
    if (D != 128 && D != 64 && D != 96 && D != 256) return 0;
    if (D == 256) Hq *= 6;   // six pairs of quarters per head
    uint32_t nchunk, rows;
    qstats_plan(Sq, B * Hq, nchunk, rows);   // (pairs of heads: half the heads, at most twice the chunks: never more than this)
    size_t need = (size_t)B * Hq * nchunk * (128 * 128 + 256) * 4 + 1024;
    if (D == 64 && Hq % 2 == 0) {
        qstats_plan(Sq, B * (Hq / 2), nchunk, rows);
        need = std::max(need, (size_t)B * (Hq / 2) * nchunk * (128 * 128 + 256) * 4 + 1024);
    }
    return need;
}

int ea_mfma_qstats(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, int dtype, int64_t B, int64_t Hq, int64_t Sq, int64_t D,
                   float* mu, float* cov, void* ws, hipStream_t stream) {
    const uint32_t pair = qstats_pairs(q_sh, Hq, D), quarters = D == 256;
    if (pair) { Hq /= 2; q_sh = 128; }   // pairs of neighbouring 64-dimensional heads as heads of 128 (ea_mfma_qstats_eligible)
    if (quarters) Hq *= 6;               // D = 256: six pairs of quarters per head (QstatArgs::quarters)
    QstatArgs a;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss;
    a.B = (uint32_t)B; a.Hq = (uint32_t)Hq; a.Sq = (uint32_t)Sq;
    qstats_plan(Sq, B * Hq, a.nchunk, a.rows_per_chunk);
    a.nch = (pair || quarters) ? 16u : (uint32_t)(D / 8);
    a.quarters = quarters;
    a.nt = !quarters && (uint64_t)B * Hq * Sq * 256 > (192ull << 20);   // streaming Q loads when Q cannot stay in the memory-side cache anyway (as kvp_gather_kv); quarters: every row is read by three pairs -- let that cache keep it
    const size_t nbh = (size_t)B * Hq;
    a.s2 = static_cast<float*>(ws);
    a.dsum = a.s2 + nbh * a.nchunk * 16384;
    a.m0 = a.dsum + nbh * a.nchunk * 128;
    const dim3 grid((uint32_t)Hq, a.nchunk, (uint32_t)B);
    // ring of three 64-token tiles, three workgroups per CU (measured 211 us at 128k x 32 heads; 4 x 2: 215, 5 x 2: 230, 2 x 4: 213; 6 x 1: 256)
    if (dtype == KVP_BF16) KVP_LAUNCH("ea_qstats_mfma", stream, (ea_qstats_tr_kernel<KVP_BF16, 3, 3><<<grid, EM_THREADS, 0, stream>>>(a)));
    else KVP_LAUNCH("ea_qstats_mfma", stream, (ea_qstats_tr_kernel<KVP_F16, 3, 3><<<grid, EM_THREADS, 0, stream>>>(a)));
    const size_t sm = ((size_t)a.nchunk * 256 + 128) * 4;
    KVP_LAUNCH("ea_qstats_combine", stream, ea_qstats_combine<<<dim3(16, (uint32_t)nbh), 256, sm, stream>>>(a.s2, a.dsum, a.m0, a.Sq, a.nchunk, a.rows_per_chunk, mu, cov, pair, (pair || quarters) ? 128u : (uint32_t)D, quarters));
    KVP_CHECK_LAUNCH("ea_qstats_mfma");
    return KVP_OK;
}

bool ea_mfma_logits_eligible(const EaArgs& a, int dtype) {
    return (dtype == KVP_BF16 || dtype == KVP_F16) && (a.D == 128 || a.D == 64 || a.D == 96 || a.D == 256) && a.Sp >= 64 && (uintptr_t)a.k % 16 == 0 && aligned8(a.k_sb) &&
           aligned8(a.k_sh) && aligned8(a.k_ss) && a.G <= 65535;
}
size_t ea_mfma_logits_scratch_bytes(int64_t, int64_t, int64_t) { return 0; }
// Keys per workgroup: 4096, doubled while the grid exceeds ONE resident round (two 4-wave workgroups per CU).  A workgroup's start-up
// -- every wave packs its strips of the covariance into hi / lo MFMA fragments: ~3000 VALU instructions -- was a third of the
// kernel's VALU work at 32 tiles per workgroup (10 432 instructions per wave against 204 per tile in the loop: round 6), and on this
// SIMD VALU and matrix instructions do not overlap: 128k tokens x 32 heads now run as 512 workgroups of 64 tiles.
static uint32_t ea_mfma_logits_chunk(const EaArgs& a) {
    const uint64_t heads = (uint64_t)a.B * a.Hkv * a.G;
    if (a.D == 256) {   // (2d): one workgroup per CU; 2048 keys each, doubled while the grid exceeds two rounds
        uint32_t chunk = 2048;
        while (chunk < 65536 && (uint64_t)((a.Sp + chunk - 1) / chunk) * heads > 512) chunk *= 2;
        return chunk;
    }
    if (a.D != 128) {   // the small-head kernel (2c).  tools/ea_small_lab.py (ea_score, 32 heads; chunk x workgroup limit): 32k tokens D = 64 43.8 us at
                        // (2048, 768) against 49.8 at (4096, .) and 51-85 with more, shorter workgroups; 128k 120 against 129; D = 96 112 against 115-197
#ifdef KVP_EA_SMALL_LAB   // (lab build of tools/ea_small_lab.py only)
        uint32_t chunk = (uint32_t)std::max(128, kvp_env_int("KVP_EA_SMALL_CHUNK", 2048)) / 128 * 128;
        const uint64_t limit = (uint64_t)std::max(1, kvp_env_int("KVP_EA_SMALL_WGS", 768));
#else
        uint32_t chunk = 2048;
        const uint64_t limit = 768;
#endif
        while (chunk < 65536 && (uint64_t)((a.Sp + chunk - 1) / chunk) * heads > limit) chunk *= 2;
        return chunk;
    }
    uint32_t chunk = EL_CHUNK;
    while (chunk < 65536 && (uint64_t)((a.Sp + chunk - 1) / chunk) * heads > 512) chunk *= 2;
    return chunk;
}
uint32_t ea_mfma_logits_nblk(const EaArgs& a) { const uint32_t c = ea_mfma_logits_chunk(a); return (a.Sp + c - 1) / c; }

int ea_mfma_logits(const EaArgs& a, int dtype, float* logits, uint32_t nblk, float* part_m, float* part_z, void*, hipStream_t stream) {
    const uint32_t ck = ea_mfma_logits_chunk(a);
    KVP_CHECK_ARG(nblk == (a.Sp + ck - 1) / ck, "ea_logits_mfma: nblk %u does not match the chunk plan", nblk);
    const uint64_t units = (uint64_t)nblk * a.B * a.Hkv;
    KVP_CHECK_ARG((units + 7) / 8 * 8 * a.G < ((uint64_t)1 << 31), "ea_logits_mfma: grid too large");
    const dim3 grid((uint32_t)((units + 7) / 8 * 8 * a.G));   // (unit, head-in-group) -> linear id: see the kernel
    if (a.D == 256) {   // (2d)
        const void* fns[4] = {(const void*)ea_logits_mfma_big_kernel<KVP_BF16, true>, (const void*)ea_logits_mfma_big_kernel<KVP_BF16, false>,
                              (const void*)ea_logits_mfma_big_kernel<KVP_F16, true>, (const void*)ea_logits_mfma_big_kernel<KVP_F16, false>};
        const void* fn = fns[(dtype == KVP_BF16 ? 0 : 2) + (a.cov ? 0 : 1)];
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EB_LDS_BYTES) != hipSuccess) {
            kvp_set_error("ea_logits_mfma: cannot raise the dynamic LDS limit to %zu bytes", EB_LDS_BYTES);
            return KVP_EHIP;
        }
        if (dtype == KVP_BF16) {
            if (a.cov) KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_big_kernel<KVP_BF16, true><<<grid, EB_THREADS, EB_LDS_BYTES, stream>>>(a, logits, nblk, ck, part_m, part_z)));
            else KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_big_kernel<KVP_BF16, false><<<grid, EB_THREADS, EB_LDS_BYTES, stream>>>(a, logits, nblk, ck, part_m, part_z)));
        } else {
            if (a.cov) KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_big_kernel<KVP_F16, true><<<grid, EB_THREADS, EB_LDS_BYTES, stream>>>(a, logits, nblk, ck, part_m, part_z)));
            else KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_big_kernel<KVP_F16, false><<<grid, EB_THREADS, EB_LDS_BYTES, stream>>>(a, logits, nblk, ck, part_m, part_z)));
        }
        KVP_CHECK_LAUNCH("ea_logits_mfma");
        return KVP_OK;
    }
    if (a.D != 128) {   // head sizes 64 and 96: (2c)
#define KVP_EL_SMALL(DTV, DKV)                                                                                                                         \
    do {                                                                                                                                               \
        if (a.cov) KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_small_kernel<DTV, DKV, true><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z))); \
        else KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_small_kernel<DTV, DKV, false><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z)));    \
    } while (0)
        if (dtype == KVP_BF16) { if (a.D == 64) KVP_EL_SMALL(KVP_BF16, 4); else KVP_EL_SMALL(KVP_BF16, 6); }
        else { if (a.D == 64) KVP_EL_SMALL(KVP_F16, 4); else KVP_EL_SMALL(KVP_F16, 6); }
#undef KVP_EL_SMALL
        KVP_CHECK_LAUNCH("ea_logits_mfma");
        return KVP_OK;
    }
    if (a.cov) {   // the quadratic form on the doubled upper triangle of the covariance (2b): exact for any matrix, 40 instead of 64 MFMAs per tile and wave
        if (dtype == KVP_BF16) KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_tri_kernel<KVP_BF16><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z)));
        else KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_tri_kernel<KVP_F16><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z)));
    } else if (dtype == KVP_BF16) {   // use_covariance = False: the mean term only
        KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_kernel<KVP_BF16, false><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z)));
    } else {
        KVP_LAUNCH("ea_logits_mfma", stream, (ea_logits_mfma_kernel<KVP_F16, false><<<grid, EM_THREADS, 0, stream>>>(a, logits, nblk, ck, part_m, part_z)));
    }
    KVP_CHECK_LAUNCH("ea_logits_mfma");
    return KVP_OK;
}
