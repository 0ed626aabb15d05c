// SnapKV window-attention passes on the gfx950 matrix cores (bf16 / f16, D = 128, W = 64).
//
// Work decomposition (per launch): workgroup = (tile set, kv-head [x group-block], batch), 8 waves;
// wave w owns HALF a q-head of the GQA group (q-head w/2, window rows 32*(w&1) .. +32), whose Q
// fragments (32 rows x 128 dims = 8 x dwordx4 per lane) stay in registers for the whole launch.
// The kernels are VALU-bound (softmax math: ~3.6 VALU per logit vs 1 MFMA per 512 logits), so the
// design goal is occupancy: <= 128 VGPRs -> 4 waves per SIMD, MFMA results written straight to
// VGPRs (-mllvm -amdgpu-mfma-vgpr-form: no v_accvgpr_read), so that one wave's exp/max/add stream
// runs under another wave's MFMAs.  K streams HBM -> registers -> LDS in 64-key tiles (16 KiB, full 256-B rows, coalesced
// dwordx4), double buffered, ONE barrier per tile; the next tile's global loads are issued
// before the current tile's MFMAs (issue-early / write-late).  All four waves read the same
// K tile from LDS (ds_read_b128, rows XOR-swizzled by (row & 15) << 4 so every 16-lane service
// group of the read hits 16 distinct 16-byte slots -> conflict-free), so K crosses HBM once
// per pass and the LDS read traffic is 4x the HBM rate (40 of 256 B/clk/CU).
//
// v_mfma_f32_32x32x16 with operands swapped between the passes so that each pass's reduction
// axis is lane-local (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)):
//   pass 1  C^T = K_tile . Q^T   -> a lane holds ONE q row and 16 keys per MFMA: running
//           (max, sum-exp) per lane, 2 states per wave-lane, no cross-lane traffic in the loop;
//   pass 2  C   = Q . K_tile^T   -> a lane holds ONE key and 16 q rows per MFMA: the column sum
//           over rows is an in-lane add chain + one xor-32 shuffle; the per-row normalisers
//           a_r = M + log2 Z are 32 registers loaded once.
// Both passes read identical fragments (same registers / same LDS addresses); only the operand
// order changes.  The causal mask exists only in pass 1 and only in the last tiles of a row.
#include "kvp_common.h"
#include "softmax_stats.h"
#include "snapkv_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 256;      // 4 waves: 2 q-heads x 2 halves of the 64-row window
constexpr int MF_WAVES = MF_THREADS / 64;
constexpr int MF_HPB = 2;            // q-heads per workgroup
constexpr int MF_TILE = 64;          // keys per LDS tile
constexpr int MF_CHUNK = 1024;       // keys per workgroup (16 tiles, interleaved across the workgroups of a head)
constexpr int MF_ROWB = 256;         // bytes per key row in HBM (D = 128, 2-byte elements)
// LDS rows are padded by 16 B: a ds_read_b128 service group (16 lanes = 16 distinct keys mod 16, same
// k-chunk) then hits 16 distinct 16-byte slots of the 256-B bank row -> conflict-free, and every fragment
// address is ONE per-lane base plus an immediate (an XOR swizzle needs 8 address registers + VALU).
constexpr int MF_LROW = MF_ROWB + 16;
constexpr int MF_TILEB = MF_TILE * MF_LROW;  // 17408 B

template <int DT> __device__ __forceinline__ f32x16 mma32(const uint4& a, const uint4& b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mma32<KVP_BF16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma32<KVP_F16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// --- K tile staging -------------------------------------------------------------------------------
// thread t moves 4 x 16 B: rows (t >> 4) + 16 i, 16-byte column t & 15  (a wave = 4 full rows = 1 KiB)
struct Stage {
    uint4 v[4];
};
// Loads are UNCONDITIONAL (row index clamped to S-1): straight-line code lets hipcc emit counted
// s_waitcnt vmcnt(N) instead of draining to 0 at every branch join.  Rows past S are duplicates of
// the last row; they are masked (pass 1) or never stored (pass 2).
__device__ __forceinline__ Stage stage_load(const char* __restrict__ kb, int64_t k_ssb, uint32_t key0, uint32_t S) {
    const uint32_t r0 = threadIdx.x >> 4, ch = threadIdx.x & 15;
    Stage st;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t kk = min(key0 + r0 + 16 * i, S - 1);
        st.v[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)kk * k_ssb + ch * 16);
    }
    return st;
}
__device__ __forceinline__ void stage_store(const Stage st, unsigned char* buf) {
    unsigned char* p = buf + (threadIdx.x >> 4) * MF_LROW + (threadIdx.x & 15) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(p + i * 16 * MF_LROW) = st.v[i];
}
// fragment of the 32-key sub-tile `sub` for k-step ks: lane (n = lane & 31, kg = lane >> 5);
// fbase = n * MF_LROW + kg * 16 is computed once per kernel
__device__ __forceinline__ uint4 kfrag(const unsigned char* buf, uint32_t fbase, int sub, int ks) {
    return *reinterpret_cast<const uint4*>(buf + fbase + sub * 32 * MF_LROW + ks * 32);
}

// Q fragments of 32 window rows of one q-head: lane (n, kg) holds row row0+n, dims ks*16+kg*8..+8
__device__ __forceinline__ void load_qfrags(uint4 (&qf)[8], const char* __restrict__ qrow0, int64_t q_swb, uint32_t n, uint32_t kg) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        qf[ks] = *reinterpret_cast<const uint4*>(qrow0 + (int64_t)n * q_swb + (ks * 16 + kg * 8) * 2);
}

// Tile -> workgroup mapping is INTERLEAVED: workgroup `chunk` of the nchunk workgroups of a kv-head takes
// tiles chunk, chunk + nchunk, ...: the workgroups running concurrently read one contiguous, advancing
// region of K (nchunk x 16 KiB) instead of nchunk streams 256 KiB apart.
struct TileWalk {
    uint32_t ntiles, tstride, kbeg, klast;
    __device__ TileWalk(uint32_t chunk, uint32_t nchunk, uint32_t nkeys) {
        const uint32_t total = (nkeys + MF_TILE - 1) / MF_TILE;
        ntiles = chunk < total ? (total - chunk + nchunk - 1) / nchunk : 0;
        tstride = nchunk * MF_TILE;
        kbeg = chunk * MF_TILE;
        klast = kbeg + (ntiles ? ntiles - 1 : 0) * tstride;  // prefetches past the end re-read the last tile (L2 hits, never stored)
    }
};

// =================================================================================================
// Software pipeline shared by both passes.
//
// A wave's work per 32-key sub-tile is 8 dependent MFMAs (32 cycles apart on the matrix pipe) and
// ~62 VALU instructions of softmax math on the 16 logits the MFMAs produce (exp = 8 cycles, the rest
// 4).  Issued back to back (MFMA chain, then softmax) the two pipes never overlap inside a wave, and
// the per-tile barrier puts all waves of a workgroup in the same phase (measured: VALU busy 47 %,
// MFMA busy 32 %, sum ~ kernel time).  So every wave runs a two-stage pipeline instead:
//
//     step k:   MFMAs of sub-tile k+1  ||  softmax of sub-tile k     (one branch-free basic block:
//               MFMA, 2 logits of softmax, MFMA, 2 logits, ...: ~32 VALU cycles per MFMA gap)
//
// with the two accumulators alternating.  The step that crosses a tile boundary sits right after
// the tile's barrier (it needs the next LDS buffer but only REGISTERS of the previous tile), so two
// LDS buffers still suffice.  Tiles that need the causal mask / sequence tail (the last tiles of a
// head) are not pipelined: they run the simple path after the pipeline has drained.
// =================================================================================================
#define KVP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SGB_VALU = 0x2, SGB_MFMA = 0x8, SGB_DSRD = 0x100;

// ---- pass 1 pieces --------------------------------------------------------------------------------
struct P1State {
    float m, z;  // raw-logit running max / sum-exp (relative to m) of this lane's window row
};

// softmax-update of 16 finished logits `ap` interleaved with the 8 MFMAs producing `ac` from (buf, sub)
template <int DT>
__device__ __forceinline__ void p1_step(const unsigned char* buf, int sub, const uint4 (&qf)[8], uint32_t fbase,
                                        const f32x16& ap, f32x16& ac, P1State& st, float c) {
    uint4 kf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, fbase, sub, ks);
    float tm = ap[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ap[r]);
    const float mn = fmaxf(st.m, tm);
    const float off = -mn * c;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ac[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        ac = mma32<DT>(kf[ks], qf[ks], ac);  // C[key][q row]
        s0 += fast_exp2(fmaf(ap[2 * ks], c, off));
        s1 += fast_exp2(fmaf(ap[2 * ks + 1], c, off));
    }
    st.z = st.z * fast_exp2(fmaf(st.m, c, off)) + (s0 + s1);
    st.m = mn;
    // schedule: 4 fragment reads first, the max chain while they land, then per MFMA gap: 1 MFMA,
    // 1 more fragment read (keeps only ~5 fragments live), 6 VALU (2 logits of fma/exp/add)
    KVP_SGB(SGB_DSRD, 4);
    KVP_SGB(SGB_VALU, 12);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        KVP_SGB(SGB_MFMA, 1);
        if (ks < 4) KVP_SGB(SGB_DSRD, 1);
        KVP_SGB(SGB_VALU, 6);
    }
}
template <int DT>
__device__ __forceinline__ void mma_only(const unsigned char* buf, int sub, const uint4 (&qf)[8], uint32_t fbase,
                                         f32x16& ac, bool q_is_a) {
    uint4 kf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, fbase, sub, ks);
#pragma unroll
    for (int i = 0; i < 16; ++i) ac[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ac = q_is_a ? mma32<DT>(qf[ks], kf[ks], ac) : mma32<DT>(kf[ks], qf[ks], ac);
}
__device__ __forceinline__ void p1_softmax_only(const f32x16& ap, P1State& st, float c) {
    float tm = ap[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ap[r]);
    const float mn = fmaxf(st.m, tm);
    const float off = -mn * c;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s0 += fast_exp2(fmaf(ap[r], c, off));
        s1 += fast_exp2(fmaf(ap[r + 1], c, off));
    }
    st.z = st.z * fast_exp2(fmaf(st.m, c, off)) + (s0 + s1);
    st.m = mn;
}

// =================================================================================================
// pass 1: per (row, chunk) partial max / sum-exp (log2 units)
// =================================================================================================
template <int DT>
__global__ __launch_bounds__(MF_THREADS, 3) void snapkv_p1_mfma(SnapArgs a, uint32_t ngb, uint32_t nchunk,
                                                                float* __restrict__ part_m, float* __restrict__ part_z) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MF_TILEB];
    // blockIdx.x enumerates (chunk, group-block) so that the ngb workgroups that read the SAME K tiles sit on
    // the same XCD (workgroup i runs on XCD i % 8; slots i/8 of one XCD are dispatched back to back) and
    // share its L2: K crosses HBM once per pass although it is staged into LDS ngb times.
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t gb = slot % ngb, chunk = (slot / ngb) * 8 + xcd;
    const uint32_t h = blockIdx.y, b = blockIdx.z;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t fbase = n * MF_LROW + kg * 16;
    const uint32_t rg = gb * MF_HPB + (wv >> 1);  // q-head inside the GQA group
    const uint32_t row0 = (wv & 1) * 32;     // first of this wave's 32 window rows
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);

    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const int64_t k_ssb = a.k_ss * 2;
    uint4 qf[8];
    load_qfrags(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)row0 * a.q_sw) * 2,
                a.q_sw * 2, n, kg);

    const TileWalk tw(chunk, nchunk, a.S);
    // tiles entirely below the causal-mask region (and the sequence end) take the pipelined path
    uint32_t ntf = 0;
    if (a.S - a.W >= (uint32_t)(MF_TILE - 1)) {
        const uint32_t last_fast_key0 = a.S - a.W - (MF_TILE - 1);  // key0 + 63 <= S - W
        if (tw.kbeg <= last_fast_key0) ntf = min(tw.ntiles, (last_fast_key0 - tw.kbeg) / tw.tstride + 1);
    }
    P1State st{KVP_NEG_INF, 0.f};
    const float c = a.c;
    const uint32_t w = row0 + n;  // window row: token S-W+w sees keys <= S-W+w
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + MF_TILEB;

    if (ntf > 0) {
        // K streams HBM -> registers -> LDS with two tiles in flight behind the one being computed
        // (issue-early, write-late; counted vmcnt, loads unconditional).
        Stage st1, st2;  // st1: tile t+1 (in flight / landed), st2: tile t+2 (just issued)
        f32x16 accA, accB;
        unsigned char* bufc = lds;             // tile t
        unsigned char* bufn = lds + MF_TILEB;  // tile t+1
        stage_store(stage_load(kb, k_ssb, tw.kbeg, a.S), bufc);
        st1 = stage_load(kb, k_ssb, min(tw.kbeg + tw.tstride, tw.klast), a.S);
        __syncthreads();
        if (active) mma_only<DT>(bufc, 0, qf, fbase, accA, false);  // pipeline fill: sub-tile 0 of tile 0
        for (uint32_t t = 0; t < ntf; ++t) {
            const uint32_t key0 = tw.kbeg + t * tw.tstride;
            st2 = stage_load(kb, k_ssb, min(key0 + 2 * tw.tstride, tw.klast), a.S);
            __builtin_amdgcn_sched_barrier(0);  // issue-early
            if (active) p1_step<DT>(bufc, 1, qf, fbase, accA, accB, st, c);  // MFMA sub 1 of t || softmax sub 0 of t
            __builtin_amdgcn_sched_barrier(0);  // write-late: the LDS store of tile t+1 stays behind the MFMAs
            stage_store(st1, bufn);
            __syncthreads();
            if (t + 1 >= ntf) {
                if (active) p1_softmax_only(accB, st, c);  // drain
                break;
            }
            if (active) p1_step<DT>(bufn, 0, qf, fbase, accB, accA, st, c);  // MFMA sub 0 of t+1 || softmax sub 1 of t
            unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
            st1 = st2;
        }
    }

    // ---- masked / tail tiles (at most the last few of a head): simple, synchronous path ---------------
    for (uint32_t t = ntf; t < tw.ntiles; ++t) {
        const uint32_t key0 = tw.kbeg + t * tw.tstride;
        __syncthreads();
        stage_store(stage_load(kb, k_ssb, key0, a.S), buf0);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 acc;
                mma_only<DT>(buf0, sub, qf, fbase, acc, false);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t kk = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (kk >= a.S || kk > a.S - a.W + w) acc[r] = KVP_NEG_INF;
                }
                float tm = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, acc[r]);
                const float mn = fmaxf(st.m, tm);
                if (mn != KVP_NEG_INF) {
                    const float off = -mn * c;
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += fast_exp2(fmaf(acc[r], c, off));
                    st.z = st.z * fast_exp2(fmaf(st.m, c, off)) + s;
                    st.m = mn;
                }
            }
        }
    }

    if (active) {
        float mm = st.m == KVP_NEG_INF ? KVP_NEG_INF : st.m * c, zz = st.z;
        const float m2 = __shfl_xor(mm, 32), z2 = __shfl_xor(zz, 32);
        softmax_merge(mm, zz, m2, z2);
        if (kg == 0) {
            const size_t o = ((size_t)(b * a.Hq + hq) * a.W + w) * nchunk + chunk;
            part_m[o] = mm;
            part_z[o] = zz;
        }
    }
}

// =================================================================================================
// pass 2: colsum[b,h,key] = sum over the group's G*64 rows of 2^(L2 - a_row), keys < S - W
// (no mask needed: every window row sees every key < S - W; keys >= S - W are simply not stored)
// =================================================================================================
// column sums of P = 2^(L2 - a_row) over the 16 finished logits `ap` (32 keys x this wave's 32 q rows)
// -> red[key], interleaved with the 8 MFMAs producing `ac` from (buf, sub)
template <int DT>
__device__ __forceinline__ void p2_step(const unsigned char* buf, int sub, const uint4 (&qf)[8], uint32_t fbase, uint32_t n,
                                        uint32_t kg, const f32x16& ap, f32x16& ac, const float (&ar)[16], float c, float* red_dst) {
    uint4 kf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, fbase, sub, ks);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ac[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        ac = mma32<DT>(qf[ks], kf[ks], ac);  // C[q row][key]
        s0 += fast_exp2(fmaf(ap[2 * ks], c, ar[2 * ks]));
        s1 += fast_exp2(fmaf(ap[2 * ks + 1], c, ar[2 * ks + 1]));
    }
    float s = s0 + s1;
    s += __shfl_xor(s, 32);
    if (kg == 0) red_dst[n] = s;
    KVP_SGB(SGB_DSRD, 4);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        KVP_SGB(SGB_MFMA, 1);
        if (ks < 4) KVP_SGB(SGB_DSRD, 1);
        KVP_SGB(SGB_VALU, 6);
    }
}
__device__ __forceinline__ void p2_colsum_only(const f32x16& ap, const float (&ar)[16], float c, uint32_t n, uint32_t kg,
                                               float* red_dst) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        s0 += fast_exp2(fmaf(ap[r], c, ar[r]));
        s1 += fast_exp2(fmaf(ap[r + 1], c, ar[r + 1]));
    }
    float s = s0 + s1;
    s += __shfl_xor(s, 32);
    if (kg == 0) red_dst[n] = s;
}

template <int DT>
__global__ __launch_bounds__(MF_THREADS, 3) void snapkv_p2_mfma(SnapArgs a, uint32_t ngb, uint32_t nchunk,
                                                                const float* __restrict__ rowstat, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MF_TILEB];
    __shared__ float red[3][MF_WAVES][MF_TILE];  // by tile % 3: a tile is flushed one barrier after it completes
    // blockIdx.x enumerates (chunk, group-block) so that the ngb workgroups that read the SAME K tiles sit on
    // the same XCD (workgroup i runs on XCD i % 8; slots i/8 of one XCD are dispatched back to back) and
    // share its L2: K crosses HBM once per pass although it is staged into LDS ngb times.
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t gb = slot % ngb, chunk = (slot / ngb) * 8 + xcd;
    const uint32_t h = blockIdx.y, b = blockIdx.z;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t fbase = n * MF_LROW + kg * 16;
    const uint32_t rg = gb * MF_HPB + (wv >> 1);
    const uint32_t row0 = (wv & 1) * 32;
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);
    const uint32_t Sm = a.S - a.W;

    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const int64_t k_ssb = a.k_ss * 2;
    uint4 qf[8];
    load_qfrags(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)row0 * a.q_sw) * 2,
                a.q_sw * 2, n, kg);
    // normalisers of the 16 q rows this lane sees in the C layout: row = row0 + (r&3) + 8*(r>>2) + 4*kg
    float ar[16];
    const float* ars = rowstat + (size_t)(b * a.Hq + hq) * a.W + row0;
#pragma unroll
    for (int r = 0; r < 16; ++r) ar[r] = -ars[(r & 3) + 8 * (r >> 2) + 4 * kg];

    const TileWalk tw(chunk, nchunk, Sm);
    const float c = a.c;
    float* cs = colsum + ((size_t)gb * a.B * a.Hkv + (size_t)(b * a.Hkv + h)) * Sm;  // plane gb: summed by the pool kernel
    const uint32_t nact = 2 * min((uint32_t)MF_HPB, a.G - gb * MF_HPB);  // active waves in this workgroup
    if (tw.ntiles == 0) return;

    // threads 0..63 add the active waves' partials of one finished tile and store 64 column sums
    auto flush = [&](uint32_t key0, int par) {
        if (threadIdx.x < MF_TILE) {
            const uint32_t kk = key0 + threadIdx.x;
            if (kk < Sm) {
                float s = red[par][0][threadIdx.x];
                for (uint32_t w = 1; w < nact; ++w) s += red[par][w][threadIdx.x];
                cs[kk] = s;
            }
        }
    };

    // Pipeline as in pass 1.  The column sums of tile t are complete after the step that follows tile t's
    // barrier (sub-tile 1 rides with the next tile's first MFMAs), so tile t is flushed after the NEXT
    // barrier.  One K tile is in flight behind the one being computed (register budget: 168 for 3 waves/SIMD).
    Stage st1;
    f32x16 accA, accB;
    unsigned char* bufc = lds;
    unsigned char* bufn = lds + MF_TILEB;
    const uint32_t nt = tw.ntiles;
    stage_store(stage_load(kb, k_ssb, tw.kbeg, a.S), bufc);
    __syncthreads();
    if (active) mma_only<DT>(bufc, 0, qf, fbase, accA, true);
    uint32_t par = 0;  // t % 3
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t key0 = tw.kbeg + t * tw.tstride;
        st1 = stage_load(kb, k_ssb, min(key0 + tw.tstride, tw.klast), a.S);
        __builtin_amdgcn_sched_barrier(0);
        if (active) p2_step<DT>(bufc, 1, qf, fbase, n, kg, accA, accB, ar, c, &red[par][wv][0]);   // colsum sub 0 of t
        __builtin_amdgcn_sched_barrier(0);
        stage_store(st1, bufn);
        __syncthreads();  // tile t+1 visible; tile t-1's column sums complete
        if (t > 0) flush(key0 - tw.tstride, par == 0 ? 2 : par - 1);
        if (t + 1 >= nt) {
            if (active) p2_colsum_only(accB, ar, c, n, kg, &red[par][wv][32]);
            __syncthreads();
            flush(key0, par);
            break;
        }
        if (active) p2_step<DT>(bufn, 0, qf, fbase, n, kg, accB, accA, ar, c, &red[par][wv][32]);  // colsum sub 1 of t
        unsigned char* tmp = bufc; bufc = bufn; bufn = tmp;
        par = par == 2 ? 0 : par + 1;
    }
}

}  // namespace

bool snapkv_mfma_eligible(const SnapArgs& a, int dtype) {
    if (dtype != KVP_BF16 && dtype != KVP_F16) return false;
    if (a.D != 128 || a.W != 64 || a.G > 8) return false;
    auto al8 = [](int64_t x) { return x % 8 == 0; };
    if (((uintptr_t)a.q % 16) || ((uintptr_t)a.k % 16)) return false;
    return al8(a.q_sb) && al8(a.q_sh) && al8(a.q_sw) && al8(a.k_sb) && al8(a.k_sh) && al8(a.k_ss);
}

static uint32_t round8(uint32_t x) { return (x + 7) / 8 * 8; }
// workgroups per (kv-head, group-block): a multiple of 8 (the XCD-paired mapping enumerates chunk = 8*j + xcd)
uint32_t snapkv_mfma_nchunk(const SnapArgs& a) { return round8((a.S + MF_CHUNK - 1) / MF_CHUNK); }
uint32_t snapkv_mfma_nplanes(const SnapArgs& a) { return (a.G + MF_HPB - 1) / MF_HPB; }

int snapkv_mfma_p1(const SnapArgs& a, int dtype, uint32_t nchunk, float* part_m, float* part_z, hipStream_t stream) {
    const uint32_t ngb = snapkv_mfma_nplanes(a);
    const dim3 grid(nchunk * ngb, a.Hkv, a.B);
    if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p1_mfma", stream, snapkv_p1_mfma<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z));
    else KVP_LAUNCH("snapkv_p1_mfma", stream, snapkv_p1_mfma<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z));
    KVP_CHECK_LAUNCH("snapkv_p1_mfma");
    return KVP_OK;
}

// colsum: [nplanes][B][Hkv][S-W]; plane gb holds the column sums of q-heads gb*2, gb*2+1 of every group
int snapkv_mfma_p2(const SnapArgs& a, int dtype, const float* rowstat, float* colsum, hipStream_t stream) {
    const uint32_t ngb = snapkv_mfma_nplanes(a);
    const uint32_t Sm = a.S - a.W;
    const uint32_t nchunk = round8((Sm + MF_CHUNK - 1) / MF_CHUNK);
    const dim3 grid(nchunk * ngb, a.Hkv, a.B);
    if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p2_mfma", stream, snapkv_p2_mfma<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, rowstat, colsum));
    else KVP_LAUNCH("snapkv_p2_mfma", stream, snapkv_p2_mfma<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, rowstat, colsum));
    KVP_CHECK_LAUNCH("snapkv_p2_mfma");
    return KVP_OK;
}
