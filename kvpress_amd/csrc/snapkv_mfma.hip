// SnapKV window-attention passes on the gfx950 matrix cores (bf16 / f16; D = 64 / 96 / 128 / 256; any window size: blocks of 64 padded rows).
// (The text below describes the D = 128 geometry the kernels were designed on; KGeo<DK> holds the numbers of the other head size, and
// "windows of any size" further down how a window that is not 64 rows maps onto the 64-row blocks.  The hand-scheduled loops are D = 128 with
// G % 4 == 0; everything else runs snapkv_p1_mfma / snapkv_p2_mfma, templated on the head size.)
//
// Work decomposition (per launch): workgroup = (tile set, kv-head [x group-block], batch), 8 waves, ONE workgroup
// per CU; wave w owns half a q-head of the GQA group (q-head w/2, window rows 32*(w&1) .. +32), whose Q fragments
// (32 rows x 128 dims = 8 x dwordx4 per lane) stay in registers for the whole launch.
//
// K stream: 128-key tiles (32 KiB) travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: every lane names a
// 16-byte global chunk, the wave's 64 chunks land contiguously at an LDS address taken from M0) into a ring of
// three buffers; tile t+2 is requested while tile t is computed, ONE request per thread per 32-key sub-tile step,
// so the requests trickle through the address path instead of arriving as a burst, no VGPRs hold K in transit and
// there is no ds_write phase.  (The earlier global -> VGPR -> ds_write staging cost as much as all the math:
// tools/ubench_steps.hip, 894 vs 530 ns per sub-tile and wave.)  vmcnt is managed by hand (the requests are inline
// asm: with the builtin hipcc makes every ds_read wait for vmcnt(0)): at the end of tile t `s_waitcnt vmcnt(4)`
// leaves only tile t+2's four requests pending, then one barrier publishes tile t+1.
// Rows are XOR-swizzled in LDS (16-byte slot p of row r holds chunk p ^ (r & 15)); the swizzle is applied on the
// GLOBAL side of the DMA (lane (r, p) fetches chunk p ^ (r & 15)), which keeps the 256-byte row fully coalesced,
// and makes every ds_read_b128 of the fragment reads conflict-free.
//
// v_mfma_f32_32x32x16 with operands swapped between the passes so that each pass's reduction
// axis is lane-local (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)):
//   pass 1  C^T = K_tile . Q^T   -> a lane holds ONE q row and 16 keys per MFMA: running
//           (max, sum-exp) per lane, 2 states per wave-lane, no cross-lane traffic in the loop;
//   pass 2  C   = Q . K_tile^T   -> a lane holds ONE key and 16 q rows per MFMA: the column sum
//           over rows is an in-lane add chain + one xor-32 shuffle; the per-row normalisers
//           a_r = M + log2 Z are 16 registers loaded once.
// Inside a tile the MFMA chain of sub-tile s is interleaved with the softmax update of sub-tile s-1 (double-buffered
// accumulators and fragment registers).  On a gfx950 SIMD MFMAs and VALU instructions do not overlap
// (tools/ubench_valu.hip, ubench_overlap.hip), so the per-sub-tile cost is ~272 cycles of MFMA + ~290 of VALU.
// The causal mask exists only in pass 1 and only in the last tiles of a row.
#include "kvp_common.h"
#include "softmax_stats.h"
#include "snapkv_internal.h"
#include "snapkv_asm.inc"

#include <vector>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 512;      // 8 waves
constexpr int MF_WAVES = MF_THREADS / 64;
constexpr int MF_TILE = 128;         // keys per LDS tile
constexpr int MF_SUBS = MF_TILE / 32;  // 32-key MFMA sub-tiles per tile == DMA requests per thread per tile
constexpr int MF_NBUF = 3;           // LDS ring: computing t, landed/landing t+1, landing t+2
constexpr int MF_CHUNK = 128;        // minimum keys per workgroup: one tile (measured: S=4k p1 22->13 us, p2 22->9 us vs 1024)
constexpr int MF_ROWB = 256;         // bytes per key row (D = 128, 2-byte elements)
constexpr int MF_TILEB = MF_TILE * MF_ROWB;
static_assert(MF_TILEB / 16 / MF_THREADS == MF_SUBS, "one DMA request per thread per sub-tile step");
// Geometry of a K tile for a head of DK k-steps of 16 elements (round 6: D = 128 -> DK = 8, D = 64 -> DK = 4; D = 96 -> DK = 6 keeps the
// 256-byte LDS rows of DK = 8 and simply never requests the four chunks a 192-byte key row does not have; D = 256 -> DK = 16: 64 KiB
// tiles, so a ring of TWO buffers (one tile of prefetch) and single-buffered fragment registers; 2-byte elements).
template <int DK> struct KGeo {
    static constexpr int ROWB = DK == 6 ? 256 : DK * 32;   // bytes per key row IN LDS
    static constexpr int NCH = DK * 2;              // 16-byte chunks a key row really has
    static constexpr int CPR = ROWB / 16;           // 16-byte slots per LDS row
    static constexpr int RPW = 1024 / ROWB;         // rows one wave's request moves (64 lanes x 16 bytes)
    static constexpr int RPR = (MF_THREADS / 64) * RPW;   // rows one request of the workgroup moves
    static constexpr int NREQ = MF_TILE / RPR;      // requests per thread and tile
    static constexpr int TILEB = MF_TILE * ROWB;
    static constexpr int NBUF = DK <= 8 ? 3 : 2;    // LDS ring: NBUF - 1 tiles of prefetch (three 64 KiB tiles would not fit the CU's 160 KiB)
    static constexpr int NKF = DK <= 8 ? 2 : 1;     // fragment register sets: the next sub-tile's reads overlap this one's chain where they fit
    static_assert(RPR % 16 == 0 && MF_TILE % RPR == 0, "the swizzle of a row (period: 16 rows) depends on its index inside a request only");
    // XOR swizzle of a tile row's 16-byte slots: 16 consecutive rows of a fragment read must hit 16 distinct slots of the 256-byte
    // bank line -- rows of 256 bytes: the row's low four bits; rows of 128 bytes (two per bank line): bits 1 .. 3
    static __device__ __forceinline__ uint32_t sw(uint32_t row) { return DK == 4 ? ((row >> 1) & 7u) : (row & 15u); }
    static __device__ __forceinline__ int ring_next(int b) { return b + 1 == NBUF ? 0 : b + 1; }
    static __device__ __forceinline__ int ring_prev(int b) { return b == 0 ? NBUF - 1 : b - 1; }
};

template <int DT> __device__ __forceinline__ f32x16 mma32(const uint4& a, const uint4& b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mma32<KVP_BF16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma32<KVP_F16>(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// --- K tile stream (LDS-DMA) ----------------------------------------------------------------------
// Request i (0..MF_SUBS-1) of a tile: wave w moves rows 32 i + 4 w .. +4 (1 KiB); lane (lr = lane >> 4, p = lane & 15)
// fetches chunk p ^ (row & 15) of row 32 i + 4 w + lr, which the DMA drops into LDS slot p of that row.
// Row indices are clamped to S-1 (unconditional requests): rows past S are copies of the last row; they are masked
// (pass 1) or never stored (pass 2).
// (written out for D = 128: DK = 8, 4 rows per wave and request, 4 requests per tile; KGeo<DK> holds the numbers of the other head sizes)
template <int DK>
struct KStreamT {
    using Geo = KGeo<DK>;
    const char* kb;     // K[b, h, 0, 0]
    int64_t k_ssb;      // bytes between keys
    uint32_t S;
    uint32_t lrow;      // RPW w + lr
    uint32_t choff;     // byte offset of this lane's chunk inside a row: (p ^ sw(lrow)) << 4  (a request's first row is a multiple of the swizzle period)
    uint32_t ldsrow;    // RPW w: first row of this wave's 1 KiB block inside a request
    bool has_chunk;     // (D = 96: slots whose chunk lies beyond the 192-byte row stay empty -- those lanes request nothing)
    __device__ KStreamT(const char* kb_, int64_t k_ssb_, uint32_t S_) : kb(kb_), k_ssb(k_ssb_), S(S_) {
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        lrow = wv * Geo::RPW + lane / Geo::CPR;
        const uint32_t chunk = (lane % Geo::CPR) ^ Geo::sw(lrow);
        choff = chunk << 4;
        has_chunk = chunk < (uint32_t)Geo::NCH;
        ldsrow = wv * Geo::RPW;
    }
    __device__ __forceinline__ void request(unsigned char* buf, uint32_t key0, int i) const {
        const uint32_t kk = min(key0 + i * Geo::RPR + lrow, S - 1);
        const char* g = kb + (int64_t)kk * k_ssb + choff;
        const uint32_t la = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(buf + (i * Geo::RPR + ldsrow) * Geo::ROWB));
        // M0 is a reserved register that hipcc never keeps values in (nothing else in these kernels uses it)
        if (Geo::NCH == Geo::CPR) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(g) : "memory");
        } else {
            asm volatile("s_mov_b32 m0, %0" ::"s"(la) : "memory");   // (outside the divergent region: M0 is per wave)
            if (has_chunk) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(g) : "memory");
        }
    }
    __device__ __forceinline__ void request_tile(unsigned char* buf, uint32_t key0) const {
#pragma unroll
        for (int i = 0; i < Geo::NREQ; ++i) request(buf, key0, i);
    }
    // the requests of a tile that are issued beside sub-tile `sub` of the tile being computed (they trickle instead of arriving as a burst)
    __device__ __forceinline__ void request_part(unsigned char* buf, uint32_t key0, int sub) const {
#pragma unroll
        for (int i = (sub * Geo::NREQ) / MF_SUBS; i < ((sub + 1) * Geo::NREQ) / MF_SUBS; ++i) request(buf, key0, i);
    }
};
typedef KStreamT<8> KStream;
// s_waitcnt through the builtin (simm16: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8): unlike an asm string, hipcc's own
// wait-count bookkeeping sees it, so it stops re-waiting for the Q-fragment loads inside the tile loop.
// only the newest tile's MF_SUBS requests of this wave may still be in flight
__device__ __forceinline__ void wait_tile_landed() { __builtin_amdgcn_s_waitcnt(0x0F70 | MF_SUBS); }
// (only the requests of tiles newer than t + 1 may still be in flight: (NBUF - 2) tiles' worth)
template <int DK> __device__ __forceinline__ void wait_tile_landed_t() { __builtin_amdgcn_s_waitcnt(0x0F70 | ((KGeo<DK>::NBUF - 2) * KGeo<DK>::NREQ)); }
__device__ __forceinline__ void wait_all_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// fragment of the 32-key sub-tile `sub` for k-step ks: lane (n = lane & 31, kg = lane >> 5)
template <int DK>
__device__ __forceinline__ uint4 kfrag_t(const unsigned char* buf, uint32_t sub, uint32_t ks, uint32_t n, uint32_t kg) {
    const uint32_t row = sub * 32 + n;
    return *reinterpret_cast<const uint4*>(buf + row * KGeo<DK>::ROWB + (((ks * 2 + kg) ^ KGeo<DK>::sw(row)) << 4));
}
__device__ __forceinline__ uint4 kfrag(const unsigned char* buf, uint32_t sub, uint32_t ks, uint32_t n, uint32_t kg) { return kfrag_t<8>(buf, sub, ks, n, kg); }

// Q fragments of 32 window rows of one q-head: lane (n, kg) holds row row0+n, dims ks*16+kg*8..+8
// (qrow: this lane's query row, see mf_qrow)
template <int DK>
__device__ __forceinline__ void load_qfrags(uint4 (&qf)[DK], const char* __restrict__ qrow, uint32_t kg) {
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qrow + (ks * 16 + kg * 8) * 2);
}

// Tile -> workgroup mapping is INTERLEAVED: workgroup `chunk` of the nchunk workgroups of a kv-head takes
// tiles chunk, chunk + nchunk, ...: the workgroups running concurrently read one contiguous, advancing
// region of K (nchunk x 32 KiB) instead of nchunk streams far apart.
struct TileWalk {
    uint32_t ntiles, tstride, kbeg, klast;
    __device__ TileWalk(uint32_t chunk, uint32_t nchunk, uint32_t nkeys) {
        const uint32_t total = (nkeys + MF_TILE - 1) / MF_TILE;
        ntiles = chunk < total ? (total - chunk + nchunk - 1) / nchunk : 0;
        tstride = nchunk * MF_TILE;
        kbeg = chunk * MF_TILE;
        klast = kbeg + (ntiles ? ntiles - 1 : 0) * tstride;  // requests past the end re-fetch the last tile (L2 hits, never read)
    }
    // a contiguous range of tiles [first, first + n) (pass 2 with shares from pass 1's clock: snapkv_internal.h)
    __device__ TileWalk(uint32_t first, uint32_t n, int) {
        ntiles = n;
        tstride = MF_TILE;
        kbeg = first * MF_TILE;
        klast = kbeg + (ntiles ? ntiles - 1 : 0) * tstride;
    }
    __device__ __forceinline__ uint32_t key0(uint32_t t) const { return min(kbeg + t * tstride, klast); }
};
__device__ __forceinline__ int ring_next(int b) { return b + 1 == MF_NBUF ? 0 : b + 1; }
__device__ __forceinline__ int ring_prev(int b) { return b == 0 ? MF_NBUF - 1 : b - 1; }

// ---- windows of any size (SnapArgs: Wp, rblk) ------------------------------------------------------------------------------------
// The kernels work on ONE block of 64 padded window rows per launch.  Row r (0 .. 63) of block a.rblk is padded row p = 64 rblk + r:
__device__ __forceinline__ uint32_t mf_weff(const SnapArgs& a) { return a.Wp - 64u * a.rblk; }         // row r may attend keys <= S - weff + r
__device__ __forceinline__ uint32_t mf_prow(const SnapArgs& a, uint32_t r) { return 64u * a.rblk + r; }  // index into [.., Wp] statistics
// the query row it reads: real row p - (Wp - W); a padding row reads real row 0 (valid memory; its results are never used)
__device__ __forceinline__ uint32_t mf_qrow(const SnapArgs& a, uint32_t r) {
    const uint32_t p = mf_prow(a, r), pad = a.Wp - a.W;
    return p >= pad ? p - pad : 0u;
}

// =================================================================================================
// pass 1: per (row, chunk) partial max / sum-exp (log2 units)
// =================================================================================================
template <int DT, int DK>
__global__ __launch_bounds__(MF_THREADS, 2) void snapkv_p1_mfma(SnapArgs a, uint32_t ngb, uint32_t nchunk,
                                                                float* __restrict__ part_m, float* __restrict__ part_z) {
    using Geo = KGeo<DK>;
    constexpr int TILEB = Geo::TILEB;
    constexpr int NBUF = Geo::NBUF, NKF = Geo::NKF;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * TILEB];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t rg = gb * 4 + (wv >> 1);  // q-head inside the GQA group
    const uint32_t row0 = (wv & 1) * 32;     // first of this wave's 32 window rows
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);

    const KStreamT<DK> ks_(static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2, a.k_ss * 2, a.S);
    const TileWalk tw(chunk, nchunk, a.S);
    // the first K tiles are requested BEFORE the Q fragments: one memory round trip for all
    if (tw.ntiles > 0) {
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i) ks_.request_tile(lds + i * TILEB, tw.key0(i));
    }
    uint4 qf[DK];
    load_qfrags<DK>(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)mf_qrow(a, row0 + n) * a.q_sw) * 2, kg);
    wait_all_landed();  // K tiles 0 and 1 and the Q fragments are in; from here on vmcnt only counts the K stream

    float m = KVP_NEG_INF, z = 0.f;  // raw-logit running max / sum-exp of window row row0 + n over this lane's keys
    const float c = a.c;
    const uint32_t w = row0 + n;     // row of this launch's 64-row block: it sees keys <= S - weff + w
    // row r of this block may attend keys <= mlim + r.  SIGNED: with a padded window S - Wp can be negative (S = 253, W = 200: Wp = 256);
    // the unsigned form classified such tiles as unmasked (found by tools/snapkv_shape_fuzz.py)
    const int32_t mlim = (int32_t)a.S - (int32_t)mf_weff(a);

    // softmax-update of the 16 finished logits of one sub-tile (lane's q row: running max m, sum-exp z);
    // MASKED: causal mask / sequence tail handled per element (only the last tiles of a head)
    auto softmax16 = [&](f32x16& acc, uint32_t key0, int sub, bool masked) {
        if (masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t kk = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (kk >= a.S || (int32_t)kk > mlim + (int32_t)w) acc[r] = KVP_NEG_INF;
            }
        }
        float tm = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tm = fmaxf(tm, acc[r]);
        const float mn = fmaxf(m, tm);
        if (!masked || mn != KVP_NEG_INF) {
            const float off = -mn * c;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                s0 += fast_exp2(fmaf(acc[r], c, off));
                s1 += fast_exp2(fmaf(acc[r + 1], c, off));
            }
            z = z * fast_exp2(fmaf(m, c, off)) + (s0 + s1);
            m = mn;
        }
    };

    // One tile = MF_SUBS sub-tiles of 32 keys; per sub-tile step: one DMA request for the tile two ahead, the LDS
    // fragment reads of the next sub-tile, and the MFMA chain of this sub-tile interleaved (1 MFMA : 6 VALU) with
    // the softmax of the previous one.  Straight-line code (no masks: every tile but the last ones of a head).
    auto compute_fast = [&](const unsigned char* buf, unsigned char* bufr, uint32_t keyr) {
        // (the 16 logits of the previous sub-tile are exponentiated beside the DK MFMAs of this one: elements 16 ks / DK .. 16 (ks + 1) / DK - 1
        // beside MFMA ks -- two per MFMA for D = 128, four for D = 64, three or two for D = 96)
        uint4 kf[NKF][DK];
        f32x16 acc[2];
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) kf[0][ks] = kfrag_t<DK>(buf, 0, ks, n, kg);
#pragma unroll
        for (int sub = 0; sub < MF_SUBS; ++sub) {
            if (NKF == 2 && sub + 1 < MF_SUBS) {
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) kf[(sub + 1) & (NKF - 1)][ks] = kfrag_t<DK>(buf, sub + 1, ks, n, kg);
            }
            if (NKF == 1 && sub > 0) {   // (one register set: this sub-tile's fragments are read now, after the previous chain)
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) kf[0][ks] = kfrag_t<DK>(buf, sub, ks, n, kg);
            }
            ks_.request_part(bufr, keyr, sub);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[sub & 1][i] = 0.f;
            if (sub == 0) {
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) acc[0] = mma32<DT>(kf[0][ks], qf[ks], acc[0]);  // C[key][q row]
            } else {
                // MFMA chain of this sub-tile || softmax of the previous one
                const f32x16& ap = acc[(sub - 1) & 1];
                float tm = ap[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, ap[r]);
                const float mn = fmaxf(m, tm);
                const float off = -mn * c;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    acc[sub & 1] = mma32<DT>(kf[sub & (NKF - 1)][ks], qf[ks], acc[sub & 1]);
#pragma unroll
                    for (int e = (16 * ks) / DK; e < (16 * (ks + 1)) / DK; ++e) {   // (the same two interleaved sums for every head size: element i goes to sum i & 1)
                        const float x = fast_exp2(fmaf(ap[e], c, off));
                        if (e & 1) s1 += x; else s0 += x;
                    }
                }
                z = z * fast_exp2(fmaf(m, c, off)) + (s0 + s1);
                m = mn;
                __builtin_amdgcn_sched_group_barrier(0x2, 12, 0);   // max chain + offsets while the fragments land
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x2, 3 * ((16 + DK - 1) / DK), 0);  // (fma, exp, add) of its share of the 16 logits
                }
            }
        }
        f32x16 last = acc[(MF_SUBS - 1) & 1];
        softmax16(last, 0, MF_SUBS - 1, false);
    };
    // the last tiles of a head (causal mask over the window, keys past S): chain, then the masked update, per sub-tile
    auto compute_masked = [&](uint32_t key0, const unsigned char* buf, unsigned char* bufr, uint32_t keyr) {
        for (int sub = 0; sub < MF_SUBS; ++sub) {
            uint4 kf[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kf[ks] = kfrag_t<DK>(buf, sub, ks, n, kg);
            ks_.request_part(bufr, keyr, sub);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) acc = mma32<DT>(kf[ks], qf[ks], acc);
            softmax16(acc, key0, sub, true);
        }
    };
    auto compute = [&](uint32_t key0, const unsigned char* buf, unsigned char* bufr, uint32_t keyr) {
        if ((int32_t)(key0 + (MF_TILE - 1)) > mlim) compute_masked(key0, buf, bufr, keyr);  // some (row, key) is masked / past S
        else compute_fast(buf, bufr, keyr);
    };

    if (tw.ntiles > 0) {
        __syncthreads();
        int bc = 0;
        for (uint32_t t = 0; t < tw.ntiles; ++t) {
            unsigned char* bufr = lds + Geo::ring_prev(bc) * TILEB;  // tile t-1's buffer: everybody left it at the last barrier
            if (active) compute(tw.key0(t), lds + bc * TILEB, bufr, tw.key0(t + NBUF - 1));
            else ks_.request_tile(bufr, tw.key0(t + NBUF - 1));
            __builtin_amdgcn_sched_barrier(0);
            wait_tile_landed_t<DK>();  // this wave's part of tile t+1 is in LDS ...
            __syncthreads();     // ... and so is everybody else's; all fragment reads of tile t are done
            bc = Geo::ring_next(bc);
        }
        wait_all_landed();
    }

    if (active) {
        float mm = m == KVP_NEG_INF ? KVP_NEG_INF : m * c, zz = z;
        const float m2 = __shfl_xor(mm, 32), z2 = __shfl_xor(zz, 32);
        softmax_merge(mm, zz, m2, z2);
        if (kg == 0) {
            const size_t o = ((size_t)(b * a.Hq + hq) * a.Wp + mf_prow(a, w)) * nchunk + chunk;
            part_m[o] = mm;
            part_z[o] = zz;
        }
    }
}

// values handed to an asm block through "s" constraints must sit in SGPRs: make their uniformity explicit
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float uni(float x) { return __uint_as_float(uni(__float_as_uint(x))); }
template <typename T> __device__ __forceinline__ T* uni(T* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return (T*)(uintptr_t)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}

// =================================================================================================
// pass 1 with the hand-scheduled tile loop (snapkv_asm.inc, generated by tools/gen_stage_asm.py)
// =================================================================================================
// Same work decomposition, K stream and partial statistics as snapkv_p1_mfma.  The steady state -- every group of three
// tiles that needs no mask and has its two prefetch tiles inside the workgroup's walk -- runs as ONE asm block with fixed
// registers: per 32-key sub-tile ("stage") the MFMA chain of sub-tile s, the row maximum of s-1 and the exp / sum of s-2
// (three accumulators), the fragment reads of s+1 and one LDS-DMA request for the tile two ahead, interleaved so that the
// VALU / transcendental work issues in the shadow of the MFMAs (tools/ubench_issue.hip, ubench_stage.hip).  The last tiles
// of a walk (>= 2, plus the masked ones at the end of a head) take the C++ path below with the statistics handed over.
// Requires all eight waves active (G % 4 == 0).
template <int DT>
__global__ __launch_bounds__(MF_THREADS, 2) void snapkv_p1_asm(SnapArgs a, uint32_t ngb, uint32_t nchunk,
                                                               float* __restrict__ part_m, float* __restrict__ part_z, uint32_t* __restrict__ ticks) {
    const uint64_t t_start = ticks ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz wall clock: this workgroup's time -> pass 2's tile shares
    constexpr int NB = KVP_P1_NBUF;   // ring depth of this kernel: NB - 1 tiles of 32 KiB in flight per workgroup (cold K from HBM
                                      // needs more than the two of the three-buffer ring: measured 3.5 TB/s with two)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NB * MF_TILEB];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t hq = h * a.G + gb * 4 + (wv >> 1);
    const uint32_t row0 = (wv & 1) * 32;

    const char* kbase = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const KStream ks_(kbase, a.k_ss * 2, a.S);
    const TileWalk tw(chunk, nchunk, a.S);
    if (tw.ntiles > 0) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) ks_.request_tile(lds + i * MF_TILEB, tw.key0(i));
    }
    const char* qrow = static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)mf_qrow(a, row0 + n) * a.q_sw) * 2 + kg * 16;
    // row r of this block may attend keys <= mlim + r.  SIGNED: with a padded window S - Wp can be negative (S = 253, W = 200: Wp = 256);
    // the unsigned form classified such tiles as unmasked (found by tools/snapkv_shape_fuzz.py)
    const int32_t mlim = (int32_t)a.S - (int32_t)mf_weff(a);
    // tiles for the asm loop: the leading unmasked ones (every key <= S - weff).  Its requests run NB - 1 tiles ahead with the tile
    // index clamped to the walk's last tile and no per-row clamp: if that last tile is ragged (rows past S), the C++ loop below
    // must be the one that requests it.
    uint32_t nfast = 0;
    while (nfast < tw.ntiles && (int32_t)(tw.kbeg + nfast * tw.tstride + (MF_TILE - 1)) <= mlim) ++nfast;
    const bool last_full = tw.klast + (MF_TILE - 1) <= a.S - 1;
    const uint32_t nasm = last_full ? nfast : min(nfast, tw.ntiles >= (uint32_t)NB ? tw.ntiles - NB : 0u);

    float m = KVP_NEG_INF, z = 0.f;
    const float c = a.c;
    if (nasm) {
        const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
        uint32_t la[8], vo[4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) la[ks] = ldsbase + n * MF_ROWB + (((ks * 2 + kg) ^ (n & 15)) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) vo[i] = (uint32_t)((i * 32 + ks_.lrow) * ks_.k_ssb) + ks_.choff;
        const uint32_t m0base = ldsbase + wv * 1024;
        const char* gnext = kbase + (int64_t)tw.key0(NB - 1) * ks_.k_ssb;   // first tile the block requests (clamped)
        const uint32_t gstride = (uint32_t)(tw.tstride * ks_.k_ssb);
        const uint32_t nadv = tw.ntiles >= (uint32_t)NB ? tw.ntiles - NB : 0;   // the request address advances while the next tile exists
        if (DT == KVP_BF16)
            asm volatile(KVP_P1_ASM_BF16
                         : "=&v"(m), "=&v"(z)
                         : "v"(qrow), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "v"(la[4]),
                           "v"(la[5]), "v"(la[6]), "v"(la[7]), "s"(uni(m0base)), "s"(uni(gnext)), "s"(uni(gstride)), "s"(uni(nasm)), "s"(uni(c)), "s"(uni(nadv)), "s"(uni(wv))
                         : KVP_P1_ASM_CLOBBERS);
        else
            asm volatile(KVP_P1_ASM_F16
                         : "=&v"(m), "=&v"(z)
                         : "v"(qrow), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "v"(la[4]),
                           "v"(la[5]), "v"(la[6]), "v"(la[7]), "s"(uni(m0base)), "s"(uni(gnext)), "s"(uni(gstride)), "s"(uni(nasm)), "s"(uni(c)), "s"(uni(nadv)), "s"(uni(wv))
                         : KVP_P1_ASM_CLOBBERS);
    } else if (tw.ntiles > 0) {
        wait_all_landed();
        __syncthreads();
    }

    // ---- the remaining tiles (and every masked one): plain path, same protocol (tile t visible, tiles t+1 .. t+NB-2 in flight) ----
    if (nasm < tw.ntiles) {
        uint4 qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qrow + ks * 32);
        const uint32_t w = row0 + n;
        int bc = nasm % NB;   // ring position of tile nasm
        for (uint32_t t = nasm; t < tw.ntiles; ++t) {
            const unsigned char* buf = lds + bc * MF_TILEB;
            unsigned char* bufr = lds + ((bc + NB - 1) % NB) * MF_TILEB;   // the previous tile's buffer: everybody left it at the last barrier
            const uint32_t key0 = tw.key0(t), keyr = tw.key0(t + NB - 1);
            const bool masked = (int32_t)(key0 + (MF_TILE - 1)) > mlim;
            for (int sub = 0; sub < MF_SUBS; ++sub) {
                uint4 kf[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, sub, ks, n, kg);
                ks_.request(bufr, keyr, sub);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = mma32<DT>(kf[ks], qf[ks], acc);
                if (masked) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t kk = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (kk >= a.S || (int32_t)kk > mlim + (int32_t)w) acc[r] = KVP_NEG_INF;
                    }
                }
                float tm = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, acc[r]);
                const float mn = fmaxf(m, tm);
                if (mn != KVP_NEG_INF) {
                    const float off = -mn * c;
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        s0 += fast_exp2(fmaf(acc[r], c, off));
                        s1 += fast_exp2(fmaf(acc[r + 1], c, off));
                    }
                    z = z * fast_exp2(fmaf(m, c, off)) + (s0 + s1);
                    m = mn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70 | (MF_SUBS * (NB - 2)));   // only the newer tiles' requests may still be in flight
            __syncthreads();
            bc = (bc + 1) % NB;
        }
    }
    wait_all_landed();

    float mm = m == KVP_NEG_INF ? KVP_NEG_INF : m * c, zz = z;
    const float m2 = __shfl_xor(mm, 32), z2 = __shfl_xor(zz, 32);
    softmax_merge(mm, zz, m2, z2);
    if (kg == 0) {
        const size_t o = ((size_t)(b * a.Hq + hq) * a.Wp + mf_prow(a, row0 + n)) * nchunk + chunk;
        part_m[o] = mm;
        part_z[o] = zz;
    }
    if (ticks && threadIdx.x == 0)
        ticks[((size_t)b * gridDim.y + blockIdx.y) * nchunk + chunk] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_start);
}

// =================================================================================================
// pass 2: colsum[b,h,key] = sum over the group's G*64 rows of 2^(L2 - a_row), keys < S - W
// =================================================================================================
template <int DT, int DK>
__global__ __launch_bounds__(MF_THREADS, 2) void snapkv_p2_mfma(SnapArgs a, uint32_t ngb, const float* __restrict__ rowstat,
                                                                float* __restrict__ colsum, float* __restrict__ colsum2) {
    using Geo = KGeo<DK>;
    constexpr int TILEB = Geo::TILEB;
    constexpr int NBUF = Geo::NBUF, NKF = Geo::NKF;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * TILEB];
    __shared__ float red[2][MF_WAVES][MF_TILE];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t rg = gb * 4 + (wv >> 1);
    const uint32_t row0 = (wv & 1) * 32;
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);
    const uint32_t Sm = a.S - a.W;

    const KStreamT<DK> ks_(static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2, a.k_ss * 2, a.S);
    const TileWalk tw(chunk, gridDim.x, Sm);
    if (tw.ntiles == 0) return;
    // the first K tiles are requested BEFORE the Q fragments and normalisers: one memory round trip for all
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) ks_.request_tile(lds + i * TILEB, tw.key0(i));
    uint4 qf[DK];
    load_qfrags<DK>(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)mf_qrow(a, row0 + n) * a.q_sw) * 2, kg);
    // normalisers of the 16 q rows this lane sees in the C layout: row = row0 + (r&3) + 8*(r>>2) + 4*kg
    float ar[16];
    const float* ars = rowstat + (size_t)(b * a.Hq + hq) * a.Wp + mf_prow(a, row0);
#pragma unroll
    for (int r = 0; r < 16; ++r) ar[r] = -ars[(r & 3) + 8 * (r >> 2) + 4 * kg];
    wait_all_landed();  // K tiles 0 and 1, Q fragments and normalisers are in; from here on vmcnt only counts the K stream (+ the flush stores)

    const float c = a.c;
    // group-block 0 owns colsum, group-block 1 (G > 4) its own slab: no float atomics, the two are added afterwards in a fixed order
    float* cs = (gb == 0 ? colsum : colsum2 + (size_t)(gb - 1) * a.B * a.Hkv * Sm) + (size_t)(b * a.Hkv + h) * Sm;   // (a slab per further group-block)
    const uint32_t nact = 2 * min(4u, a.G - gb * 4);  // active waves in this workgroup

    // column sums of P = 2^(L2 - a_row) over this wave's 32 q rows for the 32 keys of one finished sub-tile
    auto colsum16 = [&](const f32x16 acc, int par, int sub) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0 += fast_exp2(fmaf(acc[r], c, ar[r]));
            s1 += fast_exp2(fmaf(acc[r + 1], c, ar[r + 1]));
        }
        float s = s0 + s1;
        s += __shfl_xor(s, 32);
        if (kg == 0) red[par][wv][sub * 32 + n] = s;
    };
    // one tile, software-pipelined like pass 1: MFMA chain of sub-tile s || exp/add stream of sub-tile s-1
    auto compute = [&](const unsigned char* buf, int par, unsigned char* bufr, uint32_t keyr) {
        uint4 kf[NKF][DK];
        f32x16 acc[2];
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) kf[0][ks] = kfrag_t<DK>(buf, 0, ks, n, kg);
#pragma unroll
        for (int sub = 0; sub < MF_SUBS; ++sub) {
            if (NKF == 2 && sub + 1 < MF_SUBS) {
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) kf[(sub + 1) & (NKF - 1)][ks] = kfrag_t<DK>(buf, sub + 1, ks, n, kg);
            }
            if (NKF == 1 && sub > 0) {
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) kf[0][ks] = kfrag_t<DK>(buf, sub, ks, n, kg);
            }
            ks_.request_part(bufr, keyr, sub);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[sub & 1][i] = 0.f;
            if (sub == 0) {
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) acc[0] = mma32<DT>(qf[ks], kf[0][ks], acc[0]);  // C[q row][key]
            } else {
                const f32x16& ap = acc[(sub - 1) & 1];
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    acc[sub & 1] = mma32<DT>(qf[ks], kf[sub & (NKF - 1)][ks], acc[sub & 1]);
#pragma unroll
                    for (int e = (16 * ks) / DK; e < (16 * (ks + 1)) / DK; ++e) {
                        const float x = fast_exp2(fmaf(ap[e], c, ar[e]));
                        if (e & 1) s1 += x; else s0 += x;
                    }
                }
                float s = s0 + s1;
                s += __shfl_xor(s, 32);
                if (kg == 0) red[par][wv][(sub - 1) * 32 + n] = s;
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 3 * ((16 + DK - 1) / DK), 0);
                }
            }
        }
        colsum16(acc[(MF_SUBS - 1) & 1], par, MF_SUBS - 1);
    };
    // after the tile's barrier: threads 0..127 add the active waves' partials and store the tile's column sums
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

    __syncthreads();
    int bc = 0;
    for (uint32_t t = 0; t < tw.ntiles; ++t) {
        unsigned char* bufr = lds + Geo::ring_prev(bc) * TILEB;
        if (active) compute(lds + bc * TILEB, t & 1, bufr, tw.key0(t + NBUF - 1));
        else ks_.request_tile(bufr, tw.key0(t + NBUF - 1));
        __builtin_amdgcn_sched_barrier(0);
        wait_tile_landed_t<DK>();
        __syncthreads();
        flush(tw.key0(t), t & 1);
        bc = Geo::ring_next(bc);
    }
    wait_all_landed();
}

}  // namespace

bool snapkv_mfma_eligible(const SnapArgs& a, int dtype) {
    if (dtype != KVP_BF16 && dtype != KVP_F16) return false;
    if ((a.D != 256 && a.D != 128 && a.D != 96 && a.D != 64) || a.W < 1 || a.W > 4096 || a.G > 16) return false;   // (any window: blocks of 64 rows, snapkv_internal.h)
    auto al8 = [](int64_t x) { return x % 8 == 0; };
    if (((uintptr_t)a.q % 16) || ((uintptr_t)a.k % 16)) return false;
    return al8(a.q_sb) && al8(a.q_sh) && al8(a.q_sw) && al8(a.k_sb) && al8(a.k_sh) && al8(a.k_ss);
}

// Workgroups per (batch, kv-head, group-block).  The grid is sized to ONE resident round (1 workgroup of 8 waves
// per CU x 256 CUs): every workgroup pays its start-up (Q fragments, first two K tiles) once and there is
// no second dispatch round; each workgroup then walks its interleaved tile list (TileWalk).
static uint32_t mfma_nchunk_for(const SnapArgs& a, uint32_t nkeys) {
    const uint32_t ngb = (a.G + 3) / 4;
    const uint32_t planes = std::max<uint32_t>(1, a.B * a.Hkv * ngb);
    const uint32_t by_keys = (nkeys + MF_CHUNK - 1) / MF_CHUNK;   // >= MF_CHUNK keys per workgroup
    // one 8-wave workgroup per CU.  KVP_SK_SLOTS lowers it: fewer, longer tile walks -- for a process that owns
    // only part of the device (CU masks), and the way the tests reach 128k-token walk lengths on small inputs
    const int slots = std::max(1, kvp_env_int("KVP_SK_SLOTS", 256));
    const uint32_t by_cus = std::max<uint32_t>(1, (uint32_t)slots / planes);
    return std::max<uint32_t>(1, std::min(std::min(by_keys, by_cus), 256u));  // 256: what the partial-statistics workspace is sized for
}
uint32_t snapkv_mfma_nchunk(const SnapArgs& a) { return mfma_nchunk_for(a, a.S); }

// pass 2 can take its tile ranges from pass 1's workgroup times when both passes run the hand-scheduled kernels on the same grid and
// every walk is long (>= 8 tiles: a range never shrinks below the three tiles a ragged tail hands to the plain path).
// KVP_SK_BALANCE=0 keeps the static interleaved walk (A/B runs).
bool snapkv_p2_shares_plan(const SnapArgs& a, uint32_t nchunk_p1) {
    if (a.G % 4 != 0 || a.D != 128 || a.S <= a.W) return false;
    const uint32_t Sm = a.S - a.W;
    const uint32_t ntiles = (Sm + MF_TILE - 1) / MF_TILE;
    return mfma_nchunk_for(a, Sm) == nchunk_p1 && nchunk_p1 >= 2 && nchunk_p1 <= 64 && ntiles >= 8 * nchunk_p1 && kvp_env_int("KVP_SK_BALANCE", 1) != 0;
}

int snapkv_mfma_p1(const SnapArgs& a0, int dtype, uint32_t nchunk, float* part_m, float* part_z, uint32_t* p1_ticks, hipStream_t stream) {
    const uint32_t ngb = (a0.G + 3) / 4;
    const dim3 grid(nchunk, a0.Hkv * ngb, a0.B);
    SnapArgs a = a0;
    for (a.rblk = 0; a.rblk < a.Wp / 64; ++a.rblk) {   // one launch per block of 64 (padded) window rows
        if (a.G % 4 == 0 && a.D == 128) {   // hand-scheduled tile loop: all eight waves of a workgroup own a q-head half
            if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p1_asm", stream, snapkv_p1_asm<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z, p1_ticks));
            else KVP_LAUNCH("snapkv_p1_asm", stream, snapkv_p1_asm<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z, p1_ticks));
            KVP_CHECK_LAUNCH("snapkv_p1_asm");
            continue;
        }
        // G = 1, 2, 3, 5, 6, 7 (partially filled workgroups) and head size 64: the compiler-scheduled kernels
#define KVP_P1_MFMA(DTV, DKV) KVP_LAUNCH("snapkv_p1_mfma", stream, (snapkv_p1_mfma<DTV, DKV><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z)))
        if (a.D == 256) { if (dtype == KVP_BF16) KVP_P1_MFMA(KVP_BF16, 16); else KVP_P1_MFMA(KVP_F16, 16); }
        else if (a.D == 128) { if (dtype == KVP_BF16) KVP_P1_MFMA(KVP_BF16, 8); else KVP_P1_MFMA(KVP_F16, 8); }
        else if (a.D == 96) { if (dtype == KVP_BF16) KVP_P1_MFMA(KVP_BF16, 6); else KVP_P1_MFMA(KVP_F16, 6); }
        else { if (dtype == KVP_BF16) KVP_P1_MFMA(KVP_BF16, 4); else KVP_P1_MFMA(KVP_F16, 4); }
#undef KVP_P1_MFMA
        KVP_CHECK_LAUNCH("snapkv_p1_mfma");
    }
    return KVP_OK;
}

// =================================================================================================
// pass 2 with the hand-scheduled tile loop (snapkv_asm.inc): MFMA chain of sub-tile s || exp / column sums of s-2
// =================================================================================================
// Every lane half of every wave writes its partial column sum of a 32-key sub-tile to its own LDS slot (16 per key); after
// the barrier that ends the NEXT tile, waves 0-1 add the 16 slots in a fixed order and store the tile's 128 column sums
// (deterministic; three tiles of slots in flight).  Tiles that reach past S - W, the walk's last tiles when they are ragged,
// take the C++ path below.  Requires all eight waves active (G % 4 == 0).
template <int DT>
__global__ __launch_bounds__(MF_THREADS, 2) void snapkv_p2_asm(SnapArgs a, uint32_t ngb, const float* __restrict__ rowstat,
                                                               float* __restrict__ colsum, float* __restrict__ colsum2, const uint32_t* __restrict__ ticks) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[MF_NBUF * MF_TILEB];
    __shared__ __attribute__((aligned(16))) unsigned char red3[KVP_P2_RED_BYTES];
    __shared__ float red[2][MF_WAVES][MF_TILE];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t hq = h * a.G + gb * 4 + (wv >> 1);
    const uint32_t row0 = (wv & 1) * 32;
    const uint32_t Sm = a.S - a.W;

    const char* kbase = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const KStream ks_(kbase, a.k_ss * 2, a.S);
    // This workgroup's tiles: the static interleaved walk, or (ticks: pass 1's workgroup times of this plane, gridDim.x <= 64 of them) a
    // contiguous range proportional to 1 / time -- every wave of every workgroup of the plane runs the same few instructions on the same
    // numbers (times clamped to [tmin, 2 tmin]: a workgroup held up for a reason of its own must not starve; speed = 1 / time; inclusive
    // scan over the plane's workgroups; end_c = round(ntiles * prefix_c / total)), so neighbours agree on their common boundary, the
    // ranges are monotone and the last one ends at ntiles.  Either way a tile's column sums are computed by one workgroup in a fixed
    // order: the same bits.  (Round 6, first version: the ranges came out of extra blocks of the combine launch -- double precision,
    // block-wide scans: 4.9 -> 7.8 us on the critical path; here it is ~50 VALU instructions behind one L2 load in the prologue.)
    uint32_t r_first = 0, r_count = 0;
    if (ticks) {
        const uint32_t nch = gridDim.x, total_tiles = (Sm + MF_TILE - 1) / MF_TILE;
        const uint32_t* tp = ticks + ((size_t)b * gridDim.y + blockIdx.y) * nch;
        const uint32_t tk = lane < nch ? max(tp[lane], 1u) : 0xFFFFFFFFu;
        uint32_t tmin = tk;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tmin = min(tmin, (uint32_t)__shfl_xor((int)tmin, o));
        const float speed = lane < nch ? 1.0f / (float)min(tk, 2u * tmin) : 0.f;
        float incl = speed;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o);
            if ((int)lane >= o) incl += up;
        }
        const float total = __shfl(incl, 63);
        const float prev = __shfl_up(incl, 1);
        const uint32_t end = lane + 1 >= nch ? total_tiles : min((uint32_t)__float2uint_rn((float)total_tiles * (incl / total)), total_tiles);
        const uint32_t beg = lane == 0 ? 0u : min((uint32_t)__float2uint_rn((float)total_tiles * (prev / total)), total_tiles);
        r_first = uni((uint32_t)__shfl((int)beg, (int)chunk));
        r_count = uni((uint32_t)__shfl((int)(end > beg ? end - beg : 0u), (int)chunk));
    }
    const TileWalk tw = ticks ? TileWalk(r_first, r_count, 0) : TileWalk(chunk, gridDim.x, Sm);
    if (tw.ntiles == 0) return;
    ks_.request_tile(lds, tw.key0(0));
    ks_.request_tile(lds + MF_TILEB, tw.key0(1));
    const char* qrow = static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh + (int64_t)mf_qrow(a, row0 + n) * a.q_sw) * 2 + kg * 16;
    const float* ars = rowstat + (size_t)(b * a.Hq + hq) * a.Wp + mf_prow(a, row0);
    float* cs = (gb == 0 ? colsum : colsum2 + (size_t)(gb - 1) * a.B * a.Hkv * Sm) + (size_t)(b * a.Hkv + h) * Sm;   // (a slab per further group-block)
    const float c = a.c;

    uint32_t nfast = 0;   // tiles stored completely (all 128 keys < S - W)
    while (nfast < tw.ntiles && tw.kbeg + nfast * tw.tstride + MF_TILE <= Sm) ++nfast;
    const bool last_full = tw.klast + (MF_TILE - 1) <= a.S - 1;
    const uint32_t nasm = last_full ? nfast : min(nfast, tw.ntiles >= 3 ? tw.ntiles - 3 : 0u);

    if (nasm) {
        const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
        const uint32_t redbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)red3;
        uint32_t la[8], vo[4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) la[ks] = ldsbase + n * MF_ROWB + (((ks * 2 + kg) ^ (n & 15)) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) vo[i] = (uint32_t)((i * 32 + ks_.lrow) * ks_.k_ssb) + ks_.choff;
        const uint32_t m0base = __builtin_amdgcn_readfirstlane(ldsbase + wv * 1024);
        const uint32_t wvs = __builtin_amdgcn_readfirstlane(wv);
        const char* gnext = kbase + (int64_t)tw.key0(2) * ks_.k_ssb;
        const uint32_t gstride = (uint32_t)(tw.tstride * ks_.k_ssb);
        const uint32_t nadv = tw.ntiles >= 3 ? tw.ntiles - 3 : 0;
        const float* arp = ars + 4 * kg;
        float* cs0 = cs + tw.kbeg;                               // column sums of this walk's first tile
        const uint32_t csstride = tw.tstride * 4;
        const uint32_t redw = redbase + (wv * 2 + kg) * 512 + n * 4;   // this lane's slot for a sub-tile's key n
        const uint32_t flr = redbase + (threadIdx.x & 127) * 4;        // flush: slot 0 of key threadIdx.x (waves 0, 1)
        const uint32_t flo = (threadIdx.x & 127) * 4;
        if (DT == KVP_BF16)
            asm volatile(KVP_P2_ASM_BF16
                         :
                         : "v"(qrow), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "v"(la[4]),
                           "v"(la[5]), "v"(la[6]), "v"(la[7]), "s"(uni(m0base)), "s"(uni(gnext)), "s"(uni(gstride)), "s"(uni(nasm)), "s"(uni(c)), "s"(uni(nadv)), "s"(uni(wvs)),
                           "s"(uni(cs0)), "v"(arp), "s"(uni(csstride)), "v"(redw), "v"(flr), "v"(flo)
                         : KVP_P2_ASM_CLOBBERS);
        else
            asm volatile(KVP_P2_ASM_F16
                         :
                         : "v"(qrow), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "v"(la[4]),
                           "v"(la[5]), "v"(la[6]), "v"(la[7]), "s"(uni(m0base)), "s"(uni(gnext)), "s"(uni(gstride)), "s"(uni(nasm)), "s"(uni(c)), "s"(uni(nadv)), "s"(uni(wvs)),
                           "s"(uni(cs0)), "v"(arp), "s"(uni(csstride)), "v"(redw), "v"(flr), "v"(flo)
                         : KVP_P2_ASM_CLOBBERS);
    } else {
        wait_all_landed();
        __syncthreads();
    }

    if (nasm < tw.ntiles) {
        uint4 qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qrow + ks * 32);
        float ar[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = -ars[(r & 3) + 8 * (r >> 2) + 4 * kg];
        int bc = nasm % 3;
        for (uint32_t t = nasm; t < tw.ntiles; ++t) {
            const unsigned char* buf = lds + bc * MF_TILEB;
            unsigned char* bufr = lds + ring_prev(bc) * MF_TILEB;
            const uint32_t key0 = tw.key0(t), keyr = tw.key0(t + 2);
            const int par = t & 1;
            for (int sub = 0; sub < MF_SUBS; ++sub) {
                uint4 kf[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, sub, ks, n, kg);
                ks_.request(bufr, keyr, sub);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = mma32<DT>(qf[ks], kf[ks], acc);
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    s0 += fast_exp2(fmaf(acc[r], c, ar[r]));
                    s1 += fast_exp2(fmaf(acc[r + 1], c, ar[r + 1]));
                }
                float sm = s0 + s1;
                sm += __shfl_xor(sm, 32);
                if (kg == 0) red[par][wv][sub * 32 + n] = sm;
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_tile_landed();
            __syncthreads();
            if (threadIdx.x < MF_TILE) {
                const uint32_t kk = key0 + threadIdx.x;
                if (kk < Sm) {
                    float sm = red[par][0][threadIdx.x];
                    for (uint32_t w = 1; w < MF_WAVES; ++w) sm += red[par][w][threadIdx.x];
                    cs[kk] = sm;
                }
            }
            bc = ring_next(bc);
        }
    }
    wait_all_landed();
}

// x += y[0] + y[1] + ... (nslab slabs of n floats, added one after the other in slab order: run-to-run identical)
namespace {
__global__ __launch_bounds__(256) void add_slab_kernel(float* __restrict__ x, const float* __restrict__ y, size_t n, uint32_t nslab = 1) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = x[i];
        for (uint32_t j = 0; j < nslab; ++j) s += y[(size_t)j * n + i];
        x[i] = s;
    }
}
}  // namespace

int snapkv_mfma_p2(const SnapArgs& a0, int dtype, const float* rowstat, float* colsum, float* colsum2, float* colsumx, const uint32_t* p1_ticks,
                   hipStream_t stream) {
    const uint32_t ngb = (a0.G + 3) / 4;
    const uint32_t Sm = a0.S - a0.W;
    const uint32_t nrblk = a0.Wp / 64;
    KVP_CHECK_ARG(ngb == 1 || colsum2, "snapkv_p2_mfma: G = %u needs the column-sum slabs of its further group-blocks", a0.G);
    KVP_CHECK_ARG(nrblk == 1 || colsumx, "snapkv_p2_mfma: W = %u needs the row-block slab", a0.W);
    const dim3 grid(mfma_nchunk_for(a0, Sm), a0.Hkv * ngb, a0.B);
    const size_t nsum = (size_t)a0.B * a0.Hkv * Sm;
    const unsigned ablocks = (unsigned)std::min<size_t>((nsum + 255) / 256, 4096);
    SnapArgs a = a0;
    for (a.rblk = 0; a.rblk < nrblk; ++a.rblk) {
        // the first 64-row block writes the column sums, every later one its own slab that is then added: always in block order
        // (run-to-run identical scores; float atomics were not)
        float* cs = a.rblk == 0 ? colsum : colsumx;
#define KVP_P2_MFMA(DTV, DKV) KVP_LAUNCH("snapkv_p2_mfma", stream, (snapkv_p2_mfma<DTV, DKV><<<grid, MF_THREADS, 0, stream>>>(a, ngb, rowstat, cs, colsum2)))
        if (a.G % 4 == 0 && a.D == 128) {   // hand-scheduled tile loop
            if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p2_asm", stream, snapkv_p2_asm<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, rowstat, cs, colsum2, p1_ticks));
            else KVP_LAUNCH("snapkv_p2_asm", stream, snapkv_p2_asm<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, rowstat, cs, colsum2, p1_ticks));
        } else if (a.D == 256) { if (dtype == KVP_BF16) KVP_P2_MFMA(KVP_BF16, 16); else KVP_P2_MFMA(KVP_F16, 16); }
        else if (a.D == 128) { if (dtype == KVP_BF16) KVP_P2_MFMA(KVP_BF16, 8); else KVP_P2_MFMA(KVP_F16, 8); }
        else if (a.D == 96) { if (dtype == KVP_BF16) KVP_P2_MFMA(KVP_BF16, 6); else KVP_P2_MFMA(KVP_F16, 6); }
        else { if (dtype == KVP_BF16) KVP_P2_MFMA(KVP_BF16, 4); else KVP_P2_MFMA(KVP_F16, 4); }
#undef KVP_P2_MFMA
        KVP_CHECK_LAUNCH("snapkv_p2_mfma");
        if (ngb > 1) {  // cs += colsum2[0] + colsum2[1] + ... (the further group-blocks' sums, in block order)
            KVP_LAUNCH("add_slab_kernel", stream, add_slab_kernel<<<ablocks, 256, 0, stream>>>(cs, colsum2, nsum, ngb - 1));
            KVP_CHECK_LAUNCH("snapkv_p2_mfma(add)");
        }
        if (a.rblk > 0) {
            KVP_LAUNCH("add_slab_kernel", stream, add_slab_kernel<<<ablocks, 256, 0, stream>>>(colsum, colsumx, nsum));
            KVP_CHECK_LAUNCH("snapkv_p2_mfma(add rows)");
        }
    }
    return KVP_OK;
}
