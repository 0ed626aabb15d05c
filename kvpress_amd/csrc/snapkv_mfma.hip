// SnapKV window-attention passes on the gfx950 matrix cores (bf16 / f16, D = 128, W = 64).
//
// Work decomposition (per launch): workgroup = (1024-key chunk, kv-head [x group-block], batch),
// 4 waves; wave w owns ONE q-head of the GQA group = the 64 window rows of that head, whose
// Q fragments (64 rows x 128 dims = 16 x dwordx4 per lane) stay in registers for the whole
// chunk.  K streams HBM -> registers -> LDS in 64-key tiles (16 KiB, full 256-B rows, coalesced
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

constexpr int MF_THREADS = 256;
constexpr int MF_TILE = 64;          // keys per LDS tile
constexpr int MF_CHUNK = 1024;       // keys per workgroup
constexpr int MF_ROWB = 256;         // bytes per key row (D = 128, 2-byte elements)
constexpr int MF_TILEB = MF_TILE * MF_ROWB;

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
    const uint32_t r0 = threadIdx.x >> 4, ch = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = r0 + 16 * i;
        *reinterpret_cast<uint4*>(buf + row * MF_ROWB + ((ch ^ (row & 15)) << 4)) = st.v[i];
    }
}
// fragment of the 32-key sub-tile `sub` for k-step ks: lane (n = lane & 31, kg = lane >> 5)
__device__ __forceinline__ uint4 kfrag(const unsigned char* buf, uint32_t sub, uint32_t ks, uint32_t n, uint32_t kg) {
    const uint32_t row = sub * 32 + n;
    return *reinterpret_cast<const uint4*>(buf + row * MF_ROWB + (((ks * 2 + kg) ^ (row & 15)) << 4));
}

// Q fragments of one q-head: [half (32 rows)][k-step] ; lane (n, kg) holds row half*32+n, dims ks*16+kg*8..+8
__device__ __forceinline__ void load_qfrags(uint4 (&qf)[2][8], const char* __restrict__ qhead, int64_t q_swb, uint32_t n, uint32_t kg) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[hf][ks] = *reinterpret_cast<const uint4*>(qhead + (int64_t)(hf * 32 + n) * q_swb + (ks * 16 + kg * 8) * 2);
}

// =================================================================================================
// pass 1: per (row, chunk) partial max / sum-exp (log2 units)
// =================================================================================================
template <int DT>
__global__ __launch_bounds__(MF_THREADS) void snapkv_p1_mfma(SnapArgs a, uint32_t ngb, uint32_t nchunk,
                                                             float* __restrict__ part_m, float* __restrict__ part_z) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MF_TILEB];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t rg = gb * 4 + wv;  // q-head inside the GQA group
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);

    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const int64_t k_ssb = a.k_ss * 2;
    uint4 qf[2][8];
    load_qfrags(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh) * 2, a.q_sw * 2, n, kg);

    // Tile -> workgroup mapping is INTERLEAVED: workgroup `chunk` of the nchunk workgroups of this kv-head
    // takes tiles chunk, chunk + nchunk, chunk + 2 nchunk, ...  The workgroups that run concurrently then read
    // one contiguous, advancing region of K (nchunk x 16 KiB = 2 MiB per head) that covers every HBM
    // channel evenly; giving each workgroup its own contiguous 256-KiB chunk instead puts all concurrent
    // streams 256 KiB apart in lockstep (same low address bits -> same channels): measured 2.6 TB/s.
    const uint32_t total_tiles = (a.S + MF_TILE - 1) / MF_TILE;
    const uint32_t ntiles = chunk < total_tiles ? (total_tiles - chunk + nchunk - 1) / nchunk : 0;
    const uint32_t tstride = nchunk * MF_TILE;           // keys between this workgroup's consecutive tiles
    const uint32_t kbeg = chunk * MF_TILE;

    float m[2] = {KVP_NEG_INF, KVP_NEG_INF};  // raw-logit running max for q rows n and 32+n
    float z[2] = {0.f, 0.f};
    const float c = a.c;

    // one 64-key tile: 32 MFMAs + the running (max, sum-exp) update of this lane's two q rows.
    // Schedule: all 8 K fragments of a 32-key sub-tile are read from LDS before its 16 MFMAs
    // (counted lgkmcnt instead of read->wait->2 MFMAs), and BOTH sub-tiles' MFMAs are issued
    // before any softmax VALU so the exp/max work of sub-tile 0 runs under sub-tile 1's MFMAs.
    auto compute = [&](uint32_t key0, const unsigned char* buf) {
        const bool need_mask = key0 + (MF_TILE - 1) > a.S - a.W;  // some (row, key) of this tile is masked / past S
        f32x16 acc[2][2];  // [sub][hf]
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[sub][hf][i] = 0.f;
            uint4 kf[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, sub, ks, n, kg);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc[sub][0] = mma32<DT>(kf[ks], qf[0][ks], acc[sub][0]);  // C[key][q row]
                acc[sub][1] = mma32<DT>(kf[ks], qf[1][ks], acc[sub][1]);
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (need_mask) {
                    const uint32_t w = hf * 32 + n;  // window row: token S-W+w sees keys <= S-W+w
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t kk = key0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (kk >= a.S || kk > a.S - a.W + w) acc[sub][hf][r] = KVP_NEG_INF;
                    }
                }
                float tm = acc[sub][hf][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tm = fmaxf(tm, acc[sub][hf][r]);
                const float mn = fmaxf(m[hf], tm);
                if (!need_mask || mn != KVP_NEG_INF) {
                    const float off = -mn * c;
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += fast_exp2(fmaf(acc[sub][hf][r], c, off));
                    z[hf] = z[hf] * fast_exp2(fmaf(m[hf], c, off)) + s;
                    m[hf] = mn;
                }
            }
        }
    };

    // K streams HBM -> registers -> LDS with TWO tiles in flight behind the one being computed:
    // stA / stB alternate; each tile's loads have two compute phases to land (issue-early, write-late).
    Stage stA, stB;
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + MF_TILEB;
    if (ntiles > 0) {
        const uint32_t klast = kbeg + (ntiles - 1) * tstride;  // prefetches past the end re-read the last tile (L2 hits, never stored)
        stA = stage_load(kb, k_ssb, kbeg, a.S);
        stage_store(stA, buf0);
        stA = stage_load(kb, k_ssb, min(kbeg + tstride, klast), a.S);
        __syncthreads();
        for (uint32_t t = 0; t < ntiles; t += 2) {
            const uint32_t key0 = kbeg + t * tstride;
            stB = stage_load(kb, k_ssb, min(key0 + 2 * tstride, klast), a.S);
            __builtin_amdgcn_sched_barrier(0);  // issue-early
            if (active) compute(key0, buf0);
            __builtin_amdgcn_sched_barrier(0);  // keep the LDS write of the older stage BEHIND this tile's MFMAs (write-late)
            if (t + 1 < ntiles) stage_store(stA, buf1);
            __syncthreads();
            if (t + 1 >= ntiles) break;
            stA = stage_load(kb, k_ssb, min(key0 + 3 * tstride, klast), a.S);
            __builtin_amdgcn_sched_barrier(0);
            if (active) compute(key0 + tstride, buf1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntiles) stage_store(stB, buf0);
            __syncthreads();
        }
    }

    if (active) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float mm = m[hf] == KVP_NEG_INF ? KVP_NEG_INF : m[hf] * c, zz = z[hf];
            const float m2 = __shfl_xor(mm, 32), z2 = __shfl_xor(zz, 32);
            softmax_merge(mm, zz, m2, z2);
            if (kg == 0) {
                const size_t o = ((size_t)(b * a.Hq + hq) * a.W + hf * 32 + n) * nchunk + chunk;
                part_m[o] = mm;
                part_z[o] = zz;
            }
        }
    }
}

// =================================================================================================
// pass 2: colsum[b,h,key] = sum over the group's G*64 rows of 2^(L2 - a_row), keys < S - W
// =================================================================================================
template <int DT>
__global__ __launch_bounds__(MF_THREADS) void snapkv_p2_mfma(SnapArgs a, uint32_t ngb, const float* __restrict__ rowstat,
                                                             float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * MF_TILEB];
    __shared__ float red[2][4][MF_TILE];
    const uint32_t chunk = blockIdx.x, b = blockIdx.z;
    const uint32_t h = blockIdx.y / ngb, gb = blockIdx.y - h * ngb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = lane & 31, kg = lane >> 5;
    const uint32_t rg = gb * 4 + wv;
    const bool active = rg < a.G;
    const uint32_t hq = h * a.G + (active ? rg : 0);
    const uint32_t Sm = a.S - a.W;

    const char* kb = static_cast<const char*>(a.k) + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh) * 2;
    const int64_t k_ssb = a.k_ss * 2;
    uint4 qf[2][8];
    load_qfrags(qf, static_cast<const char*>(a.q) + ((int64_t)b * a.q_sb + (int64_t)hq * a.q_sh) * 2, a.q_sw * 2, n, kg);
    // normalisers of the 32 q rows this lane sees in the C layout: row = hf*32 + (r&3) + 8*(r>>2) + 4*kg
    float ar[2][16];
    const float* ars = rowstat + (size_t)(b * a.Hq + hq) * a.W;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[hf][r] = -ars[hf * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];

    // interleaved tile -> workgroup mapping (see pass 1)
    const uint32_t nchunk = gridDim.x;
    const uint32_t total_tiles = (Sm + MF_TILE - 1) / MF_TILE;
    const uint32_t ntiles = chunk < total_tiles ? (total_tiles - chunk + nchunk - 1) / nchunk : 0;
    const uint32_t tstride = nchunk * MF_TILE;
    const uint32_t kbeg = chunk * MF_TILE;
    const float c = a.c;
    float* cs = colsum + (size_t)(b * a.Hkv + h) * Sm;
    const uint32_t nact = min(4u, a.G - gb * 4);  // active waves in this workgroup

    // one 64-key tile: 32 MFMAs, P = 2^(L2 - a_row), column sums over this wave's 64 q rows -> red[par][wave][key]
    // (same schedule as pass 1: fragment reads batched, both sub-tiles' MFMAs ahead of the exp work)
    auto compute = [&](const unsigned char* buf, int par) {
        f32x16 acc[2][2];  // [sub][hf]
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[sub][hf][i] = 0.f;
            uint4 kf[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks] = kfrag(buf, sub, ks, n, kg);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc[sub][0] = mma32<DT>(qf[0][ks], kf[ks], acc[sub][0]);  // C[q row][key]
                acc[sub][1] = mma32<DT>(qf[1][ks], kf[ks], acc[sub][1]);
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float s = 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += fast_exp2(fmaf(acc[sub][hf][r], c, ar[hf][r]));
            s += __shfl_xor(s, 32);
            if (kg == 0) red[par][wv][sub * 32 + n] = s;
        }
    };
    // after the tile's barrier: threads 0..63 add the active waves' partials and store 64 column sums
    auto flush = [&](uint32_t key0, int par) {
        if (threadIdx.x < MF_TILE) {
            const uint32_t kk = key0 + threadIdx.x;
            if (kk < Sm) {
                float s = red[par][0][threadIdx.x];
                for (uint32_t w = 1; w < nact; ++w) s += red[par][w][threadIdx.x];
                if (ngb == 1) cs[kk] = s;
                else atomicAdd(&cs[kk], s);
            }
        }
    };

    Stage stA, stB;
    unsigned char* buf0 = lds;
    unsigned char* buf1 = lds + MF_TILEB;
    if (ntiles == 0) return;
    const uint32_t klast = kbeg + (ntiles - 1) * tstride;
    stA = stage_load(kb, k_ssb, kbeg, a.S);
    stage_store(stA, buf0);
    stA = stage_load(kb, k_ssb, min(kbeg + tstride, klast), a.S);
    __syncthreads();
    for (uint32_t t = 0; t < ntiles; t += 2) {
        const uint32_t key0 = kbeg + t * tstride;
        stB = stage_load(kb, k_ssb, min(key0 + 2 * tstride, klast), a.S);
        __builtin_amdgcn_sched_barrier(0);
        if (active) compute(buf0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntiles) stage_store(stA, buf1);
        __syncthreads();
        flush(key0, 0);
        if (t + 1 >= ntiles) break;
        stA = stage_load(kb, k_ssb, min(key0 + 3 * tstride, klast), a.S);
        __builtin_amdgcn_sched_barrier(0);
        if (active) compute(buf1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntiles) stage_store(stB, buf0);
        __syncthreads();
        flush(key0 + tstride, 1);
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

uint32_t snapkv_mfma_nchunk(const SnapArgs& a) { return (a.S + MF_CHUNK - 1) / MF_CHUNK; }

int snapkv_mfma_p1(const SnapArgs& a, int dtype, uint32_t nchunk, float* part_m, float* part_z, hipStream_t stream) {
    const uint32_t ngb = (a.G + 3) / 4;
    const dim3 grid(nchunk, a.Hkv * ngb, a.B);
    if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p1_mfma", stream, snapkv_p1_mfma<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z));
    else KVP_LAUNCH("snapkv_p1_mfma", stream, snapkv_p1_mfma<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, nchunk, part_m, part_z));
    KVP_CHECK_LAUNCH("snapkv_p1_mfma");
    return KVP_OK;
}

int snapkv_mfma_p2(const SnapArgs& a, int dtype, const float* rowstat, float* colsum, hipStream_t stream) {
    const uint32_t ngb = (a.G + 3) / 4;
    const uint32_t Sm = a.S - a.W;
    if (ngb > 1) {
        if (hipMemsetAsync(colsum, 0, (size_t)a.B * a.Hkv * Sm * 4, stream) != hipSuccess) {
            kvp_set_error("snapkv_p2_mfma: memset failed");
            return KVP_EHIP;
        }
    }
    const dim3 grid((Sm + MF_CHUNK - 1) / MF_CHUNK, a.Hkv * ngb, a.B);
    if (dtype == KVP_BF16) KVP_LAUNCH("snapkv_p2_mfma", stream, snapkv_p2_mfma<KVP_BF16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, rowstat, colsum));
    else KVP_LAUNCH("snapkv_p2_mfma", stream, snapkv_p2_mfma<KVP_F16><<<grid, MF_THREADS, 0, stream>>>(a, ngb, rowstat, colsum));
    KVP_CHECK_LAUNCH("snapkv_p2_mfma");
    return KVP_OK;
}
