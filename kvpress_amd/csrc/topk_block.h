// Block-wide scan / digit search shared by the one-workgroup-per-row select (topk.hip) and the cluster select
// (topk_cluster.hip): 1024-thread workgroups, thread 0 owns the HIGHEST bins of a histogram.
#pragma once
#include "kvp_common.h"

constexpr int TR_THREADS = 1024;
constexpr int TR_WAVES = TR_THREADS / 64;

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / row broadcasts (the classic gfx9 sequence: three row_shr
// of the input, row_shr:4 / :8 with bank masks, row_bcast:15 / :31 with row masks): 7 VALU-speed steps instead of the 6 dependent
// ds_bpermute round trips of a __shfl_up ladder (~100+ cycles each) -- the scans are on the critical path of the latency-bound
// select kernels.  update_dpp(old = 0, ...) yields 0 for lanes that the masks disable or whose source lies outside the row.
#define KVP_DPP0(v, ctrl, row_mask, bank_mask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (row_mask), (bank_mask), false))
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += KVP_DPP0(x, 0x111, 0xf, 0xf);   // row_shr:1
    v += KVP_DPP0(x, 0x112, 0xf, 0xf);   // row_shr:2
    v += KVP_DPP0(x, 0x113, 0xf, 0xf);   // row_shr:3   -> sum of a lane and its three predecessors inside the 16-lane row
    v += KVP_DPP0(v, 0x114, 0xf, 0xe);   // row_shr:4 into lanes 4..15 of every row
    v += KVP_DPP0(v, 0x118, 0xf, 0xc);   // row_shr:8 into lanes 8..15              -> inclusive scan of every row
    v += KVP_DPP0(v, 0x142, 0xa, 0xf);   // row_bcast:15 into rows 1 and 3
    v += KVP_DPP0(v, 0x143, 0xc, 0xf);   // row_bcast:31 into rows 2 and 3          -> inclusive scan of the wave
    return v;
}
#undef KVP_DPP0

// exclusive prefix sum over the 1024 threads of the block; lds: >= TR_WAVES words.
// The wave index is wave-uniform (readfirstlane) and the second level -- the scan over the 16 wave totals -- runs in the lanes
// (every 16-lane row scans the same 16 values; the wave picks its offset with v_readlane): no per-wave comparison masks, which
// the compiler would otherwise keep alive in 2 x 15 scalar registers across every scan of a kernel.
__device__ __forceinline__ uint32_t row_excl_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    const uint32_t x = wave_incl_scan(lds[lane & (TR_WAVES - 1)]);   // lanes 0..15 (row 0): the inclusive scan of the 16 wave totals
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)x, TR_WAVES - 1);
    const uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)x, w > 0 ? w - 1 : 0);
    const uint32_t woff = w > 0 ? below : 0u;
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// Digit search on bin counts held in registers: thread t (< nowners) holds the PERB consecutive bins
// (nowners - 1 - t) * PERB + i in loc[i] (thread 0 the highest ones), every other thread zeros.  Finds the bin that holds
// the k-th largest element (k >= 1): count(d > bin) < k <= count(d >= bin); krem = k - count(d > bin).
// lds: >= TR_WAVES + 2 words.
template <int PERB>
__device__ __forceinline__ void row_find_bin_regs(const uint32_t (&loc)[PERB], uint32_t nowners, uint32_t k, uint32_t* lds, uint32_t& bin,
                                                  uint32_t& krem) {
    const bool owner = threadIdx.x < nowners;
    const uint32_t rg = nowners - 1 - threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < PERB; ++i) sum += loc[i];
    uint32_t total;
    const uint32_t excl = row_excl_scan(sum, lds, &total);
    if (owner && excl < k && k <= excl + sum) {
        uint32_t c = excl;
#pragma unroll
        for (int i = PERB - 1; i >= 0; --i) {
            if (k > c && k <= c + loc[i]) {
                lds[TR_WAVES] = rg * PERB + i;
                lds[TR_WAVES + 1] = k - c;
            }
            c += loc[i];
        }
    }
    __syncthreads();
    bin = lds[TR_WAVES];
    krem = lds[TR_WAVES + 1];
    __syncthreads();
}

// the same on a histogram in (LDS or global) memory
template <int NB>
__device__ __forceinline__ void row_find_bin(const uint32_t* hist, uint32_t k, uint32_t* lds, uint32_t& bin, uint32_t& krem) {
    constexpr int PERB = NB >= TR_THREADS ? NB / TR_THREADS : 1;
    constexpr uint32_t NOWN = NB >= TR_THREADS ? TR_THREADS : NB;
    const bool owner = threadIdx.x < NOWN;
    const uint32_t rg = NOWN - 1 - threadIdx.x;
    uint32_t loc[PERB];
#pragma unroll
    for (int i = 0; i < PERB; ++i) loc[i] = owner ? hist[rg * PERB + i] : 0u;
    row_find_bin_regs<PERB>(loc, NOWN, k, lds, bin, krem);
}
