// Internal interface of the radix select (topk.hip) for the fused compress entry points (compress.hip) and for the
// scorers that accumulate the FIRST histogram pass while they write their scores (rownorm.hip, snapkv.hip).
#pragma once
#include "kvp_common.h"

constexpr int TK_THREADS = 256;
#ifndef KVP_TK_PER
#define KVP_TK_PER 4
#endif
constexpr int TK_PER = KVP_TK_PER;
// 1024 scores per workgroup: the passes are chains of dependent steps (loads, histogram scan, block scans), so many small
// workgroups beat few large ones (8 x 131072, last two passes: 8.2 + 7.6 us with 4 scores per thread, 9.6 + 8.2 with 8,
// 11.0 + 9.1 with 16)
constexpr int TK_CHUNK = TK_THREADS * TK_PER;
// cluster select (topk_cluster.hip): one row cluster of TC_SLOTS workgroups per row, one launch per select.  TC_CLUSTERS clusters (256
// workgroups, one per CU) are resident at once; a launch carries up to TC_MAXC of them (round 6: rows 9 .. 32 of a batch > 1 ride in
// the same launch and become resident in dispatch order as the first clusters retire).
constexpr int TC_CLUSTERS = 8;
constexpr int TC_MAXC = 32;
constexpr int TC_SLOTS = 32;
// two-hop form of the cluster select (topk_cluster.hip): bins of the sample-steered first digit, words per candidate record
constexpr int TC_WB = 256;
constexpr int TC_REC = 128;

struct TopkWs {
    uint32_t* hist1;       // [R][4096]
    uint32_t* hist2;       // [R][4096]
    uint32_t* hist3;       // [R][256]
    uint32_t* bar;         // [TC_MAXC][32] cluster select, one 128-byte line per row cluster: [0] its monotonic arrival counter,
                           // [1] its give-up code (0 = none; persistent until the workspace is zero-filled again)
    uint32_t* histw;       // [R][TC_WB] cluster select: histogram of the sample-steered window digit (zeroed region, self-cleaning)
    uint32_t* cand;        // [R][TC_SLOTS][TC_REC] cluster select: per-slot candidate records (count, keys above the bin, candidate keys)
                           // (both only for row counts the cluster select takes: topk_ws_has_cluster)
    uint32_t* sel;         // [R][4] : b1, k1, b2, k2
    uint32_t* chunk_hist;  // [R][nchunks][257] suffix counts of the last digit: [d] = #(digit >= d), [256] = 0
    uint32_t* chunk_gt;    // [R][nchunks]
    size_t zero_bytes;     // leading bytes that must be zeroed per call (hist1..hist3)
    uint32_t kmask;        // XOR-ed into every key: 0 = k largest, 0xFFFFFFFF = k smallest
    size_t total_bytes;
    uint32_t* base;        // the workspace itself and the (R, ntab) its layout was computed for (topk_ws_layout)
    uint32_t lay_R, lay_ntab;
};

// The workspace layout in 4-byte words from its base (every region starts on a 256-byte boundary).  __host__ __device__: the
// cluster select carries only the base and (R, ntab) and derives a region's address where it uses it -- nine pointers in scalar
// registers for the whole kernel were what pushed it over the scalar register file.
struct TopkWsLayout {
    uint32_t hist1, hist2, hist3, bar, histw, sel, chunk_hist, chunk_gt, cand;
    uint32_t zero_words, total_words;
};
// the cluster select's own regions (window histograms, candidate records) exist only for row counts it can run on: the segmented /
// per-chunk selects hand over thousands of short rows and would zero-fill and carry them for nothing.  A function of R
// alone: the leading zero-filled part of the layout must not depend on the row length (compress.hip sizes it before it knows ntab).
__host__ __device__ inline bool topk_ws_has_cluster(uint32_t R) { return R <= (uint32_t)TC_MAXC; }
// the layout in 64-bit arithmetic: the same regions as topk_ws_layout below, only the total (host side: size queries, range check)
inline uint64_t topk_ws_total_words64(uint64_t R, uint64_t ntab) {
    auto up = [](uint64_t w) { return (w + 63u) / 64u * 64u; };
    const bool cl = R <= (uint64_t)TC_MAXC;
    return 2 * up(R * 4096u) + up(R * 256u) + up((uint64_t)(TC_MAXC * 32 + 32 + TC_SLOTS * 32)) + (cl ? up(R * TC_WB) : 0) + up(R * 4u) +
           up(R * ntab * 257u) + up(R * ntab) + (cl ? up(R * (uint64_t)(TC_SLOTS * TC_REC)) : 0);
}
__host__ __device__ inline TopkWsLayout topk_ws_layout(uint32_t R, uint32_t ntab) {
    TopkWsLayout l;
    uint32_t off = 0;
    auto take = [&](uint32_t words) {
        const uint32_t at = off;
        off += (words + 63u) / 64u * 64u;
        return at;
    };
    const bool cl = topk_ws_has_cluster(R);
    l.hist1 = take(R * 4096u);
    l.hist2 = take(R * 4096u);
    l.hist3 = take(R * 256u);
    l.bar = take((uint32_t)(TC_MAXC * 32 + 32 + TC_SLOTS * 32));   // (the tail: form markers of the test twin / phase stamps of tools/make_tc_timing.py's lab build)
    l.histw = take(cl ? R * (uint32_t)TC_WB : 0u);
    l.zero_words = off;
    l.sel = take(R * 4u);
    // per-chunk tables: the (chunk, row) passes index them by 1024-score chunk, the cluster select by its TC_SLOTS slots
    l.chunk_hist = take(R * ntab * 257u);
    l.chunk_gt = take(R * ntab);
    l.cand = take(cl ? R * (uint32_t)(TC_SLOTS * TC_REC) : 0u);
    l.total_words = off;
    return l;
}

inline TopkWs topk_carve_ws(void* ws, int64_t R, int64_t nchunks) {
    TopkWs w;
    const uint32_t ntab = (uint32_t)std::max<int64_t>(nchunks, TC_SLOTS);
    // (callers check topk_ws_total_words64(R, ntab) < 2^32 before they LAUNCH on a carved workspace: topk_select_impl)
    const TopkWsLayout l = topk_ws_layout((uint32_t)R, ntab);
    uint32_t* base = static_cast<uint32_t*>(ws);
    auto at = [&](uint32_t words) { return base ? base + words : nullptr; };
    w.hist1 = at(l.hist1);
    w.hist2 = at(l.hist2);
    w.hist3 = at(l.hist3);
    w.bar = at(l.bar);
    w.histw = at(l.histw);
    w.zero_bytes = (size_t)l.zero_words * 4;
    w.sel = at(l.sel);
    w.chunk_hist = at(l.chunk_hist);
    w.chunk_gt = at(l.chunk_gt);
    w.cand = at(l.cand);
    w.total_bytes = (size_t)topk_ws_total_words64((uint64_t)R, ntab) * 4;   // 64-bit: a size query never wraps
    w.kmask = 0;
    w.base = base;
    w.lay_R = (uint32_t)R;
    w.lay_ntab = ntab;
    return w;
}


// ---- pass-1 histogram inside a score-producing kernel -------------------------------------------------------------
// lds_hist: 4096 words, zeroed (and __syncthreads()-ed) by the caller; every thread of the wave must call add()
// together (`valid` = this lane has a score).  Scores of one layer crowd into a handful of the 4096 (sign, exponent,
// 3 mantissa bits) bins, so plain LDS atomics serialise; two rounds of wave-level aggregation take the two most
// common bins out first.
template <int ROUNDS = 2>
__device__ __forceinline__ void topk_hist_add_bin(uint32_t* lds_hist, uint32_t bin, bool valid) {
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const uint64_t todo = __ballot(valid);
        if (todo == 0) return;
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);  // leader is wave-uniform: no LDS round trip (ds_bpermute)
        const uint64_t same = __ballot(valid && bin == b0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lds_hist[b0], (uint32_t)__popcll(same));
        valid = valid && bin != b0;
    }
    if (valid) atomicAdd(&lds_hist[bin], 1u);
}
__device__ __forceinline__ void topk_hist1_add(uint32_t* lds_hist, float score, bool valid) {
    topk_hist_add_bin(lds_hist, float_to_key(score) >> 20, valid);
}
// after a __syncthreads(): add the block's counts to the row's global histogram
__device__ __forceinline__ void topk_hist1_flush(const uint32_t* lds_hist, uint32_t* __restrict__ hist1_row) {
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) {
        const uint32_t c = lds_hist[i];
        if (c) atomicAdd(&hist1_row[i], c);
    }
}

// Radix select of the k largest of scores[r, 0..S) per row r into idx[r * idx_stride + 0..k), ascending positions,
// followed by tail_n extra indices tail_start, tail_start + 1, ... (pad columns that are kept by construction).
// hist1_ready: w.hist1 already holds this call's first-pass histogram (a fused scorer produced it).
int topk_select_impl(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int32_t* idx, int64_t idx_stride,
                     uint32_t tail_start, uint32_t tail_n, void* ws, size_t ws_bytes, bool ws_clean, bool hist1_ready,
                     hipStream_t stream, uint32_t nseg = 1, uint32_t seg_len = 0, uint32_t pos_base = 0, bool smallest = false);
// S this short: one launch, one workgroup per row, no workspace
bool topk_row_eligible(int64_t S);
// Cluster select (topk_cluster.hip): the whole select of up to 32 rows of 16385 .. 262144 scores in ONE launch (8 rows resident at once).  mode: where the keys
// come from -- the score rows, SnapKV's un-pooled column sums (avg_pool1d of width 5 + scale `inv` in the loader), or
// -||x[b,h,s,:]|| computed from 256-byte rows of a 2-byte dtype (fused Knorm compress).  Returns KVP_OK, an error code, or
// 1 = not launched (the device cannot hold the 256 workgroups at once): the caller falls back to the (chunk, row) passes.
enum { TOPK_CLUSTER_SCORES = 0, TOPK_CLUSTER_POOL5 = 1, TOPK_CLUSTER_KNORM = 2 };
bool topk_cluster_eligible(int64_t R, int64_t S);
bool topk_cluster_launchable();   // the current device can hold the cluster kernel's 256 workgroups at once
int topk_cluster_select(int mode, const float* scores, int64_t row_stride, float inv, const void* x, int dtype, int64_t x_sb, int64_t x_sh,
                        int64_t x_ss, int64_t H, float scale, int64_t R, int64_t S, int64_t k, int32_t* idx, int64_t idx_stride,
                        uint32_t tail_start, uint32_t tail_n, const TopkWs& w, bool hist1_ready, hipStream_t stream, uint32_t nseg = 1,
                        uint32_t seg_len = 0, uint32_t pos_base = 0);
// should a score-writing kernel accumulate the first histogram for a select over S columns?  (S > 16384)
bool topk_fused_hist_wanted(int64_t S);
// select from un-pooled SnapKV column sums (kernel_size 5 pooling + scale `inv` inside the loader); rows as above
bool topk_pooled_rows_eligible(int64_t Sm, int kernel_size);
int topk_select_pooled_rows(const float* colsum, int64_t R, int64_t Sm, float inv, int64_t k, int32_t* idx, int64_t idx_stride,
                            uint32_t tail_start, uint32_t tail_n, hipStream_t stream);
// nseg > 1: rows are (outer row, segment) pairs and every reported position gets (row % nseg) * seg_len + pos_base added

// ---- KVP_ORDER_SCORE: the selection in descending-score order (topk_order.hip: a hand-written segmented sort) -------------------------
// idx [R][k] holds a select's ascending positions and is rewritten in place.  mode 0: `scores` are score rows of S columns
// (row_stride); mode 1: SnapKV's un-pooled column sums [R][ncols] (row_stride = ncols; 5-tap average * inv recomputed per kept
// position).  Positions ncols .. S - 1 (ncols < 0: none) count as the largest scores, by position.
size_t topk_order_workspace_bytes(int64_t R, int64_t k);
int topk_order_by_score(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int32_t* idx, bool smallest, void* ws,
                        size_t ws_bytes, hipStream_t stream, int mode = 0, int64_t ncols = -1, float inv = 1.f);
