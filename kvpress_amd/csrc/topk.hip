// kvp_topk_select: indices of the k largest scores per row, ties -> lowest position, -0.0 == +0.0.
// Replaces `scores.topk(n_kept, dim=-1).indices` (kvpress/presses/scorer_press.py:95).
//
// The score matrix is tiny (B*H_kv rows x S floats: 4 MiB at 8 x 131072) and L2-resident, but it
// has FEW rows, so one-workgroup-per-row would leave 248 of 256 CUs idle.  Every pass is therefore
// a (chunk, row) grid of 2048-element chunks and rows are combined through global histograms:
//
//   K1 hist<12 bits>  : histogram of key>>20 per row                    (key = order-preserving u32)
//   K2 hist<12 bits>  : find digit b1 holding the k-th largest; histogram of (key>>8)&0xFFF among key>>20==b1
//   K3 hist< 8 bits>  : find b2; per-CHUNK histogram of key&0xFF among key>>8==prefix24 (+ global one)
//                       and per-chunk count of keys with a larger 24-bit prefix
//   K4 write          : find b3 -> threshold T and tie quota q (= how many keys == T to keep, lowest
//                       positions first); each chunk derives its output offset from the per-chunk
//                       tables of the chunks before it, scans its keep flags and writes positions
//                       in ascending order.
//
// The row histograms are SELF-CLEANING: K4 zeroes hist1/hist2 (nobody reads them after K3) and K2 zeroes
// hist3 (filled by K3, read by K4), so a workspace that was zero once stays valid call after call and
// no per-call memset is needed (kvp_topk_select takes a `ws_is_clean` flag; the binding caches a zeroed
// workspace per device and stream).  K1 can be skipped altogether when the kernel that WROTE the scores already
// accumulated hist1 (topk_internal.h: topk_hist1_add / topk_hist1_flush; used by the fused compress entry points).
//
// Kernel boundaries order the passes (1.5-1.9 us each on MI355X).  A single cooperative launch with two grid-wide
// barriers (arrive counter + spin, agent-scope fences = L2 write-back / invalidate on every one of the 8 XCDs) was
// measured at 51 us (128 workgroups) to 131 us (512 workgroups) against 26-32 us for the three launches: not an option;
// nor are direct global atomics for the (sparse) second-pass histogram: 260 k of them take 48 us against 14 us for the
// workgroup-private LDS histograms + flush; nor is ONE launch with one 1024-thread workgroup per row that streams its row
// from L2 once per digit and once for the compaction (round 2: 91 us at 8 x 131072, 29 us at 8 x 32768 against 31 / 23 us
// for the three launches -- a single CU pulls ~25 GB/s through such a dependent walk);
// the only inter-workgroup traffic inside a launch is atomicAdd into the row histograms.
//
// SHORT rows (S <= 32768: BASELINE config 2, decode-time compression, per-chunk / per-block selection, short prompts) are launch-bound in
// that scheme (3-4 dependent launches of ~5 us each), so they take ONE launch instead: topk_row_kernel, one 1024-thread
// workgroup per row, the row's keys in registers, three digit passes (8 + 12 + 12 bits) on an LDS histogram, same tie rule.
#include "kvp_common.h"
#include "topk_internal.h"
#include "topk_block.h"

namespace {

// exclusive prefix sum over the 256 threads of the block (4 waves); lds: >= 4 words
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);   // DPP row shifts / broadcasts (topk_block.h)
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TK_THREADS / 64; ++i) {
        const uint32_t x = lds[i];
        if (i < w) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// Find the digit bin that holds the k-th largest element (k >= 1) of a NB-bin histogram:
// count(d > bin) < k <= count(d >= bin);  krem = k - count(d > bin).   lds: >= 8 words.
template <int NB>
__device__ __forceinline__ void find_bin(const uint32_t* __restrict__ hist, uint32_t k, uint32_t* lds, uint32_t& bin,
                                         uint32_t& krem) {
    constexpr int PER = NB / TK_THREADS;
    const uint32_t rg = TK_THREADS - 1 - threadIdx.x;  // thread 0 owns the highest bins
    uint32_t loc[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        loc[i] = hist[rg * PER + i];
        sum += loc[i];
    }
    uint32_t total;
    const uint32_t excl = block_excl_scan(sum, lds, &total);  // # elements in bins above mine
    if (excl < k && k <= excl + sum) {
        uint32_t c = excl;
#pragma unroll
        for (int i = PER - 1; i >= 0; --i) {
            if (k > c && k <= c + loc[i]) {
                lds[4] = rg * PER + i;
                lds[5] = k - c;
            }
            c += loc[i];
        }
    }
    __syncthreads();
    bin = lds[4];
    krem = lds[5];
    __syncthreads();
}

__device__ __forceinline__ uint32_t load_key(const float* __restrict__ row, uint32_t i, uint32_t S, bool& valid) {
    valid = i < S;
    return valid ? float_to_key(row[i]) : 0u;
}

// ---- K1 / K2: 12-bit histograms ---------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(TK_THREADS) void topk_hist12_kernel(const float* __restrict__ scores, int64_t row_stride,
                                                                 uint32_t S, uint32_t k, TopkWs w) {
    __shared__ uint32_t lh[4096];
    __shared__ uint32_t scr[8];
    const uint32_t row = blockIdx.y, chunk = blockIdx.x;
    const float* rp = scores + (int64_t)row * row_stride;
    // the scores are requested first so that their latency overlaps the histogram scan below
    const uint32_t base = chunk * TK_CHUNK;
    uint32_t keys[TK_PER];
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const uint32_t i = base + j * TK_THREADS + threadIdx.x;
        keys[j] = i < S ? (float_to_key(rp[i]) ^ w.kmask) : 0u;
    }
    for (int i = threadIdx.x; i < 4096; i += TK_THREADS) lh[i] = 0;
    uint32_t b1 = 0, k1 = 0;
    if (PASS == 2) {
        if (chunk == 0) w.hist3[(size_t)row * 256 + threadIdx.x] = 0;  // self-cleaning: filled by K3, read by K4
        find_bin<4096>(w.hist1 + (size_t)row * 4096, k, scr, b1, k1);
        if (chunk == 0 && threadIdx.x == 0) {
            w.sel[row * 4 + 0] = b1;
            w.sel[row * 4 + 1] = k1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const bool valid = base + j * TK_THREADS + threadIdx.x < S;
        if (PASS == 1) {
            topk_hist_add_bin(lh, keys[j] >> 20, valid);
        } else {
            if (valid && (keys[j] >> 20) == b1) atomicAdd(&lh[(keys[j] >> 8) & 0xFFFu], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = (PASS == 1 ? w.hist1 : w.hist2) + (size_t)row * 4096;
    for (int i = threadIdx.x; i < 4096; i += TK_THREADS) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// ---- K3: 8-bit histogram of the last digit, per chunk and per row ----------------------------
__global__ __launch_bounds__(TK_THREADS) void topk_hist8_kernel(const float* __restrict__ scores, int64_t row_stride,
                                                                uint32_t S, uint32_t nchunks, TopkWs w) {
    __shared__ uint32_t lh[256];
    __shared__ uint32_t scr[8];
    const uint32_t row = blockIdx.y, chunk = blockIdx.x;
    const float* rp = scores + (int64_t)row * row_stride;
    const uint32_t base = chunk * TK_CHUNK;
    uint32_t keys[TK_PER];
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const uint32_t i = base + j * TK_THREADS + threadIdx.x;
        keys[j] = i < S ? (float_to_key(rp[i]) ^ w.kmask) : 0u;
    }
    lh[threadIdx.x] = 0;
    const uint32_t b1 = w.sel[row * 4 + 0], k1 = w.sel[row * 4 + 1];
    uint32_t b2, k2;
    find_bin<4096>(w.hist2 + (size_t)row * 4096, k1, scr, b2, k2);
    if (chunk == 0 && threadIdx.x == 0) {
        w.sel[row * 4 + 2] = b2;
        w.sel[row * 4 + 3] = k2;
    }
    const uint32_t prefix = (b1 << 12) | b2;
    __syncthreads();
    uint32_t ngt = 0;
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const bool valid = base + j * TK_THREADS + threadIdx.x < S;
        const uint32_t p = keys[j] >> 8;
        if (valid && p > prefix) ++ngt;
        if (valid && p == prefix) atomicAdd(&lh[keys[j] & 0xFFu], 1u);
    }
    uint32_t tot;
    block_excl_scan(ngt, scr, &tot);  // contains the barrier that also covers the LDS atomics
    // thread t owns bin 255 - t: the exclusive scan over threads counts the keys in HIGHER bins, so
    // suffix[d] = #(last digit >= d) among this chunk's prefix-matching keys
    const uint32_t d = 255u - threadIdx.x;
    const uint32_t c = lh[d];
    uint32_t tot2;
    const uint32_t above = block_excl_scan(c, scr, &tot2);
    w.chunk_hist[((size_t)row * nchunks + chunk) * 257 + d] = above + c;
    if (threadIdx.x == 0) {
        w.chunk_hist[((size_t)row * nchunks + chunk) * 257 + 256] = 0;  // suffix[256]
        w.chunk_gt[(size_t)row * nchunks + chunk] = tot;
    }
    if (c) atomicAdd(&w.hist3[(size_t)row * 256 + d], c);
}

// ---- K4: ordered compaction --------------------------------------------------------------------
__global__ __launch_bounds__(TK_THREADS) void topk_write_kernel(const float* __restrict__ scores, int64_t row_stride,
                                                                uint32_t S, uint32_t k, uint32_t nchunks, TopkWs w,
                                                                int32_t* __restrict__ idx, int64_t idx_stride,
                                                                uint32_t tail_start, uint32_t tail_n, uint32_t nseg,
                                                                uint32_t seg_len, uint32_t pos_base) {
    __shared__ uint32_t scr[8];
    const uint32_t row = blockIdx.y, chunk = blockIdx.x;
    const float* rp = scores + (int64_t)row * row_stride;
    // this thread's TK_PER consecutive positions (requested before the threshold is resolved)
    const uint32_t p0 = chunk * TK_CHUNK + threadIdx.x * TK_PER;
    uint32_t keys[TK_PER];
    const bool fast = (p0 + TK_PER <= S) && ((((uintptr_t)(rp + p0)) & 15u) == 0);
    if (fast) {
#pragma unroll
        for (int q = 0; q < TK_PER / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(rp + p0 + 4 * q);
            keys[4 * q + 0] = float_to_key(a.x) ^ w.kmask; keys[4 * q + 1] = float_to_key(a.y) ^ w.kmask;
            keys[4 * q + 2] = float_to_key(a.z) ^ w.kmask; keys[4 * q + 3] = float_to_key(a.w) ^ w.kmask;
        }
    } else {
#pragma unroll
        for (int j = 0; j < TK_PER; ++j) keys[j] = (p0 + j < S) ? (float_to_key(rp[p0 + j]) ^ w.kmask) : 0u;
    }
    const uint32_t b1 = w.sel[row * 4 + 0], b2 = w.sel[row * 4 + 2], k2 = w.sel[row * 4 + 3];
    uint32_t b3, quota;
    find_bin<256>(w.hist3 + (size_t)row * 256, k2, scr, b3, quota);
    const uint32_t T = (((b1 << 12) | b2) << 8) | b3;

    // kept elements in the chunks before this one, from their suffix tables: thread j handles chunk j
    uint32_t gt_part = 0, eq_part = 0;
    for (uint32_t j = threadIdx.x; j < chunk; j += TK_THREADS) {
        const uint32_t* sf = w.chunk_hist + ((size_t)row * nchunks + j) * 257;
        const uint32_t ge = sf[b3], gt = sf[b3 + 1];
        gt_part += w.chunk_gt[(size_t)row * nchunks + j] + gt;
        eq_part += ge - gt;
    }
    uint32_t gt_before, eq_before;
    block_excl_scan(gt_part, scr, &gt_before);
    block_excl_scan(eq_part, scr, &eq_before);

    // self-cleaning: hist1 / hist2 of this row are dead now (hist3 is zeroed by the next call's K1)
    for (uint32_t i = chunk * TK_THREADS + threadIdx.x; i < 4096; i += nchunks * TK_THREADS) {
        w.hist1[(size_t)row * 4096 + i] = 0;
        w.hist2[(size_t)row * 4096 + i] = 0;
    }

    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const bool valid = p0 + j < S;
        cg += (valid && keys[j] > T) ? 1u : 0u;
        ce += (valid && keys[j] == T) ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cg | (ce << 16), scr, &tot);  // chunk <= 4096: both fields < 65536
    uint32_t g = gt_before + (ex & 0xFFFFu);
    uint32_t e = eq_before + (ex >> 16);
    int32_t* out = idx + (int64_t)row * idx_stride;
    // segmented select: row = (outer row, segment); positions are reported relative to the outer row
    const uint32_t off = pos_base + (nseg > 1 ? (row % nseg) * seg_len : 0u);
    if (chunk == 0)  // columns kept by construction (pad region after the selected ones)
        for (uint32_t j = threadIdx.x; j < tail_n; j += TK_THREADS) out[k + j] = (int32_t)(off + tail_start + j);
#pragma unroll
    for (int j = 0; j < TK_PER; ++j) {
        const bool valid = p0 + j < S;
        const bool isg = valid && keys[j] > T;
        const bool ise = valid && keys[j] == T;
        if (isg || (ise && e < quota)) {
            const uint32_t rank = g + (e < quota ? e : quota);
            if (rank < k) out[rank] = (int32_t)(off + p0 + j);
        }
        g += isg ? 1u : 0u;
        e += ise ? 1u : 0u;
    }
}

// ---- one workgroup per row (short rows): block scan / digit search in topk_block.h ----------------------------

// ---- K2 with wide workgroups --------------------------------------------------------------------------------------
// The second 12-bit histogram only counts the keys of the threshold's first-digit bin.  Flat score rows (SnapKV averages
// over 256 window rows x 5 pooled keys: half of a row sits in that bin) spread those candidates evenly over the 4096 second
// digits, so a 2048-key workgroup flushes ~900 different bins with one global atomic each (~460 k atomics at 8 x 131072: most
// of the pass's 15.6 us).  1024 threads x PER keys aggregate PER / 2 times as many candidates per bin before the flush.
template <int PER>
__global__ __launch_bounds__(TR_THREADS) void topk_hist12_wide_kernel(const float* __restrict__ scores, int64_t row_stride, uint32_t S,
                                                                      uint32_t k, TopkWs w) {
    __shared__ uint32_t lh[4096];
    __shared__ uint32_t scr[TR_WAVES + 2];
    const uint32_t row = blockIdx.y, base = blockIdx.x * (TR_THREADS * PER);
    const float* rp = scores + (int64_t)row * row_stride;
    uint32_t keys[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = base + j * TR_THREADS + threadIdx.x;
        keys[j] = i < S ? (float_to_key(rp[i]) ^ w.kmask) : 0u;
    }
    for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 256) w.hist3[(size_t)row * 256 + threadIdx.x] = 0;  // self-cleaning: filled by K3, read by K4
    uint32_t b1, k1;
    row_find_bin<4096>(w.hist1 + (size_t)row * 4096, k, scr, b1, k1);   // (its barriers also publish the zeroed histogram)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.sel[row * 4 + 0] = b1;
        w.sel[row * 4 + 1] = k1;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool valid = base + j * TR_THREADS + threadIdx.x < S;
        if (valid && (keys[j] >> 20) == b1) atomicAdd(&lh[(keys[j] >> 8) & 0xFFFu], 1u);
    }
    __syncthreads();
    uint32_t* gh = w.hist2 + (size_t)row * 4096;
    for (int i = threadIdx.x; i < 4096; i += TR_THREADS) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// thread t owns the PER consecutive positions t * PER ..: S <= 1024 * PER.
// PAD >= 0 (fused SnapKV compress): `scores` holds the un-pooled column sums and the score of position p is
// inv * (x[p-PAD] + ... + x[p+PAD]) with zeros outside the row -- snapkv_pool_kernel's arithmetic, term for term -- so the
// pooling launch and the score round trip through memory disappear.
// HIST1 (fused compress, rows of 16385 .. 32768 scores): the kernel that wrote the scores already accumulated the histogram of
// the first 12-bit digit (key >> 20) in hist1[row][4096] (topk_internal.h); the digits are then 12 + 12 + 8 bits as in the
// multi-workgroup passes and only the ~5-10 % of the keys that share the threshold's first digit touch the LDS histogram
// again (with the kernel's own 8-bit first digit nearly every key of a layer does).  Measured at 8 x 32768 (Knorm, config 2):
// 19.1 us against 20.7 us with the kernel's own digits and 22.9 us for the three (chunk, row) launches; of the 19 us, ~4 are the
// launch, ~3.5 the key loads, 2 per digit (block-wide scans over the histogram) and 5.6 the ordered compaction.
template <int PER, int PAD, bool HIST1 = false>
__global__ __launch_bounds__(TR_THREADS) void topk_row_kernel(const float* __restrict__ scores, int64_t row_stride, uint32_t S, uint32_t k,
                                                              uint32_t kmask, float inv, int32_t* __restrict__ idx, int64_t idx_stride,
                                                              uint32_t tail_start, uint32_t tail_n, uint32_t nseg, uint32_t seg_len,
                                                              uint32_t pos_base, uint32_t* __restrict__ hist1 = nullptr) {
    // histogram, then the staged output; with PAD >= 0 first the staged input row (one pad word per 16: conflict-free reads)
    __shared__ uint32_t lh[TR_THREADS * PER + (PAD >= 0 ? TR_THREADS * PER / 16 + 1 : 0) > 4096 ? TR_THREADS * PER + (PAD >= 0 ? TR_THREADS * PER / 16 + 1 : 0) : 4096];
    __shared__ uint32_t scr[TR_WAVES + 2];
    const uint32_t row = blockIdx.x;
    const float* rp = scores + (int64_t)row * row_stride;
    const uint32_t p0 = threadIdx.x * PER;
    // positions past S carry key 0 and real keys are >= 1 (the lowest two NaN encodings share key 1): the padding sits at
    // the bottom of every histogram, where a select of k <= S never reaches, so T >= 1 and no range checks are needed below
    uint32_t keys[PER];
    if (PAD >= 0) {
        // the row goes through LDS: coalesced global loads, then every thread reads its PER + 2 PAD consecutive inputs
        // (direct loads would be 4-byte reads PER words apart across the wave, one cache line per lane and instruction)
        constexpr int NIN = PER + 2 * (PAD > 0 ? PAD : 0);
        float* st = reinterpret_cast<float*>(lh);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t e = i * TR_THREADS + threadIdx.x;
            if (e < S) st[e + (e >> 4)] = rp[e];
        }
        __syncthreads();
        float in[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int32_t pos = (int32_t)p0 + i - PAD;
            const uint32_t inside = (uint32_t)(((pos - (int32_t)S) >> 31) & ~(pos >> 31));   // all ones iff 0 <= pos < S
            const uint32_t c = (uint32_t)min(max(pos, 0), (int32_t)S - 1);
            in[i] = __uint_as_float(__float_as_uint(st[c + (c >> 4)]) & inside);
        }
        __syncthreads();  // lh is the histogram from here on
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int d = 0; d <= 2 * (PAD > 0 ? PAD : 0); ++d) sum += in[j + d];
            sum *= inv;
            const uint32_t inside = (uint32_t)((int32_t)(p0 + j - S) >> 31);
            keys[j] = max(float_to_key(sum) ^ kmask, 1u) & inside;
        }
    } else if (PER % 4 == 0 && p0 + PER <= S && ((((uintptr_t)(rp + p0)) & 15u) == 0)) {
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(rp + p0 + 4 * q);
            keys[4 * q + 0] = max(float_to_key(a.x) ^ kmask, 1u); keys[4 * q + 1] = max(float_to_key(a.y) ^ kmask, 1u);
            keys[4 * q + 2] = max(float_to_key(a.z) ^ kmask, 1u); keys[4 * q + 3] = max(float_to_key(a.w) ^ kmask, 1u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) {  // clamped address + arithmetic mask: no per-element lane masks held in SGPRs
            const uint32_t inside = (uint32_t)((int32_t)(p0 + j - S) >> 31);  // all ones iff p0 + j < S  (S < 2^31)
            keys[j] = max(float_to_key(rp[min(p0 + j, S - 1)]) ^ kmask, 1u) & inside;
        }
    }
    // Digits of 8 + 12 + 12 bits on an LDS histogram.  The kernel is bound by instruction issue (16 waves on 4 SIMDs), so
    // a thread whose PER keys all fall into ONE bin adds them with a single weighted atomic: the first digit (sign + 7
    // exponent bits) is then one wave-aggregated add per wave for typical scores, and rows of equal scores never
    // serialise on one LDS address.  Everything else takes one plain LDS atomic per key.
    const bool full = p0 + PER <= S;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        kmin = min(kmin, keys[j]);
        kmax = max(kmax, keys[j]);
    }
    uint32_t T, quota;
    if (HIST1) {
        // digit 1 (key >> 20) from the score-writing kernel; self-cleaning: the next fused call accumulates into zeros
        uint32_t* gh = hist1 + (size_t)row * 4096;
#pragma unroll
        for (int i = 0; i < 4096 / TR_THREADS; ++i) {
            lh[threadIdx.x + i * TR_THREADS] = gh[threadIdx.x + i * TR_THREADS];
            gh[threadIdx.x + i * TR_THREADS] = 0;
        }
        __syncthreads();
        uint32_t b1, k1;
        row_find_bin<4096>(lh, k, scr, b1, k1);
        for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
        __syncthreads();
        if (full && (kmin >> 8) == (kmax >> 8)) {
            if ((kmin >> 20) == b1) atomicAdd(&lh[(kmin >> 8) & 0xFFFu], (uint32_t)PER);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if ((keys[j] >> 20) == b1 && keys[j]) atomicAdd(&lh[(keys[j] >> 8) & 0xFFFu], 1u);   // (key 0 = padding past S)
        }
        __syncthreads();
        uint32_t b2, k2;
        row_find_bin<4096>(lh, k1, scr, b2, k2);
        const uint32_t prefix = (b1 << 12) | b2;
        if (threadIdx.x < 256) lh[threadIdx.x] = 0;
        __syncthreads();
        if (full && kmin == kmax) {
            if ((kmin >> 8) == prefix) atomicAdd(&lh[kmin & 0xFFu], (uint32_t)PER);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if ((keys[j] >> 8) == prefix && keys[j]) atomicAdd(&lh[keys[j] & 0xFFu], 1u);
        }
        __syncthreads();
        uint32_t b3;
        row_find_bin<256>(lh, k2, scr, b3, quota);
        T = (prefix << 8) | b3;
    } else {
    if (threadIdx.x < 256) lh[threadIdx.x] = 0;
    __syncthreads();
    {
        const bool one = full && (kmin >> 24) == (kmax >> 24);
        // (inline weighted form of topk_hist_add_bin<1>: the leader adds PER per matching lane)
        const uint64_t todo = __ballot(one);
        bool left = one;
        if (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)(kmin >> 24), leader);
            const uint64_t same = __ballot(one && (kmin >> 24) == b0);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[b0], (uint32_t)__popcll(same) * PER);
            left = one && (kmin >> 24) != b0;
        }
        if (left) atomicAdd(&lh[kmin >> 24], (uint32_t)PER);
        if (!one) {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                atomicAdd(&lh[keys[j] >> 24], 1u);
        }
    }
    __syncthreads();
    uint32_t b1, k1;
    row_find_bin<256>(lh, k, scr, b1, k1);
    for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
    __syncthreads();
    if (full && (kmin >> 12) == (kmax >> 12)) {
        if ((kmin >> 24) == b1) atomicAdd(&lh[(kmin >> 12) & 0xFFFu], (uint32_t)PER);
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if ((keys[j] >> 24) == b1) atomicAdd(&lh[(keys[j] >> 12) & 0xFFFu], 1u);
    }
    __syncthreads();
    uint32_t b2, k2;
    row_find_bin<4096>(lh, k1, scr, b2, k2);
    const uint32_t prefix = (b1 << 12) | b2;
    for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
    __syncthreads();
    if (full && kmin == kmax) {
        if ((kmin >> 12) == prefix) atomicAdd(&lh[kmin & 0xFFFu], (uint32_t)PER);
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if ((keys[j] >> 12) == prefix) atomicAdd(&lh[keys[j] & 0xFFFu], 1u);
    }
    __syncthreads();
    uint32_t b3;
    row_find_bin<4096>(lh, k2, scr, b3, quota);
    T = (prefix << 12) | b3;

    }

    // ordered compaction: keys > T, and the first `quota` keys == T
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        cg += keys[j] > T ? 1u : 0u;
        ce += keys[j] == T ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t ex = row_excl_scan(cg | (ce << 16), scr, &tot);  // S <= 32768: both fields < 65536
    uint32_t g = ex & 0xFFFFu, e = ex >> 16;
    int32_t* out = idx + (int64_t)row * idx_stride;
    const uint32_t off = pos_base + (nseg > 1 ? (row % nseg) * seg_len : 0u);
    for (uint32_t j = threadIdx.x; j < tail_n; j += TR_THREADS) out[k + j] = (int32_t)(off + tail_start + j);
    // the kept positions are ranked into LDS (the histogram is dead) and leave as one coalesced stream: a thread's own
    // ranks are consecutive, so direct global stores would be 4-byte writes ~PER/2 words apart across the wave
    int32_t* ob = reinterpret_cast<int32_t*>(lh);
    // (the threshold through an SGPR: otherwise the 2 x PER lane masks of the counting loop above are kept alive across
    // the scan for reuse here, which spills SGPRs)
    const uint32_t Ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool isg = keys[j] > Ts;
        const bool ise = keys[j] == Ts;
        if (isg || (ise && e < quota)) {
            const uint32_t rank = g + (e < quota ? e : quota);
            if (rank < k) ob[rank] = (int32_t)(off + p0 + j);
        }
        g += isg ? 1u : 0u;
        e += ise ? 1u : 0u;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < k; i += TR_THREADS) out[i] = ob[i];
}

// k == S: every position is kept (compression_ratio so small that int(S*(1-r)) == S)
// (also the k == 0 case of a call with a tail: only the tail columns are written)
__global__ void topk_iota_kernel(int32_t* __restrict__ idx, int64_t idx_stride, uint32_t k, uint32_t tail_start, uint32_t tail_n,
                                 uint32_t nseg, uint32_t seg_len, uint32_t pos_base) {
    int32_t* out = idx + (int64_t)blockIdx.y * idx_stride;
    const uint32_t off = pos_base + (nseg > 1 ? (blockIdx.y % nseg) * seg_len : 0u);
    const uint32_t n = k + tail_n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (int32_t)(off + (i < k ? i : tail_start + (i - k)));
}

}  // namespace


extern "C" size_t kvp_topk_order_workspace_bytes(int64_t R, int64_t S, int64_t k) {
    return kvp_topk_workspace_bytes(R, S, k) + topk_order_workspace_bytes(R, k);
}

extern "C" size_t kvp_topk_workspace_bytes(int64_t R, int64_t S, int64_t k) {
    (void)k;
    if (R <= 0 || S <= 0) return 256;
    const int64_t nchunks = (S + TK_CHUNK - 1) / TK_CHUNK;
    return topk_carve_ws(nullptr, R, nchunks).total_bytes;
}

// rows this short are selected by one workgroup each (topk_row_kernel); the scorers then skip their fused histogram
// should the kernel that writes the scores accumulate the first 12-bit histogram?  Not for rows of <= 16384 scores (the plain
// one-launch select with its own digits is faster there); longer: the (chunk, row) passes start at their second pass
bool topk_fused_hist_wanted(int64_t S) { return S > 16384; }

bool topk_row_eligible(int64_t S) {
    return S >= 1 && S <= 32768;  // 1024 threads x 32 keys
}

int topk_select_impl(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int32_t* idx, int64_t idx_stride,
                     uint32_t tail_start, uint32_t tail_n, void* ws, size_t ws_bytes, bool ws_clean, bool hist1_ready,
                     hipStream_t stream, uint32_t nseg, uint32_t seg_len, uint32_t pos_base, bool smallest) {
    if (R == 0 || k + tail_n == 0) return KVP_OK;
    KVP_CHECK_ARG(S < ((int64_t)1 << 31) && R <= 65535, "topk: S=%ld or R=%ld too large", (long)S, (long)R);
    KVP_CHECK_ARG(scores && idx, "topk: null pointer");
    KVP_CHECK_ARG(row_stride >= S && idx_stride >= k + tail_n, "topk: row_stride %ld < S %ld or idx_stride %ld too small", (long)row_stride,
                  (long)S, (long)idx_stride);
    const int64_t nchunks = (S + TK_CHUNK - 1) / TK_CHUNK;
    // the device-side layout is computed in 32-bit words (topk_ws_layout): refuse shapes whose workspace does not fit it
    KVP_CHECK_ARG(topk_ws_total_words64((uint64_t)R, (uint64_t)std::max<int64_t>(nchunks, TC_SLOTS)) < ((uint64_t)1 << 32),
                  "topk: R=%ld x S=%ld needs a workspace beyond 16 GiB (32-bit word offsets)", (long)R, (long)S);
    TopkWs w = topk_carve_ws(ws, R, nchunks);
    w.kmask = smallest ? 0xFFFFFFFFu : 0u;  // the k SMALLEST = the k largest of the complemented order-preserving keys
    if (k == S || k == 0) {  // every position / only the tail is kept: no selection needed
        if (hist1_ready && ws && hipMemsetAsync(w.hist1, 0, (size_t)R * 4096 * 4, stream) != hipSuccess) {  // leave the workspace clean
            kvp_set_error("topk: hipMemsetAsync failed");
            return KVP_EHIP;
        }
        const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((k + tail_n + 255) / 256, 256));
        KVP_LAUNCH("topk_iota_kernel", stream, topk_iota_kernel<<<dim3(bx, (uint32_t)R), 256, 0, stream>>>(idx, idx_stride, (uint32_t)k, tail_start, tail_n, nseg, seg_len, pos_base));
        KVP_CHECK_LAUNCH("topk(iota)");
        return KVP_OK;
    }
    // (A one-launch variant that starts from the fused first-digit histogram up to 32768 scores was measured slower than the passes
    // below -- Knorm config 2: 0.0538 against 0.0525 ms per layer -- and removed in round 5.)
    if (!hist1_ready && topk_row_eligible(S)) {  // short rows: one launch, no workspace
        const uint32_t km = w.kmask;
#define KVP_TR_CASE(P)                                                                                                                   \
    KVP_LAUNCH("topk_row_kernel", stream, (topk_row_kernel<P, -1><<<(uint32_t)R, TR_THREADS, 0, stream>>>(scores, row_stride, (uint32_t)S, (uint32_t)k, km, 1.f, idx, \
                                                                                                           idx_stride, tail_start, tail_n, nseg, seg_len, pos_base)))
        if (S <= 1024) KVP_TR_CASE(1);
        else if (S <= 2048) KVP_TR_CASE(2);
        else if (S <= 4096) KVP_TR_CASE(4);
        else if (S <= 8192) KVP_TR_CASE(8);
        else if (S <= 16384) KVP_TR_CASE(16);
        else KVP_TR_CASE(32);
#undef KVP_TR_CASE
        KVP_CHECK_LAUNCH("topk(row)");
        return KVP_OK;
    }
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("topk: workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    if (!ws_clean) {
        KVP_CHECK_ARG(!hist1_ready, "topk: a fused first pass needs a clean workspace");
        if (hipMemsetAsync(ws, 0, w.zero_bytes, stream) != hipSuccess) {
            kvp_set_error("topk: hipMemsetAsync failed");
            return KVP_EHIP;
        }
    }
    // long rows: the whole select in one launch (topk_cluster.hip); rc 1 = the device cannot hold its grid -> the passes below
    if (topk_cluster_eligible(R, S)) {
        const int rc = topk_cluster_select(TOPK_CLUSTER_SCORES, scores, row_stride, 1.f, nullptr, 0, 0, 0, 0, 1, 0.f, R, S, k, idx, idx_stride, tail_start,
                                           tail_n, w, hist1_ready, stream, nseg, seg_len, pos_base);
        if (rc != 1) return rc;
    }
    const dim3 grid((uint32_t)nchunks, (uint32_t)R);
    if (!hist1_ready)
        KVP_LAUNCH("topk_hist12_kernel", stream, topk_hist12_kernel<1><<<grid, TK_THREADS, 0, stream>>>(scores, row_stride, (uint32_t)S, (uint32_t)k, w));
    // Second pass: 1024-thread workgroups of 8 scores per thread.  Measured at 8 x 131072 on flat SnapKV scores / Knorm norms: this
    // pass's own (chunk, row) grid 15.4-15.8 us, 4 per thread 9.3-9.8, 8: 9.2-9.6, 16: 11.3-11.9, 32: 17.0-17.7 (too few workgroups).
    const dim3 gw((uint32_t)((S + (int64_t)TR_THREADS * 8 - 1) / ((int64_t)TR_THREADS * 8)), (uint32_t)R);
    KVP_LAUNCH("topk_hist12_wide_kernel", stream, topk_hist12_wide_kernel<8><<<gw, TR_THREADS, 0, stream>>>(scores, row_stride, (uint32_t)S, (uint32_t)k, w));
    KVP_LAUNCH("topk_hist8_kernel", stream, topk_hist8_kernel<<<grid, TK_THREADS, 0, stream>>>(scores, row_stride, (uint32_t)S, (uint32_t)nchunks, w));
    KVP_LAUNCH("topk_write_kernel", stream, topk_write_kernel<<<grid, TK_THREADS, 0, stream>>>(scores, row_stride, (uint32_t)S, (uint32_t)k, (uint32_t)nchunks, w, idx, idx_stride, tail_start, tail_n, nseg, seg_len, pos_base));
    KVP_CHECK_LAUNCH("topk");
    return KVP_OK;
}

// Fused SnapKV compress, short rows: select straight from the un-pooled column sums colsum[R][Sm] (avg_pool1d of width 5
// + scale inside the select's loader).  Same indices as pooling first and selecting from the written scores.
// (measured: saves the 5 us pooling launch up to 4096 columns -- 37.5 -> 36 us per SnapKV compress at 4k, 34 -> 32 us at
// 1k; beyond that the loader's extra instructions in this issue-bound kernel cost as much as the launch they save)
bool topk_pooled_rows_eligible(int64_t Sm, int kernel_size) { return kernel_size == 5 && Sm <= 4096 && topk_row_eligible(Sm); }
int topk_select_pooled_rows(const float* colsum, int64_t R, int64_t Sm, float inv, int64_t k, int32_t* idx, int64_t idx_stride,
                            uint32_t tail_start, uint32_t tail_n, hipStream_t stream) {
    if (R == 0 || k + tail_n == 0) return KVP_OK;
    KVP_CHECK_ARG(colsum && idx && R <= 65535 && k >= 0 && k <= Sm && topk_pooled_rows_eligible(Sm, 5), "topk(pooled): bad arguments");
    if (k == Sm || k == 0) return topk_select_impl(colsum, R, Sm, Sm, k, idx, idx_stride, tail_start, tail_n, nullptr, 0, true, false, stream);
#define KVP_TRP_CASE(P)                                                                                                                  \
    KVP_LAUNCH("topk_row_kernel", stream, (topk_row_kernel<P, 2><<<(uint32_t)R, TR_THREADS, 0, stream>>>(colsum, Sm, (uint32_t)Sm, (uint32_t)k, 0u, inv, idx, \
                                                                                                          idx_stride, tail_start, tail_n, 1u, 0u, 0u)))
    if (Sm <= 1024) KVP_TRP_CASE(1);
    else if (Sm <= 2048) KVP_TRP_CASE(2);
    else KVP_TRP_CASE(4);
#undef KVP_TRP_CASE
    KVP_CHECK_LAUNCH("topk(pooled rows)");
    return KVP_OK;
}

extern "C" int kvp_topk_select(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int order,
                               int32_t* idx, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    if (int rc = kvp_async_check("kvp_topk_select")) return rc;
    KVP_CHECK_ARG(R >= 0 && S >= 0 && k >= 0 && k <= S, "topk: bad shape R=%ld S=%ld k=%ld", (long)R, (long)S, (long)k);
    const int ord = order & ~(KVP_TOPK_WS_CLEAN | KVP_TOPK_SMALLEST);
    KVP_CHECK_ARG(ord == KVP_ORDER_POSITION || ord == KVP_ORDER_SCORE, "topk: bad order %d", order);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool smallest = (order & KVP_TOPK_SMALLEST) != 0;
    const size_t sel_bytes = kvp_topk_workspace_bytes(R, S, k);
    if (int rc = topk_select_impl(scores, R, S, row_stride, k, idx, k, 0, 0, ws, std::min(ws_bytes, sel_bytes), (order & KVP_TOPK_WS_CLEAN) != 0,
                                  false, stream, 1, 0, 0, smallest))
        return rc;
    if (ord == KVP_ORDER_SCORE) {  // descending score (ascending for KVP_TOPK_SMALLEST), ties by position: sort the selection
        KVP_CHECK_ARG(ws && ws_bytes >= sel_bytes, "topk: workspace too small for KVP_ORDER_SCORE");
        return topk_order_by_score(scores, R, S, row_stride, k, idx, smallest, static_cast<char*>(ws) + sel_bytes, ws_bytes - sel_bytes, stream);
    }
    return KVP_OK;
}

// Segmented select (ChunkPress, kvpress/presses/chunk_press.py:67-85): every row of scores[R, nseg * seg_len] is cut into
// nseg chunks, the k largest of EACH chunk are selected, idx[R, nseg * k] holds them chunk after chunk as positions
// in the row (+ pos_base) -- ascending overall.  One launch set for all R * nseg chunks.
extern "C" size_t kvp_topk_segmented_workspace_bytes(int64_t R, int64_t nseg, int64_t seg_len, int64_t k) {
    return kvp_topk_workspace_bytes(R * nseg, seg_len, k);
}
extern "C" int kvp_topk_select_segmented(const float* scores, int64_t R, int64_t nseg, int64_t seg_len, int64_t k, int64_t pos_base,
                                         int order, int32_t* idx, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    if (int rc = kvp_async_check("kvp_topk_select_segmented")) return rc;
    KVP_CHECK_ARG(R >= 0 && nseg >= 1 && seg_len >= 1 && k >= 0 && k <= seg_len && pos_base >= 0, "topk_segmented: bad shape R=%ld nseg=%ld seg_len=%ld k=%ld",
                  (long)R, (long)nseg, (long)seg_len, (long)k);
    KVP_CHECK_ARG((order & ~KVP_TOPK_WS_CLEAN) == KVP_ORDER_POSITION, "topk_segmented: only KVP_ORDER_POSITION");
    KVP_CHECK_ARG(R * nseg <= 65535 && pos_base + nseg * seg_len < ((int64_t)1 << 31), "topk_segmented: too many chunks (%ld) or positions", (long)(R * nseg));
    // rows of the flat view: (r, segment) at scores + (r * nseg + segment) * seg_len
    return topk_select_impl(scores, R * nseg, seg_len, seg_len, k, idx, k, 0, 0, ws, ws_bytes, (order & KVP_TOPK_WS_CLEAN) != 0, false,
                            static_cast<hipStream_t>(stream_), (uint32_t)nseg, (uint32_t)seg_len, (uint32_t)pos_base, false);
}

// ---- scores[r, idx[r, j]] = value for j < n (AdaKVPress's safeguard: `scores.scatter_(-1, top_indices, finfo.max)`,
// kvpress/presses/adakv_press.py:62-63) ----------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void fill_at_kernel(float* __restrict__ scores, int64_t row_stride, const int32_t* __restrict__ idx,
                                                      uint32_t n, uint32_t S, float value) {
    float* row = scores + (int64_t)blockIdx.y * row_stride;
    const int32_t* ir = idx + (size_t)blockIdx.y * n;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const int32_t p = ir[j];
        if (p >= 0 && (uint32_t)p < S) row[p] = value;
    }
}
}  // namespace
extern "C" int kvp_scores_fill_at(float* scores, int64_t R, int64_t S, int64_t row_stride, const int32_t* idx, int64_t n, float value,
                                  kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(R >= 0 && S >= 0 && n >= 0 && R <= 65535 && row_stride >= S, "fill_at: bad shape R=%ld S=%ld n=%ld", (long)R, (long)S, (long)n);
    if (R * n == 0) return KVP_OK;
    KVP_CHECK_ARG(scores && idx, "fill_at: null pointer");
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
    KVP_LAUNCH("fill_at_kernel", stream, fill_at_kernel<<<dim3(bx, (uint32_t)R), 256, 0, stream>>>(scores, row_stride, idx, (uint32_t)n, (uint32_t)S, value));
    KVP_CHECK_LAUNCH("fill_at");
    return KVP_OK;
}
