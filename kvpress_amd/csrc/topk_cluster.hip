// Cluster select: the whole radix select of long score rows (16384 < S <= 262144) in ONE launch.
// Replaces `scores.topk(n_kept, dim=-1).indices` (kvpress/presses/scorer_press.py:95); an exact radix select on the
// order-preserving key with the same tie rule (lowest position first), hence the same indices as the (chunk, row) passes of
// topk.hip, which stay as the fallback (devices with fewer than 256 CUs, rows beyond 262144 scores, more than 32 rows; KVP_TK_CLUSTER=0
// selects them on any device).
//
// Structure.  The (chunk, row) passes are chains of dependent launches over an L2-resident row: load the keys, count a digit,
// flush, kernel boundary, load the keys again ...  (4 launches of 5-7 us for ~2 us of work each).  Here a row belongs to a
// CLUSTER of TC_SLOTS = 32 workgroups of 1024 threads; a workgroup keeps its 1024 * PER keys in registers from the first
// load to the compaction, and the steps of the select are separated by in-kernel synchronisation instead of kernel boundaries.
// Two forms, same indices:
//   * two-hop (round 5; rows whose keys this kernel loads itself): a SAMPLE of 128 keys of the row brackets the k-th largest; the
//     first digit is cut out of that bracket (256 bins of 2^s keys, everything outside in the two end bins), counted into the row's
//     histogram (complete when its total is S: polled, no barrier); the threshold bin's few dozen keys -- the candidates -- are
//     published per slot with plain stores, ONE cluster barrier later every workgroup reads them all and finishes locally (rounds of
//     <= 12 bits in LDS).  Hop 1 = the histogram, hop 2 = the candidates;
//   * three rounds (12 + 12 + 8 bits of the key: hop per digit; cluster barriers = one monotonic arrival counter per cluster) when
//     the scorer already accumulated the first digit (HIST1), and whenever the two-hop form DECLINES: the threshold lies outside the
//     sample's bracket (k at an extreme of the row) or a slot has more candidates than its record holds (rows of few distinct
//     values).  Every workgroup of the row takes that decision from the same words, so all of them switch together.
// 256 resident workgroups = 8 clusters (a launch may carry up to 32: batches > 1); a row's 32 workgroups are CONSECUTIVE blocks
// (block / 32 = cluster), i.e. spread over all XCDs -- measured, a hop costs the same ~0.8 us round trip from anywhere
// (profiles/r03_select_cluster_lab.txt), and consecutive blocks make partial residency harmless (see the kernel).  Correctness is
// placement-independent: every word another workgroup reads -- the row histograms, the candidate records, the per-slot suffix
// tables, the counter -- is written AND read with agent-scope (sc1) atomics / loads / stores, every wave drains its vector-memory
// counter before the arrival, no fences, nothing relies on two workgroups sharing an L2 (cdna_hip_programming.md
// Guideline 16, the "agent atomics on both sides" form).  One launch selects up to 32 rows (8 resident at a time); more take the (chunk, row) passes.
//
// Failure is LOUD (torch.topk cannot return wrong indices silently, scorer_press.py:95).  The barrier spins are bounded
// (KVP_TC_TIMEOUT_US, default 1 s) so that a cluster that never becomes co-resident cannot hang the GPU; a spin that times out
// sets the cluster's PERSISTENT flag in the workspace and the process-wide host-pinned status word (kvp_async_check: the next call
// of the library returns KVP_EASYNC), and every workgroup of that cluster writes -1 instead of indices (kvp_gather_kv* turn a
// negative index into a NaN row).  A workspace that is handed in again as "clean" after such a failure without having been
// zero-filled still carries the flag: its rows are poisoned again AND the host word is raised again (stale reuse is reported, not
// only poisoned).  Reports carry the launch's sequence number, so the 32 stores of one failed launch make ONE report.
// The host launches this kernel only on a device with at least 256 CUs.
//
// Key sources (MODE):  SCORES  the row is read from memory (kvp_topk_select, every scorer);
//                      POOL5   SnapKV's un-pooled column sums: avg_pool1d(kernel 5) + scale in the loader, term for term the
//                              arithmetic of snapkv_pool_kernel (snapkv_press.py:96) -- no pooling launch, no score round trip;
//                      KNORM   the keys are -||k||_2 of the workgroup's own 1024 * PER rows of K, computed with the lanes /
//                              shuffles / rounding of rownorm_vec_kernel (knorm_press.py:38) -- the fused Knorm compress
//                              never writes its scores.
// HIST1: the kernel that wrote the scores already accumulated the first digit's histogram (topk_internal.h): the first
// step and its barrier are skipped.
#include "kvp_common.h"
#include <mutex>
#include "topk_block.h"
#include "topk_internal.h"

namespace {

#define TC_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ uint32_t tc_ld(const uint32_t* p) { return __hip_atomic_load(p, TC_RLX); }
__device__ __forceinline__ void tc_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, TC_RLX); }
__device__ __forceinline__ void tc_add(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, TC_RLX); }

// Where a cluster's barrier state lives: bar + cluster * 32 (its own 128-byte line): [0] the monotonic arrival counter,
// [1] the cluster's give-up code (0 = none; persistent: only a zero-fill of the workspace clears it).
struct ClusterSync {
    uint32_t* ctr;
    uint32_t* cl_flag;
    uint32_t* host_flag;      // process-wide pinned status word (kvp_async_flag(); may be null)
    uint32_t report;          // launch sequence number << 8: OR-ed with the give-up code in the host word
    uint32_t timeout_ticks;   // of the 100 MHz real-time counter
#ifdef KVP_TC_FAULT_INJECTION
    uint32_t delay_ticks;     // fault-injection build only (tests): this workgroup arrives at its first barrier this late
#endif
};
__device__ __forceinline__ void tc_report(const ClusterSync& cs, uint32_t code) {
    tc_st(cs.cl_flag, code);
    if (cs.host_flag) __hip_atomic_store(cs.host_flag, cs.report | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Arrive at / wait for the cluster's next barrier.  Every wave first drains its own agent-scope stores and atomics
// (they are what the other workgroups read after the barrier).  A spin that times out REPORTS it -- the cluster's flag (read by
// every workgroup of the cluster before it writes indices: see the poison path of the kernel) and the host's status word -- and
// goes on, so that every workgroup that reaches a barrier still makes all its arrivals.
__device__ __forceinline__ void cluster_barrier(const ClusterSync& cs, uint32_t code, uint32_t* lds_gave_up, bool first) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
#ifdef KVP_TC_FAULT_INJECTION
        if (first && cs.delay_ticks) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < cs.delay_ticks) __builtin_amdgcn_s_sleep(8);
        }
#else
        (void)first;
#endif
        const uint32_t old = __hip_atomic_fetch_add(cs.ctr, 1u, TC_RLX);
        const uint32_t target = (old & ~(uint32_t)(TC_SLOTS - 1)) + TC_SLOTS;
        if ((int32_t)(tc_ld(cs.ctr) - target) < 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            for (uint32_t it = 1; (int32_t)(tc_ld(cs.ctr) - target) < 0; ++it) {
                __builtin_amdgcn_s_sleep(1);
                // A sibling that gave up BEFORE it reached this barrier (a polled round that timed out) never arrives here: its report is
                // in the cluster's flag, looked at every 32 spins (~25 us), so that the others end now instead of after the full timeout
                // (the counter is then off by its missing arrivals -- a failed workspace has to be zero-filled anyway).
                if ((it & 31u) == 0u && tc_ld(cs.cl_flag) != 0u) {
                    *lds_gave_up = code;
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() - t0 > cs.timeout_ticks) {  // the cluster is not co-resident: give up, loudly
                    *lds_gave_up = code;
                    tc_report(cs, code);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// Digit search as row_find_bin_regs (topk_block.h), also returning the histogram's TOTAL and the found bin's own count: the
// polling protocol below uses the total as its completion signal and the count as the next round's expected total.
// lds: >= TR_WAVES + 3 words.
template <int PERB>
__device__ __forceinline__ uint32_t tc_find(const uint32_t (&loc)[PERB], uint32_t nowners, uint32_t k, uint32_t* lds, uint32_t& bin, uint32_t& krem,
                                            uint32_t& cnt) {
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
                lds[TR_WAVES + 2] = loc[i];
            }
            c += loc[i];
        }
    }
    __syncthreads();
    bin = lds[TR_WAVES];
    krem = lds[TR_WAVES + 1];
    cnt = lds[TR_WAVES + 2];
    __syncthreads();
    return total;
}

struct ClusterArgs {
    const float* scores;   // SCORES: [R][row_stride]; POOL5: colsum [R][row_stride] (un-pooled column sums)
    int64_t row_stride;
    uint32_t R, row_base, S, k, kmask;
    float inv;             // POOL5: 1 / (G * W * kernel_size)
    int32_t* idx;
    int64_t idx_stride;
    uint32_t tail_start, tail_n, nseg, seg_len, pos_base;
    uint32_t* ws0;             // workspace base; regions by topk_ws_layout(wsR, ws_ntab) at the point of use (TCW)
    uint32_t wsR, ws_ntab;
    // KNORM
    const void* x;         // [B,H,S,D], 256-byte rows
    int64_t x_sb, x_sh, x_ss;  // element strides
    uint32_t H;
    float scale;
    // barrier protocol
    uint32_t* host_flag;
    uint32_t report;           // launch sequence number << 8 (kvp_async_next_seq)
    uint32_t timeout_ticks;
#ifdef KVP_TC_FAULT_INJECTION
    int32_t test_delay_slot;   // slot (of cluster 0) that arrives 2 x timeout late at its first barrier; -1 = none
#endif
};

#ifndef TC_KN_UNROLL
#define TC_KN_UNROLL 4   // 64-row steps of the Knorm stream in flight per workgroup (L = 1024 PER is a multiple of 64 * 8)
#endif
enum { TC_SCORES = 0, TC_POOL5 = 1, TC_KNORM_BF16 = 2, TC_KNORM_F16 = 3 };
constexpr int TC_NS = 128;   // sampled keys per row (two-hop form)

template <int DT>
__device__ __forceinline__ float tc_sumsq16(const uint4& v) {   // = rownorm.hip's sumsq16 (same fma chain)
    float f[Elem<DT>::PER16];
    unpack16<DT>(v, f);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < Elem<DT>::PER16; ++i) a = fmaf(f[i], f[i], a);
    return a;
}

// topk_hist_add_bin (topk_internal.h) with a per-lane weight of `w` for the lanes flagged `heavy` and 1 for the others: two rounds of
// wave-level aggregation, then plain LDS atomics.  Every thread of the wave must call it together.
__device__ __forceinline__ void tc_hist_add_weighted(uint32_t* lds_hist, uint32_t bin, bool valid, bool heavy, uint32_t w) {
    const uint64_t hv = __ballot(heavy);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const uint64_t todo = __ballot(valid);
        if (todo == 0) return;
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const uint64_t same = __ballot(valid && bin == b0);
        if ((int)(threadIdx.x & 63) == leader)
            atomicAdd(&lds_hist[b0], (uint32_t)__popcll(same & hv) * w + (uint32_t)__popcll(same & ~hv));
        valid = valid && bin != b0;
    }
    if (valid) atomicAdd(&lds_hist[bin], heavy ? w : 1u);
}

// a value that every lane holds identically, moved to a scalar register: branches on it are scalar branches (no saved exec masks)
__device__ __forceinline__ uint32_t tc_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// out-of-range positions carry key 0; real keys are >= 1 (as in topk_row_kernel)
__device__ __forceinline__ uint32_t tc_key(float f, uint32_t kmask) { return max(float_to_key(f) ^ kmask, 1u); }

#define TCW(field) (a.ws0 + topk_ws_layout(a.wsR, a.ws_ntab).field)
template <int PER, int MODE, bool HIST1>
__global__ __launch_bounds__(TR_THREADS) void topk_cluster_kernel(ClusterArgs a) {
    constexpr uint32_t L = TR_THREADS * PER;   // keys per workgroup
    __shared__ __attribute__((aligned(16))) uint32_t lh[L > 4096 ? L : 4096];   // histogram; KNORM: first the staged scores; last the staged output
    __shared__ uint32_t scr[TR_WAVES + 3];
    __shared__ uint32_t fh[HIST1 ? 1 : 4096];   // two-hop form: the local rounds' histogram (<= 12 bits per round); first the sample
    __shared__ uint32_t fg[64];                 // ... and its sums over groups of 64 bins
    __shared__ uint32_t res2[2];                // verdict of a one-wave search (its own words: scr belongs to the block scans)
    __shared__ uint32_t s_fail[2];   // [0] this workgroup gave up at a barrier, [1] the cluster's flag as read after the last barrier
    // A row's 32 workgroups are CONSECUTIVE blocks: whatever part of the grid the device can hold at once, whole clusters become
    // resident in dispatch order and finish, so a device with fewer free CUs than the grid (CU masking, a busy neighbour) makes
    // the launch slower, not stuck.  (This leans on the dispatcher handing out blocks in index order -- observed, not promised:
    // if it ever did not, the bounded spins turn the would-be deadlock into the loud give-up path below.)
    const uint32_t cluster = blockIdx.x / TC_SLOTS;
    const uint32_t slot = blockIdx.x % TC_SLOTS;
    ClusterSync cs;
    cs.ctr = TCW(bar) + cluster * 32;
    cs.cl_flag = cs.ctr + 1;
    cs.host_flag = a.host_flag;
    cs.report = a.report;
    cs.timeout_ticks = a.timeout_ticks;
#ifdef KVP_TC_FAULT_INJECTION
    cs.delay_ticks = (cluster == 0 && (int32_t)slot == a.test_delay_slot) ? 2u * a.timeout_ticks : 0u;
#endif
    if (threadIdx.x < 2) s_fail[threadIdx.x] = 0;   // (ordered before its first use by the barriers of the key loaders / histograms)
    // Round 1 without a counter barrier (rounds that count their own first digit: not HIST1).  What the barrier buys --
    // "every workgroup's atomics have landed" -- the histogram itself can say: the row's first histogram is complete exactly when
    // its total equals the row's S valid keys, so a workgroup flushes and then repeats {read the histogram, search} until the
    // total is right: no drain, no arrival atomic, no separate poll (-1.6 us, profiles/r04_select_poll_lab.txt).  Rounds 2 and 3
    // keep their counter barriers: on flat rows round 2 drains ~2400 atomics per workgroup, and pollers re-reading the
    // histogram while those are in flight slow them down (measured: +1.9 us); round 3 publishes plain tables anyway.
    constexpr bool poll = !HIST1;
    const uint32_t max_poll = a.timeout_ticks / 100u + 8u;   // iterations: each costs >= one L2 round trip (~1 us)
    const uint32_t S = a.S, k = a.k, kmask = a.kmask;
    const uint32_t p0 = slot * L + threadIdx.x * PER;   // this thread's PER consecutive positions

    // (one row per cluster and launch: a row loop in here makes the compiler hoist every thread-index comparison of the body
    // into lane masks that overflow the scalar register file; more rows = more clusters in the launch instead)
    const uint32_t row = a.row_base + cluster;
    if (row >= a.R) return;
    {
        // ---- sample (two-hop form, not HIST1): TC_NS scores of the row at fixed positions, requested BEFORE the workgroup's own keys
        // so that their latency hides behind the key loads / the Knorm stream.  Every workgroup of the row reads the same positions.
        constexpr bool KN = MODE == TC_KNORM_BF16 || MODE == TC_KNORM_F16;
        // wave-uniform conditions (scalar branches) instead of thread-index comparisons: the compiler hoists the latter into lane
        // masks that live across the whole kernel and overflow the scalar register file
        const uint32_t wvu = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        float smp_f = 0.f;          // SCORES / POOL5: thread t < TC_NS holds sample t
        uint4 smp_v[KN ? TC_NS / 64 : 1];   // KNORM: 16 lanes per sampled row, 64 rows per step
        const uint32_t smp_stride = S / TC_NS;   // S > 16384: >= 64
        if (!HIST1) {
            if (KN) {
                constexpr int DT = MODE == TC_KNORM_BF16 ? KVP_BF16 : KVP_F16;
                using T = typename Elem<DT>::T;
                const uint32_t b = row / a.H, h = row - b * a.H;
                const T* __restrict__ base = static_cast<const T*>(a.x) + (int64_t)b * a.x_sb + (int64_t)h * a.x_sh;
#pragma unroll
                for (int u = 0; u < TC_NS / 64; ++u) {
                    const uint32_t sp = min((u * 64 + (threadIdx.x >> 4)) * smp_stride + smp_stride / 2, S - 1);
                    smp_v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)sp * a.x_ss + (size_t)(threadIdx.x & 15u) * 8);
                }
            } else if (wvu < (uint32_t)(TC_NS / 64)) {
                const float* rp = a.scores + (int64_t)row * a.row_stride;
                const uint32_t sp = min(threadIdx.x * smp_stride + smp_stride / 2, S - 1);
                if (MODE == TC_POOL5) {
#pragma unroll
                    for (int d = 0; d < 5; ++d) {
                        const int32_t pos = (int32_t)sp + d - 2;
                        const uint32_t inside = (uint32_t)(((pos - (int32_t)S) >> 31) & ~(pos >> 31));
                        smp_f += __uint_as_float(__float_as_uint(rp[min(max(pos, 0), (int32_t)S - 1)]) & inside);
                    }
                    smp_f *= a.inv;
                } else {
                    smp_f = rp[sp];
                }
            }
        }
        // ---- window of the two-hop form from the sample (see below); runs while this workgroup's own key loads are in flight ----
        uint32_t sft = 0, wbase = 0;
        auto window_setup = [&]() {
            uint32_t* smp = fh;                 // [TC_NS] sampled keys
            uint32_t* swin = fh + TC_NS;        // [2] Lk, Hk
            uint32_t* whist = fh + 2 * TC_NS;   // [TC_WB] this workgroup's window histogram
            static_assert(TC_NS <= 256 && TC_NS % 64 == 0 && TC_WB % 64 == 0 && TC_WB <= TR_THREADS, "sample ranking / one window bin per thread");
            if (KN) {
                constexpr int DT = MODE == TC_KNORM_BF16 ? KVP_BF16 : KVP_F16;
#pragma unroll
                for (int u = 0; u < (KN ? TC_NS / 64 : 1); ++u) {
                    float acc = tc_sumsq16<DT>(smp_v[u]);
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                    const uint32_t si = u * 64 + (threadIdx.x >> 4);
                    if ((threadIdx.x & 15u) == 0) smp[si] = tc_key(a.scale * sqrtf(acc), kmask);
                }
            } else if (wvu < (uint32_t)(TC_NS / 64)) {
                smp[threadIdx.x] = tc_key(smp_f, kmask);
            }
            if (wvu < (uint32_t)(TC_WB / 64)) whist[threadIdx.x] = 0;
            __syncthreads();
            {
                // descending rank of sample i = number of samples that are larger, or equal with a lower index (a permutation): LPS lanes
                // share a sample and split the 16-byte chunks of the sample array (chunk LPS t + q: the lanes' reads are contiguous)
                constexpr uint32_t LPS = TR_THREADS / TC_NS, NCH = TC_NS / 4;
                const uint32_t i = threadIdx.x / LPS, q = threadIdx.x % LPS;
                const uint32_t xi = smp[i];
                const uint4* s4 = reinterpret_cast<const uint4*>(smp);
                uint32_t c = 0;
#pragma unroll
                for (uint32_t t = 0; t < NCH / LPS; ++t) {
                    const uint4 x = s4[LPS * t + q];
                    const uint32_t j0 = (LPS * t + q) * 4u;
                    c += (x.x > xi || (x.x == xi && j0 + 0u < i)) ? 1u : 0u;
                    c += (x.y > xi || (x.y == xi && j0 + 1u < i)) ? 1u : 0u;
                    c += (x.z > xi || (x.z == xi && j0 + 2u < i)) ? 1u : 0u;
                    c += (x.w > xi || (x.w == xi && j0 + 3u < i)) ? 1u : 0u;
                }
#pragma unroll
                for (uint32_t o = 1; o < LPS; o <<= 1) c += __shfl_xor(c, o);
                const float pq = (float)k / (float)S;
                const int32_t rstar = min((int32_t)(pq * (float)TC_NS), TC_NS - 1);
                const int32_t delta = (int32_t)(4.5f * sqrtf((float)TC_NS * pq * (1.f - pq))) + 3;
                const uint32_t r_hi = (uint32_t)max(rstar - delta, 0), r_lo = (uint32_t)min(rstar + delta, TC_NS - 1);
                if (q == 0 && c == r_lo) swin[0] = xi;
                if (q == 0 && c == r_hi) swin[1] = xi;
            }
            __syncthreads();
            const uint32_t Lk = tc_uni(swin[0]), Hk = tc_uni(swin[1]);
            while (((Hk >> sft) - (Lk >> sft)) > (uint32_t)(TC_WB - 3)) ++sft;
            wbase = Lk >> sft;
        };
        auto wbin = [&](uint32_t key) -> uint32_t {   // the window digit of a key (valid after window_setup)
            const uint32_t v = key >> sft;
            return v < wbase ? 0u : min(v - wbase + 1u, (uint32_t)(TC_WB - 1));
        };
        // ---- keys ------------------------------------------------------------------------------------------------
        uint32_t keys[PER];
        if (MODE == TC_SCORES) {
            const float* rp = a.scores + (int64_t)row * a.row_stride;
            float raw[PER];
            if (PER % 4 == 0 && p0 + PER <= S && ((((uintptr_t)(rp + p0)) & 15u) == 0)) {
#pragma unroll
                for (int q = 0; q < PER / 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + p0 + 4 * q);
                    raw[4 * q + 0] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < PER; ++j) raw[j] = rp[min(p0 + j, S - 1)];
            }
            if (!HIST1) window_setup();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t inside = (uint32_t)((int32_t)(p0 + j - S) >> 31);  // all ones iff p0 + j < S
                keys[j] = tc_key(raw[j], kmask) & inside;
            }
        } else if (MODE == TC_POOL5) {
            const float* rp = a.scores + (int64_t)row * a.row_stride;
            constexpr int NIN = PER + 4;
            float in[NIN];
            if (PER % 2 == 0 && p0 >= 2 && p0 + PER + 2 <= S && ((((uintptr_t)(rp + p0 - 2)) & 7u) == 0)) {
#pragma unroll
                for (int i = 0; i < NIN / 2; ++i) {
                    const float2 v = *reinterpret_cast<const float2*>(rp + p0 - 2 + 2 * i);
                    in[2 * i] = v.x; in[2 * i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NIN; ++i) {
                    const int32_t pos = (int32_t)p0 + i - 2;
                    const uint32_t inside = (uint32_t)(((pos - (int32_t)S) >> 31) & ~(pos >> 31));   // all ones iff 0 <= pos < S
                    in[i] = __uint_as_float(__float_as_uint(rp[min(max(pos, 0), (int32_t)S - 1)]) & inside);
                }
            }
            if (!HIST1) window_setup();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int d = 0; d < 5; ++d) sum += in[j + d];
                sum *= a.inv;
                const uint32_t inside = (uint32_t)((int32_t)(p0 + j - S) >> 31);
                keys[j] = tc_key(sum, kmask) & inside;
            }
        } else {
            // -||k||: 16 lanes per 256-byte row, 64 rows per step of the workgroup, TC_KN_UNROLL steps in flight (rownorm_vec_kernel's
            // loads, fma chain, xor-shuffle order and rounding); the scores go through LDS to their owning threads
            constexpr int DT = MODE == TC_KNORM_BF16 ? KVP_BF16 : KVP_F16;
            using T = typename Elem<DT>::T;
            const uint32_t b = row / a.H, h = row - b * a.H;
            const T* __restrict__ base = static_cast<const T*>(a.x) + (int64_t)b * a.x_sb + (int64_t)h * a.x_sh;
            float* st = reinterpret_cast<float*>(lh);
            const uint32_t lir = threadIdx.x & 15u, g = threadIdx.x >> 4;
            const uint32_t r0 = slot * L;
#pragma unroll 1
            for (uint32_t it = 0; it < L; it += 64 * TC_KN_UNROLL) {
                uint4 v[TC_KN_UNROLL];
#pragma unroll
                for (int u = 0; u < TC_KN_UNROLL; ++u) {
                    const uint32_t s = r0 + it + u * 64 + g;
                    v[u] = make_uint4(0, 0, 0, 0);
                    if (s < S) v[u] = *reinterpret_cast<const uint4*>(base + (int64_t)s * a.x_ss + (size_t)lir * 8);
                }
                if (!HIST1 && it == 0) window_setup();   // (the sample's loads are older than this step's: they have landed or land first)
#pragma unroll
                for (int u = 0; u < TC_KN_UNROLL; ++u) {
                    float acc = tc_sumsq16<DT>(v[u]);
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                    if (lir == 0) {
                        const float sc = a.scale * sqrtf(acc);
                        st[it + u * 64 + g] = sc;
                        // two-hop form: the window is known since step 0, so the norm is counted where it is produced (4 lanes per
                        // wave and step: no aggregation needed) instead of in a pass over the finished keys
                        if (!HIST1 && r0 + it + u * 64 + g < S) atomicAdd(&fh[2 * TC_NS + wbin(tc_key(sc, kmask))], 1u);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t inside = (uint32_t)((int32_t)(p0 + j - S) >> 31);
                keys[j] = tc_key(st[threadIdx.x * PER + j], kmask) & inside;
            }
            __syncthreads();  // lh becomes the histogram
        }
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            kmin = min(kmin, keys[j]);
            kmax = max(kmax, keys[j]);
        }
        const bool full = kmin != 0u;   // all PER positions inside the row

        // (polling) repeat {read the row's histogram, search} until its total is `expect`; gives up like a barrier after max_poll rounds
#define TC_POLL_GIVE_UP(code)                                                                                              \
    do {                                                                                                                   \
        if (threadIdx.x == 0) {                                                                                            \
            s_fail[0] = (code);                                                                                            \
            tc_report(cs, (code));                                                                                         \
        }                                                                                                                  \
        __syncthreads();                                                                                                   \
    } while (0)
        // results of either form of the select: the threshold key, how many keys equal to it are kept, and the kept keys in the
        // slots before this one
        uint32_t T = 0, quota = 0, gt_before = 0, eq_before = 0;
        bool done = false;
        // ==== two-hop form ==============================================================================================
        // The 12 + 12 + 8-bit digits below cost three all-to-all rounds, and on the flat rows of a real layer the first digit (sign,
        // exponent, 3 mantissa bits) separates nothing: 60 % of a row share the threshold's bin (LAB R4.2).  Here the FIRST digit is
        // steered by a sample: the TC_NS sampled keys bracket the k-th largest between two of them (+- 4.5 sigma of the sample
        // quantile), the bracket [Lk, Hk] is cut into TC_WB - 2 bins of 2^s keys (s = the smallest shift that fits), everything below /
        // above lands in bin 0 / TC_WB - 1.  Any such binning is monotone in the key, so the digit search is as exact as before; what
        // the sample buys is that the threshold's bin now holds a few dozen keys of the row instead of most of it.  Those CANDIDATES are
        // published per slot (plain stores, no atomics), ONE cluster barrier later every workgroup holds all of them and finishes the
        // select locally (rounds of <= 12 bits in LDS over <= 4032 keys, no further hop).  Same threshold, same tie rule, same indices.
        // Whenever the sample misleads (threshold outside the bracket, a slot with more candidates than its record holds: rows of
        // mostly equal scores) every workgroup of the row sees that in the same words and all of them take the three-round form below.
        if (!HIST1) {
            uint32_t* whist = fh + 2 * TC_NS;   // (zeroed by window_setup)
            uint32_t* hw = TCW(histw) + (size_t)row * TC_WB;
            if (!KN) {   // (Knorm: the producing lanes counted their norms while they streamed K)
                // most keys lie outside the bracket (bins 0 and TC_WB - 1): those are counted in registers and added once per wave;
                // the keys inside spread over the bins in between (plain LDS atomics)
                uint32_t nout = 0;   // below in bits 0-15, above in bits 16-31 (<= 64 * PER per wave)
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const uint32_t bj = wbin(keys[j]);
                    if (keys[j] != 0u) {
                        if (bj == 0u) nout += 1u;
                        else if (bj == (uint32_t)(TC_WB - 1)) nout += 1u << 16;
                        else atomicAdd(&whist[bj], 1u);
                    }
                }
                const uint32_t ws_ = wave_incl_scan(nout);
                if ((threadIdx.x & 63u) == 63u) {
                    if (ws_ & 0xFFFFu) atomicAdd(&whist[0], ws_ & 0xFFFFu);
                    if (ws_ >> 16) atomicAdd(&whist[TC_WB - 1], ws_ >> 16);
                }
            }
            __syncthreads();
#ifdef KVP_TC_FAULT_INJECTION
            if (cs.delay_ticks && threadIdx.x == 0) {                         // fault injection: this workgroup flushes late
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                while (__builtin_amdgcn_s_memrealtime() - t0 < 2ull * cs.delay_ticks) __builtin_amdgcn_s_sleep(8);
            }
            if (cs.delay_ticks) __syncthreads();
#endif
            if (wvu < (uint32_t)(TC_WB / 64)) {
                const uint32_t c = whist[threadIdx.x];
                if (c) tc_add(&hw[threadIdx.x], c);
            }
            // complete when the histogram's total is the row's S keys (the polling protocol of round 1, see above)
            // (the search runs in ONE wave, which polls on its own -- lane l holds the TC_WB / 64 bins below bin TC_WB - 1 - l * (TC_WB / 64),
            // the prefix over the lanes is a DPP scan -- ; the other waves wait at ONE block barrier for its verdict)
            if (wvu == 0u) {
                constexpr int PB = TC_WB / 64;
                const uint32_t lane = threadIdx.x & 63u, top = (63u - lane) * PB;   // this lane's bins: top + PB - 1 .. top (descending)
                for (uint32_t it = 0;; ++it) {
                    uint32_t loc[PB], sum = 0;
#pragma unroll
                    for (int i = 0; i < PB; ++i) loc[i] = tc_ld(&hw[top + i]);
#pragma unroll
                    for (int i = 0; i < PB; ++i) sum += loc[i];
                    const uint32_t inc = wave_incl_scan(sum), excl = inc - sum;
                    if (tc_uni((uint32_t)__builtin_amdgcn_readlane((int)inc, 63)) == S) {
                        if (excl < k && k <= inc) {
                            uint32_t c = excl;
#pragma unroll
                            for (int i = PB - 1; i >= 0; --i) {
                                if (k > c && k <= c + loc[i]) {
                                    res2[0] = top + i;
                                    res2[1] = k - c;
                                }
                                c += loc[i];
                            }
                        }
                        break;
                    }
                    if (it >= max_poll) {   // gives up like a barrier
                        if (lane == 0u) {
                            s_fail[0] = 1u;
                            tc_report(cs, 1u);
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            uint32_t d1 = res2[0], k1 = res2[1];
            d1 = tc_uni(d1);
            k1 = tc_uni(k1);
            // (a poll that gave up: straight to the poison branch -- the workgroups that did see the complete histogram find this one's
            // report in the cluster's flag while they spin at the candidates' barrier, which this one never reaches, and end there too)
            if (tc_uni(s_fail[0]) != 0) {
                done = true;
            } else if (d1 >= 1u && d1 <= (uint32_t)(TC_WB - 2)) {
                // ---- candidates of this slot: keys in bin d1; ngt: keys in higher bins ----------------------------------------
                uint32_t ncand = 0, ngt = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const uint32_t bj = wbin(keys[j]);
                    ncand += (keys[j] && bj == d1) ? 1u : 0u;
                    ngt += (keys[j] && bj > d1) ? 1u : 0u;
                }
                uint32_t tot;
                const uint32_t ex = row_excl_scan(ncand | (ngt << 16), scr, &tot);   // L <= 8192: both fields < 65536
                const uint32_t cand_tot = tot & 0xFFFFu;
                uint32_t* rec = TCW(cand) + ((size_t)row * TC_SLOTS + slot) * TC_REC;
                if (cand_tot <= (uint32_t)(TC_REC - 2)) {
                    uint32_t o = 2u + (ex & 0xFFFFu);
#pragma unroll
                    for (int j = 0; j < PER; ++j)
                        if (keys[j] && wbin(keys[j]) == d1) tc_st(&rec[o++], keys[j]);
                }
                if (threadIdx.x == 0) {
                    tc_st(&rec[0], cand_tot <= (uint32_t)(TC_REC - 2) ? cand_tot : 0xFFFFFFFFu);
                    tc_st(&rec[1], tot >> 16);
                }
                cluster_barrier(cs, 2, &s_fail[0], true);
                // ---- the row's 32 records: 4096 words, 4 per thread (kept in registers: the rounds below run on them); the headers
                // (count, keys above) also go through LDS --------------------------------------------------------------------------
                uint32_t v[4];
                {
                    const uint32_t* rr = TCW(cand) + (size_t)row * TC_SLOTS * TC_REC;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = tc_ld(&rr[threadIdx.x * 4 + i]);
                    if (threadIdx.x == 0) s_fail[1] = tc_ld(cs.cl_flag);   // rides on the same round trip
                }
                static_assert(TC_REC == 128 && TC_SLOTS * TC_REC == 4 * TR_THREADS, "a thread's 4 words lie in one record");
                const uint32_t wrec = threadIdx.x >> 5, pos0 = (threadIdx.x & 31u) * 4u;   // record (= slot) and first word of this thread
                uint32_t* hdr = lh;   // [TC_SLOTS][2]
                if (pos0 == 0) {
                    hdr[wrec * 2] = v[0];
                    hdr[wrec * 2 + 1] = v[1];
                }
                __syncthreads();
                const bool ovf = hdr[(threadIdx.x & (TC_SLOTS - 1)) * 2] == 0xFFFFFFFFu;   // (every wave looks at all 32 records)
                const bool any_ovf = tc_uni((uint32_t)__syncthreads_or(ovf)) != 0u;
                if (tc_uni(s_fail[0] | s_fail[1])) {
                    done = true;   // poisoned below
                } else if (!any_ovf) {
                    const uint32_t cnt = hdr[wrec * 2];
                    bool cv[4];    // word i of this thread is a candidate key
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[i] = pos0 + i >= 2u && pos0 + i - 2u < cnt;
                    // ---- finish locally: the k1-th largest candidate.  Bits >= sft are the bin's; rounds of <= 12 bits over the rest ---
                    uint32_t Tpre = (wbase + d1 - 1u) << sft, krem = k1, hi = sft;
                    while (hi > 0u) {
                        const uint32_t wd = min(hi, 12u), lo = hi - wd;
#pragma unroll
                        for (int i = 0; i < 4; ++i) fh[threadIdx.x * 4 + i] = 0;
                        if (wvu == 0u) fg[threadIdx.x] = 0;
                        __syncthreads();
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (cv[i] && (v[i] >> hi) == (Tpre >> hi)) {
                                const uint32_t dg = (v[i] >> lo) & ((1u << wd) - 1u);
                                atomicAdd(&fh[dg], 1u);
                                atomicAdd(&fg[dg >> 6], 1u);   // the 64 groups of 64 bins
                            }
                        __syncthreads();
                        // two-level search in ONE wave (group, then bin inside the group: one value per lane, DPP scans): three block
                        // barriers per round where the 1024-thread search takes six
                        if (wvu == 0u) {
                            const uint32_t lane = threadIdx.x & 63u;
                            const uint32_t cgp = fg[63u - lane];
                            const uint32_t ig = wave_incl_scan(cgp);
                            const uint64_t hg = __ballot(ig - cgp < krem && krem <= ig);   // exactly one lane: krem <= the candidates' count
                            const int lg = __ffsll((unsigned long long)hg) - 1;
                            const uint32_t grp = 63u - (uint32_t)lg;
                            const uint32_t kr2 = krem - (uint32_t)__builtin_amdgcn_readlane((int)(ig - cgp), lg);
                            const uint32_t cb = fh[grp * 64u + (63u - lane)];
                            const uint32_t ib = wave_incl_scan(cb);
                            if (ib - cb < kr2 && kr2 <= ib) {
                                res2[0] = grp * 64u + (63u - lane);
                                res2[1] = kr2 - (ib - cb);
                            }
                        }
                        __syncthreads();
                        Tpre |= tc_uni(res2[0]) << lo;
                        krem = tc_uni(res2[1]);
                        hi = lo;
                    }
                    T = Tpre;
                    quota = krem;
                    // kept keys in the slots before this one: gt in bits 0-19 (<= 262144), eq in bits 20-31 (<= 4032)
                    uint32_t part = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (cv[i] && wrec < slot) part += (v[i] > T ? 1u : 0u) + (v[i] == T ? (1u << 20) : 0u);
                    if (threadIdx.x < slot) part += hdr[threadIdx.x * 2 + 1];
                    uint32_t tot;
                    row_excl_scan(part, scr, &tot);
                    gt_before = tot & 0xFFFFFu;
                    eq_before = tot >> 20;
                    done = true;
                }
            }
        }
        // ==== three-round form (HIST1, or the two-hop form declined) =========================================================
        if (!done) {
        uint32_t* h1 = TCW(hist1) + (size_t)row * 4096;
        uint32_t* h2 = TCW(hist2) + (size_t)row * 4096;
        uint32_t* h3 = TCW(hist3) + (size_t)row * 256;
        // ---- digit 1: key >> 20 ----------------------------------------------------------------------------------
        if (!HIST1) {
            for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
            __syncthreads();
            if (PER == 1) {
                topk_hist_add_bin(lh, keys[0] >> 20, keys[0] != 0u);
            } else {
                // a thread whose PER consecutive keys share their first digit (the usual case: neighbouring scores of a row) adds them
                // in ONE wave-aggregated step with weight PER; only the threads with mixed digits take part in the steps 1 .. PER-1
                const bool uni = full && (kmin >> 20) == (kmax >> 20);
                tc_hist_add_weighted(lh, keys[0] >> 20, keys[0] != 0u, uni, (uint32_t)PER);
#pragma unroll
                for (int j = 1; j < PER; ++j) topk_hist_add_bin(lh, keys[j] >> 20, !uni && keys[j] != 0u);
            }
            __syncthreads();
#ifdef KVP_TC_FAULT_INJECTION
            if (cs.delay_ticks && threadIdx.x == 0) {                         // fault injection: this workgroup flushes late
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                while (__builtin_amdgcn_s_memrealtime() - t0 < 2ull * cs.delay_ticks) __builtin_amdgcn_s_sleep(8);   // (a poll round is ~1.6 us, not 1)
            }
            if (cs.delay_ticks) __syncthreads();
#endif
            for (int i = threadIdx.x; i < 4096; i += TR_THREADS) {
                const uint32_t c = lh[i];
                if (c) tc_add(&h1[i], c);
            }
        }
        uint32_t b1, k1, c1;
        for (uint32_t it = 0;; ++it) {
            uint32_t loc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) loc[i] = tc_ld(&h1[(TR_THREADS - 1 - threadIdx.x) * 4 + i]);
            const uint32_t total = tc_find<4>(loc, TR_THREADS, k, scr, b1, k1, c1);
            if (!poll || total == S) break;
            if (it >= max_poll) { TC_POLL_GIVE_UP(1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        // ---- digit 2: (key >> 8) & 0xFFF among key >> 20 == b1 -----------------------------------------------------
        if (slot == 0 && threadIdx.x < 256) tc_st(&h3[threadIdx.x], 0u);   // self-cleaning: filled below, read after the last barrier
        for (int i = threadIdx.x; i < 4096; i += TR_THREADS) lh[i] = 0;
        __syncthreads();
        if (full && (kmin >> 8) == (kmax >> 8)) {   // all of this thread's keys in one bin (rows of equal scores): one weighted add
            if ((kmin >> 20) == b1) atomicAdd(&lh[(kmin >> 8) & 0xFFFu], (uint32_t)PER);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if ((keys[j] >> 20) == b1 && keys[j]) atomicAdd(&lh[(keys[j] >> 8) & 0xFFFu], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += TR_THREADS) {
            const uint32_t c = lh[i];
            if (c) tc_add(&h2[i], c);
        }
        cluster_barrier(cs, 2, &s_fail[0], true);   // (the first counter barrier of the launch: round 1 is polled or was done by the scorer)
        uint32_t b2, k2;
        {
            uint32_t loc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) loc[i] = tc_ld(&h2[(TR_THREADS - 1 - threadIdx.x) * 4 + i]);
            row_find_bin_regs<4>(loc, TR_THREADS, k1, scr, b2, k2);
        }
        const uint32_t prefix = (b1 << 12) | b2;
        // ---- digit 3: key & 0xFF among key >> 8 == prefix; per-slot suffix table + count of larger prefixes ------------
        if (threadIdx.x < 256) lh[threadIdx.x] = 0;
        __syncthreads();
        uint32_t ngt = 0;
        if (full && kmin == kmax) {
            if ((kmin >> 8) == prefix) atomicAdd(&lh[kmin & 0xFFu], (uint32_t)PER);
            ngt = (kmin >> 8) > prefix ? (uint32_t)PER : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t p = keys[j] >> 8;
                ngt += (keys[j] && p > prefix) ? 1u : 0u;
                if (keys[j] && p == prefix) atomicAdd(&lh[keys[j] & 0xFFu], 1u);
            }
        }
        uint32_t ngt_tot;
        row_excl_scan(ngt, scr, &ngt_tot);   // (its barriers also cover the LDS atomics)
        {
            // thread t < 256 owns bin 255 - t: the exclusive scan over threads counts the keys in HIGHER bins
            const uint32_t d = 255u - threadIdx.x;
            const uint32_t c = threadIdx.x < 256 ? lh[d & 255u] : 0u;
            uint32_t tot2;
            const uint32_t above = row_excl_scan(c, scr, &tot2);
            uint32_t* tab = TCW(chunk_hist) + ((size_t)row * TC_SLOTS + slot) * 257;
            if (threadIdx.x < 256) {
                tc_st(&tab[d], above + c);   // suffix[d] = #(digit >= d)
                if (c) tc_add(&h3[d], c);
            }
            if (threadIdx.x == 0) {
                tc_st(&tab[256], 0u);
                tc_st(&TCW(chunk_gt)[(size_t)row * TC_SLOTS + slot], ngt_tot);
            }
        }
        cluster_barrier(cs, 3, &s_fail[0], false);
        uint32_t b3;
        {
            uint32_t loc[1];
            loc[0] = threadIdx.x < 256 ? tc_ld(&h3[255u - threadIdx.x]) : 0u;
            // the cluster's give-up flag rides on the same round trip as the histogram read-back (thread 256 has no bin to fetch)
            if (threadIdx.x == 256) s_fail[1] = tc_ld(cs.cl_flag);
            row_find_bin_regs<1>(loc, 256, k2, scr, b3, quota);   // (its barriers publish s_fail[1])
        }
        if ((s_fail[0] | s_fail[1]) == 0) {
            T = (prefix << 8) | b3;
            // kept elements in the slots before this one
            uint32_t gt_part = 0, eq_part = 0;
            if (threadIdx.x < slot) {
                const uint32_t* sf = TCW(chunk_hist) + ((size_t)row * TC_SLOTS + threadIdx.x) * 257;
                const uint32_t ge = tc_ld(&sf[b3]), gt = tc_ld(&sf[b3 + 1]);
                gt_part = tc_ld(&TCW(chunk_gt)[(size_t)row * TC_SLOTS + threadIdx.x]) + gt;
                eq_part = ge - gt;
            }
            row_excl_scan(gt_part, scr, &gt_before);
            row_excl_scan(eq_part, scr, &eq_before);
            // self-cleaning: this row's hist1 / hist2 are dead (every workgroup read them before the last barrier)
            for (uint32_t i = slot * TR_THREADS + threadIdx.x; i < 4096; i += TC_SLOTS * TR_THREADS) {
                tc_st(&h1[i], 0u);
                tc_st(&h2[i], 0u);
            }
        }
        }   // three-round form
        int32_t* out = a.idx + (int64_t)row * a.idx_stride;
        // ---- a barrier of this cluster timed out: NO index of this row is trustworthy ------------------------------------------
        // Every give-up of a cluster happens before any of its workgroups gets past the last barrier legitimately (that takes all
        // 32 arrivals, and a workgroup arrives at barrier 3 only after its own earlier give-ups), and a workgroup that gave up at
        // barrier 3 itself knows it from s_fail[0]: so every workgroup of the cluster takes this branch, none writes an index, and
        // together they fill the row's k + tail_n entries with -1 (kvp_gather_kv: a row of NaN instead of somebody else's token).
        // (A polled round 1 that timed out is one more "earlier give-up".)
        if (s_fail[0] | s_fail[1]) {
            // (also reached with a STALE flag: a workspace reused as "clean" after a failure without a zero-fill -- report that too)
            if (threadIdx.x == 0 && slot == 0) tc_report(cs, s_fail[0] ? s_fail[0] : s_fail[1]);
            const uint32_t tot_out = k + a.tail_n, per = (tot_out + TC_SLOTS - 1) / TC_SLOTS;
            for (uint32_t j = slot * per + threadIdx.x; j < min((slot + 1) * per, tot_out); j += TR_THREADS) out[j] = -1;
            return;
        }
#ifdef KVP_TC_FAULT_INJECTION
        // test twin only (tests/_fault_child.py "paths"): which form finished this (cluster, slot) -- 2 = two-hop, 1 = three rounds --
        // in the spare words behind the barrier lines
        if (threadIdx.x == 0) tc_st(&TCW(bar)[TC_MAXC * 32 + 32 + slot * 32 + cluster], done ? 2u : 1u);
#endif
        // self-cleaning: the window histogram is dead -- every workgroup of the row searched its COMPLETE state before it arrived at
        // the barrier(s) this workgroup has passed since (the candidates' barrier, or barriers 2 and 3 of the three-round form)
        if (!HIST1 && slot == 0 && wvu < (uint32_t)(TC_WB / 64)) tc_st(&(TCW(histw) + (size_t)row * TC_WB)[threadIdx.x], 0u);
        // ---- ordered compaction: keys > T, and the first `quota` keys == T ---------------------------------------------
        uint32_t cg = 0, ce = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            cg += keys[j] > T ? 1u : 0u;
            ce += keys[j] == T ? 1u : 0u;
        }
        uint32_t tot;
        const uint32_t ex = row_excl_scan(cg | (ce << 16), scr, &tot);   // L <= 8192: both fields < 65536
        uint32_t g = gt_before + (ex & 0xFFFFu), e = eq_before + (ex >> 16);
        // ranks of this workgroup: [rank0, rank0 + nmine)
        const uint32_t rank0 = gt_before + min(eq_before, quota);
        const uint32_t nmine = (tot & 0xFFFFu) + (min(eq_before + (tot >> 16), quota) - min(eq_before, quota));
        const uint32_t off = a.pos_base + (a.nseg > 1 ? (row % a.nseg) * a.seg_len : 0u);
        if (slot == 0)
            for (uint32_t j = threadIdx.x; j < a.tail_n; j += TR_THREADS) out[k + j] = (int32_t)(off + a.tail_start + j);
        int32_t* ob = reinterpret_cast<int32_t*>(lh);   // staged: the ranks of one thread are consecutive, the stores below coalesced
        const uint32_t Ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool isg = keys[j] > Ts;
            const bool ise = keys[j] == Ts;
            if (isg || (ise && e < quota)) {
                const uint32_t rank = g + (e < quota ? e : quota);
                if (rank < k) ob[rank - rank0] = (int32_t)(off + p0 + j);
            }
            g += isg ? 1u : 0u;
            e += ise ? 1u : 0u;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nmine; i += TR_THREADS)
            if (rank0 + i < k) out[rank0 + i] = ob[i];
#undef TC_POLL_GIVE_UP
    }
}

// ONE cluster select in flight per device (round 6, LAB R6.12).  Two of them dispatched at the same moment from different queues can each be handed
// the CUs the other's rows wait for (a row's 32 workgroups sit on four CUs of every XCD; the XCDs' dispatchers pick queues independently): a circular
// wait that only the time-out ends.  Within a process the library keeps the contract itself: a cluster launch on a stream OTHER than the previous
// one's first waits (hipStreamWaitEvent: on the device, no host sync) for an event recorded after that previous launch.  A process that only ever
// uses one stream -- the reference's hook -- pays a mutex and a pointer compare: no event is created, recorded or waited for.  Streams under graph
// capture are left alone.  Across PROCESSES sharing a GPU nothing here helps: KVP_TK_CLUSTER=0.
struct ClusterStreamGuard {
    std::mutex mu;
    hipStream_t last = nullptr;
    hipEvent_t ev = nullptr;
    bool have = false, multi = false;
};
ClusterStreamGuard g_cluster_guard[64];

struct ClusterLaunchScope {   // held across the launch: enter (wait if the stream changed) .. leave (record once a second stream has been seen)
    ClusterStreamGuard* g = nullptr;
    hipStream_t s;
    bool track = false;
    explicit ClusterLaunchScope(hipStream_t stream) : s(stream) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
        g = &g_cluster_guard[dev];
        g->mu.lock();
        track = true;
        if (g->have && g->last != s) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); track = false; return; }
            if (!g->ev && hipEventCreateWithFlags(&g->ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g->ev = nullptr; return; }
            if (!g->multi) {   // first change of stream: nothing was recorded after the previous launch yet -- record behind everything enqueued on its stream so far
                g->multi = true;
                if (hipEventRecord(g->ev, g->last) != hipSuccess) { (void)hipGetLastError(); return; }   // (a destroyed stream: nothing left to wait for)
            }
            if (hipStreamWaitEvent(s, g->ev, 0) != hipSuccess) (void)hipGetLastError();
        }
    }
    ~ClusterLaunchScope() {
        if (!g) return;
        if (track) {
            if (g->multi && g->ev && hipEventRecord(g->ev, s) != hipSuccess) (void)hipGetLastError();
            g->last = s;
            g->have = true;
        }
        g->mu.unlock();
    }
};

template <int PER, int MODE, bool HIST1>
int launch_one(const ClusterArgs& a, hipStream_t stream) {
    if (!topk_cluster_launchable()) return 1;
    ClusterLaunchScope scope(stream);
    ClusterArgs b = a;
    b.host_flag = kvp_async_flag();
    b.timeout_ticks = (uint32_t)std::min<int64_t>(std::max<int64_t>(kvp_env_int("KVP_TC_TIMEOUT_US", 1000000), 100), 20000000) * 100u;
#ifdef KVP_TC_FAULT_INJECTION
    b.test_delay_slot = kvp_env_int("KVP_TC_TEST_DELAY_SLOT", -1);
#endif
    // one launch carries up to TC_MAXC rows: TC_CLUSTERS clusters fit the device at once, the others start in dispatch order as the
    // first retire (a row's workgroups are consecutive blocks: see the kernel)
    for (b.row_base = 0; b.row_base < b.R; b.row_base += TC_MAXC) {
        b.report = kvp_async_next_seq() << 8;
        const uint32_t nclusters = std::min<uint32_t>(b.R - b.row_base, (uint32_t)TC_MAXC);
        KVP_LAUNCH("topk_cluster_kernel", stream, (topk_cluster_kernel<PER, MODE, HIST1><<<nclusters * TC_SLOTS, TR_THREADS, 0, stream>>>(b)));
    }
    return 0;
}

template <int MODE, bool HIST1>
int launch_per(const ClusterArgs& a, hipStream_t stream) {
    const int64_t per = ((int64_t)a.S + (int64_t)TC_SLOTS * TR_THREADS - 1) / ((int64_t)TC_SLOTS * TR_THREADS);
    if (per <= 1) return launch_one<1, MODE, HIST1>(a, stream);
    if (per <= 2) return launch_one<2, MODE, HIST1>(a, stream);
    if (per <= 4) return launch_one<4, MODE, HIST1>(a, stream);
    return launch_one<8, MODE, HIST1>(a, stream);   // (16 keys per thread spill scalar registers: rows beyond 262144 take the passes)
}

}  // namespace

// All TC_CLUSTERS * TC_SLOTS workgroups must be resident at once (the barriers spin).  Every instantiation fits an empty CU
// (1024 threads, <= 128 VGPRs, <= 50 KB of LDS), so the question is whether the current device has that many CUs.
bool topk_cluster_launchable() {
    static int ok[64] = {0};   // per device: 0 unknown, 1 yes, -1 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (ok[dev] == 0) {
        int cus = 0;
        ok[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= TC_CLUSTERS * TC_SLOTS) ? 1 : -1;
    }
    return ok[dev] > 0;
}

// (few long rows: the structure buys latency, not throughput.  Round 3, three-round form, profiles/r03_select_cluster_lab.txt: 8 x 131008
// flat scores 22.5 us = the four (chunk, row) launches; 1 row 18 against 24 us; 16 rows (two launches) 44 against 29 us -- hence 8 rows
// then.  Round 6: with the two-hop form (13.6 us per 8 rows) and all rows in ONE launch, batches of 2 - 4 elements (R = 16 .. 32) take
// it too: what it saves there is SnapKV's pooling launch / score round trip (POOL5 loader) and Knorm's norms never leaving the chip.)
bool topk_cluster_eligible(int64_t R, int64_t S) {
    return R <= TC_MAXC && S > 16384 && S <= (int64_t)TC_SLOTS * TR_THREADS * 8 && kvp_env_int("KVP_TK_CLUSTER", 1) != 0;
}

// returns KVP_OK, an error, or 1 = "not launched" (the device cannot hold the grid: the caller takes the (chunk, row) passes)
int topk_cluster_select(int mode, const float* scores, int64_t row_stride, float inv, const void* x, int dtype, int64_t x_sb, int64_t x_sh,
                        int64_t x_ss, int64_t H, float scale, int64_t R, int64_t S, int64_t k, int32_t* idx, int64_t idx_stride,
                        uint32_t tail_start, uint32_t tail_n, const TopkWs& w, bool hist1_ready, hipStream_t stream, uint32_t nseg,
                        uint32_t seg_len, uint32_t pos_base) {
    ClusterArgs a;
    a.scores = scores; a.row_stride = row_stride;
    a.R = (uint32_t)R; a.S = (uint32_t)S; a.k = (uint32_t)k; a.kmask = w.kmask;
    a.inv = inv;
    a.idx = idx; a.idx_stride = idx_stride;
    a.tail_start = tail_start; a.tail_n = tail_n; a.nseg = nseg; a.seg_len = seg_len; a.pos_base = pos_base;
    a.ws0 = w.base; a.wsR = w.lay_R; a.ws_ntab = w.lay_ntab;
    a.x = x; a.x_sb = x_sb; a.x_sh = x_sh; a.x_ss = x_ss; a.H = (uint32_t)std::max<int64_t>(1, H); a.scale = scale;
    int rc;
    switch (mode) {
        case TOPK_CLUSTER_POOL5: rc = launch_per<TC_POOL5, false>(a, stream); break;
        case TOPK_CLUSTER_KNORM: rc = dtype == KVP_BF16 ? launch_per<TC_KNORM_BF16, false>(a, stream) : launch_per<TC_KNORM_F16, false>(a, stream); break;
        default: rc = hist1_ready ? launch_per<TC_SCORES, true>(a, stream) : launch_per<TC_SCORES, false>(a, stream); break;
    }
    if (rc == 0) KVP_CHECK_LAUNCH("topk(cluster)");
    return rc;
}
