// KVP_ORDER_SCORE: the retained indices in DESCENDING SCORE order (what `scores.topk(n_kept).indices` returns with
// sorted=True, kvpress/presses/scorer_press.py:95), ties in ascending position.  This is the order in which the reference stores
// K' / V' (scorer_press.py:96-100): `press.kept_order = "score"` reproduces its tensors.
//
// A hand-written segmented sort (round 5; rounds 3-4 called rocPRIM's device-wide radix sort: 138 us for 8 x 65536 pairs).
// Input: kvp_topk_select's position-ordered selection idx[R][k].  Every kept index becomes ONE 64-bit composite
//     (~order_key(score) << 32) | position          -- ascending composite = descending score, ties by ascending position --
// (a poisoned index, -1 from a select that reported a failure, becomes (0 << 32) | 0x80000000 | slot: first in its row, written
// back as -1; padding slots are (0xFFFFFFFF << 32) | 0x80000000 | slot: last).  Composites are UNIQUE within a row, which is what
// bounds the bucket sizes below whatever the score distribution is (flat rows put half of their scores into one 12-bit radix bin).
//
// Sort by regular sampling (PSRS), two launches for up to 32 tiles per row:
//   order_tiles_kernel   grid (tiles, rows), 1024 threads: a tile of T = 1024 E composites (E = 2 or 4 per thread) is fetched
//                        (index -> score -> key), sorted by a bitonic network that lives in registers (strides inside a thread: plain
//                        compare-exchange; inside a wave: shuffles; across waves: one LDS exchange per step), and written back sorted
//                        together with its regular samples (every 64th element: NS = 32 or 64 per tile).  One tile per row: the indices
//                        are written directly.
//   order_buckets_kernel grid (tiles, rows): workgroup b sorts the row's <= 32 NS samples itself (every workgroup of a row does the
//                        same few microseconds of work instead of a third launch + hop), takes the samples of rank NS b - 1 and
//                        NS (b + 1) - 1 as its two pivots, finds per tile -- ballot over the tile's samples, then ONE coalesced load
//                        of the 64 elements between two samples -- how many elements lie below each pivot, gathers those
//                        < 2 T composites (the PSRS bound; typically 1.2 T) into LDS, sorts them with the same
//                        network at 2 E per thread and writes the positions at the bucket's offset (= the sum of its lower counts).
// More than 32 tiles of 4096 (k > 131072): a plain global merge network over sorted 2048-tiles (log^2 launches; correct for any k,
// used by no configuration of this package's benchmarks).
#include "kvp_common.h"
#include "topk_block.h"
#include "topk_internal.h"

namespace {

typedef unsigned long long u64;
constexpr u64 OS_PAD_HI = 0xFFFFFFFFull << 32;
constexpr int OS_GROUP = 64;        // elements between two regular samples of a sorted tile: T / 64 samples per tile (32 or 64)
constexpr int OS_MAX_TILES = 32;    // tiles per row on the two-launch path: 32 tiles x T / 64 samples = one 1024- / 2048-element sort

__device__ __forceinline__ u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a < b ? b : a; }
// value of lane (lane ^ D) for a compile-time lane distance: quad permutes (VALU speed) for 1 / 2, the LDS crossbar's bit-mode swizzle
// for 4 / 8 / 16 (no address register), ds_bpermute for 32
template <int D> __device__ __forceinline__ uint32_t xor_lane32(uint32_t x) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);         // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    else if constexpr (D <= 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (D << 10) | 0x1F);           // xor mask D, and mask 0x1f
    else return (uint32_t)__shfl_xor((int)x, D);
}
template <int D> __device__ __forceinline__ u64 xor_lane64(u64 v) {
    return ((u64)xor_lane32<D>((uint32_t)(v >> 32)) << 32) | xor_lane32<D>((uint32_t)v);
}

// Bitonic sort of the 1024 * E composites of a 1024-thread workgroup, ascending; thread t holds elements E t .. E t + E - 1 before and
// after.  xch: LDS, 1024 * E words of 8 bytes (used only by the steps whose partner sits in another wave).  Composites are unique,
// so ONE 64-bit compare decides a compare-exchange: keep the partner's value iff (partner < mine) == (I keep the minimum).
// Every (size, stride) step is its own template instance: lane distances are compile-time (DPP / swizzle instead of ds_bpermute).
template <int E, uint32_t SIZE, uint32_t STRIDE>
__device__ __forceinline__ void bitonic_step(u64 (&v)[E], u64* xch, uint32_t t_) {
    // (opaque copy of the thread index per step: otherwise the compiler hoists every step's lane masks out of the network and
    // keeps ~80 of them alive in scalar registers -- spills)
    uint32_t t = t_;
    asm volatile("" : "+v"(t));
    if constexpr (STRIDE >= (uint32_t)E) {
        constexpr uint32_t D = STRIDE / E;                   // partner thread = t ^ D, same slot
        const bool keep_min = ((t & D) == 0) == (((E * t) & SIZE) == 0);   // (lower element of the pair) == (ascending block); SIZE >= 2 E here
        if constexpr (D < 64) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const u64 p = xor_lane64<(int)D>(v[i]);
                v[i] = ((p < v[i]) == keep_min) ? p : v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) xch[E * t + i] = v[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const u64 p = xch[E * (t ^ D) + i];
                v[i] = ((p < v[i]) == keep_min) ? p : v[i];
            }
            __syncthreads();
        }
    } else {                                                 // both elements in this thread's registers
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if ((i & STRIDE) == 0) {
                const bool asc = ((E * t + i) & SIZE) == 0;
                const u64 a = v[i], b = v[i | STRIDE];
                const bool swap = (b < a) == asc;
                v[i] = swap ? b : a;
                v[i | STRIDE] = swap ? a : b;
            }
        }
    }
    if constexpr (STRIDE > 1) bitonic_step<E, SIZE, STRIDE / 2>(v, xch, t_);
}
template <int E, uint32_t SIZE>
__device__ __forceinline__ void bitonic_phase(u64 (&v)[E], u64* xch, uint32_t t) {
    bitonic_step<E, SIZE, SIZE / 2>(v, xch, t);
    if constexpr (SIZE < 1024u * E) bitonic_phase<E, SIZE * 2>(v, xch, t);
}
template <int E>
__device__ __forceinline__ void bitonic_sort_block(u64 (&v)[E], u64* xch) {
    bitonic_phase<E, 2>(v, xch, threadIdx.x);
}

// Where a kept position's score comes from.  ORDER_SCORES: the score row.  ORDER_POOL5: SnapKV's un-pooled column sums -- the fused
// compress never writes its pooled scores, so the 5-tap zero-padded average is recomputed here with the additions and the scale of
// the select's loader (topk_cluster.hip TC_POOL5, snapkv_pool_kernel: the same bits, hence the order the modular sequence gives).
// Positions >= ncols are the window columns the reference pads with max + 1 (snapkv_press.py:103): they sort first, by position.
enum { ORDER_SCORES = 0, ORDER_POOL5 = 1 };
template <int MODE>
__device__ __forceinline__ u64 order_composite(const float* __restrict__ row, int32_t p, uint32_t S, uint32_t ncols, float inv, uint32_t kmask,
                                               uint32_t slot) {
    if ((uint32_t)p >= S) return (u64)(0x80000000u | slot);                    // poisoned: sorts first, written back as -1
    if ((uint32_t)p >= ncols) return (u64)(uint32_t)p;                         // kept by construction: ahead of every scored position
    float sc;
    if (MODE == ORDER_POOL5) {
        float sum = 0.f;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int32_t q = p - 2 + d;
            sum += (q >= 0 && (uint32_t)q < ncols) ? row[q] : 0.f;
        }
        sc = sum * inv;
    } else {
        sc = row[p];
    }
    return ((u64)(~(float_to_key(sc) ^ kmask)) << 32) | (uint32_t)p;
}
__device__ __forceinline__ int32_t order_position(u64 c) { return (c & 0x80000000ull) ? -1 : (int32_t)(uint32_t)c; }

struct OrderArgs {
    const float* scores;
    int64_t row_stride;
    int32_t* idx;          // [R][k] in: ascending positions; out: descending score
    uint32_t S, k, kmask, ntiles;
    uint32_t ncols;        // scored columns of a row (positions ncols .. S - 1 are kept by construction); ORDER_SCORES callers pass S
    float inv;             // ORDER_POOL5: 1 / (G * W * kernel_size)
    u64* tiles;            // [R][ntiles * T] sorted tiles
    u64* samples;          // [R][ntiles][T / OS_GROUP]; nullptr: none wanted (the global network)
};

// ---- launch 1: sort the tiles ---------------------------------------------------------------------------------------------------
template <int E, int MODE>
__global__ __launch_bounds__(1024) void order_tiles_kernel(OrderArgs a) {
    constexpr uint32_t T = 1024u * E;
    __shared__ __attribute__((aligned(16))) u64 xch[T];
    const uint32_t tile = blockIdx.x, r = blockIdx.y, t = threadIdx.x;
    const float* row = a.scores + (int64_t)r * a.row_stride;
    int32_t* ir = a.idx + (size_t)r * a.k;
    u64 v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const uint32_t j = tile * T + E * t + i;
        v[i] = j < a.k ? order_composite<MODE>(row, ir[j], a.S, a.ncols, a.inv, a.kmask, j) : (OS_PAD_HI | 0x80000000u | j);
    }
    bitonic_sort_block<E>(v, xch);
    if (a.ntiles == 1) {                                        // the whole row: done
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t j = E * t + i;
            if (j < a.k) ir[j] = order_position(v[i]);
        }
        return;
    }
    u64* out = a.tiles + ((size_t)r * a.ntiles + tile) * T;
#pragma unroll
    for (int i = 0; i < E; ++i) out[E * t + i] = v[i];
    // sample m = element 64 m + 63: the last slot of every (64 / E)-th thread
    constexpr uint32_t G = OS_GROUP, NS = T / G;
    if (a.samples && (t % (G / E)) == (G / E) - 1) a.samples[((size_t)r * a.ntiles + tile) * NS + t / (G / E)] = v[E - 1];
}

// ---- launch 2: one bucket per workgroup -----------------------------------------------------------------------------------------
// Bucket sizes: pivot j is the sample of rank NS j - 1, so between j T and j T + tiles (G - 1) elements of the row are <= pivot j
// (a tile with c samples <= pivot has between G c and G c + G - 1 such elements): a bucket holds at most T + 32 * 63 < 2 T = CAP.
template <int E>   // E = composites per thread of launch 1; this kernel sorts 2 E per thread
__global__ __launch_bounds__(1024) void order_buckets_kernel(OrderArgs a) {
    constexpr uint32_t T = 1024u * E, G = OS_GROUP, NS = T / G, CAP = 2 * T;
    constexpr int ES = (int)(OS_MAX_TILES * NS / 1024);                        // samples per thread of the sample sort: 1 (E = 2) or 2 (E = 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char os_lds[];
    u64* stage = reinterpret_cast<u64*>(os_lds);                               // [CAP]: sample sort, then the bucket
    u64* smp = stage + CAP;                                                    // [OS_MAX_TILES][NS] the tiles' samples, unsorted
    __shared__ uint32_t cnt[2][OS_MAX_TILES];                                  // per tile: elements <= lower pivot / <= upper pivot
    __shared__ uint32_t pre[OS_MAX_TILES + 2];                                 // exclusive scan of the piece sizes; [nt] = bucket size; [nt + 1] = output offset
    const uint32_t b = blockIdx.x, r = blockIdx.y, t = threadIdx.x, nt = a.ntiles;
    const uint32_t lane = t & 63, wv = t >> 6;
    const u64* tiles = a.tiles + (size_t)r * nt * T;
    const u64* sg = a.samples + (size_t)r * nt * NS;

    // (1) the row's samples: registers for the sort, an unsorted copy in LDS for the per-tile searches
    u64 sv[ES];
#pragma unroll
    for (int i = 0; i < ES; ++i) {
        const uint32_t j = ES * t + i;
        sv[i] = j < nt * NS ? sg[j] : (OS_PAD_HI | 0xC0000000u | j);           // (above every real composite and every tile pad)
        if (j < nt * NS) smp[j] = sv[i];
    }
    bitonic_sort_block<ES>(sv, stage);
#pragma unroll
    for (int i = 0; i < ES; ++i) stage[ES * t + i] = sv[i];
    __syncthreads();
    // (2) pivots of bucket b: samples of rank NS b - 1 (exclusive lower bound) and NS (b + 1) - 1 (inclusive upper bound)
    const bool has_lo = b > 0, has_hi = b + 1 < nt;
    const u64 plo = has_lo ? stage[NS * b - 1] : 0ull;
    const u64 phi = has_hi ? stage[NS * (b + 1) - 1] : ~0ull;
    __syncthreads();                                                           // stage is reused below
    // (3) per tile and pivot: number of elements <= pivot.  A wave per search: ballot over the tile's NS samples, then the G = 64
    //     elements between the last sample <= pivot and the next one (one coalesced 512-byte load).
    for (uint32_t q = wv; q < 2 * nt; q += TR_WAVES) {
        const uint32_t tile = q >> 1, which = q & 1;
        uint32_t count;
        if (which == 0 && !has_lo) count = 0;
        else if (which == 1 && !has_hi) count = T;
        else {
            const u64 piv = which ? phi : plo;
            const uint32_t c = (uint32_t)__popcll(__ballot(lane < NS && smp[tile * NS + (lane & (NS - 1))] <= piv));   // samples ascend: the first c
            count = T;
            if (c < NS) {
                const u64 x = tiles[(size_t)tile * T + G * c + lane];
                count = G * c + (uint32_t)__popcll(__ballot(x <= piv));
            }
        }
        if (lane == 0) cnt[which][tile] = count;
    }
    __syncthreads();
    // (4) piece sizes -> offsets inside the bucket, bucket size, output offset (wave 0; nt <= 32 <= 64 lanes)
    if (wv == 0) {
        const uint32_t lo = lane < nt ? cnt[0][lane] : 0u, hi = lane < nt ? cnt[1][lane] : 0u;
        const uint32_t n = hi - lo;
        const uint32_t inc = wave_incl_scan(n), inclo = wave_incl_scan(lo);
        if (lane < nt) pre[lane] = inc - n;
        if (lane == 63) {
            pre[nt] = inc;
            pre[nt + 1] = inclo;
        }
    }
    __syncthreads();
    const uint32_t total = pre[nt], gofs = pre[nt + 1];
    // (5) gather the pieces (each a contiguous range of a sorted tile) into LDS; pad to the capacity
    for (uint32_t tile = wv; tile < nt; tile += TR_WAVES) {
        const uint32_t lo = cnt[0][tile], n = cnt[1][tile] - lo, o = pre[tile];
        for (uint32_t j = lane; j < n && o + j < CAP; j += 64) stage[o + j] = tiles[(size_t)tile * T + lo + j];   // (o + n <= 1.5 T by the PSRS bound)
    }
    for (uint32_t j = total + t; j < CAP; j += 1024) stage[j] = OS_PAD_HI | 0xC0000000u | j;
    __syncthreads();
    u64 v[2 * E];
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) v[i] = stage[2 * E * t + i];
    __syncthreads();
    // (6) sort the bucket, write the positions
    bitonic_sort_block<2 * E>(v, stage);
    int32_t* out = a.idx + (size_t)r * a.k;
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) {
        const uint32_t j = 2 * E * t + i;
        if (j < total && gofs + j < a.k) out[gofs + j] = order_position(v[i]);
    }
}

// ---- any k: global merge network over sorted 2048-tiles ---------------------------------------------------------------------------
// tiles: [R][npad] with npad = a power of two >= k, pads last, every `size / 2`-long run ascending.  Merging two ascending runs into
// one of length `size`: ONE mirrored compare-exchange (element i of the left run against element size - 1 - i of the pair) leaves
// every left element <= every right element and both halves bitonic; half-cleaners of stride size / 4 .. 1, all ascending, finish.
__global__ __launch_bounds__(256) void order_global_mirror_kernel(u64* __restrict__ tiles, uint32_t npad, uint32_t size) {
    u64* row = tiles + (size_t)blockIdx.y * npad;
    const uint32_t half = size / 2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npad / 2; i += gridDim.x * blockDim.x) {
        const uint32_t blk = i / half, off = i % half;
        const uint32_t e = blk * size + off, f = blk * size + size - 1 - off;
        const u64 x = row[e], y = row[f];
        row[e] = umin64(x, y);
        row[f] = umax64(x, y);
    }
}
__global__ __launch_bounds__(256) void order_global_step_kernel(u64* __restrict__ tiles, uint32_t npad, uint32_t stride) {
    u64* row = tiles + (size_t)blockIdx.y * npad;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npad / 2; i += gridDim.x * blockDim.x) {
        const uint32_t e = ((i / stride) * 2 * stride) + (i % stride);        // lower element of pair i
        const u64 x = row[e], y = row[e + stride];
        row[e] = umin64(x, y);
        row[e + stride] = umax64(x, y);
    }
}
// half-cleaners of stride 1024 .. 1 inside a 2048-tile, ascending; last = 1: the positions go to idx instead
__global__ __launch_bounds__(1024) void order_global_tail_kernel(u64* __restrict__ tiles, uint32_t npad, int32_t* __restrict__ idx, uint32_t k, uint32_t last) {
    __shared__ __attribute__((aligned(16))) u64 xch[2048];
    u64* row = tiles + (size_t)blockIdx.y * npad;
    const uint32_t t = threadIdx.x, base = blockIdx.x * 2048u;
    xch[t] = row[base + t];
    xch[t + 1024] = row[base + t + 1024];
    __syncthreads();
    for (uint32_t stride = 1024; stride >= 1; stride >>= 1) {
        const uint32_t e = ((t / stride) * 2 * stride) + (t % stride);
        const u64 x = xch[e], y = xch[e + stride];
        xch[e] = umin64(x, y);
        xch[e + stride] = umax64(x, y);
        __syncthreads();
    }
    if (last) {
        int32_t* out = idx + (size_t)blockIdx.y * k;
        if (base + t < k) out[base + t] = order_position(xch[t]);
        if (base + t + 1024 < k) out[base + t + 1024] = order_position(xch[t + 1024]);
    } else {
        row[base + t] = xch[t];
        row[base + t + 1024] = xch[t + 1024];
    }
}

struct OrderPlan {
    int e;               // composites per thread of launch 1 (2 or 4); 0 = the global network
    uint32_t T, ntiles;
    size_t tiles_bytes, samples_bytes, total_bytes;
};
OrderPlan order_plan(int64_t R, int64_t k) {
    OrderPlan p{};
    const size_t rows = (size_t)std::max<int64_t>(1, R);
    if (k <= 2048 * OS_MAX_TILES) p.e = 2;
    else if (k <= 4096 * OS_MAX_TILES) p.e = 4;
    if (p.e) {
        p.T = 1024u * p.e;
        p.ntiles = (uint32_t)std::max<int64_t>(1, (k + p.T - 1) / p.T);
        p.tiles_bytes = kvp_align_up(rows * p.ntiles * p.T * 8, 256);
        p.samples_bytes = kvp_align_up(rows * p.ntiles * (p.T / OS_GROUP) * 8, 256);
    } else {
        uint64_t npad = 4096;
        while ((int64_t)npad < k) npad <<= 1;
        p.T = 2048;
        p.ntiles = (uint32_t)(npad / 2048);
        p.tiles_bytes = kvp_align_up(rows * npad * 8, 256);
        p.samples_bytes = 256;
    }
    p.total_bytes = p.tiles_bytes + p.samples_bytes;
    return p;
}

template <int E>
int launch_psrs(const OrderArgs& a, int mode, int64_t R, hipStream_t stream) {
    const dim3 grid(a.ntiles, (uint32_t)R);
    if (mode == ORDER_POOL5) KVP_LAUNCH("order_tiles_kernel", stream, (order_tiles_kernel<E, ORDER_POOL5><<<grid, 1024, 0, stream>>>(a)));
    else KVP_LAUNCH("order_tiles_kernel", stream, (order_tiles_kernel<E, ORDER_SCORES><<<grid, 1024, 0, stream>>>(a)));
    KVP_CHECK_LAUNCH("topk(order: tiles)");
    if (a.ntiles == 1) return KVP_OK;
    const size_t lds = (size_t)(2 * 1024 * E + OS_MAX_TILES * (1024 * E / OS_GROUP)) * 8;
    static bool raised[2] = {false, false};                      // (per instantiation; the attribute is per function, set once)
    if (lds > 48 * 1024 && !raised[E / 4]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(order_buckets_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            kvp_set_error("topk(order): cannot raise the dynamic LDS limit to %zu bytes", lds);
            return KVP_EHIP;
        }
        raised[E / 4] = true;
    }
    KVP_LAUNCH("order_buckets_kernel", stream, (order_buckets_kernel<E><<<grid, 1024, lds, stream>>>(a)));
    KVP_CHECK_LAUNCH("topk(order: buckets)");
    return KVP_OK;
}

}  // namespace

size_t topk_order_workspace_bytes(int64_t R, int64_t k) { return (R <= 0 || k <= 0) ? 0 : order_plan(R, k).total_bytes; }

// idx [R, k] (contiguous, ascending positions from the select) is rewritten in descending-score order.  mode / ncols / inv: see
// order_composite (defaults: plain score rows of S columns).
int topk_order_by_score(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int32_t* idx, bool smallest, void* ws,
                        size_t ws_bytes, hipStream_t stream, int mode, int64_t ncols, float inv) {
    if (R == 0 || k == 0) return KVP_OK;
    KVP_CHECK_ARG(R <= 65535 && k < ((int64_t)1 << 30) && S < ((int64_t)1 << 31), "topk(order): shape too large (R=%ld k=%ld)", (long)R, (long)k);
    const OrderPlan p = order_plan(R, k);
    if (!ws || ws_bytes < p.total_bytes) {
        kvp_set_error("topk(order): workspace too small (%zu < %zu)", ws_bytes, p.total_bytes);
        return KVP_EWORKSPACE;
    }
    OrderArgs a;
    a.scores = scores; a.row_stride = row_stride; a.idx = idx;
    a.S = (uint32_t)S; a.k = (uint32_t)k; a.kmask = smallest ? 0xFFFFFFFFu : 0u; a.ntiles = p.ntiles;
    a.ncols = (uint32_t)(ncols < 0 ? S : ncols); a.inv = inv;
    a.tiles = static_cast<u64*>(ws);
    a.samples = reinterpret_cast<u64*>(static_cast<char*>(ws) + p.tiles_bytes);
    if (p.e == 2) return launch_psrs<2>(a, mode, R, stream);
    if (p.e == 4) return launch_psrs<4>(a, mode, R, stream);
    // any k: sorted 2048-tiles (launch 1 with padding tiles, no samples), then the global merge network
    const uint32_t npad = p.ntiles * 2048u;
    a.samples = nullptr;
    const dim3 gs(std::min<uint32_t>(npad / 512, 2048), (uint32_t)R), gt(p.ntiles, (uint32_t)R);
    if (mode == ORDER_POOL5) KVP_LAUNCH("order_tiles_kernel", stream, (order_tiles_kernel<2, ORDER_POOL5><<<gt, 1024, 0, stream>>>(a)));
    else KVP_LAUNCH("order_tiles_kernel", stream, (order_tiles_kernel<2, ORDER_SCORES><<<gt, 1024, 0, stream>>>(a)));
    for (uint32_t size = 4096; size <= npad; size <<= 1) {
        KVP_LAUNCH("order_global_mirror_kernel", stream, (order_global_mirror_kernel<<<gs, 256, 0, stream>>>(a.tiles, npad, size)));
        for (uint32_t stride = size / 4; stride >= 2048; stride >>= 1)
            KVP_LAUNCH("order_global_step_kernel", stream, (order_global_step_kernel<<<gs, 256, 0, stream>>>(a.tiles, npad, stride)));
        KVP_LAUNCH("order_global_tail_kernel", stream, (order_global_tail_kernel<<<gt, 1024, 0, stream>>>(a.tiles, npad, idx, a.k, size == npad ? 1u : 0u)));
    }
    KVP_CHECK_LAUNCH("topk(order: global network)");
    return KVP_OK;
}
