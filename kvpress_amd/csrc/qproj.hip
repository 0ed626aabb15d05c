// kvp_snapkv_qproj_rope: RoPE'd window queries straight from the hidden states of the last W = 64 tokens.
// Replaces `get_prerope_query_states(module, hidden_states[:, -W:])` (kvpress/utils.py:43-46: q_proj, view, transpose) followed
// by `q * cos + rotate_half(q) * sin` (snapkv_press.py:53-58) for a plain nn.Linear q_proj without bias.
//
// A 64 x hidden by hidden x (H_q * 128) product: 32 MiB of bf16 weights for Llama-3.1-8B, 2 GFLOP -- a pure streaming problem whose
// bound is NOT HBM but what each CU has to pull through its L2 port (~40 GB/s per CU with 256 CUs pulling: round 4, R4.1).  Rounds 1-4
// gave every workgroup 16 output columns over the WHOLE hidden dimension: 128 KiB of weights + the whole 512 KiB hidden window per CU =
// 160 MiB in flight chip-wide, 15 us (the library GEMM's 16 x 64 tiling moves the same bytes in the same time).  Round 5 splits the
// hidden dimension instead:
//   qproj_splitk_kernel   workgroup = (64 output columns -- 32 dims of a head and their rotate_half partners --, one of NS = 4 ranges of
//                         the hidden dimension): 128 KiB of weights + 128 KiB of the window per CU (64 MiB chip-wide).  K is walked
//                         in tiles of 128 elements: 64 weight rows + 64 hidden rows (32 KiB) by LDS-DMA into a ring of four buffers
//                         (XOR-swizzled on the global side); wave w multiplies k-step w & 3 of the tile for column blocks 2 (w >> 2),
//                         2 (w >> 2) + 1 and all four row blocks (v_mfma_f32_16x16x32, 8 per tile and wave).  The four k-step partials
//                         of a workgroup are added in a fixed order through LDS and written as float32 partial products
//                         [b][split][column block][64 rows][64 columns] (workspace).  Workgroups of one split share an XCD (block
//                         index % 4 = split), so an L2 sees one quarter of the window.
//   qproj_reduce_rope_kernel  adds the NS partials in split order (fixed: deterministic), rounds to the model dtype like a GEMM
//                         output, applies RoPE with torch's per-op rounding (rope_elem) and writes [B, H_q, W, D].
// Two launches instead of an in-launch seam: a last-arriver ticket would need zeroed tickets in a caller-owned workspace (no such
// contract on kvp_snapkv_score_hidden) and a cluster barrier a failure path; the second launch costs ~3 us on the chain.
// History of the single-launch kernel (16 columns x whole K per workgroup; rounds 1-4, profiles/r04_qproj_lab.txt): 20 us in isolation,
// 17.3 with a rotated tile walk; weights in registers / deeper rings / register staging did not help (tools/lab_patches/qproj_v2_v3.diff)
// -- every variant moved the same 640 KiB per CU.
#include "kvp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QP_THREADS = 512;
constexpr int QP_ROWS = 64;     // window
constexpr int QP_COLS = 64;     // output columns per workgroup: dims d0 .. d0 + 31 of a head and d0 + 64 .. d0 + 95
constexpr int QP_KT = 128;      // K elements per tile = 4 k-steps of 32
constexpr int QP_ROWB = QP_KT * 2;                         // 256 B per tile row = 16 chunks of 16 B
constexpr int QP_TILEB = (QP_COLS + QP_ROWS) * QP_ROWB;    // 16 KiB of weights + 16 KiB of hidden states
constexpr int QP_REQ = QP_TILEB / 16 / QP_THREADS;         // LDS-DMA requests per thread and tile (4)
constexpr int QP_NBUF = 4;      // three tiles (96 KiB) in flight per CU: the walk is only 8 tiles long
constexpr int QP_MAXSPLIT = 4;
static_assert(QP_TILEB % (16 * QP_THREADS) == 0, "whole requests");

template <int DT> __device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<KVP_BF16>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<KVP_F16>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT> __device__ __forceinline__ void st_dt(typename Elem<DT>::T* p, float x);
template <> __device__ __forceinline__ void st_dt<KVP_F16>(_Float16* p, float x) { *p = (_Float16)x; }
template <> __device__ __forceinline__ void st_dt<KVP_BF16>(uint16_t* p, float x) { *p = (uint16_t)(__float_as_uint(round_dt<KVP_BF16>(x)) >> 16); }

struct QprojArgs {
    const char* x;      // hidden window [B, 64, K]
    int64_t x_sb, x_sw; // BYTE strides
    const char* w;      // weight [Hq * 128, K] row-major, contiguous
    const void* cosp;   // [1 or B, 64, 128]
    const void* sinp;
    int64_t cs_sb, cs_sw;  // element strides
    void* out;          // [B, Hq, 64, 128] contiguous
    float* part;        // [B][nsplit][Hq * 2][64 rows][64 cols] float32 partial products
    uint32_t Hq, K, nsplit;
};

// Tile rows 0..63 = the 64 weight rows (columns of the output block), rows 64..127 = the 64 hidden-state rows; 256 B per row,
// 16-byte slot p of row r holds chunk p ^ (r & 15) (XOR swizzle applied on the global side of the DMA): the 16 lanes of a
// fragment read (16 consecutive rows, same chunk) hit 16 distinct slots.
template <int DT>
__global__ __launch_bounds__(QP_THREADS) void qproj_splitk_kernel(QprojArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // QP_NBUF * QP_TILEB; reused for the k-step partials
    const uint32_t split = blockIdx.x % a.nsplit, cb = blockIdx.x / a.nsplit, b = blockIdx.y;
    const uint32_t h = cb >> 1, d0 = (cb & 1) * 32;   // output columns: dims d0 .. d0 + 31 of head h, then d0 + 64 .. d0 + 95
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t l16 = lane & 15, kq = lane >> 4;   // fragment row / column, k-slot group (8 elements)
    const uint32_t kstep = wv & 3, chalf = wv >> 2;   // this wave: k-step of every tile, column blocks 2 chalf and 2 chalf + 1
    const uint32_t ktiles = a.K / a.nsplit / QP_KT;   // tiles of this workgroup's range of the hidden dimension
    const int64_t kbase = (int64_t)split * ktiles * QP_ROWB;   // byte offset of the range inside a row

    // ---- LDS-DMA requests: request i of a tile moves chunks e = i * 512 + t; a wave's 64 chunks = 4 rows x 16 slots
    const char* gsrc[QP_REQ];
#pragma unroll
    for (int i = 0; i < QP_REQ; ++i) {
        const uint32_t e = i * QP_THREADS + threadIdx.x, row = e >> 4, slot = e & 15;
        const uint32_t chunk = slot ^ (row & 15);
        const char* rowp;
        if (row < QP_COLS) {
            const uint32_t wrow = h * 128 + (row < 32 ? d0 + row : d0 + 64 + (row - 32));
            rowp = a.w + (int64_t)wrow * a.K * 2;
        } else {
            rowp = a.x + (int64_t)b * a.x_sb + (int64_t)(row - QP_COLS) * a.x_sw;
        }
        gsrc[i] = rowp + kbase + chunk * 16;
    }
    auto request_tile = [&](uint32_t t, uint32_t buf) {
        const uint32_t tt = min(t, ktiles - 1);   // past the end: re-fetch the last tile (never read)
#pragma unroll
        for (int i = 0; i < QP_REQ; ++i) {
            const char* g = gsrc[i] + (int64_t)tt * QP_ROWB;
            const uint32_t la = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds + buf * QP_TILEB + (i * QP_THREADS + wv * 64) * 16));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(g) : "memory");
        }
    };
#pragma unroll
    for (int p = 0; p < QP_NBUF - 1; ++p) request_tile(p, p);
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((QP_NBUF - 2) * QP_REQ));  // tile 0 landed (the newer tile may still be in flight)
    __syncthreads();

    f32x4 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][m] = {0.f, 0.f, 0.f, 0.f};
    uint32_t bc = 0;
    for (uint32_t t = 0; t < ktiles; ++t) {
        const unsigned char* buf = lds + bc * QP_TILEB;
        request_tile(t + QP_NBUF - 1, bc == 0 ? QP_NBUF - 1 : bc - 1);  // into the buffer tile t-1 just left
        const uint32_t sl = ((kstep * 4 + kq) ^ l16) << 4;   // this wave's k-step: chunk kstep * 4 + kq, swizzled by the row's low bits
        uint4 bfrag[2], afrag[4];
#pragma unroll
        for (int c = 0; c < 2; ++c)   // weight rows (output columns) 16 (2 chalf + c) + l16
            bfrag[c] = *reinterpret_cast<const uint4*>(buf + ((2 * chalf + c) * 16 + l16) * QP_ROWB + sl);
#pragma unroll
        for (int m = 0; m < 4; ++m)   // hidden rows 16 m + l16
            afrag[m] = *reinterpret_cast<const uint4*>(buf + (QP_COLS + 16 * m + l16) * QP_ROWB + sl);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[c][m] = mma16<DT>(afrag[m], bfrag[c], acc[c][m]);  // C[row 16 m + 4 kq + r][col 16 (2 chalf + c) + l16]
        __builtin_amdgcn_s_waitcnt(0x0070 | ((QP_NBUF - 2) * QP_REQ));  // lgkmcnt(0) + tile t+1 landed
        __syncthreads();
        bc = bc + 1 == QP_NBUF ? 0 : bc + 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the DMA before the buffers are reused for the partial sums
    __syncthreads();

    float (*kp)[QP_ROWS][QP_COLS + 1] = reinterpret_cast<float (*)[QP_ROWS][QP_COLS + 1]>(lds);  // [k-step][row][col] 66.5 KiB
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) kp[kstep][m * 16 + kq * 4 + r][(2 * chalf + c) * 16 + l16] = acc[c][m][r];
    __syncthreads();
    // the four k-step partials in a fixed order -> this workgroup's partial product (coalesced: 64 columns of a row per 64 threads)
    float* pout = a.part + ((((size_t)b * a.nsplit + split) * (a.Hq * 2) + cb) * QP_ROWS) * QP_COLS;
    for (uint32_t e = threadIdx.x; e < QP_ROWS * QP_COLS; e += QP_THREADS) {
        const uint32_t row = e >> 6, col = e & 63;
        pout[e] = ((kp[0][row][col] + kp[1][row][col]) + kp[2][row][col]) + kp[3][row][col];
    }
}

// One thread per RoPE pair (b, head, window row, dim d < 64 and its partner d + 64).
template <int DT>
__global__ __launch_bounds__(256) void qproj_reduce_rope_kernel(QprojArgs a, uint32_t B) {
    using T = typename Elem<DT>::T;
    const uint32_t total = B * a.Hq * QP_ROWS * 64;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t d = i & 63, row = (i >> 6) & 63, h = (i >> 12) % a.Hq, b = (i >> 12) / a.Hq;
        const uint32_t cb = h * 2 + (d >> 5), c = d & 31;
        float a0 = 0.f, a1 = 0.f;
        for (uint32_t s = 0; s < a.nsplit; ++s) {   // fixed order over the splits; the first add to 0.f is exact
            const float* pp = a.part + ((((size_t)b * a.nsplit + s) * (a.Hq * 2) + cb) * QP_ROWS + row) * QP_COLS;
            a0 += pp[c];
            a1 += pp[32 + c];
        }
        const float q0 = round_dt<DT>(a0), q1 = round_dt<DT>(a1);   // the GEMM's output rounding
        const T* cr = static_cast<const T*>(a.cosp) + (int64_t)b * a.cs_sb + (int64_t)row * a.cs_sw;
        const T* sr = static_cast<const T*>(a.sinp) + (int64_t)b * a.cs_sb + (int64_t)row * a.cs_sw;
        T* orow = static_cast<T*>(a.out) + (((size_t)b * a.Hq + h) * QP_ROWS + row) * 128;
        st_dt<DT>(orow + d, rope_elem<DT>(q0, Elem<DT>::ld(cr + d), -q1, Elem<DT>::ld(sr + d)));
        st_dt<DT>(orow + d + 64, rope_elem<DT>(q1, Elem<DT>::ld(cr + d + 64), q0, Elem<DT>::ld(sr + d + 64)));
    }
}

uint32_t qproj_nsplit(int64_t K) {   // the largest split whose ranges are whole 128-element tiles
    for (uint32_t n = QP_MAXSPLIT; n > 1; n >>= 1)
        if (K % ((int64_t)n * QP_KT) == 0) return n;
    return 1;
}

}  // namespace

size_t kvp_qproj_rope_ws_bytes(int64_t B, int64_t Hq) {   // partial products of the largest split
    return kvp_align_up((size_t)std::max<int64_t>(1, B) * QP_MAXSPLIT * (size_t)std::max<int64_t>(1, Hq) * 2 * QP_ROWS * QP_COLS * 4, 256);
}

bool kvp_qproj_rope_eligible(int dtype, int64_t W, int64_t D, int64_t K, const void* x, int64_t x_sb, int64_t x_sw, const void* w,
                             const void* cosp, const void* sinp, int64_t cs_sb, int64_t cs_sw) {
    if (dtype != KVP_BF16 && dtype != KVP_F16) return false;
    if (W != QP_ROWS || D != 128 || K < 256 || K % 256 != 0) return false;
    auto al8 = [](int64_t v) { return v % 8 == 0; };
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)cosp % 2) || ((uintptr_t)sinp % 2)) return false;
    return al8(x_sb) && al8(x_sw) && cs_sb >= 0 && cs_sw >= 0;
}

// out: [B, Hq, 64, 128] contiguous in the input dtype.  Strides in elements.  part: kvp_qproj_rope_ws_bytes(B, Hq) bytes of scratch.
int kvp_qproj_rope_launch(const void* x, int64_t x_sb, int64_t x_sw, const void* w, const void* cosp, const void* sinp, int64_t cs_sb,
                          int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t K, void* out, void* part, hipStream_t stream) {
    QprojArgs a;
    a.x = static_cast<const char*>(x); a.x_sb = x_sb * 2; a.x_sw = x_sw * 2;
    a.w = static_cast<const char*>(w);
    a.cosp = cosp; a.sinp = sinp; a.cs_sb = cs_sb; a.cs_sw = cs_sw;
    a.out = out; a.part = static_cast<float*>(part);
    a.Hq = (uint32_t)Hq; a.K = (uint32_t)K; a.nsplit = qproj_nsplit(K);
    const size_t lds = (size_t)QP_NBUF * QP_TILEB;   // 128 KiB (>= the 66.5 KiB of the k-step partials that reuse it)
    static_assert(QP_NBUF * QP_TILEB >= 4 * QP_ROWS * (QP_COLS + 1) * 4, "the k-step partials reuse the ring");
    static bool raised[2] = {false, false};
    const int di = dtype == KVP_BF16 ? 0 : 1;
    if (!raised[di]) {
        const void* fn = dtype == KVP_BF16 ? reinterpret_cast<const void*>(qproj_splitk_kernel<KVP_BF16>) : reinterpret_cast<const void*>(qproj_splitk_kernel<KVP_F16>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            kvp_set_error("qproj_rope: cannot raise the dynamic LDS limit to %zu bytes", lds);
            return KVP_EHIP;
        }
        raised[di] = true;
    }
    const dim3 grid((uint32_t)(Hq * 2 * a.nsplit), (uint32_t)B);
    const uint32_t pairs = (uint32_t)(B * Hq * QP_ROWS * 64);
    const uint32_t rblocks = std::min<uint32_t>((pairs + 255) / 256, 2048);
    if (dtype == KVP_BF16) {
        KVP_LAUNCH("qproj_rope_splitk_kernel", stream, qproj_splitk_kernel<KVP_BF16><<<grid, QP_THREADS, lds, stream>>>(a));
        KVP_LAUNCH("qproj_rope_reduce_kernel", stream, qproj_reduce_rope_kernel<KVP_BF16><<<rblocks, 256, 0, stream>>>(a, (uint32_t)B));
    } else {
        KVP_LAUNCH("qproj_rope_splitk_kernel", stream, qproj_splitk_kernel<KVP_F16><<<grid, QP_THREADS, lds, stream>>>(a));
        KVP_LAUNCH("qproj_rope_reduce_kernel", stream, qproj_reduce_rope_kernel<KVP_F16><<<rblocks, 256, 0, stream>>>(a, (uint32_t)B));
    }
    KVP_CHECK_LAUNCH("qproj_rope");
    return KVP_OK;
}

extern "C" size_t kvp_snapkv_qproj_rope_workspace_bytes(int64_t B, int64_t Hq) { return kvp_qproj_rope_ws_bytes(B, Hq); }

extern "C" int kvp_snapkv_qproj_rope(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, const void* cosp,
                                     const void* sinp, int64_t cs_sb, int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t W,
                                     int64_t D, int64_t hidden, void* q_rot, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && B <= 65535 && Hq * 8 < ((int64_t)1 << 28), "qproj_rope: bad shape B=%ld Hq=%ld", (long)B, (long)Hq);
    KVP_CHECK_ARG(hidden_win && wq && cosp && sinp && q_rot, "qproj_rope: null pointer");
    if (!kvp_qproj_rope_eligible(dtype, W, D, hidden, hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw)) {
        kvp_set_error("qproj_rope: needs bf16/f16, W = 64, D = 128, hidden %% 256 == 0 and 16-byte aligned rows");
        return KVP_EUNSUPPORTED;
    }
    if (!ws || ws_bytes < kvp_qproj_rope_ws_bytes(B, Hq)) {
        kvp_set_error("qproj_rope: workspace too small (%zu < %zu)", ws_bytes, kvp_qproj_rope_ws_bytes(B, Hq));
        return KVP_EWORKSPACE;
    }
    return kvp_qproj_rope_launch(hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw, dtype, B, Hq, hidden, q_rot, ws,
                                 static_cast<hipStream_t>(stream_));
}
