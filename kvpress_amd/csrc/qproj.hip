// kvp_snapkv_qproj_rope: RoPE'd window queries straight from the hidden states of the last W = 64 tokens.
// Replaces `get_prerope_query_states(module, hidden_states[:, -W:])` (kvpress/utils.py:43-46: q_proj, view, transpose) followed
// by `q * cos + rotate_half(q) * sin` (snapkv_press.py:53-58) for a plain nn.Linear q_proj without bias.
//
// A 64 x hidden by hidden x (H_q * 128) product: 32 MiB of bf16 weights for Llama-3.1-8B.  One workgroup per 16 output
// columns -- the 8 dims d0..d0+7 of a head AND their rotate_half partners d0+64..d0+71, so the RoPE pairs meet in the same
// workgroup -- 256 workgroups for 32 heads.  K is walked in tiles of 256 elements: the 16 weight rows and the 64
// hidden-state rows of a tile (40 KiB) land in LDS by LDS-DMA (coalesced 512-byte row segments, ring of three buffers);
// each of the 8 waves multiplies one 32-element k-step of the tile (v_mfma_f32_16x16x32: 4 row blocks x 1 column block,
// fragments by ds_read_b128 from XOR-swizzled rows).  The eight k-step partials are summed in a fixed order through LDS
// (deterministic), rounded to the model dtype like a GEMM output, rotated with torch's per-op rounding (rope_elem) and
// written as [B, H_q, W, D].
//
// Measured (Llama-3.1-8B window, MI355X).  Round 1, in isolation: 20 us, on par with the library GEMM + RoPE launch it replaces
// (17.6 + 5 us).  Round 4, inside bench.py's loop (two boxes, alternating runs, profiles/r04_qproj_lab.txt): the step is 3-4 us
// SHORTER with this kernel (271.3 / 271.6 / 268.7 us against 274.5 / 275.9 / 273.0 / 272.0 with hipBLASLt 15.1 + RoPE 4.8), so the
// presses now project the window here by default (kvpress_amd/_native.py USE_LIBRARY_QPROJ).  A rotated tile walk (workgroup j of an
// XCD starts at K tile j % 16, so the 32 CUs of an XCD do not ask their L2 for the same hidden-window lines at the same moment)
// is worth another ~0.8 us of the kernel, 1.6 us of the step (three alternating A/B pairs: 271.9 against 273.5 us).
// What bounds it is the traffic between the L2s and the CUs, not HBM and not the matrix pipe: every workgroup pulls the whole hidden
// window (512 KiB) next to its 128 KiB weight slice, 160 MiB in total, and that path delivers ~10 TB/s chip-wide when all CUs read
// the same lines -- the library GEMM's 16 x 64 tiles move the same 160 MiB and take the same ~15 us.  Round 4 tried to hide it and
// could not (same file history, tools/lab_patches/qproj_v2_v3.diff): the weight slice in registers with ALL of it in flight from the first
// microsecond + the hidden window through a 4-deep LDS-DMA ring fed by 8 loader waves: 21.6 us (18.4 for this kernel, same run); the
// same with the window staged through registers (global_load_dwordx4 -> ds_write_b128, 12 loads per lane in flight): 26 us.  Deeper
// prefetch does not help a stream that is bandwidth-bound where it enters the CU.  Earlier variants: fragments straight from
// global memory (16-byte pieces of 16 rows per instruction: address-path bound, 27 us); 64-column tiles on 64 workgroups (64 MiB of
// traffic but 1 MiB per CU: 28 us); a fourth buffer (no change); split-K with a second reduction pass (18.4 against 15.4 us in
// isolation).  Round 5 built the split over the hidden dimension properly (64 columns x K/4 per workgroup: 64 MiB of traffic, float32
// partials + a reduce / RoPE kernel; tools/lab_patches/qproj_splitk.diff): same step time on the same box (0.2705 vs 0.2695 ms,
// profiles/r05_ab_qproj.txt) and +3 us on short caches -- the fixed cost of 256 short-lived streaming workgroups, not the L2 -> CU bytes, is what is left.
#include "kvp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QP_THREADS = 512;
constexpr int QP_WAVES = QP_THREADS / 64;  // one 32-element k-step of every K tile per wave
constexpr int QP_ROWS = 64;     // window
constexpr int QP_COLS = 16;     // 8 dims + their 8 rotate_half partners
constexpr int QP_KT = 256;      // K elements per tile = QP_WAVES k-steps of 32
constexpr int QP_ROWB = QP_KT * 2;                         // 512 B per tile row = 32 chunks of 16 B
constexpr int QP_TILEB = (QP_COLS + QP_ROWS) * QP_ROWB;    // 8 KiB of weights + 32 KiB of hidden states
constexpr int QP_REQ = QP_TILEB / 16 / QP_THREADS;         // LDS-DMA requests per thread and tile (5)
constexpr int QP_NBUF = 3;
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
    uint32_t Hq, K;
};

// Tile rows 0..15 = the 16 weight rows (columns of the output tile), rows 16..79 = the 64 hidden-state rows; 512 B per row,
// 16-byte slot p of row r holds chunk p ^ (r & 15) (XOR swizzle applied on the global side of the DMA): the 16 lanes of a
// fragment read (16 rows, same chunk) hit 16 distinct slots.
template <int DT>
__global__ __launch_bounds__(QP_THREADS) void qproj_rope_kernel(QprojArgs a) {
    using T = typename Elem<DT>::T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[QP_NBUF * QP_TILEB];
    const uint32_t nt = blockIdx.x, b = blockIdx.y;
    const uint32_t h = nt >> 3, d0 = (nt & 7) * 8;  // output columns: dims d0..d0+7 of head h, then d0+64..d0+71
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t l16 = lane & 15, kq = lane >> 4;  // fragment row / column, k-slot group (8 elements)
    const uint32_t ntiles = a.K / QP_KT;
    const uint32_t rot = (blockIdx.x >> 3) % ntiles;

    // ---- LDS-DMA requests: request i of a tile moves chunks e = i * 512 + t; a wave's 64 chunks = 2 rows x 32 slots
    const char* gsrc[QP_REQ];     // global address of this thread's chunk in K tile 0
#pragma unroll
    for (int i = 0; i < QP_REQ; ++i) {
        const uint32_t e = i * QP_THREADS + threadIdx.x, row = e >> 5, slot = e & 31;
        const uint32_t chunk = slot ^ (row & 15);
        const char* rowp;
        if (row < QP_COLS) {
            const uint32_t wrow = h * 128 + (row < 8 ? d0 + row : d0 + 64 + (row - 8));
            rowp = a.w + (int64_t)wrow * a.K * 2;
        } else {
            rowp = a.x + (int64_t)b * a.x_sb + (int64_t)(row - QP_COLS) * a.x_sw;
        }
        gsrc[i] = rowp + chunk * 16;
    }
    auto request_tile = [&](uint32_t t, uint32_t buf) {
        // Every workgroup reads the SAME hidden window; walking it in lockstep makes the 32 CUs of an XCD ask their L2 for the same
        // lines at the same moment.  A rotated walk (workgroup j of an XCD starts at tile j % ntiles; blocks b, b + 8, ... share an
        // XCD) spreads them over the window.  The sum over the tiles then runs in a rotated order per workgroup: still a fixed order.
        uint32_t tt = min(t, ntiles - 1);  // past the end: re-fetch the last tile (never read)
        tt += rot;
        tt -= tt >= ntiles ? ntiles : 0u;
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
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((QP_NBUF - 2) * QP_REQ));  // tile 0 landed (the newer tiles may still be in flight)
    __syncthreads();

    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = {0.f, 0.f, 0.f, 0.f};
    uint32_t bc = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        const unsigned char* buf = lds + bc * QP_TILEB;
        request_tile(t + QP_NBUF - 1, bc == 0 ? QP_NBUF - 1 : bc - 1);  // into the buffer tile t-1 just left
        const uint32_t sl = ((wv * 4 + kq) ^ l16) << 4;  // this wave's k-step: chunk wv * 4 + kq, swizzled by the row's low bits
        const uint4 bfrag = *reinterpret_cast<const uint4*>(buf + l16 * QP_ROWB + sl);
        uint4 afrag[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)   // row 16 + 16 m + l16: (row & 15) == l16
            afrag[m] = *reinterpret_cast<const uint4*>(buf + (QP_COLS + 16 * m + l16) * QP_ROWB + sl);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = mma16<DT>(afrag[m], bfrag, acc[m]);  // C[row 16 m + 4 kq + r][col l16]
        __builtin_amdgcn_s_waitcnt(0x0070 | ((QP_NBUF - 2) * QP_REQ));  // lgkmcnt(0) + tile t+1 landed
        __syncthreads();
        bc = bc + 1 == QP_NBUF ? 0 : bc + 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the DMA before the buffers are reused for the partial sums

    float (*part)[QP_ROWS][QP_COLS + 1] = reinterpret_cast<float (*)[QP_ROWS][QP_COLS + 1]>(lds);  // [k-step][row][col]
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wv][m * 16 + kq * 4 + r][l16] = acc[m][r];
    __syncthreads();
    if (threadIdx.x >= 256) return;

    // epilogue: thread t -> row t / 4, pairs p = (t % 4) * 2 + {0, 1}
    const uint32_t row = threadIdx.x >> 2;
    const T* cr = static_cast<const T*>(a.cosp) + (int64_t)b * a.cs_sb + (int64_t)row * a.cs_sw;
    const T* sr = static_cast<const T*>(a.sinp) + (int64_t)b * a.cs_sb + (int64_t)row * a.cs_sw;
    T* orow = static_cast<T*>(a.out) + (((size_t)b * a.Hq + h) * QP_ROWS + row) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t p = (threadIdx.x & 3) * 2 + i;
        // fixed summation order over the eight k-step slices, then the GEMM's output rounding
        float a0 = part[0][row][p], a1 = part[0][row][p + 8];
#pragma unroll
        for (int w = 1; w < QP_WAVES; ++w) {
            a0 += part[w][row][p];
            a1 += part[w][row][p + 8];
        }
        const float q0 = round_dt<DT>(a0), q1 = round_dt<DT>(a1);
        const uint32_t d = d0 + p;
        st_dt<DT>(orow + d, rope_elem<DT>(q0, Elem<DT>::ld(cr + d), -q1, Elem<DT>::ld(sr + d)));
        st_dt<DT>(orow + d + 64, rope_elem<DT>(q1, Elem<DT>::ld(cr + d + 64), q0, Elem<DT>::ld(sr + d + 64)));
    }
}


}  // namespace

bool kvp_qproj_rope_eligible(int dtype, int64_t W, int64_t D, int64_t K, const void* x, int64_t x_sb, int64_t x_sw, const void* w,
                             const void* cosp, const void* sinp, int64_t cs_sb, int64_t cs_sw) {
    if (dtype != KVP_BF16 && dtype != KVP_F16) return false;
    if (W != QP_ROWS || D != 128 || K < QP_KT || K % QP_KT != 0) return false;
    auto al8 = [](int64_t v) { return v % 8 == 0; };
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)cosp % 2) || ((uintptr_t)sinp % 2)) return false;
    return al8(x_sb) && al8(x_sw) && cs_sb >= 0 && cs_sw >= 0;
}

// out: [B, Hq, 64, 128] contiguous in the input dtype.  Strides in elements.
int kvp_qproj_rope_launch(const void* x, int64_t x_sb, int64_t x_sw, const void* w, const void* cosp, const void* sinp, int64_t cs_sb,
                          int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t K, void* out, hipStream_t stream) {
    QprojArgs a;
    a.x = static_cast<const char*>(x); a.x_sb = x_sb * 2; a.x_sw = x_sw * 2;
    a.w = static_cast<const char*>(w);
    a.cosp = cosp; a.sinp = sinp; a.cs_sb = cs_sb; a.cs_sw = cs_sw;
    a.out = out; a.Hq = (uint32_t)Hq; a.K = (uint32_t)K;
    const dim3 grid((uint32_t)(Hq * 8), (uint32_t)B);
    if (dtype == KVP_BF16) KVP_LAUNCH("qproj_rope_kernel", stream, qproj_rope_kernel<KVP_BF16><<<grid, QP_THREADS, 0, stream>>>(a));
    else KVP_LAUNCH("qproj_rope_kernel", stream, qproj_rope_kernel<KVP_F16><<<grid, QP_THREADS, 0, stream>>>(a));
    KVP_CHECK_LAUNCH("qproj_rope");
    return KVP_OK;
}

extern "C" int kvp_snapkv_qproj_rope(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, const void* cosp,
                                     const void* sinp, int64_t cs_sb, int64_t cs_sw, int dtype, int64_t B, int64_t Hq, int64_t W,
                                     int64_t D, int64_t hidden, void* q_rot, kvp_stream_t stream_) {
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && B <= 65535 && Hq * 8 < ((int64_t)1 << 31), "qproj_rope: bad shape B=%ld Hq=%ld", (long)B, (long)Hq);
    KVP_CHECK_ARG(hidden_win && wq && cosp && sinp && q_rot, "qproj_rope: null pointer");
    if (!kvp_qproj_rope_eligible(dtype, W, D, hidden, hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw)) {
        kvp_set_error("qproj_rope: needs bf16/f16, W = 64, D = 128, hidden %% 256 == 0 and 16-byte aligned rows");
        return KVP_EUNSUPPORTED;
    }
    return kvp_qproj_rope_launch(hidden_win, x_sb, x_sw, wq, cosp, sinp, cs_sb, cs_sw, dtype, B, Hq, hidden, q_rot,
                                 static_cast<hipStream_t>(stream_));
}
