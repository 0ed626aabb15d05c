// kvp_cur_score: CURPress.score (kvpress/presses/cur_press.py:32-66), approximate leverage scores of keys and values.
//   k2 = sum_d k^2, v2 = sum_d v^2                                   (:40-41)   one launch over both tensors (rownorm.hip, squared)
//   local approximation: each is divided by its sum over windows of `w` consecutive tokens (zero-padded tail)   (:43-48)
//   combined by leverage type: key | value | (k2 + v2) / 2 | k2 * v2                                              (:50-59)
//   normalised by the row sum, first `num_sinks` positions set to 1                                               (:61-62)
// The window / combine / normalise step works on the two [B*H, S] float vectors (L2-resident) in two launches over a
// (blocks, rows) grid: combine + per-block partial sums, then the division by the row total (partials added in block order:
// deterministic), fp32 throughout.  (One workgroup per row, as first written, took 400 us of the 490 us at 8 x 131072.)
#include "kvp_common.h"

int kvp_rowsumsq2_launch(const void* k, const void* v, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                         int64_t v_sb, int64_t v_sh, int64_t v_ss, float* out_k, float* out_v, hipStream_t stream);

namespace {

constexpr int CU_THREADS = 256;
constexpr int CU_MAXBLK = 256;  // workgroups per row in the combine pass (their partial sums are added in a fixed order)
static_assert(CU_MAXBLK <= CU_THREADS, "cur_normalize_kernel reads one partial per thread");

__device__ __forceinline__ float combine(int type, float a, float b) {
    switch (type) {
        case KVP_CUR_KEY: return a;
        case KVP_CUR_VALUE: return b;
        case KVP_CUR_KV_AVG: return (a + b) * 0.5f;
        default: return a * b;
    }
}

// window sum of x over the `w` tokens of position s's window (positions past S count as zero)
__device__ __forceinline__ float window_sum(const float* __restrict__ x, uint32_t s, uint32_t S, uint32_t w) {
    const uint32_t lo = s - s % w, hi = min(lo + w, S);
    float t = 0.f;
    for (uint32_t i = lo; i < hi; ++i) t += x[i];
    return t;
}

// pass 1 (grid = blocks per row x rows): window-normalise, combine, write the un-normalised score, one partial sum per block
__global__ __launch_bounds__(CU_THREADS) void cur_combine_kernel(const float* __restrict__ k2, const float* __restrict__ v2, uint32_t S, uint32_t w,
                                                                 int type, float* __restrict__ scores, float* __restrict__ partial) {
    __shared__ float red[CU_THREADS / 64];
    const float* kr = k2 + (size_t)blockIdx.y * S;
    const float* vr = v2 + (size_t)blockIdx.y * S;
    float* out = scores + (size_t)blockIdx.y * S;
    float acc = 0.f;
    for (uint32_t s = blockIdx.x * CU_THREADS + threadIdx.x; s < S; s += gridDim.x * CU_THREADS) {
        float a = kr[s], b = vr[s];
        if (w) {
            a = a / window_sum(kr, s, S, w);
            b = b / window_sum(vr, s, S, w);
        }
        const float c = combine(type, a, b);
        out[s] = c;
        acc += c;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The same for local windows whose length divides the workgroup's 256-token span: the span's energies go through LDS once, one thread
// per window adds them in window_sum's order, every token divides by its window's sum -- the same tokens per thread and the same
// arithmetic as the per-token kernel (bit for bit), without its 2 x w loads per token.
__global__ __launch_bounds__(CU_THREADS) void cur_combine_lds_kernel(const float* __restrict__ k2, const float* __restrict__ v2, uint32_t S, uint32_t w,
                                                                     int type, float* __restrict__ scores, float* __restrict__ partial) {
    __shared__ float red[CU_THREADS / 64];
    __shared__ float sk[CU_THREADS], sv[CU_THREADS], tk[CU_THREADS], tv[CU_THREADS];
    const float* kr = k2 + (size_t)blockIdx.y * S;
    const float* vr = v2 + (size_t)blockIdx.y * S;
    float* out = scores + (size_t)blockIdx.y * S;
    const uint32_t nwin = CU_THREADS / w, mywin = threadIdx.x / w;
    float acc = 0.f;
    for (uint32_t base = blockIdx.x * CU_THREADS; base < S; base += gridDim.x * CU_THREADS) {
        const uint32_t s = base + threadIdx.x;
        const float a = s < S ? kr[s] : 0.f, b = s < S ? vr[s] : 0.f;
        sk[threadIdx.x] = a;
        sv[threadIdx.x] = b;
        __syncthreads();
        if (threadIdx.x < nwin) {
            const uint32_t lo = threadIdx.x * w, hi = min(lo + w, S - base);
            float ta = 0.f, tb = 0.f;
            for (uint32_t i = lo; i < hi; ++i) {
                ta += sk[i];
                tb += sv[i];
            }
            tk[threadIdx.x] = ta;
            tv[threadIdx.x] = tb;
        }
        __syncthreads();
        if (s < S) {
            const float c = combine(type, a / tk[mywin], b / tv[mywin]);
            out[s] = c;
            acc += c;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// (Round 3: one THREAD per window straight from global memory -- its two sums once, then its w scores -- was measured at 82 us against
// the per-token kernel's 12: sixteen dependent 4-byte loads per lane, 64 bytes apart across the wave, is the worst shape for the vector
// memory path.  Hence the LDS staging above.)
// pass 2: divide by the row total (the partials added in block order), sinks = 1
__global__ __launch_bounds__(CU_THREADS) void cur_normalize_kernel(float* __restrict__ scores, const float* __restrict__ partial, uint32_t nblk, uint32_t S,
                                                                   uint32_t num_sinks) {
    // row total: the nblk <= CU_MAXBLK = CU_THREADS partials, one per thread, added in a fixed (tree) order -- every workgroup of
    // the row computes the same value (a serial loop over 256 dependent loads per thread was 25 of this kernel's 27 us)
    __shared__ float red[CU_THREADS / 64];
    float t = threadIdx.x < nblk ? partial[(size_t)blockIdx.y * nblk + threadIdx.x] : 0.f;
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    float* out = scores + (size_t)blockIdx.y * S;
    for (uint32_t s = blockIdx.x * CU_THREADS + threadIdx.x; s < S; s += gridDim.x * CU_THREADS) out[s] = s < num_sinks ? 1.0f : out[s] / tot;
}

}  // namespace

extern "C" size_t kvp_cur_workspace_bytes(int64_t B, int64_t H, int64_t S) {
    if (B < 1 || H < 1 || S < 1) return 256;
    return 2 * kvp_align_up((size_t)B * H * S * 4, 256) + kvp_align_up((size_t)B * H * CU_MAXBLK * 4, 256);
}

extern "C" int kvp_cur_score(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh,
                             int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int leverage_type,
                             int64_t local_window_size, int64_t num_sinks, float* scores, void* ws, size_t ws_bytes, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KVP_CHECK_ARG(B >= 1 && H >= 1 && S >= 1 && D >= 1 && B * H <= 65535, "cur: bad shape B=%ld H=%ld S=%ld D=%ld", (long)B, (long)H, (long)S,
                  (long)D);
    KVP_CHECK_ARG(leverage_type >= KVP_CUR_KEY && leverage_type <= KVP_CUR_KV_PRODUCT, "cur: unknown leverage type %d", leverage_type);
    KVP_CHECK_ARG(local_window_size >= 0 && num_sinks >= 0, "cur: bad window / sinks");
    KVP_CHECK_ARG(k && v && scores, "cur: null pointer");
    const size_t half = kvp_align_up((size_t)B * H * S * 4, 256);
    const size_t need = kvp_cur_workspace_bytes(B, H, S);
    if (!ws || ws_bytes < need) {
        kvp_set_error("cur: workspace too small (%zu < %zu)", ws_bytes, need);
        return KVP_EWORKSPACE;
    }
    float* k2 = static_cast<float*>(ws);
    float* v2 = reinterpret_cast<float*>(static_cast<char*>(ws) + half);
    if (int rc = kvp_rowsumsq2_launch(k, v, dtype, B, H, S, D, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, k2, v2, stream)) return rc;
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + 2 * half);
    const uint32_t R = (uint32_t)(B * H);
    const uint32_t nblk = (uint32_t)std::max<int64_t>(1, std::min<int64_t>({(S + CU_THREADS - 1) / CU_THREADS, (int64_t)CU_MAXBLK, std::max<int64_t>(1, 2048 / R)}));
    if (local_window_size >= 2 && CU_THREADS % local_window_size == 0)   // windows never straddle a workgroup's 256-token span
        KVP_LAUNCH("cur_combine_kernel", stream, cur_combine_lds_kernel<<<dim3(nblk, R), CU_THREADS, 0, stream>>>(k2, v2, (uint32_t)S, (uint32_t)local_window_size,
                                                                                                                 leverage_type, scores, partial));
    else
        KVP_LAUNCH("cur_combine_kernel", stream, cur_combine_kernel<<<dim3(nblk, R), CU_THREADS, 0, stream>>>(k2, v2, (uint32_t)S, (uint32_t)local_window_size,
                                                                                                             leverage_type, scores, partial));
    KVP_LAUNCH("cur_normalize_kernel", stream, cur_normalize_kernel<<<dim3(nblk, R), CU_THREADS, 0, stream>>>(scores, partial, nblk, (uint32_t)S,
                                                                                                             (uint32_t)std::min<int64_t>(num_sinks, S)));
    KVP_CHECK_LAUNCH("cur");
    return KVP_OK;
}
