// KVP_ORDER_SCORE: the retained indices in DESCENDING SCORE order (what `scores.topk(n_kept).indices` returns with
// sorted=True, kvpress/presses/scorer_press.py:95), ties in ascending position.
//
// Not on the hot path (no press of this package asks for it: attention is permutation-invariant over the kept tokens and the
// position order makes the gather a monotone stream), so it is built from parts: kvp_topk_select's position-ordered
// result, one kernel that fetches each kept score as a 64-bit key (row << 32 | descending-order score key), and ONE device-wide
// rocPRIM radix sort over all R * k pairs (stable: equal scores stay in ascending position; the row bits keep the rows apart).
// (Round 3 used rocPRIM's SEGMENTED sort: with 8 segments of 65536 it ran 1.3 ms -- one workgroup per segment; the device-wide
// sort of the same 524288 pairs takes tens of microseconds.)  rocPRIM ships with ROCm as HIP headers.
#include "kvp_common.h"
#include "topk_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct OrderWs {
    uint64_t* keys_in;
    uint64_t* keys_out;
    int32_t* idx_in;
    void* tmp;
    size_t tmp_bytes, total_bytes;
};

unsigned row_bits(int64_t R) {
    unsigned b = 0;
    while (((int64_t)1 << b) < R) ++b;
    return b;
}

size_t sort_tmp_bytes(int64_t R, int64_t k) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                    (size_t)(R * k), 0u, 32u + row_bits(R), (hipStream_t)0);
    return bytes;
}

OrderWs carve(void* ws, int64_t R, int64_t k) {
    OrderWs w;
    size_t off = 0;
    char* base = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += kvp_align_up(bytes, 256);
        return p;
    };
    const size_t n = (size_t)std::max<int64_t>(1, R * k);
    w.keys_in = (uint64_t*)take(n * 8);
    w.keys_out = (uint64_t*)take(n * 8);
    w.idx_in = (int32_t*)take(n * 4);
    w.tmp_bytes = sort_tmp_bytes(R, k);
    w.tmp = take(w.tmp_bytes);
    w.total_bytes = off;
    return w;
}

// keys_in[r * k + j] = r << 32 | ~(order-preserving key of scores[r, idx[r, j]])  (ascending sort = rows in order, descending scores).
// A poisoned index (-1: a select that reported a failure) sorts first in its row and stays -1.
__global__ __launch_bounds__(256) void order_keys_kernel(const float* __restrict__ scores, int64_t row_stride, const int32_t* __restrict__ idx,
                                                         uint32_t k, uint32_t S, uint32_t kmask, uint64_t* __restrict__ keys,
                                                         int32_t* __restrict__ idx_copy) {
    const uint32_t r = blockIdx.y;
    const float* row = scores + (int64_t)r * row_stride;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
        const int32_t p = idx[(size_t)r * k + j];
        const uint32_t sk = (uint32_t)p < S ? ~(float_to_key(row[p]) ^ kmask) : 0u;
        keys[(size_t)r * k + j] = ((uint64_t)r << 32) | sk;
        idx_copy[(size_t)r * k + j] = p;
    }
}

}  // namespace

size_t topk_order_workspace_bytes(int64_t R, int64_t k) { return (R <= 0 || k <= 0) ? 0 : carve(nullptr, R, k).total_bytes; }

// idx [R, k] (contiguous, ascending positions from the select) is rewritten in descending-score order
int topk_order_by_score(const float* scores, int64_t R, int64_t S, int64_t row_stride, int64_t k, int32_t* idx, bool smallest, void* ws,
                        size_t ws_bytes, hipStream_t stream) {
    if (R == 0 || k == 0) return KVP_OK;
    KVP_CHECK_ARG(R * k < ((int64_t)1 << 31), "topk(order): too many indices");
    OrderWs w = carve(ws, R, k);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("topk(order): workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((k + 255) / 256, 256));
    KVP_LAUNCH("order_keys_kernel", stream, order_keys_kernel<<<dim3(bx, (uint32_t)R), 256, 0, stream>>>(
        scores, row_stride, idx, (uint32_t)k, (uint32_t)S, smallest ? 0xFFFFFFFFu : 0u, w.keys_in, w.idx_in));
    KVP_CHECK_LAUNCH("topk(order keys)");
    size_t tmp_bytes = w.tmp_bytes;
    hipError_t e = hipSuccess;
    KVP_LAUNCH("rocprim_radix_sort_pairs", stream, (e = rocprim::radix_sort_pairs(w.tmp, tmp_bytes, (const uint64_t*)w.keys_in, w.keys_out, (const int32_t*)w.idx_in,
                                                                                idx, (size_t)(R * k), 0u, 32u + row_bits(R), stream)));
    if (e != hipSuccess) {
        kvp_set_error("topk(order): sort failed: %s", hipGetErrorString(e));
        return KVP_EHIP;
    }
    return KVP_OK;
}
