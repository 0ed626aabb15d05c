// KVP_ORDER_SCORE: the retained indices in DESCENDING SCORE order (what `scores.topk(n_kept).indices` returns with
// sorted=True, kvpress/presses/scorer_press.py:95), ties in ascending position.
//
// Not on the hot path (no press of this package asks for it: attention is permutation-invariant over the kept tokens and the
// position order makes the gather a monotone stream), so it is built from parts: kvp_topk_select's position-ordered
// result, one kernel that fetches each kept score as a descending-order key, and rocPRIM's segmented radix sort (stable:
// equal scores stay in ascending position).  rocPRIM ships with ROCm as HIP headers.
#include "kvp_common.h"
#include "topk_internal.h"

#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace {

struct OrderWs {
    uint32_t* keys_in;
    uint32_t* keys_out;
    int32_t* idx_in;
    uint32_t* offsets;  // [R + 1]
    void* tmp;
    size_t tmp_bytes, total_bytes;
};

size_t sort_tmp_bytes(int64_t R, int64_t k) {
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                              (int32_t*)nullptr, (unsigned)(R * k), (unsigned)R, (const uint32_t*)nullptr,
                                              (const uint32_t*)nullptr, 0, 32, (hipStream_t)0);
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
    w.keys_in = (uint32_t*)take(n * 4);
    w.keys_out = (uint32_t*)take(n * 4);
    w.idx_in = (int32_t*)take(n * 4);
    w.offsets = (uint32_t*)take((size_t)(R + 1) * 4);
    w.tmp_bytes = sort_tmp_bytes(R, k);
    w.tmp = take(w.tmp_bytes);
    w.total_bytes = off;
    return w;
}

// keys_in[r * k + j] = ~(order-preserving key of scores[r, idx[r, j]])  (ascending sort of these = descending scores)
__global__ __launch_bounds__(256) void order_keys_kernel(const float* __restrict__ scores, int64_t row_stride, const int32_t* __restrict__ idx,
                                                         uint32_t k, uint32_t R, uint32_t kmask, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ idx_copy, uint32_t* __restrict__ offsets) {
    const uint32_t r = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        offsets[r] = r * k;
        if (r == R - 1) offsets[R] = R * k;
    }
    const float* row = scores + (int64_t)r * row_stride;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
        const int32_t p = idx[(size_t)r * k + j];
        keys[(size_t)r * k + j] = ~(float_to_key(row[p]) ^ kmask);
        idx_copy[(size_t)r * k + j] = p;
    }
}

}  // namespace

size_t topk_order_workspace_bytes(int64_t R, int64_t k) { return (R <= 0 || k <= 0) ? 0 : carve(nullptr, R, k).total_bytes; }

// idx [R, k] (contiguous, ascending positions from the select) is rewritten in descending-score order
int topk_order_by_score(const float* scores, int64_t R, int64_t row_stride, int64_t k, int32_t* idx, bool smallest, void* ws, size_t ws_bytes,
                        hipStream_t stream) {
    if (R == 0 || k == 0) return KVP_OK;
    KVP_CHECK_ARG(R * k < ((int64_t)1 << 31), "topk(order): too many indices");
    OrderWs w = carve(ws, R, k);
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("topk(order): workspace too small (%zu < %zu)", ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    const uint32_t bx = (uint32_t)std::max<int64_t>(1, std::min<int64_t>((k + 255) / 256, 256));
    KVP_LAUNCH("order_keys_kernel", stream, order_keys_kernel<<<dim3(bx, (uint32_t)R), 256, 0, stream>>>(
        scores, row_stride, idx, (uint32_t)k, (uint32_t)R, smallest ? 0xFFFFFFFFu : 0u, w.keys_in, w.idx_in, w.offsets));
    KVP_CHECK_LAUNCH("topk(order keys)");
    size_t tmp_bytes = w.tmp_bytes;
    const hipError_t e = rocprim::segmented_radix_sort_pairs(w.tmp, tmp_bytes, (const uint32_t*)w.keys_in, w.keys_out, (const int32_t*)w.idx_in,
                                                             idx, (unsigned)(R * k), (unsigned)R, (const uint32_t*)w.offsets,
                                                             (const uint32_t*)w.offsets + 1, 0, 32, stream);
    if (e != hipSuccess) {
        kvp_set_error("topk(order): segmented sort failed: %s", hipGetErrorString(e));
        return KVP_EHIP;
    }
    return KVP_OK;
}
