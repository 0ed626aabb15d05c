// Fused ScorerPress.compress entry points: score -> top-k -> gather in ONE library call
// (kvpress/presses/scorer_press.py:76-102 with the scorer inlined).
//
// Same kernels as the modular entry points (kvp_*_score, kvp_topk_select, kvp_gather_kv) with two launches removed:
//   * the kernel that WRITES the scores also accumulates the radix select's first 12-bit histogram
//     (topk_internal.h), so the select starts at its second pass;
//   * SnapKV: the W window columns need no score at all -- they are kept by construction (the reference pads them with
//     max + 1, snapkv_press.py:103) -- so there is no global-max reduction and no pad fill: the select runs over the
//     first S - W columns for n_kept - W entries and the window positions are appended (they are the largest
//     positions, so the ascending-position order is preserved).  n_kept < W (fewer survivors than window tokens) takes
//     the unfused sequence.
//   * the select by row length: <= 16384 scores ONE launch with one workgroup per row and its own digits (topk_row_kernel; SnapKV up
//     to 4096 columns also pools + scales inside that launch's loader, topk_select_pooled_rows); 16385 .. 262144 scores of <= 8 rows
//     ONE launch of the cluster select (topk_cluster.hip) whose loader pools SnapKV's column sums (no pooling launch, no score
//     round trip) or computes Knorm's norms (the scores never reach memory); everything else -- more rows, longer rows, other
//     devices -- the (chunk, row) passes starting from the first-digit histogram that the score-writing kernel accumulated.
// Scores and indices live in the caller's workspace and never leave the device.
#include "kvp_common.h"
#include "snapkv_internal.h"
#include "topk_internal.h"

int kvp_rownorm_launch(const void* x, int dtype, int64_t B, int64_t H, int64_t S, int64_t D, int64_t sb, int64_t sh,
                       int64_t ss, float scale, float* out, hipStream_t stream, uint32_t* hist1, bool* hist1_done);

namespace {

struct CompressWs {
    void* topk;       // first: its leading zero_bytes are the part that must be clean
    size_t topk_bytes;
    float* scores;    // [R][S]
    int32_t* idx;     // [R][n_kept]
    void* scorer;     // scorer-specific scratch
    size_t scorer_bytes;
    void* order;      // KVP_ORDER_SCORE: the sort's tiles and samples (topk_order.hip)
    size_t order_bytes;
    size_t total_bytes;
};

CompressWs carve(void* ws, int64_t R, int64_t S_select, int64_t S, int64_t n_kept, size_t scorer_bytes) {
    CompressWs w;
    size_t off = 0;
    char* base = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += kvp_align_up(bytes, 256);
        return p;
    };
    // the select may run over S (unfused fallback) or S_select <= S columns: size for the larger chunk count
    w.topk_bytes = std::max(kvp_topk_workspace_bytes(R, S, n_kept), kvp_topk_workspace_bytes(R, S_select, n_kept));
    w.topk = take(w.topk_bytes);
    w.scores = (float*)take((size_t)R * S * 4);
    w.idx = (int32_t*)take((size_t)std::max<int64_t>(1, R * n_kept) * 4);
    w.scorer_bytes = scorer_bytes;
    w.scorer = take(scorer_bytes);
    w.order_bytes = topk_order_workspace_bytes(R, n_kept);
    w.order = take(w.order_bytes);
    w.total_bytes = off;
    return w;
}

int check_ws(const CompressWs& w, const void* ws, size_t ws_bytes, const char* who) {
    if (!ws || ws_bytes < w.total_bytes) {
        kvp_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total_bytes);
        return KVP_EWORKSPACE;
    }
    return KVP_OK;
}

}  // namespace

extern "C" size_t kvp_knorm_compress_workspace_bytes(int64_t B, int64_t H, int64_t S, int64_t n_kept) {
    if (B < 1 || H < 1 || S < 1 || n_kept < 0) return 256;
    return carve(nullptr, B * H, S, S, n_kept, 0).total_bytes;
}

extern "C" int kvp_knorm_compress(const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb,
                                  int64_t v_sh, int64_t v_ss, int dtype, int64_t B, int64_t H, int64_t S, int64_t D,
                                  int64_t n_kept, void* k_out, void* v_out, void* ws, size_t ws_bytes, int flags,
                                  kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = kvp_async_check("kvp_knorm_compress")) return rc;
    KVP_CHECK_ARG(B >= 1 && H >= 1 && S >= 1 && D >= 1 && n_kept >= 0 && n_kept <= S, "knorm_compress: bad shape B=%ld H=%ld S=%ld D=%ld n=%ld",
                  (long)B, (long)H, (long)S, (long)D, (long)n_kept);
    if (n_kept == 0) return KVP_OK;
    const int64_t R = B * H;
    const CompressWs w = carve(ws, R, S, S, n_kept, 0);
    if (int rc = check_ws(w, ws, ws_bytes, "knorm_compress")) return rc;
    const bool clean = (flags & KVP_TOPK_WS_CLEAN) != 0;
    if (!clean && hipMemsetAsync(w.topk, 0, topk_carve_ws(nullptr, R, 1).zero_bytes, stream) != hipSuccess) {
        kvp_set_error("knorm_compress: hipMemsetAsync failed");
        return KVP_EHIP;
    }
    // long rows of 256-byte keys (Llama: D = 128, bf16 / f16): norms, select digits and compaction in ONE launch, the scores stay
    // in registers (topk_cluster.hip); other shapes / a device that cannot hold the grid / KVP_TK_CLUSTER=0: the sequence below
    const bool by_score = (flags & KVP_ORDER_SCORE) != 0;   // K' / V' rows in descending-score order (the reference's layout, scorer_press.py:95-100)
    if (!by_score && n_kept < S && topk_cluster_eligible(R, S) && (dtype == KVP_BF16 || dtype == KVP_F16) && D == 128 && ((uintptr_t)k % 16) == 0 &&
        (k_sb * 2) % 16 == 0 && (k_sh * 2) % 16 == 0 && (k_ss * 2) % 16 == 0) {
        TopkWs tw = topk_carve_ws(w.topk, R, (S + TK_CHUNK - 1) / TK_CHUNK);
        const int rc = topk_cluster_select(TOPK_CLUSTER_KNORM, nullptr, 0, 1.f, k, dtype, k_sb, k_sh, k_ss, H, -1.0f, R, S, n_kept, w.idx, n_kept, 0, 0, tw,
                                           false, stream);
        if (rc < 0) return rc;
        if (rc == 0) return kvp_gather_kv(k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, H, S, D, w.idx, n_kept, k_out, v_out, stream_);
    }
    // -||k||, with the first radix histogram accumulated by the same kernel (vector path) -- knorm_press.py:38
    bool hist1_done = false;
    uint32_t* hist1 = (n_kept < S && topk_fused_hist_wanted(S)) ? topk_carve_ws(w.topk, R, 1).hist1 : nullptr;  // short rows: one-launch select with its own digits
    if (int rc = kvp_rownorm_launch(k, dtype, B, H, S, D, k_sb, k_sh, k_ss, -1.0f, w.scores, stream, hist1, &hist1_done)) return rc;
    if (int rc = topk_select_impl(w.scores, R, S, S, n_kept, w.idx, n_kept, 0, 0, w.topk, w.topk_bytes, true, hist1_done, stream)) return rc;
    if (by_score)   // (the norms are in w.scores: the one-launch path that keeps them in registers is not taken with this flag)
        if (int rc = topk_order_by_score(w.scores, R, S, S, n_kept, w.idx, false, w.order, w.order_bytes, stream)) return rc;
    return kvp_gather_kv(k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, H, S, D, w.idx, n_kept, k_out, v_out, stream_);
}

extern "C" size_t kvp_snapkv_compress_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D,
                                                      int64_t n_kept) {
    if (B < 1 || Hq < 1 || Hkv < 1 || S < 1 || W < 1 || D < 1 || n_kept < 0) return 256;
    return carve(nullptr, B * Hkv, std::max<int64_t>(1, S - W), S, n_kept, kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D)).total_bytes;
}

// long rows, kernel_size 5: the cluster select pools SnapKV's column sums in its loader (no pooling launch, no score round trip)
static bool snapkv_cluster_pooled(int64_t R, int64_t Sm, int kernel_size) {
    return kernel_size == 5 && topk_cluster_eligible(R, Sm) && topk_cluster_launchable();
}

// select + gather after a SnapKV scorer has run (fused: hist1 holds the first pass over the S - W non-window columns)
static int snapkv_select_gather(const CompressWs& w, bool fused, const float* colsum, float inv, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss, const void* v,
                                int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype, int64_t B, int64_t Hkv, int64_t S, int64_t W, int64_t D,
                                int64_t n_kept, void* k_out, void* v_out, hipStream_t stream, bool by_score) {
    const int64_t R = B * Hkv;
    int rc;
    if (fused && colsum && !topk_pooled_rows_eligible(S - W, 5)) {  // long rows, kernel_size 5: pooling inside the cluster select's loader
        TopkWs tw = topk_carve_ws(w.topk, R, (S - W + TK_CHUNK - 1) / TK_CHUNK);
        if (n_kept - W == 0 || n_kept == S)   // only the window / everything is kept: no selection
            rc = topk_select_impl(colsum, R, S - W, S - W, n_kept - W, w.idx, n_kept, (uint32_t)(S - W), (uint32_t)W, w.topk, w.topk_bytes, true, false, stream);
        else
            rc = topk_cluster_select(TOPK_CLUSTER_POOL5, colsum, S - W, inv, nullptr, 0, 0, 0, 0, 1, 0.f, R, S - W, n_kept - W, w.idx, n_kept, (uint32_t)(S - W),
                                     (uint32_t)W, tw, false, stream);
        KVP_CHECK_ARG(rc != 1, "snapkv_compress: cluster select not launchable");   // (snapkv_cluster_pooled() asked the same question before the scorer ran)
    } else if (fused && colsum)  // short rows, kernel_size 5: pooling happens inside the select's loader
        rc = topk_select_pooled_rows(colsum, R, S - W, inv, n_kept - W, w.idx, n_kept, (uint32_t)(S - W), (uint32_t)W, stream);
    else if (fused)  // short rows carry no fused histogram: the select is one launch of its own (topk_row_eligible)
        rc = topk_select_impl(w.scores, R, S - W, S, n_kept - W, w.idx, n_kept, (uint32_t)(S - W), (uint32_t)W, w.topk, w.topk_bytes, true,
                              topk_fused_hist_wanted(S - W), stream);
    else
        rc = topk_select_impl(w.scores, R, S, S, n_kept, w.idx, n_kept, 0, 0, w.topk, w.topk_bytes, true, false, stream);
    if (rc) return rc;
    if (by_score) {
        // descending score: the window columns first (the reference's pad = max + 1, ties by position), then the scored ones.  Keys come
        // from the pooled scores where they were written, else from the un-pooled column sums (pooled again per kept position)
        if (fused && colsum) rc = topk_order_by_score(colsum, R, S, S - W, n_kept, w.idx, false, w.order, w.order_bytes, stream, 1, S - W, inv);
        else rc = topk_order_by_score(w.scores, R, S, S, n_kept, w.idx, false, w.order, w.order_bytes, stream, 0, fused ? S - W : -1);
        if (rc) return rc;
    }
    return kvp_gather_kv(k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, Hkv, S, D, w.idx, n_kept, k_out, v_out, stream);
}

// As kvp_snapkv_compress_rope, but starting from the hidden states of the last W tokens and the q_proj weight: the
// projection and the RoPE run in one kernel of the library (qproj.hip) instead of a library GEMM + a RoPE launch.
extern "C" int kvp_snapkv_compress_hidden(const void* hidden_win, int64_t x_sb, int64_t x_sw, const void* wq, int64_t hidden,
                                          const void* cosp, const void* sinp, int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb,
                                          int64_t k_sh, int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype,
                                          int64_t B, int64_t Hq, int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size,
                                          int64_t n_kept, void* k_out, void* v_out, void* ws, size_t ws_bytes, int flags,
                                          kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = kvp_async_check("kvp_snapkv_compress")) return rc;
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && W >= 1 && S > W && D >= 1 && n_kept >= 0 && n_kept <= S,
                  "snapkv_compress: bad shape B=%ld Hq=%ld Hkv=%ld S=%ld W=%ld D=%ld n=%ld", (long)B, (long)Hq, (long)Hkv, (long)S, (long)W,
                  (long)D, (long)n_kept);
    if (n_kept == 0) return KVP_OK;
    const int64_t R = B * Hkv;
    const CompressWs w = carve(ws, R, S - W, S, n_kept, kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D));
    if (int rc = check_ws(w, ws, ws_bytes, "snapkv_compress")) return rc;
    if (!(flags & KVP_TOPK_WS_CLEAN) && hipMemsetAsync(w.topk, 0, topk_carve_ws(nullptr, R, 1).zero_bytes, stream) != hipSuccess) {
        kvp_set_error("snapkv_compress: hipMemsetAsync failed");
        return KVP_EHIP;
    }
    const bool fused = n_kept >= W;
    uint32_t* hist1 = (fused && topk_fused_hist_wanted(S - W)) ? topk_carve_ws(w.topk, R, 1).hist1 : nullptr;
    const bool pooled = fused && (topk_pooled_rows_eligible(S - W, kernel_size) || snapkv_cluster_pooled(R, S - W, kernel_size));  // pool + select in one launch
    if (int rc = snapkv_score_hidden_impl(hidden_win, x_sb, x_sw, wq, hidden, cosp, sinp, cs_sb, cs_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq,
                                          Hkv, S, W, D, kernel_size, w.scores, w.scorer, w.scorer_bytes, stream, hist1,
                                          pooled ? SNAP_FINISH_COLSUM : fused ? SNAP_FINISH_NO_PAD : SNAP_FINISH_FULL))
        return rc;
    return snapkv_select_gather(w, fused, pooled ? snapkv_ws_colsum(w.scorer, B, Hq, Hkv, S, W, D) : nullptr, snapkv_pool_scale(Hq, Hkv, W, kernel_size), k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, Hkv, S, W, D, n_kept, k_out, v_out, stream, (flags & KVP_ORDER_SCORE) != 0);
}

extern "C" int kvp_snapkv_compress_rope(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sw, const void* cosp, const void* sinp,
                                        int64_t cs_sb, int64_t cs_sw, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                                        const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, int dtype, int64_t B, int64_t Hq,
                                        int64_t Hkv, int64_t S, int64_t W, int64_t D, int kernel_size, int64_t n_kept, void* k_out,
                                        void* v_out, void* ws, size_t ws_bytes, int flags, kvp_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = kvp_async_check("kvp_snapkv_compress")) return rc;
    KVP_CHECK_ARG(B >= 1 && Hq >= 1 && Hkv >= 1 && W >= 1 && S > W && D >= 1 && n_kept >= 0 && n_kept <= S,
                  "snapkv_compress: bad shape B=%ld Hq=%ld Hkv=%ld S=%ld W=%ld D=%ld n=%ld", (long)B, (long)Hq, (long)Hkv, (long)S, (long)W,
                  (long)D, (long)n_kept);
    if (n_kept == 0) return KVP_OK;
    const int64_t R = B * Hkv;
    const size_t snap_bytes = kvp_snapkv_workspace_bytes(B, Hq, Hkv, S, W, D);
    const CompressWs w = carve(ws, R, S - W, S, n_kept, snap_bytes);
    if (int rc = check_ws(w, ws, ws_bytes, "snapkv_compress")) return rc;
    const bool clean = (flags & KVP_TOPK_WS_CLEAN) != 0;
    if (!clean && hipMemsetAsync(w.topk, 0, topk_carve_ws(nullptr, R, 1).zero_bytes, stream) != hipSuccess) {
        kvp_set_error("snapkv_compress: hipMemsetAsync failed");
        return KVP_EHIP;
    }
    const bool fused = n_kept >= W;
    uint32_t* hist1 = (fused && topk_fused_hist_wanted(S - W)) ? topk_carve_ws(w.topk, R, 1).hist1 : nullptr;
    const bool pooled = fused && (topk_pooled_rows_eligible(S - W, kernel_size) || snapkv_cluster_pooled(R, S - W, kernel_size));  // pool + select in one launch
    if (int rc = snapkv_score_rope_impl(q, q_sb, q_sh, q_sw, cosp, sinp, cs_sb, cs_sw, k, k_sb, k_sh, k_ss, dtype, B, Hq, Hkv, S, W, D,
                                        kernel_size, w.scores, w.scorer, w.scorer_bytes, stream, hist1, false,
                                        pooled ? SNAP_FINISH_COLSUM : fused ? SNAP_FINISH_NO_PAD : SNAP_FINISH_FULL))
        return rc;
    return snapkv_select_gather(w, fused, pooled ? snapkv_ws_colsum(w.scorer, B, Hq, Hkv, S, W, D) : nullptr, snapkv_pool_scale(Hq, Hkv, W, kernel_size), k, k_sb, k_sh, k_ss, v, v_sb, v_sh, v_ss, dtype, B, Hkv, S, W, D, n_kept, k_out, v_out, stream, (flags & KVP_ORDER_SCORE) != 0);
}
