// ExpectedAttention on the matrix cores (bf16/f16, D = 128): query statistics (syrk) and the
// quadratic-form logits.  Round-1 status: not implemented yet -- the eligibility predicates
// return false so kvp_ea_* use the generic kernels of ea.hip.
#include "ea_internal.h"

bool ea_mfma_qstats_eligible(const void*, int64_t, int64_t, int64_t, int, int64_t) { return false; }
size_t ea_mfma_qstats_ws_bytes(int64_t, int64_t, int64_t, int64_t) { return 0; }
int ea_mfma_qstats(const void*, int64_t, int64_t, int64_t, int, int64_t, int64_t, int64_t, int64_t, float*, float*, void*,
                   hipStream_t) {
    kvp_set_error("ea_mfma_qstats: not implemented");
    return KVP_EUNSUPPORTED;
}

bool ea_mfma_logits_eligible(const EaArgs&, int) { return false; }
size_t ea_mfma_logits_scratch_bytes(int64_t, int64_t, int64_t) { return 0; }
uint32_t ea_mfma_logits_nblk(const EaArgs&) { return 0; }
int ea_mfma_logits(const EaArgs&, int, float*, uint32_t, float*, float*, void*, hipStream_t) {
    kvp_set_error("ea_mfma_logits: not implemented");
    return KVP_EUNSUPPORTED;
}
