// Optional per-kernel HIP-event timing (off by default; enabled by fastkv_profile_enable).  When enabled every kernel
// launch of the library is bracketed by two events recorded on the launch stream; fastkv_profile_read() synchronises
// them and returns, per kernel id, the number of launches and the summed duration.  Used by bench.py for the
// `roofline` figures; never enabled in the product path.
#pragma once
#include <hip/hip_runtime.h>

namespace fk {

enum KernelId { K_PREP_Q = 0, K_LOGITS, K_ROWSTATS, K_FINALIZE, K_TSP_ROWSUM, K_SELECT, K_RANK, K_COMPACT, K_GATHER, K_FUSED, K_SELECT_SPLIT, K_SP_AUX, K_DECODE, K_COUNT };

struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kid, hipStream_t s);
    ~ProfScope();
};

}  // namespace fk
