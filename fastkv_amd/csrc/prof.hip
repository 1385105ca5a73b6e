#include "prof.h"
#include <vector>
#include "../../include/fastkv_hip.h"

namespace fk {

static bool g_enabled = false;
struct Rec { hipEvent_t a, b; int kid; };
static std::vector<Rec> g_recs;     // recorded since the last read
static std::vector<Rec> g_pool;     // reusable events

static const char *const g_names[K_COUNT] = {"prep_q", "score_logits", "row_stats", "score_finalize", "tsp_rowsum",
                                             "select_topk", "rank_scatter", "compact_kv", "gather_rows", "score_fused", "select_split", "sp_aux", "decode"};

ProfScope::ProfScope(int kid, hipStream_t s) : slot(-1), st(s)
{
    if (!g_enabled) return;
    Rec r;
    if (!g_pool.empty()) { r = g_pool.back(); g_pool.pop_back(); }
    else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return; }
    r.kid = kid;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope()
{
    if (slot >= 0) (void)hipEventRecord(g_recs[slot].b, st);
}

}  // namespace fk

extern "C" {

void fastkv_profile_enable(int on) { fk::g_enabled = on != 0; }

int fastkv_profile_kernels(void) { return fk::K_COUNT; }
const char *fastkv_profile_kernel_name(int kid) { return (kid >= 0 && kid < fk::K_COUNT) ? fk::g_names[kid] : ""; }

// Synchronises on every recorded event; counts[kid] += launches, ms[kid] += summed duration; clears the records.
int fastkv_profile_read(int64_t *counts, double *ms)
{
    for (auto &r : fk::g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return FASTKV_ELAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return FASTKV_ELAUNCH;
        counts[r.kid] += 1;
        ms[r.kid] += (double)t;
        fk::g_pool.push_back(r);
    }
    fk::g_recs.clear();
    return FASTKV_OK;
}

}  // extern "C"
