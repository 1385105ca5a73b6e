// Host-side shared declarations: workspace layout and kernel launchers (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/fastkv_hip.h"

namespace fk {

constexpr int TKA = 256;        // keys per workgroup in score_logits (4 waves x 64 lanes, one key per lane)
constexpr int SEL_THREADS = 1024;
constexpr size_t CTRL_BYTES = 8192;  // control block at the start of every operator workspace (fastkv_workspace_init):
                                     // u64 magic, u32 epoch (the rest is reserved)
                                     // u64 magic, u32 epoch, u32 abort word (token of a launch that gave up waiting)
constexpr uint64_t CTRL_MAGIC = 0x66617374'6b765f31ull;
constexpr int FUSED_CU_SLOTS = 2048;  // {XCC_ID, SE, SH, CU} of HW_ID as an index: who ran on a compute unit in this launch (placement check of the fused kernel)
constexpr int EPOCH_STRIDE = 64;     // the workspace epoch advances by this much per operator call: sub-launch s of a call (< EPOCH_STRIDE) uses epoch + s, so
                                     // no two launches ever share a hand-off token (the xor-mixed sub index could alias another epoch's token)
constexpr int FUSED_MAX_WGS = 1024;  // (unit, span) pairs whose hand-off records exist at a time -- a regular launch: 512 workgroups (2 per CU) x up to 2 streams; a rolling launch: the 2 F entries its record areas rotate over (four entries of 256 workgroups at 32k): sizes the record areas
// The split selection's granule tables (select.hip): a launch takes the table path only with rows x chunks <= SPL_MAX_WGS workgroups,
// one line of SPL_LINE 8-byte granules per workgroup.  An operator call makes two such launches (the per-head rows, the TSP rows): two
// tables at FIXED offsets, never anything else in them (a launch beyond SPL_MAX_WGS runs the wait-free selection and has no table).
constexpr int SPL_LINE = 32, SPL_MAX_WGS = 1024;
constexpr size_t SELTAB_ONE_BYTES = (size_t)SPL_MAX_WGS * SPL_LINE * 8;
constexpr size_t SELTAB_FIXED_BYTES = 2 * SELTAB_ONE_BYTES;
constexpr int HIST12 = 4096;    // bins of the high-12-bit key histogram that score_finalize / tsp_rowsum build for select

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Measurement switches (FASTKV_FUSED_ROLLING_PERT / _PARTS / _F, FASTKV_FUSED_MAX_WGS, FASTKV_FUSED_STAGGER_US, FASTKV_FUSED_TUNE,
// FASTKV_FUSED_STREAMS, FASTKV_CHAIN): read from the environment only in builds with -DFK_EXPERIMENTS (FASTKV_CXXFLAGS, into a
// FASTKV_BUILD_DIR); the product library holds their defaults as constants and dispatches on nothing a test does not cover.
static inline int exp_env_int(const char *name, int dflt)
{
#ifdef FK_EXPERIMENTS
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// fastkv_problem.reserved bits 0-1.  Two arithmetic CONTRACTS for the contraction of utils.py:94 (oracle/fastkv_oracle.c, "the
// contraction"): the fp32 fma chain -- engines VALU and MFMA (v_mfma_f32_32x32x2_f32), bit-identical to each other -- and "mfma16",
// what v_mfma_f32_32x32x16_f16 computes on the fp16 operands themselves (16x the matrix rate).  AUTO = the library's default
// contract (FASTKV_CONTRACTION=fmaf | mfma16, default fmaf since round 6) with the engine that suits the shape.
enum { ENGINE_AUTO = 0, ENGINE_VALU = 1, ENGINE_MFMA = 2, ENGINE_MFMA16 = 3 };
bool default_contract_f16();      // capi.hip

struct Layout {
    int engine;                      // contraction engine of score_logits (both are the oracle's fmaf chain)
    int G, R, RB, passes, R_alloc;   // query rows per KV head (G*W), rows per pass, passes, padded rows
    int n, Sp, n_pad;                // candidates S-W, padded logits row stride, padded score row stride
    int ntA;                         // tiles of score_logits
    size_t off_qf, off_logits, off_gmax, off_rinv, off_c, off_t, off_hist, off_thist, off_arrive, off_seltab, off_fpart, off_fchain, off_idx, off_keys, total;
    int zero_words;                  // u32 words from off_hist that row_stats zeroes: histograms + arrival counters
};

static inline size_t select_ws_bytes(int64_t rows, int64_t n, int64_t k)
{
    (void)n;
    // stand-alone select with ORDER_SCORE: ascending index list + key list for the rank scatter
    const size_t kal = ((size_t)k + 7) & ~(size_t)7;
    return align_up((size_t)rows * kal * sizeof(int64_t), 256) + align_up((size_t)rows * kal * sizeof(uint16_t), 256);
}

// Per-batch-entry base addresses instead of one base + batch stride (fastkv_update_kv_ptrs_f16: the batch entries are
// separately allocated tensors, e.g. the layers of a model).  Device arrays of p.B addresses each; head / row strides are
// shared by the entries.  nullptr = not used.
struct PtrTables {
    const uint64_t *q, *k, *v;
    const uint64_t *k_out, *v_out;
};

static inline int resolve_engine(const fastkv_problem &p)
{
    const int e = p.reserved & 3, R = (p.H / p.Hkv) * p.window;
    return e != ENGINE_AUTO ? e : default_contract_f16() ? ENGINE_MFMA16 : (R >= 24 ? ENGINE_MFMA : ENGINE_VALU);
}

static inline Layout make_layout(const fastkv_problem &p)
{
    Layout L;
    L.G = p.H / p.Hkv;
    L.R = L.G * p.window;
    L.engine = p.reserved & 3;
    if (L.engine == ENGINE_AUTO)     // (fmaf contract: a 32-row MFMA block needs rows to fill it; mfma16: missing rows are zero queries)
        L.engine = default_contract_f16() ? ENGINE_MFMA16 : (L.R >= 24 ? ENGINE_MFMA : ENGINE_VALU);
    int r8 = (L.R + 7) / 8 * 8;
    if (L.engine == ENGINE_MFMA || L.engine == ENGINE_MFMA16) { L.RB = 32; L.passes = (L.R + 31) / 32; }
    else if (r8 <= 64) { L.RB = r8 <= 8 ? 8 : r8 <= 16 ? 16 : r8 <= 32 ? 32 : 64; L.passes = 1; }
    else { L.RB = 64; L.passes = (r8 + 63) / 64; }
    L.R_alloc = L.RB * L.passes;
    L.n = p.S - p.window;
    L.Sp = (p.S + 7) / 8 * 8;
    L.n_pad = (L.n + 7) / 8 * 8;
    L.ntA = (p.S + TKA - 1) / TKA;
    // Token-tagged hand-off granules live at FIXED offsets right behind the control block, in areas of fixed size that never hold
    // anything else: a reader accepts a granule by its 32-bit token, and memory that other data of ANOTHER call's layout has passed
    // through (fp16 scores, logits, indices: arbitrary bit patterns) carries the current token once in 2^32 words -- found by
    // tools/soak_rolling.py, one wrong call in 500,000 when the shapes change from call to call and the areas moved with them.
    // Here an area only ever holds granules of earlier launches (older tokens) or the zeros of fastkv_workspace_init.
    size_t o = CTRL_BYTES;
    L.off_fpart = o;  o += align_up((size_t)FUSED_MAX_WGS * (32 * 24 + 2 * 4 * 31 * 8) + FUSED_CU_SLOTS * 8 + (size_t)FUSED_MAX_WGS * 8, 256);   // fused score: row max / row sum / halo granules + one {token, unit} granule per compute unit + one "done" granule per (unit, span) of a rolling launch
    L.off_fchain = o; o += align_up((size_t)512 * 1024 * 8, 256);               // fused score, more than 4 query heads per KV head: [unit span][positions]: 512 Ki head-sum granules at most
    L.off_seltab = o; o += SELTAB_FIXED_BYTES;                                   // [per-head rows' table | TSP rows' table], SELTAB_ONE_BYTES each
    L.off_qf = o;     o += align_up((size_t)p.B * p.Hkv * L.R_alloc * p.D * 4, 256);
    L.off_logits = o; o += align_up((size_t)p.B * p.H * p.window * L.Sp * 2, 256);
    L.off_gmax = o;   o += align_up((size_t)p.B * p.H * p.window * 4, 256);
    L.off_rinv = o;   o += align_up((size_t)p.B * p.H * p.window * 4, 256);
    L.off_c = o;      o += align_up((size_t)p.B * p.Hkv * L.n_pad * 2, 256);
    L.off_t = o;      o += align_up((size_t)p.B * L.n_pad * 2, 256);
    L.off_hist = o;   o += align_up((size_t)p.B * p.Hkv * HIST12 * 4, 256);     // 12-bit key histograms of the score rows
    L.off_thist = o;  o += (size_t)p.B * HIST12 * 4;                            // ... and of the TSP rows (adjacent: zeroed together)
    L.off_arrive = o; o += align_up((size_t)p.B * (p.Hkv + 1) * 4, 256);        // split select: arrival counter per score row (zeroed too)
    L.zero_words = (int)((o - L.off_hist) / 4);
    L.off_idx = o;    o += align_up((size_t)p.B * p.Hkv * (size_t)(p.capacity > p.window ? p.capacity - p.window : 0) * 8, 256);
    L.off_keys = o;   // winners' 16-bit keys in ascending position, rows padded to a multiple of 8
    {
        const size_t kk = p.capacity > p.window ? (size_t)(p.capacity - p.window) : 0;
        o += align_up((size_t)p.B * p.Hkv * ((kk + 7) & ~(size_t)7) * 2, 256);
    }
    L.total = o;
    return L;
}

// bounded hand-off waits (fk_device.h SpinCtl): the process-wide abort flag in pinned host memory (device view; nullptr if
// the allocation failed: the launchers then take the staged path) and the wall-clock limit in s_memrealtime ticks
uint32_t *abort_flag_device();
uint64_t spin_limit_ticks();
bool no_wait_mode();              // FASTKV_FUSED=0 or the fail-safe switch after a placement violation (capi.hip)

// launchers (defined in score.hip / select.hip / compact.hip); all return hipError_t of the launch
hipError_t launch_score(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, const void *k,
                        const int64_t *ks, uint16_t *c_out, int64_t c_row_stride, uint16_t *t_out, int64_t t_row_stride,
                        char *ws, hipStream_t st, int64_t *all_idx = nullptr, uint16_t *all_keys = nullptr,
                        int64_t all_key_stride = 0, uint32_t **epoch_bump_later = nullptr, const PtrTables *pt = nullptr);
// fused logits + softmax + window-row sum + pooling/head sum (fused.hip); false = shape not covered, take the three-kernel path
bool launch_score_fused(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, const void *k,
                        const int64_t *ks, uint16_t *c_out, int64_t c_row_stride, int64_t *all_idx, uint16_t *all_keys,
                        int64_t all_key_stride, char *ws, hipStream_t st, hipError_t *err, const PtrTables *pt = nullptr);
int fused_entries_per_launch(const fastkv_problem &p);      // fused.hip: entries one fused launch holds (0: geometry off the fused path)
hipError_t launch_epoch_bump(uint32_t *epoch, hipStream_t st);
hipError_t launch_head_sum(const uint16_t *c, int64_t B, int64_t R, int64_t n, uint16_t *t_out, hipStream_t st);
hipError_t launch_pool_rows(const uint16_t *in, int64_t in_stride, int64_t rows, int64_t n, int ksize, int pooling, uint16_t *out,
                            int64_t out_stride, hipStream_t st);
hipError_t launch_sp_logits(const fastkv_problem &p, const void *q_win, const int64_t *qs, const void *k, const int64_t *ks,
                            uint16_t *logits, int Sp, int col_off, float *qf_scratch, hipStream_t st);
hipError_t launch_sp_rowstats(const fastkv_problem &p, uint16_t *logits, const fastkv_sp_window &w, int mode, float *gmax,
                              int64_t *sums, hipStream_t st);
hipError_t launch_sp_scores(const fastkv_problem &p, uint16_t *logits, const fastkv_sp_window &w, const float *gmax,
                            const int64_t *sums, uint16_t *c_out, int64_t c_row_stride, uint16_t *t_out,
                            int64_t t_row_stride, int n_own, hipStream_t st);
// The TSP row sums t[b, j] = fp16(sum over the KV heads of c[b, g, j]) (utils.py:127) + their 12-bit key histogram, computed by extra
// workgroups of the per-head selection launch instead of a launch of their own (round 5: 8-10 us for 0.5 MB, all latency, on the step's
// critical path): the selection's own workgroups come first in the grid, the extra ones behind them read the scores the launch is
// given and share nothing with the selection.  `B` batch rows of `Hkv` score rows each.
struct TspFold {
    int B, Hkv;
    uint16_t *t_out;
    int64_t t_row_stride;
    uint32_t *thist;
};
// true when launch_select will take the split selection (table or wait-free): the launches a TspFold can ride in
bool select_takes_split(const uint16_t *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, const uint32_t *hist12);
hipError_t launch_select(const uint16_t *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, int append,
                         int64_t *idx_out, int64_t idx_row_stride, uint16_t *key_out, int64_t key_row_stride,
                         const uint32_t *hist12, uint32_t *arrive, uint32_t *table, hipStream_t st, uint32_t *ctrl = nullptr,
                         const TspFold *fold = nullptr);
hipError_t launch_rank_scatter(const int64_t *idx_asc, int64_t asc_row_stride, const uint16_t *keys, int64_t key_row_stride,
                               int64_t rows, int64_t k, int64_t *out, int64_t out_row_stride, hipStream_t st);
// idx: ascending-position winners; keys != nullptr => rows are placed in ORDER_SCORE (rank by comparison counting, or -- 64 or
// more heads -- by a grouping pass that turns the key list into the slot list IN PLACE: `keys` is scratch of the call) and
// idx_sorted_out (optional) receives the indices in that order
hipError_t launch_compact(const fastkv_problem &p, const void *k, const int64_t *ks, const void *v, const int64_t *vs,
                          const int64_t *idx, const uint16_t *keys, int64_t *idx_sorted_out, void *k_out, void *v_out,
                          hipStream_t st, uint32_t *epoch_bump = nullptr, const int64_t *out_strides = nullptr,
                          const PtrTables *pt = nullptr);
hipError_t launch_winner_keys(const uint16_t *scores, int64_t row_stride, int64_t n, const int64_t *idx, int64_t rows, int kk,
                              uint16_t *keys, hipStream_t st);
hipError_t launch_gather_rows(const void *src, int64_t sbs, int64_t srs, const int64_t *idx, int64_t ibs, int64_t batches,
                              int64_t rows_out, int64_t rows_in, int64_t row_bytes, void *dst, hipStream_t st, const int64_t *pos_in = nullptr,
                              int64_t pos_batch_stride = 0, int64_t *pos_out = nullptr);

}  // namespace fk
