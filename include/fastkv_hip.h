/*
 * fastkv_hip.h -- C ABI of the MI355X (gfx950) implementation of FastKV's hot path.
 *
 * The reference (dongwonjo/FastKV) is pure Python: the operator this library replaces is
 *   FastKVCluster.update_kv            /root/reference/baselines/fastkv/utils.py:80-134
 * and its one extra consumer, the TSP hidden-state gather in the decoder layer
 *   llama_decoderlayer_forward_fastkv  /root/reference/baselines/fastkv/llama_model.py:252-259
 * The reference has no FFI of its own; these entry points are what a ctypes binding inside
 * that Python method would call (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor storage);
 *   - fp16 tensors are passed as `const void*`, strides are in ELEMENTS, the innermost
 *     (head_dim) stride must be 1 and every row must be 16-byte aligned;
 *   - logical layout of q/k/v is [B, H, S, D]; any physical layout is accepted through the
 *     strides (the attention module hands over [B,S,H,D] storage, llama_model.py:117-122);
 *   - all work is enqueued on `stream`; no allocation, no host synchronisation, no
 *     exceptions across the ABI; the calls are hipGraph-capturable;
 *   - return value 0 = success, negative = error code (fastkv_strerror).
 */
#ifndef FASTKV_HIP_H
#define FASTKV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTKV_OK 0
#define FASTKV_EINVAL (-1)     /* bad argument (shape, stride, alignment, pooling, ...) */
#define FASTKV_EWORKSPACE (-2) /* workspace too small */
#define FASTKV_ELAUNCH (-3)    /* HIP launch error */
#define FASTKV_EUNSUPPORTED (-4)
#define FASTKV_EABORTED (-5)   /* an EARLIER fused launch gave up a bounded in-kernel wait; see fastkv_workspace_init */
#define FASTKV_EOVERFLOW (-6)  /* an EARLIER static-decode step found its cache slab full (reported like FASTKV_EABORTED) */
#define FASTKV_EBOUNDS (-7)    /* FASTKV_DEBUG_BOUNDS=1 only: an EARLIER gather was handed a row index outside [0, S) (it read a
                                  clamped row, as always; reported like FASTKV_EABORTED) */
#define FASTKV_EPLACEMENT (-8) /* workgroups of an EARLIER fused scoring launch shared a compute unit with workgroups of another unit:
                                  redo the calls since the last report (see fastkv_set_placement_policy; reported like
                                  FASTKV_EABORTED) */

#define FASTKV_POOL_AVG 0 /* F.avg_pool1d(k, padding=k//2, stride=1)   utils.py:105-106 */
#define FASTKV_POOL_MAX 1 /* F.max_pool1d(k, padding=k//2, stride=1)   utils.py:107-108 */

#define FASTKV_ORDER_INDEX 0 /* selected rows in ascending position                         */
#define FASTKV_ORDER_SCORE 1 /* score descending, ties by ascending position: the order of  */
                             /* `topk(sorted=True)` (utils.py:113) with a canonical tie rule */

/* Problem description shared by the entry points (plain data, passed by pointer). */
typedef struct fastkv_problem {
    int32_t B;         /* batch */
    int32_t H;         /* query heads */
    int32_t Hkv;       /* key/value heads, H % Hkv == 0 */
    int32_t S;         /* prompt length (q_len == kv_len, utils.py:82) */
    int32_t D;         /* head dim: 64, 128 or 256 */
    int32_t window;    /* FastKVCluster.window_size */
    int32_t kernel;    /* FastKVCluster.kernel_size (odd) */
    int32_t pooling;   /* FASTKV_POOL_* */
    int32_t capacity;  /* max_capacity_prompt after the proportional rule (utils.py:86-87); window < capacity <= S */
    int32_t tsp_len;   /* 0 = no TSP on this layer; else window < tsp_len < S (utils.py:126) */
    int32_t order;     /* FASTKV_ORDER_* for the K/V rows */
    int32_t reserved;  /* 0 = the library's default contraction contract (FASTKV_CONTRACTION=fmaf | mfma16 at load; fmaf unless
                          set).  Bits 0-1 force an engine of the scoring kernels.  Two arithmetic CONTRACTS for the fp16 matmul
                          of utils.py:94 (both restated bit for bit by oracle/fastkv_oracle.c, tested against each other's oracle):
                            1 = vector ALU, 2 = FP32 matrix pipe (v_mfma_f32_32x32x2_f32): "fmaf", the fp32 fma chain in
                                ascending head-dim order; the two produce the same bits.  THE DEFAULT since round 6: it IS the
                                reference's matmul -- torch's CPU kernel accumulates exactly so: 0 of 1.0e9 fp16 logits differ
                                over the 120-case sweep (tests/test_oracle_golden.py) -- and what then remains between this
                                library and the reference is the softmax denominator's summation order alone
                            3 = "mfma16": v_mfma_f32_32x32x16_f16 on the fp16 operands, chained over ascending chunks of 16
                                dims (the opt-in fast mode: 16x the matrix rate, 0.58 instead of 0.78 ms per 32-layer step; 1e-3
                                of the logits one fp16 ulp away from the reference's; tools/probes/README.md has the
                                instruction's arithmetic) */
} fastkv_problem;

/* Bytes of scratch `fastkv_update_kv_f16` / `fastkv_score_f16` need for this problem (>= ~23 MiB: the hand-off records of the fused
 * scoring kernel and of the split selection lie in fixed-size areas at fixed offsets behind the control block, whatever the shape, so
 * that calls of DIFFERENT shapes can share one workspace: an area that holds token-tagged records never holds anything else). */
size_t fastkv_workspace_bytes(const fastkv_problem *p);

/*
 * Call ONCE per workspace allocation (256-B aligned, any size >= 8 KiB), before its first use by
 * `fastkv_update_kv_f16` / `fastkv_score_f16`: the first 8 KiB of an operator workspace are a control block (magic word,
 * call epoch of the fused scoring kernel's hand-offs) that the library keeps consistent from then on, so no per-call
 * memset is needed and graph replays are safe.  Every operator call advances the epoch, so the token its kernels tag their
 * hand-off records with is never seen again; the call also clears the rest of the workspace (stream-ordered memset), because an
 * allocation on top of an EARLIER workspace's memory would otherwise restart the token sequence over that workspace's
 * records.  It is idempotent for the control block (a live one keeps its epoch), so it may be repeated between calls or end
 * up inside a captured graph.  A workspace that was never initialised makes the scoring kernel leave at once and report FASTKV_EABORTED (at the
 * next synchronisation), never a silent wrong answer.  One workspace serves one stream at a time.
 *
 * Residency.  The fused scoring kernel and the split selection exchange partial results between workgroups INSIDE a launch,
 * so a waiting workgroup needs its partners to be resident.  The library sizes those grids to what the device holds when
 * it is otherwise idle (occupancy query), but HIP gives no residency guarantee next to other work (hipLaunchCooperativeKernel
 * only checks the grid size at launch, and costs 15-20 us per launch on this chip).  What the library guarantees instead:
 *   - another kernel holding compute units (another stream, RCCL, another process) only DELAYS the launch: its late
 *     workgroups start when units free up and the waiting ones are served then;
 *   - every in-kernel wait is bounded by wall-clock time (FASTKV_SPIN_LIMIT_MS, default 2000).  A workgroup that gives up
 *     tells the rest of the launch (which exits at its next poll) and raises a flag in pinned host memory; nothing traps,
 *     nothing hangs, the HIP context stays usable, indices are clamped so the following stages cannot fault.  The outputs
 *     of that call are invalid, and the NEXT operator call in the process (or fastkv_last_status()) returns FASTKV_EABORTED
 *     once -- the asynchronous-error convention of the HIP runtime itself -- and, under the default ("fail safe") policy of
 *     fastkv_set_placement_policy, the process switches to the no-wait kernels (what FASTKV_FUSED=0 selects): whatever held the
 *     compute units may still be there when the caller repeats the call, and the repeat must not run into the same wait;
 *   - calls on different streams of ONE process are chained by the library (an event dependency when the stream changes,
 *     taken under a lock that also covers the enqueue), so two such launches of one process never overlap;
 *   - processes that share a GPU, or graphs replayed concurrently on several streams, should set FASTKV_FUSED=0: the
 *     library then runs without ANY in-launch wait (three-kernel scoring; the split selection counts the row in every
 *     chunk instead of exchanging counters) at 2.02 ms instead of 0.77 ms per 32-layer step (1.79 ms against 0.57 under the mfma16
 *     contract; round 6); if they do not, overlapping
 *     launches end in FASTKV_EABORTED.
 */
int fastkv_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
/* FASTKV_EABORTED if a launch of this process gave up a bounded wait since the last report, FASTKV_EOVERFLOW if a static-decode
 * step (fastkv_decode_attention_f16 / fastkv_decode_step_attention_f16) found its slab full (the step then overwrites the last
 * cached row and the length stops advancing); clears the report; else 0.
 * Host-only (reads two words of pinned memory); call it after synchronising to learn about the calls just completed. */
int fastkv_last_status(void);
/*
 * A REGULAR launch of the fused scoring kernel (a call of one or two 32k layers; more entries of shorter ones) gives the two workgroups
 * that share a compute unit adjacent spans of one (batch row, kv head): which two share is the GPU's dispatch order on an idle device,
 * an observation and not a promise (docs/HISTORY.md on why the kernel cares: in round 3 a workgroup that ran a phase ahead of a
 * DIFFERENT unit's workgroup on its compute unit produced wrong sums now and then -- narrowed down to packed-fp32 instructions beside
 * the fp32-fma-chain contract's matrix phase; the library has been compiled WITHOUT packed-fp32 instructions since, which took the
 * strongest reproducer from 40 % wrong launches to 0 of 600).  Every regular launch checks the pairing: a workgroup that finds another
 * unit's workgroup of the same launch on its compute unit is counted in pinned host memory.  Launches of the fma-chain contract count
 * into the word the placement policy below acts on; launches of the "mfma16" contract, which issue neither half of that hazard, are
 * counted and never reported.  The ROLLING launch (fastkv_set_fused_rolling) runs its entries on shared compute units OUT OF STEP by
 * design, under both contracts, and arms nothing: that this is safe without packed-fp32 instructions is what the soaks of round 6
 * stand for (profiles/r06_soak_*.log: 240,000 + 200,000 random groups against the in-step launches, bit for bit).
 * Returns the count since the last reset (host only, no synchronisation; complete once the stream has been synchronised).  0 on an
 * idle GPU (the tests assert it); > 0 beside foreign kernels, or when a launch could not become resident all at once -- results were
 * bit-exact in every such test, but the guarantee of the pairing is gone: a shared GPU should run FASTKV_FUSED=0.  The count is a
 * running total (reported violations included) until `reset`.
 */
int fastkv_placement_violations(int reset);
/*
 * What a counted violation -- and, under policy 2, a given-up wait (FASTKV_EABORTED) -- leads to (process-wide; initial value from
 * FASTKV_STRICT_PLACEMENT: unset -> 2, "1" -> 1, "0" -> 0):
 *   2  fail safe (default): the next operator call / fastkv_last_status() returns FASTKV_EPLACEMENT once -- the outputs of the calls
 *      since the last report are not vouched for, redo them -- AND the process switches to the no-wait kernels (staged scoring,
 *      wait-free selection: what FASTKV_FUSED=0 selects), so that the redo, and everything after it, has no exposure left;
 *   1  strict: the report only, the fused kernels stay;
 *   0  count only.
 * Returns 0, FASTKV_EINVAL for another value.
 */
int fastkv_set_placement_policy(int policy);
/* The no-wait switch itself: fastkv_set_no_wait_mode(on) returns the previous setting (FASTKV_FUSED=0 cannot be undone by it);
 * fastkv_no_wait_mode() = 1 when no kernel with an in-launch wait will be launched (the environment variable or the switch). */
int fastkv_set_no_wait_mode(int on);
/* The rolling launch of the fused scoring kernel (csrc/fused.hip launch_score_fused): a call with more entries than the chip holds
 * at a time (8k - 32k token layers: 8 - 2 of them) scores ALL of them in one launch whose entries follow each other over the chip
 * out of step; a layer of more than 64k tokens (no regular fused launch holds it) is taken in parts of its KV heads the same way.
 * Both contraction contracts (round 6; FASTKV_FUSED_ROLLING_FMAF=0 keeps the fma chain on the in-step launches).  The hand-off
 * record areas rotate over twice the entries the chip holds, and an area changes hands EXPLICITLY: a workgroup publishes into it only
 * when every workgroup of its unit in the area's previous entry has left a "done" record -- an entry that is slower than its successors
 * (the fma chain's NaN redo, a compute unit held by a foreign kernel) keeps its records until it has read them.  Same results bit
 * for bit; on by default (FASTKV_FUSED_ROLLING=0 turns it off for the process); returns the previous setting. */
int fastkv_set_fused_rolling(int on);
int fastkv_no_wait_mode(void);

/*
 * The whole operator: replaces the compress branch of FastKVCluster.update_kv (utils.py:93-132).
 *   q, k, v        fp16, logical [B,H,S,D] / [B,Hkv,S,D] with the given element strides
 *   k_out, v_out   fp16 [B,Hkv,capacity,D] contiguous: rows 0..capacity-window-1 are the selected
 *                  rows in `order`, the last `window` rows are the window rows (utils.py:114-121)
 *   kv_idx_out     optional int64 [B,Hkv,capacity-window]: the per-head selected positions
 *                  (internal `indices` of utils.py:113; not returned by the reference)
 *   tsp_idx_out    int64 [B,tsp_len] ascending (utils.py:127-130); required iff tsp_len > 0
 *   scores_out     optional fp16 [B,Hkv,S-window]: `attn_cache` (utils.py:112), for tests
 */
int fastkv_update_kv_f16(const fastkv_problem *p,
                         const void *q, const int64_t q_strides[4],
                         const void *k, const int64_t k_strides[4],
                         const void *v, const int64_t v_strides[4],
                         void *k_out, void *v_out, int64_t *kv_idx_out, int64_t *tsp_idx_out,
                         void *scores_out, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same with strided outputs: k_out / v_out are [B,Hkv,capacity,D] views with element strides out_strides =
 * {batch, head, row} (row >= D, all multiples of 8; NULL = contiguous), e.g. the first `capacity` rows of a pre-sized
 * per-layer cache slab [B,Hkv,capacity + max_new_tokens,D]: the compaction then IS the cache write (the reference goes
 * through `past_key_value.update(...)`, a torch.cat copy, baselines/fastkv/llama_model.py:142).
 */
int fastkv_update_kv_strided_f16(const fastkv_problem *p,
                                 const void *q, const int64_t q_strides[4],
                                 const void *k, const int64_t k_strides[4],
                                 const void *v, const int64_t v_strides[4],
                                 void *k_out, void *v_out, const int64_t out_strides[3], int64_t *kv_idx_out,
                                 int64_t *tsp_idx_out, void *scores_out, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same operator over SEPARATELY ALLOCATED batch entries: p->B entries (e.g. the layers of a model whose compression was
 * deferred to the end of the forward pass -- the reference compresses layer by layer inside the attention forward,
 * /root/reference/baselines/fastkv/llama_model.py:136-142, but nothing reads a layer's compressed cache before decode), each
 * with its own q / k / v / k_out / v_out base address.  The five arrays hold p->B addresses each and live in DEVICE memory;
 * strides [1..3] (head, row, element) are shared by the entries, [0] is ignored; out_strides as in
 * fastkv_update_kv_strided_f16 (NULL: entries are contiguous [Hkv,capacity,D]).  kv_idx_out [B,Hkv,capacity-window] and
 * tsp_idx_out [B,tsp_len] are ordinary batched tensors.  One launch sequence for all entries: the per-launch latencies of
 * the small post-TSP layers (16.5 + 8.2 us each) are paid once.  Entries must be 16-B aligned (not checked: the addresses
 * are on the device).  Fused scoring path only: FASTKV_EUNSUPPORTED (nothing launched) for geometries that would take the
 * three-kernel path (fastkv_fused_entries_f16 == 0) -- call the strided entry point per entry then.  More entries than one fused
 * scoring launch holds are scored by several launches; the selection and the copy always run once over all entries.
 */
int fastkv_update_kv_ptrs_f16(const fastkv_problem *p, const void *const *q_ptrs, const int64_t q_strides[4],
                              const void *const *k_ptrs, const int64_t k_strides[4], const void *const *v_ptrs,
                              const int64_t v_strides[4], void *const *k_out_ptrs, void *const *v_out_ptrs,
                              const int64_t out_strides[3], int64_t *kv_idx_out, int64_t *tsp_idx_out, void *workspace,
                              size_t workspace_bytes, void *stream);

/*
 * How many batch entries ONE fused scoring launch holds for this geometry (p->B is ignored; 2 for Llama-3-8B at 32k, 16 at 2k), 0 when
 * the geometry takes the staged three-kernel path (a window other than 8, 5-7 query heads per KV head, FASTKV_FUSED=0, a prompt too
 * long for the resident grid).  fastkv_update_kv_ptrs_f16 accepts ANY number of entries of a geometry with a non-zero answer: it
 * scores them in launches of this many and selects / copies all of them with one launch each.  For callers that decide up front
 * whether to batch calls (fastkv_amd.cluster.DeferredCompression).
 */
int fastkv_fused_entries_f16(const fastkv_problem *p);

/*
 * Stage 1 alone: window-attention scores (utils.py:93-112 [+ :127 head sum]).
 *   scores_out      fp16 [B,Hkv,S-window] contiguous
 *   tsp_scores_out  optional fp16 [B,S-window]
 */
int fastkv_score_f16(const fastkv_problem *p, const void *q, const int64_t q_strides[4],
                     const void *k, const int64_t k_strides[4], void *scores_out, void *tsp_scores_out,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * Stage 2 alone: canonical top-k of `rows` independent fp16 score rows (utils.py:113 / :127).
 *   scores          fp16, row r starts at scores + r*row_stride (elements), n valid entries
 *   idx_out         int64 [rows, k + append]; `append` > 0 appends n, n+1, ... (the window positions,
 *                   utils.py:128-129) after the k selected entries
 *   order           FASTKV_ORDER_*
 */
int fastkv_select_f16(const void *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, int32_t order,
                      int32_t append, int64_t *idx_out, void *workspace, size_t workspace_bytes, void *stream);
size_t fastkv_select_workspace_bytes(int64_t rows, int64_t n, int64_t k);

/*
 * Stage 3 alone: gather/compact of K and V (utils.py:114-121).  idx int64 [B,Hkv,capacity-window].
 */
int fastkv_compact_f16(const fastkv_problem *p, const void *k, const int64_t k_strides[4],
                       const void *v, const int64_t v_strides[4], const int64_t *idx,
                       void *k_out, void *v_out, void *stream);

/*
 * Stage 3 alone in the reference's row order (FASTKV_ORDER_SCORE): idx_asc int64 [B,Hkv,capacity-window] are the winners
 * in ASCENDING position (fastkv_select_f16 with FASTKV_ORDER_INDEX), scores fp16 [B*Hkv rows, >= S-window] their score rows;
 * row i of a head lands in slot rank(i) (score descending, ties by ascending position: utils.py:113), found by comparison
 * counting inside the copy kernel -- no sort.  idx_sorted_out (optional) receives the positions in that order.
 * workspace: B*Hkv * round_up(capacity-window, 8) * 2 bytes, 16-B aligned.
 */
int fastkv_compact_ranked_f16(const fastkv_problem *p, const void *k, const int64_t k_strides[4],
                              const void *v, const int64_t v_strides[4], const int64_t *idx_asc, const void *scores,
                              int64_t score_row_stride, int64_t *idx_sorted_out, void *k_out, void *v_out, void *workspace,
                              size_t workspace_bytes, void *stream);

/*
 * Generic row gather: dst[b, r, :] = src[b, idx[b, r], :]; rows of `row_bytes` (multiple of 16).
 * Serves the TSP propagation of hidden states and position ids (llama_model.py:254-257).
 */
int fastkv_gather_rows(const void *src, int64_t src_batch_stride_bytes, int64_t src_row_stride_bytes,
                       const int64_t *idx, int64_t idx_batch_stride, int64_t batches, int64_t rows_out,
                       int64_t rows_in, int64_t row_bytes, void *dst, void *stream);

/*
 * The decoder layer's whole TSP propagation (llama_model.py:252-259; mistral_model.py:217-224) in ONE launch:
 *   position_ids_out[b, r] = position_ids[b, tsp_idx[b, r]]          (:254, `position_ids.gather(1, tsp_idx)`)
 *   hidden_out[b, r, :]    = hidden[b, tsp_idx[b, r], :]             (:255-257)
 * position_ids [batches, rows_in] int64 (pos_batch_stride in elements), outputs contiguous.  Indices outside [0, rows_in) read a
 * clamped row, as in fastkv_gather_rows -- so the TSP index tensor of a call that was REPORTED (FASTKV_EABORTED) cannot fault here.
 */
int fastkv_tsp_propagate(const void *hidden, int64_t hidden_batch_stride_bytes, int64_t hidden_row_stride_bytes,
                         const int64_t *position_ids, int64_t pos_batch_stride, const int64_t *tsp_idx, int64_t idx_batch_stride,
                         int64_t batches, int64_t rows_out, int64_t rows_in, int64_t row_bytes, void *hidden_out,
                         int64_t *position_ids_out, void *stream);

/*
 * ---- Sequence-sharded building blocks (one prompt split over P GPUs on the sequence axis; fastkv_amd/dist.py) ----
 * The reference has no multi-GPU path; these stages let P ranks reproduce the single-GPU result bit for bit:
 * scores are local per position, the only global quantities are the per-row softmax max / sum (two tiny all-reduces;
 * the sum is 2^-40 fixed point, i.e. exact and order-free) and the top-k threshold (one all-gather of candidates).
 *
 * A rank's logits row has `ncols` columns; column x holds global position pos0 + x.  [own_lo, own_hi) are the columns
 * the rank owns; the `kernel/2` columns on either side are halo positions owned by the neighbours (their K rows are
 * exchanged up front) that the pooling window needs.  Sp = row stride in elements (multiple of 8).
 */
typedef struct fastkv_sp_window {
    int32_t ncols, pos0, own_lo, own_hi, S_glob, Sp;
} fastkv_sp_window;

/*
 * Head sum of score rows (utils.py:127: `attn_cache.sum(dim=-2)`): t[b,j] = fp16( sum_r c[b,r,j] ), fp32 accumulation in
 * ascending r.  `c` is fp16 [B,R,n] contiguous.  The whole operator does this itself for the KV heads it holds; a
 * tensor-parallel caller (one rank = a slice of the KV heads) all-gathers the ranks' score rows and sums them here
 * (fastkv_amd/dist.py: tp_update_kv), then selects with fastkv_select_f16(append = window).
 */
int fastkv_head_sum_f16(const void *c, int64_t B, int64_t R, int64_t n, void *t_out, void *stream);

/*
 * Pooling of fp16 score rows as a stage of its own (the arithmetic of utils.py:105-108: stride 1, padding kernel/2, fp32 taps
 * in tap order, avg divides by `kernel` whatever the padding covers): out[row, j] for `rows` rows of n elements (row strides in
 * elements; in-place is NOT allowed).  The GemFilter rule needs it behind a head sum
 * (/root/reference/baselines/gemfilter/utils.py:25-38 `standard_dis_index`: last-query logits -> sum over heads -> avg_pool1d ->
 * topk = fastkv_sp_logits_f16 with window 1 -> fastkv_head_sum_f16 -> fastkv_pool_f16 -> fastkv_select_f16;
 * fastkv_amd/variants.py).
 */
int fastkv_pool_f16(const void *in, int64_t rows, int64_t in_row_stride, int64_t n, int32_t kernel, int32_t pooling, void *out,
                    int64_t out_row_stride, void *stream);

/* scratch for the calls below: fp32 query block (vector-ALU engine) + B*H*window floats */
size_t fastkv_sp_workspace_bytes(const fastkv_problem *p);
/* raw fp16 logits of the p->S keys in `k` against the window queries q_win [B,H,window,D] (utils.py:94 matmul),
 * written to logits[b,h,r, col_off + j], rows of stride Sp */
int fastkv_sp_logits_f16(const fastkv_problem *p, const void *q_win, const int64_t q_strides[4], const void *k,
                         const int64_t k_strides[4], void *logits, int64_t Sp, int64_t col_off, void *workspace,
                         size_t workspace_bytes, void *stream);
/* in place: scale by sqrt(D) + window mask on all columns (utils.py:94-101); local_max[2*B*H*window]: the maxima over the
 * owned columns, followed by one NaN flag (0/1) per row -- all-reduce the whole buffer with MAX */
int fastkv_sp_rowmax_f16(const fastkv_problem *p, void *logits, const fastkv_sp_window *w, float *local_max, void *stream);
/* local_sum[B*H*window] (int64, 2^-40 fixed point) = sum over owned columns of exp(x - global_max); all-reduce with SUM */
int fastkv_sp_rowsum_f16(const fastkv_problem *p, void *logits, const fastkv_sp_window *w, const float *global_max,
                         int64_t *local_sum, void *stream);
/* rewrites `logits` in place with the fp16 probabilities (utils.py:103), then the scores of the owned candidate
 * positions: c_out [B,Hkv,n_own] (utils.py:104-112), t_out [B,n_own] optional (utils.py:127);
 * n_own = min(own_hi, S_glob - window - pos0) - own_lo */
/* global_max: the MAX-reduced buffer of fastkv_sp_rowmax_f16 (maxima + NaN flags) */
int fastkv_sp_scores_f16(const fastkv_problem *p, void *logits, const fastkv_sp_window *w, const float *global_max,
                         const int64_t *global_sum, void *c_out, void *t_out, void *workspace, size_t workspace_bytes,
                         void *stream);

/*
 * Candidate exchange of the sequence-sharded selection (utils.py:113 / :127 across ranks): a global winner is always a
 * canonical local winner of the shard that owns it, so every rank contributes its local top-k as fixed-size records
 * {fp16 score bits << 32 | global position} (padded with {-inf, 0xffffffff}), ONE all-gather moves them (the records of
 * the per-head rows and of the TSP row share the buffer), and the final canonical top-k over the P*k candidates of a
 * row -- listed rank by rank, i.e. in ascending global position -- equals the single-device selection, ties included.
 *   pack    records_out[row, i] for i < k from the rank's ascending local winners idx_local [rows, kl] (kl <= k) of the
 *           fp16 rows `scores` (row stride in elements); pos0 = global position of the shard's first key
 *   unpack  gathered records (rank r's block at records + r*rank_stride, this row set at `offset` inside a block)
 *           -> fp16 rows scores_out [rows, P*k] for fastkv_select_f16
 *   pick    idx_out[row, j] = position of candidate sel[row, j] (j < kout), followed by `append` window positions
 *           n_glob, n_glob+1, ... (utils.py:128-129)
 */
int fastkv_sp_pack_f16(const void *scores, int64_t rows, int64_t row_stride, const int64_t *idx_local, int64_t kl, int64_t k,
                       int64_t pos0, int64_t *records_out, void *stream);
int fastkv_sp_unpack_f16(const int64_t *records, int64_t rank_stride, int64_t offset, int32_t P, int64_t rows, int64_t k,
                         void *scores_out, int64_t out_stride, void *stream);
int fastkv_sp_pick(const int64_t *records, int64_t rank_stride, int64_t offset, int64_t rows, int64_t k, const int64_t *sel,
                   int64_t kout, int64_t append, int64_t n_glob, int64_t *idx_out, void *stream);
/*
 * K/V rows of the global winners kv_idx [B,Hkv,capacity-window] (global positions) that lie in this rank's shard
 * [pos0, pos0 + S_r), written to their slots of k_out / v_out [B,Hkv,capacity,D] contiguous; slots owned by other ranks
 * are written as zeros (the ranks' outputs add up to utils.py:114-121's result).  window_owner != 0 (the last rank): the
 * shard's last `window` rows fill the last `window` slots.  k, v: the rank's [B,Hkv,S_r,D] slices, element strides.
 */
int fastkv_sp_compact_f16(int32_t B, int32_t Hkv, int32_t S_r, int32_t D, int32_t window, int32_t capacity, const void *k,
                          const int64_t k_strides[4], const void *v, const int64_t v_strides[4], const int64_t *kv_idx, int64_t pos0,
                          int32_t window_owner, void *k_out, void *v_out, void *stream);

/*
 * ---- Decode over the compressed cache (the reference appends with `past_key_value.update` -- a torch.cat
 * of the whole layer -- and attends with flash-attn: baselines/fastkv/llama_model.py:143-145, benchmark/e2e.py:72-93) ----
 * The layer's cache is a pre-sized slab [B,Hkv,rows,D] (element strides slab_strides = {batch, head, row}, the compaction
 * wrote its first `capacity` rows: fastkv_update_kv_strided_f16) whose current length is an int32 in DEVICE memory: shapes
 * are static, so a whole decode step can be captured in a HIP graph and replayed.
 *   fastkv_decode_append_f16     k_new / v_new [B,Hkv,1,D] (element strides {batch, head}) -> slab row *len_dev
 *   fastkv_decode_attention_f16  q [B,H,1,D] (strides {batch, head}) attends over rows 0 .. *len_dev (the appended row
 *                                included), G = H/Hkv in {1,2,4,8} query heads share a KV head; fp32 softmax; out fp16
 *                                [B,1,H*D] contiguous; then *len_dev += 1.  nsplit = slices per KV head (8-32: the layer is
 *                                spread over Hkv*nsplit workgroups); workspace: fastkv_decode_workspace_bytes.
 * Compared with PyTorch SDPA to fp16 tolerance, not bit for bit (a different summation order of the softmax).
 */
size_t fastkv_decode_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t nsplit);
int fastkv_decode_append_f16(int32_t B, int32_t Hkv, int32_t D, const void *k_new, const int64_t kn_strides[2], const void *v_new,
                             const int64_t vn_strides[2], void *kslab, void *vslab, const int64_t slab_strides[3], int32_t rows,
                             const int32_t *len_dev, void *stream);
int fastkv_decode_attention_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, const void *q, const int64_t q_strides[2],
                                const void *kslab, const void *vslab, const int64_t slab_strides[3], int32_t rows, int32_t *len_dev,
                                float scaling, int32_t nsplit, void *out, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The step's small operators as single launches (the stock modules issue 7 / 8 / 2 elementwise launches for them; at one
 * token per step those launches, not the bytes, are the cost).  Same arithmetic as the modules they stand in for:
 *   rmsnorm   LlamaRMSNorm / MistralRMSNorm: fp32 mean of squares, x * rsqrt(var + eps) -> fp16, * weight -> fp16
 *             x [rows, hidden] with a row stride, out [rows, hidden] contiguous
 *   rope      apply_rotary_pos_emb on q [B,H,1,D] and k [B,Hkv,1,D] IN PLACE (strides {batch, head}); cos / sin [B,1,D]
 *   silu_mul  act_fn(gate) * up of the MLP, n elements (multiple of 8)
 */
/* The attention part of a one-token step in ONE launch (csrc/decode_step.hip; RoPE + append + attention + merge: what
 * fastkv_decode_rope_f16 + fastkv_decode_append_f16 + fastkv_decode_attention_f16 do in four): q / k_new / v_new are the RAW
 * projections of the step, cos / sin [B,1,D] the rotary tables of its position; the rotated K row and the V row are written to
 * slab row *len_dev, the output is fp16 [B,1,H*D], *len_dev is advanced.  nsplit: slices per KV head, <= 0 = the library
 * chooses (one 128-row tile per slice).  `counters`: 1024 uint32 of device memory, zeroed ONCE by the caller -- word 0 (arrivals)
 * is zero again when a launch ends, word 1 is the launch epoch and only ever grows; `workspace`
 * (fastkv_decode_workspace_bytes; nsplit <= 0 there = room for every choice) holds the slices' records as {epoch token, value}
 * granules: ZERO it once, give it to ONE counters block for good, let nothing else write to it.  Graph-replayable.
 * B*Hkv <= 65535. */
int fastkv_decode_step_attention_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, const void *q, const int64_t q_strides[2],
                                     const void *k_new, const int64_t kn_strides[2], const void *v_new, const int64_t vn_strides[2],
                                     const void *cosv, const void *sinv, int64_t cs_batch_stride, void *kslab, void *vslab,
                                     const int64_t slab_strides[3], int32_t rows, int32_t *len_dev, float scaling, int32_t nsplit,
                                     void *out, void *workspace, size_t workspace_bytes, void *counters, void *stream);
int fastkv_decode_rmsnorm_f16(const void *x, int64_t rows, int64_t x_row_stride, int32_t hidden, const void *weight, float eps, void *out,
                              void *stream);
int fastkv_decode_rope_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, void *q, const int64_t q_strides[2], void *k,
                           const int64_t k_strides[2], const void *cosv, const void *sinv, int64_t cs_batch_stride, void *stream);
int fastkv_decode_silu_mul_f16(const void *gate, const void *up, int64_t n, void *out, void *stream);
/* Rotary tables of a one-token step in ONE launch: cos / sin [B,1,D] fp16 for position pos[b] (int64), the arithmetic of transformers'
 * LlamaRotaryEmbedding.forward (fp32 product inv_freq x position, cosf / sinf, x attention_scaling, -> fp16; the stock module: ten launches). */
int fastkv_decode_rotary_f16(int32_t B, int32_t D, const float *inv_freq, const int64_t *pos, float scaling, void *cosv, void *sinv, void *stream);
/* Greedy sampling of a step in ONE launch (the reference's loop: `out.logits[:, -1].argmax(-1)`, /root/reference/benchmark/e2e.py:72-93):
 * tok[b] = argmax of logits row b ([B rows of V fp16 values], row stride in elements; torch.argmax's rule: the first maximal value,
 * NaN counts as maximal); optional bookkeeping of a captured step: pos[b] += 1, log[*log_index] = tok[0] (while *log_index <
 * log_cap), *log_index += 1.  `scratch`: B + 1 uint64 of device memory, zeroed ONCE by the caller; the launch leaves them zero. */
int fastkv_decode_greedy_f16(int32_t B, int32_t V, const void *logits, int64_t row_stride, void *scratch, int64_t *tok, int64_t *pos,
                             int64_t *log, int64_t *log_index, int32_t log_cap, void *stream);

/* Weight-streaming GEMV of the decode step (csrc/gemv.hip): out[b, :] = W x[b, :] for ONE input row per batch element
 * (B = 1, 2 or 4; K % 512 == 0, B*K*2 <= 64 KiB - 256), fp16 in / fp32 accumulate / fp16 out.  Replaces the q/k/v/o and MLP
 * `nn.Linear` calls of the attention module and MLP during a one-token step
 * (/root/reference/baselines/fastkv/llama_model.py:118-120, :186; the stock LlamaMLP) -- one launch each for
 *   n_mats <= 3 matrices sharing the input, `rows[i]` rows of K contiguous fp16 each, outputs concatenated;
 *   norm_weight != NULL: the input is RMS-normalised first (LlamaRMSNorm arithmetic, `eps`);
 *   glu != 0 (n_mats == 2, rows[0] == rows[1]): out = fp16(fp16(silu(W0 x)) * fp16(W1 x)), rows[0] columns;
 *   residual != NULL: out = fp16(fp16(W x) + residual).
 * Row strides in elements.  Agrees with the stock modules to fp16 tolerance (another accumulation order), not bit for bit. */
int fastkv_decode_gemv_f16(int32_t B, int32_t K, const void *x, int64_t x_row_stride, const void *norm_weight, float eps,
                           int32_t n_mats, const void *const *weights, const int32_t *rows, int32_t glu, const void *residual,
                           int64_t res_row_stride, void *out, int64_t out_row_stride, void *stream);

/*
 * Test hook (not part of the operator): evaluates primitive `op` of the arithmetic contract element-wise
 * (0 det_exp(a), 1 a/b, 2 fp16 round trip, 3 fixed-point round trip (+raw in out64), 4 fma(a,b,out),
 * 5 fix_to_f32(bits(a)<<32|bits(b)), 6 a*b, 7 a+b, 8 scale_div(a, b), 9-11 packed twins of 0 / 3 / 8, 12 packed fp16 round trip) so tests can compare the GPU bit-for-bit with the CPU oracle.
 */
int fastkv_debug_contract(int op, const float *a, const float *b, float *out, uint64_t *out64, int n, void *stream);
/* Test hook of the "mfma16" contraction contract: the raw v_mfma_f32_32x32x16_f16, chained over dd / 16 chunks, on `ntiles` tiles
 * a [32][dd] x bt [32][dd] (fp16 bit patterns, row-major) on top of c [32][32] fp32 (NULL: +0) -> out [32][32]; the tests hold the
 * oracle's restatement of the instruction against it on the machine they run on. */
/* (test hook, host only) offsets of a problem's token-tagged hand-off areas in its workspace {fused score records, head-sum chain,
 * split-selection tables} and the workspace size: the areas lie at FIXED offsets whatever the problem's shape. */
int fastkv_debug_granule_areas(const fastkv_problem *p, size_t out[4]);
int fastkv_debug_mfma16(const void *a, const void *bt, const float *c, float *out, int ntiles, int dd, void *stream);
/* Test hook: `wgs` 256-thread workgroups that each hold `lds_bytes` of LDS for `usec` microseconds -- "another kernel is
 * holding compute units" for the residency tests of the in-launch hand-offs. */
int fastkv_debug_occupy(int wgs, int lds_bytes, int64_t usec, void *stream);
/* Test hook: `enable` != 0 makes every following fused scoring launch record, per workgroup of its linear launch order, the words
 * {HW_ID, XCC_ID, unit, span} (where it ran, what it worked on); `host` != NULL copies the first `n_words` (<= 4096) recorded words
 * out (call after synchronising).  The kernel gives two workgroups that share a compute unit adjacent spans of one unit; which two
 * share is the GPU's dispatch order, and this is how the tests check that order on the machine they run on. */
int fastkv_debug_fused_placement(int enable, unsigned int *host, size_t n_words);

/*
 * Measurement hooks (bench.py): when enabled, every kernel launch is bracketed by HIP events on its stream.
 * fastkv_profile_read adds launches / summed milliseconds per kernel id to the caller's arrays (length
 * fastkv_profile_kernels()) and clears the records; it synchronises on the recorded events.
 */
void fastkv_profile_enable(int on);
int fastkv_profile_kernels(void);
const char *fastkv_profile_kernel_name(int kernel_id);
int fastkv_profile_read(int64_t *counts, double *ms);

const char *fastkv_strerror(int code);
/* "fastkv-hip <version> gfx950" */
const char *fastkv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTKV_HIP_H */
