// Fused window scoring for the common geometries (window W == 8, G = 1-3 or 4, 8, 12, ... query heads per KV head, rows short
// enough that a wave keeps the logits of all its tiles in registers): ONE launch replaces score_logits + row_stats + score_finalize
// (/root/reference/baselines/fastkv/utils.py:93-112).  The 16 MiB of logits of the 32k shape never leave the registers
// and the window-row sums never leave LDS; what reaches memory is the score tensor c[b,g,j] (0.5 MiB) and its histogram.
//
// Same arithmetic, operation for operation, as the staged kernels (and as oracle/fastkv_oracle.c): fp32 fma chain on
// the matrix pipe -> fp16 -> true division by sqrt(D) -> fp16 -> window mask -> row max -> det_expf -> 2^-40 fixed-point
// sum -> p = fp16(e * (1/sum)) -> sequential fp32 sum over the window rows -> fp16 -> pool -> fp16 -> head sum -> fp16.
//
// A workgroup owns 4*PER consecutive tiles (64 keys, or 32 on short prompts: NB) of one (batch, kv head, virtual head of
// 4 query heads = the 32 rows of one fp32 MFMA block; more query heads per KV head are more virtual heads, chained in
// phase D).  The softmax needs two reductions over ALL
// workgroups of the head (row max, then row sum) and the pooling needs `kernel/2` neighbouring positions from the two
// adjacent workgroups.  All workgroups of the launch are resident at once (the host takes this path only when
// grid <= 2 workgroups per CU and the occupancy query agrees), so these are in-kernel hand-offs, not kernel boundaries:
//   * max / sum: every workgroup publishes its 32 partial values as 8-byte {token, value} granules, one write-through
//     (sc1) store each -- the data is the flag; one wave polls granule 0 of the head's workgroups, then everybody reads
//     the records with sc1 loads and checks the tags.  No fences, no atomics on shared words, order-free integer / max
//     combination.
//   * halo: the same granules, one-to-one between adjacent workgroups.
//   The token is a mix of the epoch in the workspace control block (fastkv_workspace_init clears the workspace once; the
//   compaction kernel of EVERY operator call advances the epoch, whichever scoring path ran: the split selection tags its
//   counters with the same token), never a launch argument that a graph replay would freeze.  The epoch advances by EPOCH_STRIDE per
//   operator call and the s-th scoring launch of a call (a batch that does not fit one launch) uses epoch + s, s < EPOCH_STRIDE: the
//   tokens of all launches are distinct values of one bijective mix, so a granule left by an earlier launch never matches.
//
// Which workgroup works for which (entry, unit, span): numbered unit by unit, span fastest -- a launch of up to one workgroup per
// compute unit is dispatched unit after unit and gets through a partly occupied GPU.  With more workgroups than compute units,
// workgroup p and workgroup p + #CUs sit on ONE compute unit; they are given adjacent spans of one unit, so that the hand-offs keep
// the two in step (see the kernel body; the measurement behind this is in docs/HISTORY.md).
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"
#include "mfma_tile.h"
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <type_traits>

namespace fk {

#ifdef FK_STAMP
__device__ unsigned long long g_fstamps[4096 * 48];
#define FKF_STAMP(slot) do { if (lane == 0) g_fstamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 4096 * 48 + (slot)] = wall_clock64(); } while (0)
#else
#define FKF_STAMP(slot) do { } while (0)
#endif

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// Test hook (fastkv_debug_fused_placement): where the workgroups of the last fused launch ran -- per workgroup of the launch's linear
// order {HW_ID, XCC_ID, unit, span}.  The pairing of workgroups that share a compute unit (see the kernel) rests on the dispatch order
// of the GPU: tests/test_hip_parity.py reads this back and checks it on the machine it runs on.
__device__ uint32_t g_placement[1024 * 4];
FKH_GLOBALS                                                      // (hunt builds only: csrc/fk_hunt.h)
constexpr int FUSED_PARTS = 8;      // 256 threads = 32 rows x 8 slices of the nblk partial records

// Hand-off records are 8-byte {token, 32-bit value} granules, written by ONE write-through (sc1) store each: the data is
// the flag (no drain, no separate flag word).  A consumer first lets one wave poll granule 0 of every producer (one 8-B
// load per lane and pass, sleeping in between: pollers share those words' memory channel with the producers' stores),
// then everybody reads the records and checks every tag (a record is one store instruction, so this almost never loops).
__device__ __forceinline__ uint64_t granule(uint32_t token, uint32_t value) { return ((uint64_t)token << 32) | value; }
// Returns false when the wait was given up (fk_device.h: SpinCtl) -- the caller leaves the kernel.
__device__ __forceinline__ bool wait_first_granules(const uint64_t *rec, int stride, int nblk, uint32_t token, int lane, const SpinCtl &sp)
{
    for (;;) {
        bool ok = true;
        for (int l = lane; l < nblk; l += 64)
            ok = ok && ((uint32_t)(__hip_atomic_load(rec + (size_t)l * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == token);
        if (__all(ok)) break;
        FKH_POLL_SLEEP();
        if (__builtin_amdgcn_readfirstlane((int)spin_failed(sp))) return false;     // a partner never arrived: give up, loudly (host flag)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler only: every load of handed-over bytes is an sc1 load
    return true;
}

// NS = streams of a workgroup: 1 (the product), or 2 (an experiment kept for reproduction, FASTKV_FUSED_STREAMS=2, head dim
// 128 only) -- the workgroup then works for TWO units (batch row, kv head, virtual head) at once, PER / NS tiles per wave
// each, and runs their phases interleaved: while the hand-off of one stream is in flight the workgroup computes for the
// other.  Schedule for NS = 2:  A0 pub | A1 pub | max0 B0 pub | max1 B1 pub | sum0 C0 pub | sum1 C1 pub | halo0 D0 | halo1 D1.
// Measured on MI355X (per-wave stamps, tools/stamp_fused.py; profiles/r02_fused_streams.md), 32k shape / 2k shape:
//   NS = 1  end 45.1 / 15.8 us;  NS = 2  end 56.3 / 29.0 us.  Bit-exact either way, and slower: the hand-offs are not what
// the waves wait for -- a hand-off completes 0.3-1.2 us behind the LAST arriving wave; what looks like waiting in a median
// wave is the other wave of its SIMD using the issue slots (vector-ALU work and fp32 MFMAs of a SIMD add up, in one wave or
// across two: tools/probes/probe_overlap.hip).  The kernel is issue bound (per SIMD: A 27, B 9, C 5, D 4 us), and two streams
// pay every hand-off's record sweep twice, with twice the producers per head.
// WPE = workgroups per compute unit (= waves per SIMD) the instantiation is built for: 2 (256 registers, 80 KiB of LDS per workgroup).
// (Round 5 built WPE = 3 -- 168 registers, <= 53 KiB, three workgroups of a compute unit in three different phases: bit-exact and 15 %
// slower, docs/HISTORY.md; its instantiations left the library in round 6.)
template <int D, int PER, int NB, int NS, bool F16, int WPE>
__device__ __forceinline__ bool score_fused_body(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                             const uint16_t *__restrict__ q, int64_t qs_b, int64_t qs_h, int64_t qs_s,
                                                             int H, int Hkv, int S, float sqrtD, float rsqrtD,
                                                             uint64_t *__restrict__ edges, uint64_t *__restrict__ pmax,
                                                             uint64_t *__restrict__ psum, uint32_t *__restrict__ ctrl,
                                                             uint32_t *__restrict__ zero_area, int zero_words, int ksize, int pooling,
                                                             uint16_t *__restrict__ c_out, int64_t c_row_stride,
                                                             int64_t *__restrict__ all_idx, uint16_t *__restrict__ all_keys,
                                                             int64_t all_key_stride, int VH, uint64_t *__restrict__ chain,
                                                             uint32_t *__restrict__ host_flag, uint64_t spin_ticks, const uint64_t *__restrict__ q_tab,
                                                             const uint64_t *__restrict__ k_tab, int HV, int b0, int BG_total, uint32_t sub, int ncu, uint32_t *__restrict__ place, uint64_t *__restrict__ cu_slots, uint64_t *__restrict__ done, int rolling, int start_delay, int parts, int tune)
{
    // NB = 32-key column blocks per wave tile: 2, or 1 on short prompts (twice the waves; a packed pair is then two query
    // rows of one column instead of two columns of one row).  NW = packed words per tile.  PS = tiles per wave and stream.
    static_assert(PER % NS == 0, "tiles split evenly over the streams");
    constexpr int W = 8, G = 4, NPH = D / DH, TK = 32 * NB, NW = 8 * NB, PS = PER / NS;
    // (the query block: fp32 values for the fma-chain contract, the fp16 rows as they are -- half the bytes -- for mfma16)
    constexpr int SLAB_BYTES = 4 * 64 * ROWB, AS_FLOATS = F16 ? (D / 4) * 64 : (D / 2) * 64;
    // Phase B's exponentials are needed again in phase C.  Up to two tiles per wave keep them in registers (64 of them); with
    // four tiles per wave (two 32k layers per launch) the other two tiles park theirs in LDS -- the K slabs and the query
    // operand are dead by then -- as one float4 per lane and word pair (conflict-free ds_write_b128 / ds_read_b128).  Round 2
    // recomputed them instead (128 more registers do not fit): 13.2 us of phase C against 3.9 us (per-wave stamps, r03).
    constexpr int E_REGS = PER < 4 ? PER : 2, E_LDS = PER - E_REGS;
    constexpr int E_BYTES = 4 * E_LDS * 64 * 2 * NW * 4;          // 64 KiB (32 KiB with 32-key tiles) per workgroup when PER == 4
    // the window-row sums of phases C / D are fp16 values: kept as fp16 bits when the parked exponentials need the room
    using tile_t = std::conditional_t<(E_LDS > 0), uint16_t, float>;
    constexpr int TWG_ = 4 * PS * TK, TW_ = TWG_ + 64;            // 31 halo columns on either side; rows stay 8-B / 16-B aligned
    constexpr int HIST_BYTES = HIST12 * 4;
    // phases C / D: [parked exponentials | histogram (inside the exponentials' area once they are consumed, behind them when
    // there are none)] [window-row-sum tiles of the streams]
    constexpr int CD_HEAD = E_BYTES >= HIST_BYTES ? E_BYTES : (E_BYTES + NS * HIST_BYTES);
    constexpr int CD_BYTES = CD_HEAD + NS * G * TW_ * (int)sizeof(tile_t);
    constexpr int A_BYTES = SLAB_BYTES + NS * AS_FLOATS * 4;
    // one block of LDS, carved twice: phase A = the waves' K slabs + the fp32 query operand of every stream; phases B-D (all
    // of phase A is over by then, on every stream) = parked exponentials, the window-row-sum tiles, the key histograms
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES > CD_BYTES ? A_BYTES : CD_BYTES];
    __shared__ float s_pf[NS][4][32];                          // per-wave row maxima of a stream (phase A -> publish)
    __shared__ uint64_t s_pu[NS][4][32];                       // per-wave fixed-point row sums (phase B -> publish)
    __shared__ uint32_t s_pb[NS][4][32];
    __shared__ float s_rf[FUSED_PARTS][32];                    // record reductions (one hand-off at a time)
    __shared__ uint64_t s_ru[FUSED_PARTS][32];
    __shared__ uint32_t s_rb[FUSED_PARTS][32];
    __shared__ float s_gm[NS][32], s_ri[NS][32];               // row maxima / reciprocal row sums of the streams' heads
    __shared__ uint32_t s_abort;
    FKH_SHARED
    const unsigned tix = threadIdx.x;
    const int lane = tix & 63, w = __builtin_amdgcn_readfirstlane(tix >> 6);
    // Entries 1 .. F-1 of a rolling launch (launch_score_fused; `rolling` = F, the entries the chip holds at a time) start late ON
    // PURPOSE, one K-streaming time apart: entries that begin together stay in step -- all stream K, then all do arithmetic -- and
    // gain nothing from sharing the chip.
    if (rolling && !(tune & 2) && start_delay > 0 && blockIdx.y > 0 && (int)blockIdx.y < rolling) { const uint64_t t_end = wall_clock64() + (uint64_t)start_delay * blockIdx.y; while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(16); }
    // A KV head with G = 4*VH query heads is worked on by VH "virtual heads" of 4 query heads each (own workgroups, own
    // softmax hand-offs, the same K rows); phase D chains them: virtual head vh continues the fp32 head sum that vh - 1
    // hands over per position (utils.py:112 adds the G pooled values in head order), the last one rounds and writes.
    // Unit = (kv head, virtual head) of this batch row; stream s of the workgroup works for unit hvp + s * UH / NS.
    const int UH = Hkv * VH, UP = UH / NS;
    // A batch that does not fit one launch is scored by several launches of `gridDim.y` entries each (launch_score_fused): `b` is the
    // entry of the WHOLE problem (addresses, score rows, histograms), `blockIdx.y` the entry inside this launch (hand-off records:
    // every launch uses the same record areas, told apart by `sub` mixed into the token)
    // Which (entry, unit, span) a workgroup works for.  With more workgroups than compute units the hardware puts workgroup p and
    // workgroup p + ncu on one unit (dispatch is round robin over the units while all of them have room; checked through HW_ID,
    // tools/probes/probe_lds_iso.hip): the two are given ADJACENT SPANS OF ONE UNIT (the host makes the span count even), so that the
    // hand-offs keep them in step -- neither can be a phase ahead of the other (maxima, sums and halo all need the partner's record).
    // Why that matters: a workgroup that runs its later phases beside one still in phase A (another entry's, held up by the NaN redo
    // or anything else) was measured to produce wrong row sums / window-row sums now and then (tools/repro_nan_mate.py; docs/HISTORY.md).
    // (a rolling launch may split a batch row's units over `parts` entries -- prompts so long that the units of one row do not fit
    // half of the chip: an entry is then UP / parts units of a row, the rows' parts follow each other in grid order)
    const int UPE = rolling ? UP / parts : UP;                   // units per entry of this launch
    const int nblk = gridDim.x / UPE;
    int hvp, blk, yb, ent = 0, part_base = 0;
    if (NS == 1 && !rolling && !FKH_OLD_NUMBERING) {             // (a rolling launch: entry = blockIdx.y, one workgroup per compute unit and entry)
        const int T = gridDim.x * gridDim.y, p = blockIdx.y * gridDim.x + blockIdx.x, P = T > ncu ? T - ncu : 0;
        const int l = p >= ncu ? 2 * (p - ncu) + 1 : (p < P ? 2 * p : 2 * P + (p - P));     // logical index: unit-major, span fastest
        const int unit = l / nblk;
        blk = l - unit * nblk;
        yb = unit / UH;
        hvp = unit - yb * UH;
    } else if (rolling) {
        // unit-major inside the entry, like the numbering above: on a chip that another kernel occupies in part the units of an entry
        // complete one after the other instead of all waiting for workgroups that have no place yet
        ent = blockIdx.y;
        yb = ent / parts;
        part_base = (ent - yb * parts) * UPE;
        hvp = blockIdx.x / nblk; blk = blockIdx.x - hvp * nblk;
        hvp += part_base;
    } else {
        hvp = blockIdx.x % UP; blk = blockIdx.x / UP; yb = blockIdx.y;
    }
    const int b = yb + b0;
    if (place && tix == 0) {
        uint32_t *pl = place + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4;
        pl[0] = __builtin_amdgcn_s_getreg(63492);                  // HW_ID (all 32 bits)
        pl[1] = __builtin_amdgcn_s_getreg(63508);                  // XCC_ID
        pl[2] = (uint32_t)(yb * UH + hvp);
        pl[3] = (uint32_t)blk;
    }
    const int BG = BG_total;
    int g_s[NS], vh_s[NS], bg_s[NS], bgv_s[NS];
    const uint16_t *kb_s[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int hv = hvp + s * UP;
        g_s[s] = hv / VH;
        vh_s[s] = hv - g_s[s] * VH;
        bg_s[s] = b * Hkv + g_s[s];
        // hand-off records are per virtual head of THIS launch (a rolling launch: of the 2 F entries that can be on the chip or
        // about to be; the entry's own token tells its records from those of entry - 2 F)
        bgv_s[s] = rolling ? (ent % (2 * rolling)) * UPE + (hv - part_base) : (yb * Hkv + g_s[s]) * VH + vh_s[s];
        kb_s[s] = (k_tab ? reinterpret_cast<const uint16_t *>(k_tab[b]) : k + b * ks_b) + (int64_t)g_s[s] * ks_h;   // (per-entry base: fk_host.h PtrTables)
    }
    const int n = S - W;
    const int nwt = (S + TK - 1) / TK;
    const int wave_id = blk * 4 + w;
    unsigned char *my = smem + w * (64 * ROWB);
    float *As = reinterpret_cast<float *>(smem + SLAB_BYTES);  // [NS][AS_FLOATS]
    const int n31 = lane & 31, hi = lane >> 5, sh = hi * 16;

    // tile t of the wave (0 .. PER-1): stream t / PS, tile wt(t) of that stream's row of tiles (contiguous keys per workgroup)
    auto tile_wt = [&](int t) { return wave_id * PS + (t % PS); };
    auto tile_valid = [&](int t) { return t < PER && tile_wt(t) < nwt; };
    auto tile_key0 = [&](int t) { const int wt = tile_wt(t); return (wt < nwt ? wt : 0) * TK; };

    // ---------------------------------------------------------------- A operands (see score_logits_mfma_kernel)
    constexpr int QV = 32 * (D / 8) / 256;
    uint4 qv[NS][QV];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int u = 0; u < QV; ++u) {
            const int item = u * 256 + tix, rowl = item / (D / 8), ch = item - rowl * (D / 8);
            const int i = rowl / W, r = rowl - i * W;
            // HV < 4 (models with 1-3 query heads per KV head; then VH == 1): the block's rows of the missing heads are zero
            // queries -- their logits, maxima and sums are computed and never used (phases C / D stop at HV heads)
            const int qh = g_s[s] * (HV < G ? HV : G * VH) + vh_s[s] * G + i;
            qv[s][u] = i < HV ? *reinterpret_cast<const uint4 *>((q_tab ? reinterpret_cast<const uint16_t *>(q_tab[b]) : q + b * qs_b) +
                                                                 (int64_t)qh * qs_h + (int64_t)(n + r) * qs_s + ch * 8)
                              : make_uint4(0u, 0u, 0u, 0u);
        }
    KStage sA, sB;
    k_fetch<NB>(sA, kb_s[0], ks_s, tile_key0(0), S, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    // (the control block is read only now: its round trip hides behind the query / first-tile loads issued above instead of
    // standing in front of them)
    // control block (fastkv_workspace_init): a missing initialisation must not turn into a silent wrong answer.  The token
    // of this launch is the epoch left by the previous one + 1 (the compaction kernel bumps it): never a launch argument,
    // which a graph replay would freeze; the granules in memory still carry earlier tokens (or whatever the allocation held).
    if (*reinterpret_cast<const uint64_t *>(ctrl) != CTRL_MAGIC) {
        // workspace never initialised (fastkv_workspace_init): no hand-off of this launch could be trusted.  Reported like an
        // abandoned wait -- the process-wide flag in pinned host memory, FASTKV_EABORTED at the next call -- and every workgroup
        // leaves at once (the condition is the same for all of them); nothing traps, the context stays usable.
        if (tix == 0) __hip_atomic_fetch_or(host_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (bit 1: not a given-up wait; the bits of several reporters combine, capi.hip take_abort_status)
        return false;
    }
    // (sub 0: the operator call's token, shared with the selection; the epoch advances by EPOCH_STRIDE per call, sub < EPOCH_STRIDE:
    // the tokens of all launches are distinct values of one bijective mix)
    const uint32_t token = handoff_token(ctrl[2] + sub + (rolling ? (uint32_t)ent : 0u));
    const SpinCtl sp = make_spin(ctrl, host_flag, token, spin_ticks);
    if (tix == 0) s_abort = 0;
    // Placement check.  The pairing above ASSUMES which workgroups share a compute unit; this notices when the assumption did not hold:
    // every workgroup swaps {token, its unit} into the slot of the compute unit it runs on (one returning atomic, asked for now, looked
    // at when the kernel ends).  Meeting this launch's token with ANOTHER unit in it = a workgroup of another unit ran on this compute
    // unit during the launch (beside this one, or before it where a launch was not resident all at once): counted in pinned word 3,
    // fastkv_placement_violations().  Never on an idle GPU (tests); expected beside a foreign kernel.
    // ---- Rolling launch: the record areas rotate over 2 F entries, so entry e writes where entry e - 2 F's records lay.  Grid-order
    // dispatch makes entry e - 2 F OLDER, not FINISHED: a unit of it that is slow -- the fma-chain contract's NaN redo takes a wave
    // through its tiles at a fraction of the speed; a foreign kernel may hold a compute unit -- can still be waiting for records of its
    // head when the entries behind it have come and gone (found by the first soak of the rolling launch under the fma chain, round 6: a
    // group of up to 20 entries with a NaN query row in one of them gave up its waits and was REPORTED, FASTKV_EABORTED).  So the
    // hand-over of an area is explicit: every workgroup leaves a {token, 1} "done" granule behind its last read of its head's records,
    // and a workgroup of entry e >= 2 F publishes its first record only when all workgroups of ITS unit in entry e - 2 F have left
    // theirs.  Those workgroups were dispatched long before this one and need nothing from it: the wait cannot deadlock.  The
    // granules are looked at once at the start (in the common case they are all there: the predecessor left the chip an entry's
    // lifetime ago) and polled before the first publish only if they were not.
    const bool area_handover = rolling && done && ent >= 2 * rolling;      // (done == nullptr: experiments builds only, FASTKV_FUSED_NO_HANDOVER=1)
    const uint32_t token_prev = area_handover ? handoff_token(ctrl[2] + sub + (uint32_t)(ent - 2 * rolling)) : 0u;
    uint64_t *done_unit = done + (size_t)bgv_s[0] * nblk;
    bool prev_done = true;
    if (area_handover && w == 0) {
        for (int l = lane; l < nblk; l += 64)
            prev_done = prev_done && ((uint32_t)(__hip_atomic_load(done_unit + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == token_prev);
        prev_done = __all(prev_done);
    }
    uint64_t cu_seen = 0;
    const uint32_t my_unit = (uint32_t)bgv_s[0] + 1u;
    if (cu_slots && tix == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(63492), xc = __builtin_amdgcn_s_getreg(63508);
        cu_seen = __hip_atomic_exchange(cu_slots + (((xc & 7u) << 8) | ((hw >> 8) & 0xffu)), granule(token, my_unit), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    }
    FKF_STAMP(0);
#ifdef FK_STAMP
    // (where the wave runs: slot 40 = XCC_ID << 32 | HW_ID, slot 41 = entry << 32 | unit << 16 | span -- tools/stamp_arrivals.py)
    if (lane == 0) {
        unsigned long long *row = g_fstamps + (size_t)(((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 4096) * 48;
        row[40] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
        row[41] = ((unsigned long long)ent << 32) | ((unsigned long long)(yb * UH + hvp) << 16) | (unsigned long long)blk;
    }
#endif
    FKH_DELAY_AT_START();
    // Zero what later stages accumulate into.  The key histogram of score row bg is filled in THIS launch (phase D) by the
    // workgroups of bg: they zero it themselves with write-through stores that are drained before their first hand-off record
    // is published, so passing the first hand-off implies the row is clean.  The TSP histograms and the arrival counters of
    // the selection are touched by later kernels only.
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (vh_s[s] == VH - 1) {                               // the group that fills the histogram in phase D
            uint32_t *hist_row = zero_area + (size_t)bg_s[s] * HIST12;
            for (int i = blk * 256 + (int)tix; i < HIST12; i += nblk * 256)
                __hip_atomic_store(hist_row + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (sub == 0) {
        const int first = BG * HIST12, rest = zero_words - first;
        const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = (rest + nwg - 1) / nwg;
        const int lo = wg * per, hi2 = min(lo + per, rest);
        for (int i = lo + (int)tix; i < hi2; i += 256) zero_area[first + i] = 0;
    }

#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int u = 0; u < QV; ++u) {
            const int item = u * 256 + tix, rowl = item / (D / 8), ch = item - rowl * (D / 8);
            if (F16) {
                // contract "mfma16": the query block stays fp16, as A-operand fragments of v_mfma_f32_32x32x16_f16 -- fragment
                // [chunk c][lane = 32 h + row] = dims 16 c + 8 h .. + 7 of the row, i.e. the row's 16-B piece ch = 2 c + h as it is
                reinterpret_cast<uint4 *>(As + s * AS_FLOATS)[((ch >> 1) * 64) + (ch & 1) * 32 + rowl] = qv[s][u];
                continue;
            }
            const uint32_t wds[4] = {qv[s][u].x, qv[s][u].y, qv[s][u].z, qv[s][u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[s * AS_FLOATS + (ch * 4 + e) * 64 + rowl] = h2f((uint16_t)(wds[e] & 0xffffu));
                As[s * AS_FLOATS + (ch * 4 + e) * 64 + 32 + rowl] = h2f((uint16_t)(wds[e] >> 16));
            }
        }
    __syncthreads();
    FKF_STAMP(16);
    if (NPH >= 2) k_fetch<NB>(sB, kb_s[0], ks_s, tile_key0(0), S, 1, lane);
    else if (tile_valid(1)) k_fetch<NB>(sB, kb_s[(1 / PS) % NS], ks_s, tile_key0(1), S, 0, lane);
    __builtin_amdgcn_sched_barrier(0);

    // lg[t][i]: a packed pair of scaled + masked fp16 logits.  NB == 2: query row m(i) = (i&3) + 8*(i>>2) + 4*hi at columns
    // key0 + n31 (low half) and key0 + 32 + n31 (high half).  NB == 1: rows m(i) (low) and m(i + 8) (high) at column
    // key0 + n31.
    static_assert(D == 64 || D == 128 || D == 256, "scale_div2_finite_h2 is exact for these divisors");
    const float rsqrtD_lo = recip_lo(sqrtD, rsqrtD);             // (what rsqrtD leaves of 1 / sqrtD: fk_device.h scale_div2_finite_h2)
    f16x8 pm0, pm1;
    perm_operands(lane, pm0, pm1);
    uint32_t lg[PER][NW];
    float ev[E_REGS][2][NW];
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ *ebuf = reinterpret_cast<f32x4_ *>(smem) + (size_t)w * E_LDS * (NW / 2) * 64 + lane;   // [tile][word pair][lane] of this wave
    uint32_t tile_nan = 0;                                       // bit t: tile t's contraction held a NaN (wave-uniform)

    // ================================================================ phase A of stream s: logits of its tiles, row maxima,
    // publication of the workgroup's 32 partial maxima as {token, value} granules
    auto phaseA = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        float mx[16];                                            // mx[r] = running maximum of row m(r) over this lane's columns
        uint32_t mx16[NW];                                       // the same for full tiles, as packed fp16 pairs
#pragma unroll
        for (int i = 0; i < 16; ++i) mx[i] = -INFINITY;
#pragma unroll
        for (int i = 0; i < NW; ++i) mx16[i] = 0xFC00FC00u;
#pragma unroll
        for (int lt = 0; lt < PS; ++lt) {
            constexpr int t0 = s * PS;
            const int t = t0 + lt;
#pragma unroll
            for (int i = 0; i < NW; ++i) lg[t][i] = 0;
            const int wt = tile_wt(t);
            if (wt < nwt) {
                const int key0 = wt * TK;
                f32x16 acc0, acc1;
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph) {
                    const bool useA = NPH >= 2 ? ((ph & 1) == 0) : ((t & 1) == 0);
                    // what to request next into the stage that is committed now: two phases ahead (the next phase pair of
                    // this tile, or of the wave's next tile -- possibly the first tile of the next stream), or two tiles
                    // ahead when a tile is a single phase
                    int nt, nph;
                    if (NPH == 1) { nt = t + 2; nph = 0; }
                    else if (ph + 2 < NPH) { nt = t; nph = ph + 2; }
                    else { nt = t + 1; nph = ph + 2 - NPH; }
                    const bool more = nt < PER && tile_wt(nt) < nwt;
                    const uint16_t *nkb = kb_s[(nt / PS) % NS];
                    const int nkey = tile_key0(nt);
                    FKH_MATE_NO_STAGING
                    if (useA) { k_commit<NB>(sA, lane, my); if (more) k_fetch<NB>(sA, nkb, ks_s, nkey, S, nph, lane); }
                    else { k_commit<NB>(sB, lane, my); if (more) k_fetch<NB>(sB, nkb, ks_s, nkey, S, nph, lane); }
                    __builtin_amdgcn_sched_barrier(0);
                    FKH_MATE_PHASE
                    if (F16) mfma_phase_f16<NB>(acc0, acc1, my, reinterpret_cast<const f16x8 *>(As + s * AS_FLOATS) + ph * 4 * 64 + lane, n31, hi);
                    else
                    mfma_phase_mx<NB>(acc0, acc1, my, As + s * AS_FLOATS + ph * (DH / 2) * 64 + lane, n31, hi, pm0, pm1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                FKF_STAMP(17 + 2 * (t % 2));
                FKH_DELAY_AFTER_TILE();
                if (F16) {
                    // the instruction follows IEEE on Inf / NaN operands, element by element, and so does the oracle's restatement of
                    // it: nothing to redo -- only remember that the tile holds a NaN (general softmax path later on)
                    bool bad = false;
#pragma unroll
                    for (int i = 0; i < 16; ++i) bad = bad || (NB == 2 ? __builtin_isunordered(acc0[i], acc1[i]) : acc0[i] != acc0[i]);
                    if (__any(bad)) tile_nan |= 1u << t;
                } else
                if (redo_tile_if_nan<NPH, NB>(acc0, acc1, kb_s[s], ks_s, key0, S, lane, my, As + s * AS_FLOATS + lane, n31, sh))
                    tile_nan |= 1u << t;
                const int jA = key0 + n31, jB = NB == 2 ? jA + 32 : jA;    // columns of the low / high half of a word
                // largest magnitude among the tile's fp32 results (NaN-free here unless redo_tile_if_nan said so): below 65520 no
                // logit rounds to an fp16 infinity, and the division needs no special cases (scale_div2_finite)
                float amax = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(acc0[i]), NB == 2 ? __builtin_fabsf(acc1[i]) : 0.0f));
                const bool tame = __all(amax < 65520.0f) && !((tile_nan >> t) & 1u);                            // wave-uniform
                if (key0 + TK <= n && tame) {
                    // tile entirely among the candidates, every logit finite: no window mask, every column counts; the running
                    // maxima stay packed fp16 pairs (v_pk_max_f16: maxNum, ignores NaN like fmaxf)
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        const uint32_t raw = f2h2(acc0[i], NB == 2 ? acc1[i] : acc0[(i + 8) & 15]);             // matmul -> fp16
                        // utils.py:94 (both contracts since round 6: the two-operation quotient on the fp16 pair as it stands)
                        const f32x2 scv = scale_div2_finite_h2(raw, rsqrtD, rsqrtD_lo);
                        const uint32_t wd = f2h2(scv.x, scv.y);
                        mx16[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h16x2, mx16[i]),
                                                                                          __builtin_bit_cast(h16x2, wd)));
                        lg[t][i] = wd;
                    }
                } else if (key0 + TK <= n) {
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        const uint32_t raw = f2h2(acc0[i], NB == 2 ? acc1[i] : acc0[(i + 8) & 15]);             // matmul -> fp16
                        const f32x2 scv = scale_div2((f32x2){h2f((uint16_t)(raw & 0xffffu)), h2f((uint16_t)(raw >> 16))}, sqrtD, rsqrtD);   // utils.py:94
                        const uint32_t wd = f2h2(scv.x, scv.y);
                        mx16[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h16x2, mx16[i]),
                                                                                          __builtin_bit_cast(h16x2, wd)));
                        lg[t][i] = wd;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        const int rw = (i & 3) + 4 * hi;                                     // window row of both rows of the word: m % W
                        const int rB = NB == 2 ? i : (i + 8) & 15;
                        // utils.py:94: matmul -> fp16, / sqrt(D) -> fp16 (both halves of the pair in one packed sequence)
                        const uint32_t raw = f2h2(acc0[i], NB == 2 ? acc1[i] : acc0[rB]);
                        const f32x2 scv = scale_div2((f32x2){h2f((uint16_t)(raw & 0xffffu)), h2f((uint16_t)(raw >> 16))}, sqrtD, rsqrtD);
                        const uint32_t sw = f2h2(scv.x, scv.y);
                        uint16_t s0 = (uint16_t)(sw & 0xffffu), s1 = (uint16_t)(sw >> 16);
                        if (jA >= n && (jA - n) > rw) s0 = f2h(h2f(s0) + (-65504.0f));       // utils.py:95-101
                        if (jB >= n && (jB - n) > rw) s1 = f2h(h2f(s1) + (-65504.0f));
                        if (jA < S) mx[i] = fmaxf(mx[i], h2f(s0));
                        if (jB < S) mx[rB] = fmaxf(mx[rB], h2f(s1));
                        lg[t][i] = (uint32_t)s0 | ((uint32_t)s1 << 16);
                    }
                }
                FKF_STAMP(18 + 2 * (t % 2));
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int rB = NB == 2 ? i : (i + 8) & 15;
            mx[i] = fmaxf(mx[i], h2f((uint16_t)(mx16[i] & 0xffffu)));
            mx[rB] = fmaxf(mx[rB], h2f((uint16_t)(mx16[i] >> 16)));
        }
        // row maxima: across the 32 lanes of a half wave, then across the 4 waves, then published
        {
            const float r = halfwave_reduce16(mx, lane, [](float a, float b2) { return fmaxf(a, b2); });
            const int i = halfwave_red_index(lane);
            if ((lane & 1) == 0) s_pf[s][w][(i & 3) + 8 * (i >> 2) + 4 * hi] = r;
        }
        FKF_STAMP(21);
        if (s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's histogram zeros have reached memory
        // (rolling launch: the area's previous user must have read its last record -- see `area_handover` above)
        if (area_handover && w == 0 && !prev_done && !wait_first_granules(done_unit, 1, nblk, token_prev, lane, sp)) s_abort = 1;
        __syncthreads();
        if (area_handover && s_abort) return;
        if (w == 0 && lane < 32)
            __hip_atomic_store(pmax + ((size_t)bgv_s[s] * nblk + blk) * 32 + lane,
                               granule(token, f32_bits(fmaxf(fmaxf(s_pf[s][0][lane], s_pf[s][1][lane]), fmaxf(s_pf[s][2][lane], s_pf[s][3][lane])))),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FKF_STAMP(22);
    };

    // ================================================================ hand-off 1 of stream s: row maxima of the head -> s_gm[s]
    // (returns false when the launch is abandoned: every thread of the workgroup leaves)
    auto read_max = [&](auto sc) -> bool {
        constexpr int s = decltype(sc)::value;
        FKH_BEFORE_FIRST_HANDOFF();
        const uint64_t *pm = pmax + (size_t)bgv_s[s] * nblk * 32;     // [nblk][32] granules: row maxima
        if (w == 0 && !wait_first_granules(pm, 32, nblk, token, lane, sp)) s_abort = 1;
        __syncthreads();
        if (s_abort) return false;
        FKF_STAMP(23);
        const int row = tix & 31, part = tix >> 5;
        for (;;) {
            float v = -INFINITY;
            bool ok = true;
            for (int base = 0; base < nblk; base += 8 * FUSED_PARTS) {    // 64 producers per round, 8 loads per thread in flight
                uint64_t rec[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bl = base + part + u * FUSED_PARTS;
                    rec[u] = __hip_atomic_load(pm + (bl < nblk ? bl : blk) * 32 + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ok = ok && (uint32_t)(rec[u] >> 32) == token;
                    v = fmaxf(v, bits_f32((uint32_t)rec[u]));
                }
            }
            s_rf[part][row] = v;
            if (FKH_SYNC_AND(ok)) break;
            if (tix == 0 && spin_failed(sp)) s_abort = 1;      // a record behind a current granule 0 is still old: rare
            __syncthreads();
            if (s_abort) return false;
        }
        FKF_STAMP(24);
        if (tix < 32) {
            float v = s_rf[0][tix];
#pragma unroll
            for (int u = 1; u < FUSED_PARTS; ++u) v = fmaxf(v, s_rf[u][tix]);
            s_gm[s][tix] = v;
        }
        __syncthreads();
        return true;
    };

    // ================================================================ phase B of stream s: e = exp(x - max), fixed-point row
    // sums, publication of the workgroup's 32 partial sums (two granules each)
    auto phaseB = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        float gm[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gm[i] = s_gm[s][(i & 3) + 8 * (i >> 2) + 4 * hi];
        // row sums in 2^-40 fixed point, per lane as two 32-bit words: ahi in units of 2^-20, alo (signed) in units of 2^-40
        // (fk_device.h exp_to_fix2_magic); `nmagic` = elements per accumulator whose raw magic-sum bits are still in there
        uint32_t ahi[16], alo[16], nanbits = 0;
        int nmagic = 0;
        bool gm_ok = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) { ahi[i] = 0; alo[i] = 0; gm_ok = gm_ok && __builtin_fabsf(gm[i]) < INFINITY; }
        const bool gm_finite = __all(gm_ok);                         // wave-uniform: +-inf or NaN row maxima send every tile the general way
        float ngm[16];                                               // (x - max as x + (-max): the mixed fma takes the addend as it is)
#pragma unroll
        for (int i = 0; i < 16; ++i) ngm[i] = -gm[i];
        f32x2 epair = {0.0f, 0.0f};                                  // the even word's exponentials, until the odd word's join them
        auto keep = [&](int t, int i, f32x2 e) {                     // t, i: compile-time after unrolling
            if (t < E_REGS) { ev[t < E_REGS ? t : 0][0][i] = e.x; ev[t < E_REGS ? t : 0][1][i] = e.y; }
            else if (i & 1) ebuf[((t - E_REGS) * (NW / 2) + (i >> 1)) * 64] = (f32x4_){epair.x, epair.y, e.x, e.y};
            else epair = e;
        };
#pragma unroll
        for (int lt = 0; lt < PS; ++lt) {
            constexpr int t0 = s * PS;
            const int t = t0 + lt;
            const int key0 = tile_wt(t) * TK;                        // >= S when the wave has no tile t: nothing is counted
            if (key0 + TK <= S && !((tile_nan >> t) & 1u) && gm_finite) {
                // Full tile of finite logits under finite row maxima (the only tiles a well-formed prompt has): every column
                // counts, x <= max and d = x - max is never NaN, so the range test of det_expf is a clamp and no NaN
                // bookkeeping is needed (det_expf2_clamped: bit-identical sums and probabilities).  Everything else -- a NaN
                // among the tile's results, a row whose maximum is +-inf (d = inf - inf), ragged tiles -- takes the general
                // path below, which selects and tracks NaNs per element.
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int rB = NB == 2 ? i : (i + 8) & 15;
                    const f32x2 dlt = (f32x2){mix_add_h0(lg[t][i], ngm[i]), mix_add_h1(lg[t][i], ngm[rB])};    // x - max, per half of the word
                    const f32x2 e = det_expf2_clamped(dlt);
                    keep(t, i, e);
                    uint32_t h0, l0, h1, l1;
                    exp_to_fix2_magic(e, h0, l0, h1, l1);            // raw bits: the magic constant comes off once, below
                    if (NB == 2) {
                        ahi[i] += h0 + h1;
                        alo[i] += l0 + l1;
                    } else {
                        ahi[i] += h0; alo[i] += l0;
                        ahi[rB] += h1; alo[rB] += l1;
                    }
                }
                nmagic += NB;                                        // every accumulator took NB elements of this tile
            } else {
                const bool inA = key0 + n31 < S, inB = NB == 2 ? key0 + 32 + n31 < S : inA;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int rB = NB == 2 ? i : (i + 8) & 15;
                    const f32x2 x = {h2f((uint16_t)(lg[t][i] & 0xffffu)), h2f((uint16_t)(lg[t][i] >> 16))};
                    const f32x2 e = det_expf2(x - (f32x2){gm[i], gm[rB]});
                    keep(t, i, e);
                    uint32_t h0, l0, h1, l1;
                    const bool nan0 = e.x != e.x, nan1 = e.y != e.y;
                    exp_to_fix2_magic((f32x2){nan0 ? 0.0f : e.x, nan1 ? 0.0f : e.y}, h0, l0, h1, l1);
                    h0 -= FIX_MAGIC_BITS; l0 -= FIX_MAGIC_BITS; h1 -= FIX_MAGIC_BITS; l1 -= FIX_MAGIC_BITS;
                    if (inA && nan0) nanbits |= 1u << i;
                    if (inB && nan1) nanbits |= 1u << rB;
                    ahi[i] += inA && !nan0 ? h0 : 0u;
                    alo[i] += inA && !nan0 ? l0 : 0u;
                    ahi[rB] += inB && !nan1 ? h1 : 0u;
                    alo[rB] += inB && !nan1 ? l1 : 0u;
                }
            }
        }
        FKF_STAMP(25);
        // per lane at most 2*PS <= 8 elements per row (hi <= 2^20, |lo| <= 2^19 each): the 32-bit lane sums are exact; they
        // are combined to the 2^-40 fixed-point value before the half-wave reduction
        {
            uint64_t tot[16];
            const uint32_t off = (uint32_t)nmagic * FIX_MAGIC_BITS;     // (wave-uniform; 32-bit wrap-around cancels)
#pragma unroll
            for (int i = 0; i < 16; ++i) tot[i] = ((uint64_t)(ahi[i] - off) << 20) + (uint64_t)(int64_t)(int32_t)(alo[i] - off);
            const uint64_t r = halfwave_reduce16(tot, lane, [](uint64_t a, uint64_t b2) { return a + b2; });
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) nanbits |= (uint32_t)__shfl_xor((int)nanbits, o, 64);
            const int i = halfwave_red_index(lane);
            if ((lane & 1) == 0) {
                const int m = (i & 3) + 8 * (i >> 2) + 4 * hi;
                s_pu[s][w][m] = r;
                s_pb[s][w][m] = (nanbits >> i) & 1u;
            }
        }
        FKF_STAMP(26);
        __syncthreads();
        if (w == 0) {
            const int row = lane >> 1;
            const uint32_t bad = s_pb[s][0][row] | s_pb[s][1][row] | s_pb[s][2][row] | s_pb[s][3][row];
            const uint64_t tot = bad ? FK_SUM_POISON : s_pu[s][0][row] + s_pu[s][1][row] + s_pu[s][2][row] + s_pu[s][3][row];
            __hip_atomic_store(psum + ((size_t)bgv_s[s] * nblk + blk) * 64 + lane, granule(token, (uint32_t)((lane & 1) ? tot >> 32 : tot)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ================================================================ hand-off 2 of stream s: row sums of the head -> s_ri[s]
    auto read_sum = [&](auto sc) -> bool {
        constexpr int s = decltype(sc)::value;
        const uint64_t *psu = psum + (size_t)bgv_s[s] * nblk * 64;    // [nblk][32][2] granules: row sums, low / high word
        if (w == 0 && !wait_first_granules(psu, 64, nblk, token, lane, sp)) s_abort = 1;
        __syncthreads();
        if (s_abort) return false;
        FKF_STAMP(27);
        const int row = tix & 31, part = tix >> 5;
        for (;;) {
            uint64_t s2 = 0;
            uint32_t bad = 0;
            bool ok = true;
            for (int base = 0; base < nblk; base += 8 * FUSED_PARTS) {
                uint64_t rl[8], rh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bl = base + part + u * FUSED_PARTS;
                    const uint64_t *g2 = psu + (bl < nblk ? bl : blk) * 64 + row * 2;
                    rl[u] = __hip_atomic_load(g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rh[u] = __hip_atomic_load(g2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ok = ok && (uint32_t)(rl[u] >> 32) == token && (uint32_t)(rh[u] >> 32) == token;
                    const uint64_t v = ((rh[u] & 0xffffffffull) << 32) | (rl[u] & 0xffffffffull);
                    if (base + part + u * FUSED_PARTS < nblk) { if (v == FK_SUM_POISON) bad = 1; else s2 += v; }
                }
            }
            s_ru[part][row] = s2;
            s_rb[part][row] = bad;
            if (FKH_SYNC_AND(ok)) break;
            if (tix == 0 && spin_failed(sp)) s_abort = 1;
            __syncthreads();
            if (s_abort) return false;
        }
        FKF_STAMP(28);
        if (tix < 32) {
            uint64_t sm = 0;
            uint32_t bad = 0;
#pragma unroll
            for (int u = 0; u < FUSED_PARTS; ++u) { sm += s_ru[u][tix]; bad |= s_rb[u][tix]; }
            s_ri[s][tix] = bad ? __builtin_nanf("") : 1.0f / fix_to_f32(sm);          // utils.py:103
        }
        __syncthreads();
        return true;
    };

    // ================================================================ phases C / D
    // Head i4 = i >> 2 of the group owns rows 8*i4 .. 8*i4+7; the lower half wave holds window rows 0-3 of every head,
    // the upper half rows 4-7.  The reference adds the 8 fp16 probabilities in ascending row order (fp32 accumulator,
    // utils.py:104): the lower half's partial sum crosses to the upper half, which finishes it and owns the result.
    // The workgroup owns the contiguous positions [lo, lo + TWG) of each stream's head: the window-row sums hs go to the
    // stream's LDS tile, column pad + local position; candidates past n hold the pooling pad value (utils.py:106,108).
    constexpr int TWG = TWG_, PADMAX = 31, TW = TW_;
    static_assert(E_LDS == 0 || NS == 1, "parked exponentials: one stream only");
    static_assert(CD_BYTES <= (int)sizeof(smem) && (sizeof(smem) + 6656) * WPE <= 160 * 1024, "WPE workgroups per CU share 160 KiB of LDS");
    const int pad = ksize / 2, lo = blk * TWG;
    const bool avg = pooling == FASTKV_POOL_AVG;
    const float padv = avg ? 0.0f : -INFINITY;
    const bool want_hist = all_idx == nullptr;
    auto tile_of = [&](int s) { return reinterpret_cast<tile_t(*)[TW]>(smem + CD_HEAD + (size_t)s * G * TW * sizeof(tile_t)); };
    auto hist_of = [&](int s) { return reinterpret_cast<uint32_t *>(smem + (E_BYTES >= HIST_BYTES ? 0 : E_BYTES) + (size_t)s * HIST_BYTES); };
    // a tile element: an fp16-valued number (or the pooling pad value: 0 / -inf), stored as float or as its fp16 bits
    // (every NaN enters the tile as THE positive quiet NaN: phase D's max pooling then is an integer maximum over the bit patterns --
    // the tile holds -inf (the pooling pad), values >= +0 and that one NaN, whose pattern is the largest, as the reference's max
    // pooling has it: any NaN in the window wins; the final scores store every NaN as 0x7e00 anyway, f2h_score)
    auto tl_put = [](tile_t &dst, float v) {
        const float c = (v != v) ? bits_f32(0x7fc00000u) : v;
        if constexpr (std::is_same<tile_t, float>::value) dst = c; else dst = f2h(c);
    };
    auto tl_get = [](const tile_t &src) -> float { if constexpr (std::is_same<tile_t, float>::value) return src; else return h2f(src); };

    auto phaseC = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        tile_t(*tile)[TW] = tile_of(s);
        uint32_t *s_hist = hist_of(s);
        float ri[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ri[i] = s_ri[s][(i & 3) + 8 * (i >> 2) + 4 * hi];
        FKF_STAMP(29);
#pragma unroll
        for (int lt = 0; lt < PS; ++lt) {
            constexpr int t0 = s * PS;
            const int t = t0 + lt;
            const int lp = (w * PS + lt) * TK + n31;                    // local position of the word's column (NB == 2: block 0; block 1 is 32 further)
#pragma unroll
            for (int i4 = 0; i4 < 2 * NB; ++i4) {
                // one packed pair per step: every operation below is per component what the scalar chain does.  NB == 2: the two
                // column blocks of head i4; NB == 1: heads i4 and i4 + 2 of the one column
                uint32_t phw[4];                                  // the step's four packed probability pairs
                f32x4_ parked[2];                                 // this step's four words of a tile whose exponentials wait in LDS
                if (t >= E_REGS) {
                    parked[0] = ebuf[((t - E_REGS) * (NW / 2) + 2 * i4) * 64];
                    parked[1] = ebuf[((t - E_REGS) * (NW / 2) + 2 * i4 + 1) * 64];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int wd = 4 * i4 + u, rB = NB == 2 ? wd : (wd + 8) & 15;
                    f32x2 ee;
                    if (t < E_REGS) ee = (f32x2){ev[t < E_REGS ? t : 0][0][wd], ev[t < E_REGS ? t : 0][1][wd]};
                    else ee = (u & 1) ? (f32x2){parked[u >> 1].z, parked[u >> 1].w} : (f32x2){parked[u >> 1].x, parked[u >> 1].y};
                    const f32x2 pr = ee * (f32x2){ri[wd], ri[rB]};
                    const uint32_t ph = f2h2(pr.x, pr.y);                                 // utils.py:103 -> fp16
                    phw[u] = ph;
                }
                f32x2 a = splat2(0.0f), c;
                // (the fp16 probabilities are added as they are: mixed fma, the same fp32 additions in the same order)
#pragma unroll
                for (int u = 0; u < 4; ++u) a = (f32x2){mix_add_h0(phw[u], a.x), mix_add_h1(phw[u], a.y)};
                c = (f32x2){__shfl_xor(a.x, 32, 64), __shfl_xor(a.y, 32, 64)};     // upper half: the lower half's sums of rows 0-3
#pragma unroll
                for (int u = 0; u < 4; ++u) c = (f32x2){mix_add_h0(phw[u], c.x), mix_add_h1(phw[u], c.y)};
                if (hi) {
                    if (NB == 2) {
                        if (i4 < HV) {
                            tl_put(tile[i4][PADMAX + lp], lo + lp < n ? h2f(f2h(c.x)) : padv);
                            tl_put(tile[i4][PADMAX + lp + 32], lo + lp + 32 < n ? h2f(f2h(c.y)) : padv);
                        }
                    } else {
                        if (i4 < HV) tl_put(tile[i4][PADMAX + lp], lo + lp < n ? h2f(f2h(c.x)) : padv);
                        if (((i4 + 2) & 3) < HV) tl_put(tile[(i4 + 2) & 3][PADMAX + lp], lo + lp < n ? h2f(f2h(c.y)) : padv);
                    }
                }
            }
        }
        FKF_STAMP(30);
        __syncthreads();
        // (the histogram may lie where the parked exponentials were: cleared only now that every wave has consumed them; phase D,
        // its first user, starts behind the barrier of read_halo)
        if (want_hist) for (int i = tix; i < HIST12; i += 256) s_hist[i] = 0;
        // halo: pooling reaches `pad` positions into the neighbouring workgroups of the head.  Every workgroup publishes its
        // first and last pad values per head as 8-byte {token, value} granules (one write-through store each: the data is the flag)
        uint64_t *eg = edges + ((size_t)bgv_s[s] * nblk + blk) * (2 * G * PADMAX);
        const int tt = tix, per_side = G * pad;
        if (tt < 2 * per_side && (tt % per_side) / pad < HV) {
            const int side = tt >= per_side, q2 = tt - side * per_side, i4 = q2 / pad, e = q2 - i4 * pad;
            const float v = tl_get(tile[i4][PADMAX + (side ? TWG - pad + e : e)]);
            __hip_atomic_store(eg + (side * G + i4) * PADMAX + e, ((uint64_t)token << 32) | f32_bits(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // the neighbours' halo granules of stream s into the pad columns of its tile; no neighbour = pooling padding
    auto read_halo = [&](auto sc) -> bool {
        constexpr int s = decltype(sc)::value;
        tile_t(*tile)[TW] = tile_of(s);
        const int tt = tix, per_side = G * pad;
        if (tt < 2 * per_side && (tt % per_side) / pad < HV) {
            const int side = tt >= per_side, q2 = tt - side * per_side, i4 = q2 / pad, e = q2 - i4 * pad;
            // side 0 of this thread = the LEFT halo of this workgroup = the right edge (side 1) of workgroup blk - 1, and vice versa
            const int nb = side ? blk + 1 : blk - 1;
            float hv = padv;
            if (nb >= 0 && nb < nblk) {
                const uint64_t *src = edges + ((size_t)bgv_s[s] * nblk + nb) * (2 * G * PADMAX) + ((side ? 0 : 1) * G + i4) * PADMAX + e;
                uint64_t x;
                while (((x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != token) {
                    __builtin_amdgcn_s_sleep(4);
                    if (spin_failed(sp)) { s_abort = 1; break; }
                }
                hv = bits_f32((uint32_t)x);
            }
            tl_put(tile[i4][side ? PADMAX + TWG + e : PADMAX - pad + e], hv);
        }
        __syncthreads();
        return !s_abort;
    };

    // phase D of stream s: pool, sum over the heads, scores + histogram (score_finalize of the three-kernel path,
    // utils.py:105-112)
    auto phaseD = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        tile_t(*tile)[TW] = tile_of(s);
        uint32_t *s_hist = hist_of(s);
        const int bg = bg_s[s], bgv = bgv_s[s], vh = vh_s[s];
        uint32_t *hist_row = zero_area + (size_t)bg * HIST12;
        uint64_t *chain_in = chain + ((size_t)(bgv - 1) * nblk + blk) * TWG;      // written by virtual head vh - 1 (vh > 0 only)
        uint64_t *chain_out = chain + ((size_t)bgv * nblk + blk) * TWG;
        const bool last_vh = vh == VH - 1;
#pragma unroll
        for (int u = 0; u * 256 < TWG; ++u) {
            const int lp = u * 256 + tix, j = lo + lp;
            const bool is_out = lp < TWG && j < n;
            float gsum = 0.0f;
            if (is_out && vh > 0) {                              // the head sum so far: {token, fp32 bits} granule of this position
                uint64_t x;
                while ((uint32_t)((x = __hip_atomic_load(chain_in + lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != token) {
                    __builtin_amdgcn_s_sleep(4);
                    if (spin_failed(sp)) break;                  // abandoned launch: the value is never used (the host reports the call)
                }
                gsum = bits_f32((uint32_t)x);
            }
            if (is_out) {
                float pv[G];
#pragma unroll
                for (int i4 = 0; i4 < G; ++i4) pv[i4] = i4 < HV ? pool_taps(tile[i4], PADMAX + lp, pad, ksize, avg) : 0.0f;
#pragma unroll
                for (int i4 = 0; i4 < G; ++i4) if (i4 < HV) gsum = gsum + h2f(f2h(pv[i4]));
                if (!last_vh) __hip_atomic_store(chain_out + lp, granule(token, f32_bits(gsum)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint16_t c16 = f2h_score(gsum);
            if (is_out && last_vh) {
                c_out[(size_t)bg * c_row_stride + j] = c16;
                if (all_idx) {                                   // capacity == S: identity selection + keys (see score_finalize)
                    all_idx[(size_t)bg * n + j] = (int64_t)j;
                    if (all_keys) all_keys[(size_t)bg * all_key_stride + j] = (uint16_t)mono16(c16);
                }
            }
            if (FKH_HIST && want_hist && last_vh) hist12_add(s_hist, mono16(c16) >> 4, is_out, lane);
        }
        FKF_STAMP(31);
        if (all_keys && last_vh && blk == 0 && (int)tix < (int)(all_key_stride - n)) all_keys[(size_t)bg * all_key_stride + n + tix] = 0;
        if (want_hist && last_vh) {
            __syncthreads();
            for (int i = tix; i < HIST12; i += 256) { const uint32_t v = s_hist[i]; if (v) atomicAdd(&hist_row[i], v); }
        }
    };

    // Phase D for the usual kernel sizes (3 / 5 / 7): FOUR consecutive positions per thread.  The 12-element window
    // [lp - 3, lp + 9) of a head's row serves all four (3 aligned 8-B / 16-B LDS reads instead of 28 scalar ones per position),
    // the four scores leave in one 8-byte store.  Per position the same operations in the same order as the scalar loop above.
    auto phaseD_vec = [&](auto sc, auto padc) {
        constexpr int s = decltype(sc)::value, PAD = decltype(padc)::value, KS = 2 * PAD + 1;
        static_assert(PAD >= 1 && PAD <= 3 && PADMAX % 4 == 3 && TW % 4 == 0, "aligned 12-element windows");
        tile_t(*tile)[TW] = tile_of(s);
        uint32_t *s_hist = hist_of(s);
        const int bg = bg_s[s], bgv = bgv_s[s], vh = vh_s[s];
        uint32_t *hist_row = zero_area + (size_t)bg * HIST12;
        uint64_t *chain_in = chain + ((size_t)(bgv - 1) * nblk + blk) * TWG;
        uint64_t *chain_out = chain + ((size_t)bgv * nblk + blk) * TWG;
        const bool last_vh = vh == VH - 1;
#pragma unroll
        for (int u = 0; u * 1024 < TWG; ++u) {
            const int lp = (u * 256 + (int)tix) * 4, j = lo + lp;
            const bool any = lp < TWG && j < n;
            float gs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (any && vh > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (j + e < n) {
                        uint64_t x;
                        while ((uint32_t)((x = __hip_atomic_load(chain_in + lp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != token) {
                            __builtin_amdgcn_s_sleep(4);
                            if (spin_failed(sp)) break;
                        }
                        gs[e] = bits_f32((uint32_t)x);
                    }
                }
            }
            if (any) {
                float pv[G][4];
#pragma unroll
                for (int i4 = 0; i4 < G; ++i4) {
                    if (i4 >= HV) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) pv[i4][e] = 0.0f;
                        continue;
                    }
                    const tile_t *row = &tile[i4][PADMAX - 3 + lp];
                    if (avg) {
                        float wv[12];
                        if constexpr (std::is_same<tile_t, float>::value) {
#pragma unroll
                            for (int v4 = 0; v4 < 3; ++v4) {
                                const float4 x = *reinterpret_cast<const float4 *>(row + 4 * v4);
                                wv[4 * v4] = x.x; wv[4 * v4 + 1] = x.y; wv[4 * v4 + 2] = x.z; wv[4 * v4 + 3] = x.w;
                            }
                        } else {
#pragma unroll
                            for (int v4 = 0; v4 < 3; ++v4) {
                                const uint2 x = *reinterpret_cast<const uint2 *>(row + 4 * v4);
                                wv[4 * v4] = h2f((uint16_t)(x.x & 0xffffu)); wv[4 * v4 + 1] = h2f((uint16_t)(x.x >> 16));
                                wv[4 * v4 + 2] = h2f((uint16_t)(x.y & 0xffffu)); wv[4 * v4 + 3] = h2f((uint16_t)(x.y >> 16));
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float p = 0.0f;
#pragma unroll
                            for (int o = 0; o < KS; ++o) p = p + wv[e + 3 - PAD + o];
                            pv[i4][e] = h2f(f2h(p / (float)KS));                              // utils.py:106 -> fp16
                        }
                    } else {
                        // max pooling as an INTEGER maximum over the elements' bit patterns (see tl_put: -inf < every value >= +0 < the one
                        // NaN, as signed integers too): v_max3_i32 over shared partial maxima, 10 instructions for the four positions of a
                        // head where the compare-and-select chain took 84; the result is an element of the tile: an fp16 value already
                        int32_t wi[12];
                        if constexpr (std::is_same<tile_t, float>::value) {
#pragma unroll
                            for (int v4 = 0; v4 < 3; ++v4) {
                                const float4 x = *reinterpret_cast<const float4 *>(row + 4 * v4);
                                wi[4 * v4] = (int32_t)f32_bits(x.x); wi[4 * v4 + 1] = (int32_t)f32_bits(x.y);
                                wi[4 * v4 + 2] = (int32_t)f32_bits(x.z); wi[4 * v4 + 3] = (int32_t)f32_bits(x.w);
                            }
                        } else {
#pragma unroll
                            for (int v4 = 0; v4 < 3; ++v4) {
                                const uint2 x = *reinterpret_cast<const uint2 *>(row + 4 * v4);
                                wi[4 * v4] = (int32_t)(int16_t)(x.x & 0xffffu); wi[4 * v4 + 1] = (int32_t)x.x >> 16;
                                wi[4 * v4 + 2] = (int32_t)(int16_t)(x.y & 0xffffu); wi[4 * v4 + 3] = (int32_t)x.y >> 16;
                            }
                        }
                        auto mx3 = [](int32_t a, int32_t b2, int32_t c2) { return max(max(a, b2), c2); };
                        int32_t r[4];
                        if constexpr (PAD == 3) {
                            const int32_t core = max(mx3(wi[3], wi[4], wi[5]), wi[6]);           // elements every position's window holds
                            r[0] = max(mx3(wi[0], wi[1], wi[2]), core);
                            r[1] = max(mx3(wi[1], wi[2], wi[7]), core);
                            r[2] = max(mx3(wi[2], wi[7], wi[8]), core);
                            r[3] = max(mx3(wi[7], wi[8], wi[9]), core);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                int32_t m = wi[e + 3 - PAD];
#pragma unroll
                                for (int o = 1; o < KS; ++o) m = max(m, wi[e + 3 - PAD + o]);
                                r[e] = m;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (std::is_same<tile_t, float>::value) pv[i4][e] = bits_f32((uint32_t)r[e]);
                            else pv[i4][e] = h2f((uint16_t)(r[e] & 0xffff));
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int i4 = 0; i4 < G; ++i4) if (i4 < HV) gs[e] = gs[e] + pv[i4][e];     // (pv: fp16 values, utils.py:106 / :108)
                    if (!last_vh && j + e < n)
                        __hip_atomic_store(chain_out + lp + e, granule(token, f32_bits(gs[e])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            uint16_t c16[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c16[e] = f2h_score(gs[e]);
            if (any && last_vh) {
                uint16_t *cp = c_out + (size_t)bg * c_row_stride + j;
                if (j + 3 < n && (reinterpret_cast<uintptr_t>(cp) & 7) == 0) {
                    *reinterpret_cast<uint2 *>(cp) = make_uint2((uint32_t)c16[0] | ((uint32_t)c16[1] << 16), (uint32_t)c16[2] | ((uint32_t)c16[3] << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (j + e < n) cp[e] = c16[e];
                }
                if (all_idx) {                                   // capacity == S: identity selection + keys (see score_finalize)
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (j + e < n) all_idx[(size_t)bg * n + j + e] = (int64_t)(j + e);
                    if (all_keys) {
                        uint16_t *kp = all_keys + (size_t)bg * all_key_stride + j;
                        if (j + 3 < n && (reinterpret_cast<uintptr_t>(kp) & 7) == 0) {
                            *reinterpret_cast<uint2 *>(kp) = make_uint2(mono16(c16[0]) | (mono16(c16[1]) << 16), mono16(c16[2]) | (mono16(c16[3]) << 16));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (j + e < n) kp[e] = (uint16_t)mono16(c16[e]);
                        }
                    }
                }
            }
            if (want_hist && last_vh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) hist12_add(s_hist, mono16(c16[e]) >> 4, any && j + e < n, lane);
            }
        }
        FKF_STAMP(31);
        if (all_keys && last_vh && blk == 0 && (int)tix < (int)(all_key_stride - n)) all_keys[(size_t)bg * all_key_stride + n + tix] = 0;
        if (want_hist && last_vh) {
            __syncthreads();
            for (int i = tix; i < HIST12; i += 256) { const uint32_t v = s_hist[i]; if (v) atomicAdd(&hist_row[i], v); }
        }
    };
    auto phaseD_any = [&](auto sc) {
        if (ksize == 7) phaseD_vec(sc, std::integral_constant<int, 3>{});
        else if (ksize == 5) phaseD_vec(sc, std::integral_constant<int, 2>{});
        else if (ksize == 3) phaseD_vec(sc, std::integral_constant<int, 1>{});
        else phaseD(sc);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NS - 1>;            // (== S0 when NS == 1: the second calls below are compiled out)
    // ---- Pacing of the rolling launch (round 6; tune bit 1, experiments builds; OFF: a measured negative).  The launch is built on the
    // idea that one entry runs phase A (matrix pipe / memory) while its partner on the compute unit runs phases B-D (vector work,
    // hand-off latency).  Nothing enforces that: per-entry stamps (profiles/r06_stamps_rolling_fmaf.log) show the two entries of the
    // chip only 8 us apart in a 66 us residency -- they contract together, do their vector work together and wait together.  With
    // this switch the offset is explicit, through records that exist anyway: a workgroup of entry e starts phase A only when its
    // unit's workgroups of entry e - 1 have PUBLISHED THEIR ROW MAXIMA (entry e - 1 was dispatched before entry e and waits for nothing
    // of it: no deadlock; its record area cannot change hands meanwhile -- entry e - 1 + 2 F publishes only behind entry e, by this very
    // rule).  Bit-exact (the rolling suite under both contracts), and 4 % SLOWER under the fma chain (276.5 against 265.4 us per eight
    // layers, alternating runs), no change under mfma16: phase A alone takes 26 us, not the 15.8 us of its matrix instructions -- its own
    // epilogue is vector work on the same lanes --, the partner's phase B then takes 23 us instead of 11 beside it, and the sum does not
    // move: the FP32 lanes are the bottleneck whoever uses them when (profiles/r06_pacing.log).
    if ((tune & 2) && rolling == 2 && ent >= 1) {
        const uint32_t token_before = handoff_token(ctrl[2] + sub + (uint32_t)(ent - 1));
        const uint64_t *pm_before = pmax + ((size_t)((ent - 1) % (2 * rolling)) * UPE + (size_t)(hvp - part_base)) * nblk * 32;
        if (w == 0 && !wait_first_granules(pm_before, 32, nblk, token_before, lane, sp)) s_abort = 1;
        __syncthreads();
        if (s_abort) return false;
    }
    FKF_STAMP(15);
    if (tune & 1) __builtin_amdgcn_s_setprio(3);
    phaseA(S0{});
    if (tune & 1) __builtin_amdgcn_s_setprio(0);
    if (tune & 4) __builtin_amdgcn_s_setprio(3);                 // (experiments builds: the phases behind phase A at raised priority instead)
    if (area_handover && s_abort) return false;                  // (uniform: read behind phase A's barrier)
    FKF_STAMP(1);
    if (NS == 2) phaseA(S1{});
    FKF_STAMP(2);
    if (!read_max(S0{})) return false;
    FKF_STAMP(3);
    phaseB(S0{});
    FKF_STAMP(4);
    if (NS == 2) { if (!read_max(S1{})) return false; FKF_STAMP(5); phaseB(S1{}); }
    FKF_STAMP(6);
    if (!read_sum(S0{})) return false;
    FKF_STAMP(7);
    phaseC(S0{});
    FKF_STAMP(8);
    if (NS == 2) { if (!read_sum(S1{})) return false; FKF_STAMP(9); phaseC(S1{}); }
    FKF_STAMP(10);
    if (!read_halo(S0{})) return false;
    FKF_STAMP(11);
    phaseD_any(S0{});
    FKF_STAMP(12);
    if (NS == 2) { if (!read_halo(S1{})) return false; FKF_STAMP(13); phaseD_any(S1{}); }
    FKF_STAMP(14);
    // this workgroup has read its last hand-off record (row maxima, row sums, halo: all behind barriers above): the area may go to
    // entry e + 2 F once every workgroup of the unit has said so
    if (rolling && done && tix == 0) __hip_atomic_store(done_unit + blk, granule(token, 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cu_slots && tix == 0 && (uint32_t)(cu_seen >> 32) == token && (uint32_t)cu_seen != my_unit)
        __hip_atomic_fetch_add(host_flag + (F16 ? 4 : 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (word 4: counted, never reported -- capi.hip)
    return true;
}

// The launch.  (Round 3 also built a variant that runs the sub-batches of a batch one after the other inside ONE launch -- a workgroup
// that has finished its unit of one sub-batch starts on the next at once, so that start-up and the spread of the waves at the end are
// paid once per launch: bit-exact, and no faster than one launch per sub-batch (0.822 vs 0.825 ms per step) once both used the same
// register allocation; the loop needs the thread index laundered through an opaque move or everything derived from it is hoisted
// and spilled.  Dropped: separate launches are simpler and cannot reuse a hand-off record too early.)
template <int D, int PER, int NB, int NS, bool F16, int WPE = 2>
__global__ void __launch_bounds__(256, WPE) score_fused_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                             const uint16_t *__restrict__ q, int64_t qs_b, int64_t qs_h, int64_t qs_s,
                                                             int H, int Hkv, int S, float sqrtD, float rsqrtD,
                                                             uint64_t *__restrict__ edges, uint64_t *__restrict__ pmax,
                                                             uint64_t *__restrict__ psum, uint32_t *__restrict__ ctrl,
                                                             uint32_t *__restrict__ zero_area, int zero_words, int ksize, int pooling,
                                                             uint16_t *__restrict__ c_out, int64_t c_row_stride,
                                                             int64_t *__restrict__ all_idx, uint16_t *__restrict__ all_keys,
                                                             int64_t all_key_stride, int VH, uint64_t *__restrict__ chain,
                                                             uint32_t *__restrict__ host_flag, uint64_t spin_ticks, const uint64_t *__restrict__ q_tab,
                                                             const uint64_t *__restrict__ k_tab, int HV, int b0, int BG_total, uint32_t sub, int ncu, uint32_t *__restrict__ place, uint64_t *__restrict__ cu_slots, uint64_t *__restrict__ done, int rolling, int start_delay, int parts, int tune)
{
    (void)score_fused_body<D, PER, NB, NS, F16, WPE>(k, ks_b, ks_h, ks_s, q, qs_b, qs_h, qs_s, H, Hkv, S, sqrtD, rsqrtD, edges, pmax, psum, ctrl, zero_area,
                                           zero_words, ksize, pooling, c_out, c_row_stride, all_idx, all_keys, all_key_stride, VH, chain, host_flag,
                                           spin_ticks, q_tab, k_tab, HV, b0, BG_total, sub, ncu, place, cu_slots, done, rolling, start_delay, parts, tune);
}

// ------------------------------------------------------------------------------------------ host side
static bool g_record_placement = false;
static uint32_t *placement_buffer()
{
    if (!g_record_placement) return nullptr;
    void *ptr = nullptr;
    return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_placement)) == hipSuccess ? static_cast<uint32_t *>(ptr) : nullptr;
}
static int device_cus()
{
    static const int cus = []() {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }();
    return cus;
}
template <int D, int PER, int NB, int NS, bool F16, int WPE> static bool fused_resident(int grid_wgs)
{
    struct Info { int wgs_per_cu, cus; };
    static const Info info = []() {                            // initialised once, thread-safe (C++11 static)
        Info r = {0, 0};
        int dev = 0, nb = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(score_fused_kernel<D, PER, NB, NS, F16, WPE>), 256, 0) ==
                hipSuccess) {
            r.wgs_per_cu = nb;
            r.cus = prop.multiProcessorCount;
        }
        return r;
    }();
    return info.wgs_per_cu >= 1 && grid_wgs <= (info.wgs_per_cu < WPE ? info.wgs_per_cu : WPE) * info.cus;
}

// One instantiation: residency check + launch.
template <int D, int PER, int NB, int NS, bool F16, int WPE = 2> struct FusedLaunch {
    static bool resident(int wgs) { return fused_resident<D, PER, NB, NS, F16, WPE>(wgs); }
    template <typename... Args> static void launch(dim3 grid, hipStream_t st, Args... args)
    {
        hipLaunchKernelGGL((score_fused_kernel<D, PER, NB, NS, F16, WPE>), grid, dim3(256), 0, st, args...);
    }
};

// Calls f(FusedLaunch<D, PER, NB, NS, F16>{}) for the runtime shape; false if that combination is not instantiated.
template <typename F> static bool fused_dispatch(int D, int per, int nb, int ns, bool f16, F &&f)
{
#define FK_CASE(DV, PV, NBV, NSV, FV) if (D == DV && per == PV && nb == NBV && ns == NSV && f16 == FV) { f(FusedLaunch<DV, PV, NBV, NSV, FV>{}); return true; }
#define FK_CASES_D(DV, FV)                                                                                          \
    FK_CASE(DV, 1, 1, 1, FV) FK_CASE(DV, 1, 2, 1, FV) FK_CASE(DV, 2, 2, 1, FV) FK_CASE(DV, 4, 2, 1, FV)      /* (32-key tiles only ever come one per wave) */
    FK_CASES_D(64, false) FK_CASES_D(128, false) FK_CASES_D(256, false)
    FK_CASES_D(64, true) FK_CASES_D(128, true) FK_CASES_D(256, true)
#ifdef FK_EXPERIMENTS
    // two streams (experiment, fp32 contract only): head dim 128, one tile per stream and wave
    FK_CASE(128, 2, 1, 2, false) FK_CASE(128, 2, 2, 2, false)
#endif
#undef FK_CASES_D
#undef FK_CASE
    return false;
}

// switches of the rolling launch (FASTKV_FUSED_TUNE, bits, experiments builds; default 1): 2 = pacing (an entry starts phase A when the
// entry before it has published its row maxima: see the kernel; round 6: measured 4 % SLOWER under the fma chain, no change under
// mfma16 -- profiles/r06_pacing.log -- and left off); 1 = the waves of an entry run phase A (K streaming) at raised
// issue priority (s_setprio 3), so that their loads / LDS commits are never queued behind the other entry's vector work -- measured in
// round 5 at -3 ... -5 us per eight-layer launch (203.7 / 206.0 / 204.5 -> 200.7 / 201.1 / 201.7 us, alternating runs on one box)
static int fused_tune()
{
    static const int t = exp_env_int("FASTKV_FUSED_TUNE", 1);
    return t;
}

// the rolling launch (launch_score_fused) is on unless FASTKV_FUSED_ROLLING=0 / fastkv_set_fused_rolling(0)
static std::atomic<int> &rolling_flag()
{
    static std::atomic<int> f{[]() { const char *e = getenv("FASTKV_FUSED_ROLLING"); return (e && e[0] == '0') ? 0 : 1; }()};
    return f;
}

// Returns true when the fused kernel was launched (and *err holds the launch status); false when the shape is not
// covered and the caller must take the three-kernel path.
struct FusedPlan { int NBV, PERT, NS, nblk, wgs; };
static bool fused_plan_for(const fastkv_problem &p, int UH, int ns_pref, int Bn, FusedPlan &pl, bool f16, int max_wgs = 2 * 256);

// The rolling launch's geometry (launch_score_fused): F entries on the chip at a time, each 1 / parts of a batch row's units at its
// smallest size (four tiles per wave).  false: no such plan (short prompts: many entries fit a regular launch anyway).
static int rolling_pert_min()
{
    static const int v = exp_env_int("FASTKV_FUSED_ROLLING_PERT", 4);     // measurement aid
    return v;
}
// the rolling launch under the fp32-fma-chain contract (round 6): FASTKV_FUSED_ROLLING_FMAF=0 restores round 5's launches of two
static bool rolling_fmaf()
{
    static const bool on = []() { const char *e = getenv("FASTKV_FUSED_ROLLING_FMAF"); return !(e && e[0] == '0'); }();
    return on;
}
static bool rolling_plan_for(const fastkv_problem &p, int UH, bool f16, int &F, int &parts, FusedPlan &ph)
{
    static const int f_cap = []() { const int v = exp_env_int("FASTKV_FUSED_ROLLING_F", 8); return v < 8 ? v : 8; }();
    F = 0; parts = 1;
    const int pert_min = rolling_pert_min();
    static const int parts_min = []() { const int v = exp_env_int("FASTKV_FUSED_ROLLING_PARTS", 1); return v >= 1 ? v : 1; }();   // measurement aid
    for (int pp = parts_min; pp <= 8 && pp <= UH && !F; pp *= 2) {
        if (UH % pp) break;
        for (int f = f_cap; f >= 2 && !F; --f)
            if (fused_plan_for(p, UH / pp, 1, 1, ph, f16, 512 / f) && ph.PERT >= pert_min && (size_t)2 * f * (UH / pp) * ph.nblk <= FUSED_MAX_WGS) { F = f; parts = pp; }
    }
    return F != 0 && fused_plan_for(p, UH / parts, 1, 1, ph, f16, 512 / F);
}

// Largest number of batch entries ONE fused scoring launch holds for this geometry (p.B is ignored), 0 = the geometry is off the
// fused path.  What a caller that batches entries itself wants to know up front (fastkv_amd.cluster.DeferredCompression).
int fused_entries_per_launch(const fastkv_problem &p)
{
    const bool disabled = no_wait_mode();
    static const int ns_pref_env = exp_env_int("FASTKV_FUSED_STREAMS", 1) == 2 ? 2 : 1;
    const int G = p.H / p.Hkv, VH = G < 4 ? 1 : G / 4;
    const bool f16 = resolve_engine(p) == ENGINE_MFMA16;
    const int ns_pref = f16 ? 1 : ns_pref_env;      // (two streams under the mfma16 contract: built and measured in round 4, 58 vs 42 us per one-layer launch: not instantiated)
    const bool engine_ok = f16 || (G < 4 ? (p.reserved & 3) != ENGINE_VALU : (p.reserved & 3) != ENGINE_VALU && G * p.window >= 24);
    if (disabled || !engine_ok || p.window != 8 || (G >= 4 && G % 4 != 0) || VH > 8 || p.kernel > 63 || !abort_flag_device()) return 0;
    FusedPlan pl;
    int best = 0;
    for (int cand = 1; cand <= 64; ++cand)
        if (fused_plan_for(p, p.Hkv * VH, ns_pref, cand, pl, f16)) best = cand;
    if (!best && (f16 || rolling_fmaf()) && VH == 1 && rolling_flag().load(std::memory_order_relaxed)) {
        // a row too long for a regular launch: the rolling launch takes it in parts, EPOCH_STRIDE entries (= parts) per call at most
        int F = 0, parts = 1;
        if (rolling_plan_for(p, p.Hkv * VH, f16, F, parts, pl) && parts > F) best = EPOCH_STRIDE / parts;
    }
    return best;
}

bool launch_score_fused(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, const void *k,
                        const int64_t *ks, uint16_t *c_out, int64_t c_row_stride, int64_t *all_idx, uint16_t *all_keys,
                        int64_t all_key_stride, char *ws, hipStream_t st, hipError_t *err, const PtrTables *pt)
{
    const bool disabled = no_wait_mode();
    // FASTKV_FUSED_STREAMS=2: the two-heads-per-workgroup experiment (measured slower, see the kernel's comment); default 1
    static const int ns_pref_env = exp_env_int("FASTKV_FUSED_STREAMS", 1) == 2 ? 2 : 1;
    // virtual heads of 4 query heads per KV head; a group of 1-3 query heads (MHA, G = 2 models, the per-query-head rule's
    // views) is ONE block whose missing heads are zero queries: a quarter to three quarters of the block's matrix work is
    // spent on rows nobody reads, and the launch still beats the three staged kernels with their logits round trip
    const int G = p.H / p.Hkv, VH = G < 4 ? 1 : G / 4, HV = G < 4 ? G : 4;
    const bool f16 = L.engine == ENGINE_MFMA16;                  // contract "mfma16": the fp16 matrix instruction itself (mfma_tile.h)
    const int ns_pref = f16 ? 1 : ns_pref_env;      // (two streams under the mfma16 contract: built and measured in round 4, 58 vs 42 us per one-layer launch: not instantiated)
    const bool engine_ok = f16 || (G < 4 ? (p.reserved & 3) != ENGINE_VALU : L.engine == ENGINE_MFMA);     // (tests force engines through `reserved`)
    if (disabled || !engine_ok || p.window != 8 || (G >= 4 && G % 4 != 0) || VH > 8 || p.kernel > 63) return false;
    uint32_t *host_flag = abort_flag_device();
    if (!host_flag) return false;                                // no way to report an abandoned launch: staged path
    const int UH = p.Hkv * VH;                                   // units (kv head, virtual head) per batch row
    // Entries per launch.  A launch holds `sb` entries of the batch: all of them when that fits the chip, else the largest divisor
    // of the batch that does (a batch of FOUR 32k layers = two launches of two, then ONE selection and ONE copy over all four: the
    // launches whose cost is latency are paid once per call, not once per launch that fits).
    FusedPlan pl;
    int sb = 0;
    for (int cand = p.B; cand >= 1; --cand)
        if (fused_plan_for(p, UH, ns_pref, cand, pl, f16)) { sb = cand; break; }
    // (sb == 0: not even one entry fits a regular launch -- prompts beyond 64k tokens at this geometry; the rolling launch below can
    // still take them, a part of a row's units at a time; without it: the staged path)
    if (sb && (p.B + sb - 1) / sb > EPOCH_STRIDE) return false;  // (more scoring launches than one call's share of hand-off tokens)
    const float sqrtD = (float)sqrt((double)p.D);
    uint64_t *pmax = reinterpret_cast<uint64_t *>(ws + L.off_fpart);
    uint64_t *psum = reinterpret_cast<uint64_t *>(ws + L.off_fpart + (size_t)FUSED_MAX_WGS * 32 * 8);
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(ws);
    const uint64_t spin_ticks = spin_limit_ticks();
    uint64_t *edges = reinterpret_cast<uint64_t *>(ws + L.off_fpart + (size_t)FUSED_MAX_WGS * 32 * 24);   // [unit span][2][4][31] halo granules
    uint32_t *zero = reinterpret_cast<uint32_t *>(ws + L.off_hist);
    // The placement check (fastkv_placement_violations) guards the compute-unit pairing of the REGULAR launches (two workgroups of one
    // head on a compute unit, kept in step by the hand-offs).  The hazard the pairing fences off needs the fp32-fma-chain contract's
    // matrix phase (K from LDS into fp16 MFMAs, those into fp32 MFMAs) beside packed-fp32 arithmetic (docs/HISTORY.md), and the library
    // holds no packed-fp32 instruction: launches of the fma chain count into the word the placement policy acts on (fail safe by
    // default, as since round 4); launches of the mfma16 contract issue neither half of the hazard and count into a word that is
    // only ever COUNTED (round 6, ADVICE r05: armed, never raised -- a harmless displacement on a shared GPU does not fail a prefill,
    // and `placement_violations: 0` in a bench line is a measurement again).  The ROLLING launch arms nothing under either contract:
    // its entries share compute units out of step by design, which the soaks of round 6 (profiles/r06_soak_*.log) stand for.
    uint64_t *cu_slots = reinterpret_cast<uint64_t *>(ws + L.off_fpart + (size_t)FUSED_MAX_WGS * (32 * 24 + 2 * 4 * 31 * 8));
    uint64_t *done = reinterpret_cast<uint64_t *>(ws + L.off_fpart + (size_t)FUSED_MAX_WGS * (32 * 24 + 2 * 4 * 31 * 8) + (size_t)FUSED_CU_SLOTS * 8);   // rolling launch: one "done" granule per (unit, span) of a record area
    if (exp_env_int("FASTKV_FUSED_NO_HANDOVER", 0)) done = nullptr;      // (experiments builds: the rolling launch as round 5 had it, to show what the hand-over is for)
    if (FKH_OLD_NUMBERING) cu_slots = nullptr;                   // (hunt builds: other units share compute units by design, every workgroup would report)
    uint64_t *chain = reinterpret_cast<uint64_t *>(ws + L.off_fchain);   // [unit span][positions of a span] head-sum granules (VH > 1)
    // Two fused launches should not overlap on a GPU (each needs ALL its workgroups resident; overlapping ones would wait
    // for each other until the spin limit and be reported as FASTKV_EABORTED).  Within this process the library sees to it:
    // when a launch comes on another stream than the previous one, an event recorded on the previous stream (now: behind
    // everything submitted there so far) makes the new stream wait.  The lock is held from the chaining to the enqueue of the
    // kernel, so that a second thread cannot slip its own launch in between.  Nothing is added while the caller stays on
    // one stream; inside a stream capture the chain is skipped (see include/fastkv_hip.h).
    // Why the event is recorded on the PREVIOUS stream at the switch and not behind every launch on its own stream: measured
    // (round 3, `FASTKV_CHAIN=0` A/B on one box) an event record behind each of the 17 fused launches of a step costs 4-8 % of
    // the step (1.03-1.08 ms against 0.996 ms): the marker packet keeps the next kernel from being dispatched back to back.
    // The previous stream's handle is therefore kept.  If the caller has destroyed that stream meanwhile, hipEventRecord
    // refuses the handle (the runtime checks every stream handle against its table of live streams and returns
    // hipErrorContextIsDestroyed / hipErrorInvalidHandle; nothing is dereferenced) and the chain is skipped: the destroyed
    // stream's work was complete, or at least ordered, when it was destroyed.  A NEW stream that re-uses the address merely
    // contributes a harmless extra dependency.
    static std::mutex mtx;
    static hipStream_t last_stream[16];
    static bool have_last[16];
    static hipEvent_t chain_ev[16];
    std::lock_guard<std::mutex> lk(mtx);
    {
        static const bool chain_off = exp_env_int("FASTKV_CHAIN", 1) == 0;   // measurement aid
        int dev = 0;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!chain_off && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && hipStreamIsCapturing(st, &cs) == hipSuccess &&
            cs == hipStreamCaptureStatusNone) {
            if (have_last[dev] && last_stream[dev] != st) {
                if (!chain_ev[dev]) (void)hipEventCreateWithFlags(&chain_ev[dev], hipEventDisableTiming);
                if (chain_ev[dev] && hipEventRecord(chain_ev[dev], last_stream[dev]) == hipSuccess)
                    (void)hipStreamWaitEvent(st, chain_ev[dev], 0);
                (void)hipGetLastError();                         // a destroyed previous stream is not this call's error
            }
            last_stream[dev] = st;
            have_last[dev] = true;
        }
    }
    *err = hipSuccess;
    // ---- The rolling launch (mfma16 contract, more entries than the chip holds at a time).  ONE launch holds ALL entries, each at its
    // smallest size (four tiles per wave: 256 workgroups for a 32k layer, one per compute unit): the chip has room for F of them (2 at
    // 32k, 4 at 16k, 8 at 8k), the hardware starts the workgroups of entry e + F as those of entry e leave, and entries 1 .. F-1 start
    // late on purpose (start_delay each) -- so at any time F entries share the chip OUT OF STEP: while one streams its K rows (memory
    // bound, the vector units idle) another runs its softmax / pooling phases (vector bound, the memory idle).  Measured: beside a partner that is a phase ahead a 32k layer takes 48 us, two of them in step (the pair launch) 59 us --
    // 24 us per layer against 29.5 (tools/exp_stagger.py, tools/trace_interleave.sh).  What it rests on: workgroups are dispatched in
    // grid order (entry e is complete on the chip before any workgroup of entry e + F starts: it takes the place of one of entry
    // e), every entry has its own token (epoch + sub + entry) and the record areas rotate over 2 F entries.  The compute-unit pairing
    // rule of the fp32 contract (a partner must be the adjacent span of the same head) does not apply: the matrix phase it protects
    // does not exist under this contract and the library holds no packed-fp32 instruction (tools/probes/README.md, erratum note);
    // the placement check is not armed for this launch.
    {
        const bool rolling_on = rolling_flag().load(std::memory_order_relaxed) != 0;
        // how far entry 1 stays behind: ~ the K streaming time of one entry alone (bytes of K at 5 TB/s), in 100 MHz ticks (6 / 12 / 19 /
        // 25 us behind at 32k: 200 / 199 / 203 / 208 us for eight layers)
        static const int stagger_env = exp_env_int("FASTKV_FUSED_STAGGER_US", -1);
        const int stagger_ticks = stagger_env >= 0 ? stagger_env * 100 : (int)((double)p.Hkv * p.S * p.D * 2.0 / 5.0e12 * 1.0e8);
        FusedPlan ph;
        // F = entries on the chip at a time: as many as fit with four tiles per wave (the smallest entry the kernel is built for).  32k
        // layers: 2 (256 workgroups each); 16k: 4 (128 each); 12k: 5 (96 each); 8k: 8.  The record areas rotate over 2 F entries.
        // Prompts whose rows do not fit half of the chip even then (beyond 32k tokens at 8 KV heads) are split: an entry is 1 / parts of a
        // row's units (4 heads at 64k, 2 at 128k, 1 at 256k), the parts of a row follow each other like entries do -- the units of a row
        // share nothing but the row's inputs.  (FASTKV_FUSED_ROLLING_F caps F: a measurement switch.)
        int F = 0, parts = 1;
        const bool have_plan = rolling_plan_for(p, UH, f16, F, parts, ph);
        const int entries = p.B * parts;
        // (VH == 1: the head-sum chain of models with more than four query heads per KV head has room for two entries' spans only)
        if (rolling_on && (f16 || rolling_fmaf()) && VH == 1 && have_plan && (sb * parts <= F || rolling_pert_min() < 4) && entries > F && entries <= EPOCH_STRIDE) {
            const dim3 grid(ph.nblk * (UH / parts), entries);
            ProfScope ps_(K_FUSED, st);
            fused_dispatch(p.D, ph.PERT, ph.NBV, 1, f16, [&](auto fl) {
                decltype(fl)::launch(grid, st, (const uint16_t *)k, ks[0], ks[1], ks[2], (const uint16_t *)q, qs[0], qs[1], qs[2], p.H, p.Hkv,
                                     p.S, sqrtD, 1.0f / sqrtD, edges, pmax, psum, ctrl, zero, L.zero_words, p.kernel, p.pooling, c_out,
                                     c_row_stride, all_idx, all_keys, all_key_stride, VH, chain, host_flag, spin_ticks, pt ? pt->q : nullptr,
                                     pt ? pt->k : nullptr, HV, 0, p.B * p.Hkv, 0u, device_cus(), (uint32_t *)nullptr /* (the placement record holds 1024 workgroups) */, (uint64_t *)nullptr, done, F, stagger_ticks / parts * 2 / F, parts, fused_tune());     // (the F starts spread over two K-streaming times of an entry: 32k 13 us apart, 16k 3.4, 8k 0.8 -- measured flat below that, worse above)
            });
            *err = hipGetLastError();
            return true;
        }
    }
    if (!sb) return false;                                       // not even one entry fits a regular launch: staged path
    uint32_t sub = 0;
    for (int b0 = 0; b0 < p.B && *err == hipSuccess; ++sub) {
        // launches of `sb` entries; the last one takes what is left (its own plan: fewer entries may mean fewer tiles per wave)
        int take = p.B - b0 < sb ? p.B - b0 : sb;
        while (take > 1 && !fused_plan_for(p, UH, ns_pref, take, pl, f16)) --take;
        if (!fused_plan_for(p, UH, ns_pref, take, pl, f16)) { *err = hipErrorLaunchFailure; break; }      // (cannot happen: one entry fits)
        const dim3 grid(pl.nblk * UH / pl.NS, take);
        ProfScope ps_(K_FUSED, st);
        fused_dispatch(p.D, pl.PERT, pl.NBV, pl.NS, f16, [&](auto fl) {
            decltype(fl)::launch(grid, st, (const uint16_t *)k, ks[0], ks[1], ks[2], (const uint16_t *)q, qs[0], qs[1], qs[2], p.H, p.Hkv,
                                 p.S, sqrtD, 1.0f / sqrtD, edges, pmax, psum, ctrl, zero, L.zero_words, p.kernel, p.pooling, c_out,
                                 c_row_stride, all_idx, all_keys, all_key_stride, VH, chain, host_flag, spin_ticks, pt ? pt->q : nullptr,
                                 pt ? pt->k : nullptr, HV, b0, p.B * p.Hkv, sub, device_cus(), placement_buffer(), cu_slots, done, 0, 0, 1, 0);
        });
        *err = hipGetLastError();
        b0 += take;
    }
    return true;
}

static bool fused_plan_for(const fastkv_problem &p, int UH, int ns_pref, int Bn, FusedPlan &pl, bool f16, int max_wgs)
{
    static const int cap_env = exp_env_int("FASTKV_FUSED_MAX_WGS", 0);     // measurement aid
    if (cap_env > 0 && cap_env < max_wgs && (int64_t)Bn * UH <= cap_env) max_wgs = cap_env;
    // 64-key wave tiles, or 32-key tiles when those would leave more than half of the chip's 1024 SIMDs without a wave
    pl.NBV = (int64_t)Bn * UH * ((p.S + 63) / 64) <= 512 ? 1 : 2;
    const int TKV = 32 * pl.NBV;
    const int nwt = (p.S + TKV - 1) / TKV;
    // one stream per workgroup: a workgroup owns 4 * PER consecutive tiles of ONE unit, at most 512 workgroups (2 per CU; 256 for a
    // launch that shares the chip with another one: the interleaved schedule)
    int nblk1 = max_wgs / (Bn * UH);
    if (nblk1 < 1) return false;
    if (nblk1 > (nwt + 3) / 4) nblk1 = (nwt + 3) / 4;
    const int per1 = (nwt + nblk1 * 4 - 1) / (nblk1 * 4);
    if (per1 > 4) return false;
    pl.PERT = per1 <= 1 ? 1 : per1 <= 2 ? 2 : 4;                // tiles per wave the kernel is instantiated for
    pl.NS = 1;
    // Two streams (experiment): the workgroup owns 4 * PER / 2 consecutive tiles of TWO units (hvp and hvp + UH/2).  Needs an
    // even number of units per batch row, rows that split into whole workgroup spans and both query operands in LDS.
    if (ns_pref == 2 && UH % 2 == 0 && p.D == 128 && pl.PERT <= 2) {
        const int span = 4 * 1 * TKV;                           // positions of a workgroup per stream (one tile per stream and wave)
        if (p.S % span == 0) { pl.NS = 2; pl.PERT = 2; }
    }
    const int PS = pl.PERT / pl.NS;
    pl.nblk = (nwt + PS * 4 - 1) / (PS * 4);                   // workgroups (spans) per unit
    // more workgroups than compute units: two share a unit, and the kernel pairs ADJACENT SPANS OF ONE UNIT there (see its comment):
    // an even span count (the extra span owns no keys: it only takes part in the hand-offs)
    const int cus = device_cus();
    if (cus <= 0) return false;
    if (pl.NS == 1 && (int64_t)Bn * UH * pl.nblk > cus && (pl.nblk & 1)) ++pl.nblk;
    const size_t vwgs = (size_t)Bn * UH * pl.nblk;              // hand-off records are per unit and span
    if (vwgs > FUSED_MAX_WGS) return false;
    pl.wgs = (int)(vwgs / pl.NS);
    if (pl.wgs > max_wgs) return false;
    bool resident = false;
    if (!fused_dispatch(p.D, pl.PERT, pl.NBV, pl.NS, f16, [&](auto fl) { resident = decltype(fl)::resident(pl.wgs); }) || !resident) return false;
    return true;
}

}  // namespace fk
extern "C" int fastkv_set_fused_rolling(int on)
{
    return fk::rolling_flag().exchange(on ? 1 : 0, std::memory_order_relaxed);
}
extern "C" int fastkv_debug_fused_placement(int enable, unsigned int *host, size_t n_words)
{
    fk::g_record_placement = enable != 0;
    if (!host || !n_words) return FASTKV_OK;
    if (n_words > 1024 * 4) return FASTKV_EINVAL;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::g_placement), n_words * sizeof(unsigned int)) == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}
#if defined(FK_HUNT) && defined(FK_DBG_DELAY)
extern "C" int fastkv_debug_set_delay(int ticks)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(fk::g_dbg_delay_ticks), &ticks, sizeof(int)) == hipSuccess ? 0 : -3;
}
#endif
namespace fk {
#ifdef FK_STAMP
}  // namespace fk
extern "C" int fastkv_debug_read_fused_stamps(unsigned long long *host, size_t n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::g_fstamps), n * sizeof(unsigned long long));
}
namespace fk {
#endif
}  // namespace fk
