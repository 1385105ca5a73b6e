// Canonical top-k selection on fp16 score rows: replaces `attn_cache.topk(k).indices`
// (/root/reference/baselines/fastkv/utils.py:113) and the TSP selection + window union + sort
// (utils.py:127-130).
//
// One 1024-thread workgroup per score row (rows are <= a few hundred KiB and L2 resident; the
// stage is latency bound by design).  Exact radix select on the 16-bit order-preserving key:
//   pass 1  4096-bin histogram of the high 12 key bits -> threshold prefix, #above.  In the fused operator the
//           histogram arrives ready-made: score_finalize / tsp_rowsum (1000+ workgroups) build it while they write
//           the scores, so this single workgroup does not have to
//   pass 2  16-bin histogram of the low 4 bits among the (few) elements carrying the threshold prefix -> k-th value
//   pass 3  ordered compaction (one block scan per 32768 positions): every element above the k-th
//           value plus the first `quota` elements equal to it in ascending position => canonical tie rule
// The winners leave this kernel in ascending position together with their 16-bit keys.  ORDER_SCORE (value
// descending, equal values in ascending position) is NOT produced by sorting here (a 4-pass radix sort on one CU
// cost 13 us): every winner's destination slot is obtained by comparison counting over the k keys
// (fk::rank_partial), in rank_scatter for the stand-alone API and inside the gather/compact kernel for the fused
// operator, which spreads the k*k comparisons over the whole chip (packed 16-bit compares, ~1.5 instr per pair).
// A row of up to 32768 positions is read ONCE (4 unconditional 16-B loads per thread, all in flight
// together) and stays in registers for the passes; longer rows are streamed per pass.
#include "fk_device.h"
#include <cstdlib>
#include "fk_host.h"
#include "prof.h"
#include "rank.h"

namespace fk {

constexpr int SEL_SUPER = SEL_THREADS * 8 * 4;

struct SelShared {
    uint32_t h12[HIST12];
    uint32_t h4[16 * 8];
    uint32_t wtot[4][16];
    uint32_t bcast[4];
};

// 4 x 8 consecutive keys per thread of super chunk `sc`: vector u covers positions sc*32768 + u*8192 + tid*8 ...
// Loads are unconditional (a predicated load costs a branch and serialises the batch): `vec` rows are 16-B aligned
// with a padded stride, so the vector that straddles n is readable; vectors wholly past n re-read the row's last one.
__device__ __forceinline__ void load_super(const uint16_t *row, int sc, int n, bool vec, uint4 (&raw)[4])
{
    const int jl = ((n - 1) >> 3) << 3;
    if (vec) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + threadIdx.x * 8;
            raw[u] = *reinterpret_cast<const uint4 *>(row + (j0 < n ? j0 : jl));
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + threadIdx.x * 8;
            uint32_t wds[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const int ja = j0 + e < n ? j0 + e : n - 1, jb = j0 + e + 1 < n ? j0 + e + 1 : n - 1;
                wds[e >> 1] = (uint32_t)row[ja] | ((uint32_t)row[jb] << 16);
            }
            raw[u] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
    }
}
__device__ __forceinline__ uint32_t key_of(const uint4 &v, int e)
{
    const uint32_t w = e < 2 ? v.x : e < 4 ? v.y : e < 6 ? v.z : v.w;
    return mono16((w >> ((e & 1) * 16)) & 0xffffu);
}

// one histogram update per lane; the lanes that agree with lane 0 are folded into a single LDS atomic
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool active, int lane)
{
    const uint32_t first = __builtin_amdgcn_readfirstlane(bin);
    const uint64_t same = __ballot(active && bin == first);
    if (active) {
        if (bin == first) {
            if (lane == __builtin_ctzll(same)) atomicAdd(&hist[first], (uint32_t)__builtin_popcountll(same));
        } else {
            atomicAdd(&hist[bin], 1u);
        }
    }
}

// Find the 12-bit bin holding the kk-th largest element: thread t owns bins 4095-4t .. 4092-4t (descending).
// out: bcast[0] = bin, bcast[1] = number of elements in bins above it.  Whole block, ends with a barrier.
__device__ __forceinline__ void find_bin12(SelShared &sh, uint32_t kk)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t c[4], loc = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = sh.h12[HIST12 - 1 - 4 * tid - u]; loc += c[u]; }
    uint32_t inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) sh.wtot[0][w] = inc;
    __syncthreads();
    uint32_t above = inc - loc;
    for (int u = 0; u < w; ++u) above += sh.wtot[0][u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (above < kk && kk <= above + c[u]) { sh.bcast[0] = HIST12 - 1 - 4 * tid - u; sh.bcast[1] = above; }
        above += c[u];
    }
    __syncthreads();
}

// Same for the 16-bin low-nibble histogram (8 privatised copies); executed by wave 0 only.
__device__ __forceinline__ void find_bin4(SelShared &sh, uint32_t kk, int lane)
{
    uint32_t c = 0;
    if (lane < 16) {
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) c += sh.h4[(15 - lane) * 8 + cp];      // lane l owns nibble 15-l (descending)
    }
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        uint32_t v = __shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    const uint32_t above = inc - c;
    if (lane < 16 && above < kk && kk <= above + c) { sh.bcast[2] = 15 - lane; sh.bcast[3] = above; }
}

__global__ void __launch_bounds__(SEL_THREADS) select_topk_kernel(const uint16_t *__restrict__ scores, int64_t row_stride, int n,
                                                                  int k, int append, int64_t *__restrict__ idx_out,
                                                                  int64_t idx_row_stride, uint16_t *__restrict__ key_out,
                                                                  int64_t key_row_stride, const uint32_t *__restrict__ hist12,
                                                                  int list_in_lds)
{
    __shared__ SelShared sh;
    // winners are collected in LDS (4-B position + 2-B key each) and written out coalesced at the end: scattered 8-B
    // global stores from the divergent emit loop were the most expensive phase of the kernel (8.8 of 19 us)
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint32_t *lidx = reinterpret_cast<uint32_t *>(dyn);
    uint16_t *lkey = reinterpret_cast<uint16_t *>(lidx + ((k + 7) & ~7));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rowi = blockIdx.x;
    const uint16_t *row = scores + (size_t)rowi * row_stride;
    int64_t *out = idx_out + (size_t)rowi * idx_row_stride;
    uint16_t *kout = key_out ? key_out + (size_t)rowi * key_row_stride : nullptr;
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0) && (row_stride >= (((int64_t)n + 7) & ~(int64_t)7));
    const int nsuper = (n + SEL_SUPER - 1) / SEL_SUPER;

    for (int a = tid; a < append; a += SEL_THREADS) out[k + a] = (int64_t)n + a;     // the window positions (utils.py:128-129)
    if (k == 0) return;

    uint4 raw[4];
    load_super(row, 0, n, vec, raw);                       // the whole row when n <= 32768
    // ---------------- pass 1: histogram of the high 12 key bits (ready-made in the fused operator)
    if (hist12) {
        const uint32_t *gh = hist12 + (size_t)rowi * HIST12;
#pragma unroll
        for (int u = 0; u < 4; ++u) sh.h12[u * SEL_THREADS + tid] = gh[u * SEL_THREADS + tid];
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) sh.h12[u * SEL_THREADS + tid] = 0;
        __syncthreads();
        for (int sc = 0; sc < nsuper; ++sc) {
            if (sc) load_super(row, sc, n, vec, raw);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + tid * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) hist_add(sh.h12, key_of(raw[u], e) >> 4, j0 + e < n, lane);
            }
        }
    }
    if (tid < 16 * 8) sh.h4[tid] = 0;
    __syncthreads();
    find_bin12(sh, (uint32_t)k);
    const uint32_t thr12 = sh.bcast[0], above12 = sh.bcast[1];
    // ---------------- pass 2: low nibble among the elements carrying the threshold prefix
    for (int sc = 0; sc < nsuper; ++sc) {
        if (nsuper > 1) load_super(row, sc, n, vec, raw);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + tid * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t key = key_of(raw[u], e);
                if ((j0 + e < n) && (key >> 4) == thr12) atomicAdd(&sh.h4[(key & 15u) * 8 + (lane & 7)], 1u);
            }
        }
    }
    __syncthreads();
    if (w == 0) find_bin4(sh, (uint32_t)k - above12, lane);
    __syncthreads();
    const uint32_t thr = (thr12 << 4) | sh.bcast[2];
    const uint32_t quota = (uint32_t)k - (above12 + sh.bcast[3]);   // elements equal to thr that are kept

    // ---------------- pass 3: ordered compaction, one block scan per super chunk
    uint32_t gt_base = 0, eq_base = 0;
    for (int sc = 0; sc < nsuper; ++sc) {
        if (nsuper > 1) load_super(row, sc, n, vec, raw);
        uint32_t pk[4], inc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + tid * 8;
            uint32_t cg = 0, ce = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t key = key_of(raw[u], e);
                const bool ok = j0 + e < n;
                cg += (ok && key > thr);
                ce += (ok && key == thr);
            }
            pk[u] = cg | (ce << 16);
            inc[u] = pk[u];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t v = __shfl_up((int)inc[u], o, 64);
                if (lane >= o) inc[u] += v;
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < 4; ++u) sh.wtot[u][w] = inc[u];
        }
        __syncthreads();
        // packed totals of everything before (u, w) in (u, w) order: lane l of every wave scans entry l of the 4x16 table
        const uint32_t tv = sh.wtot[lane >> 4][lane & 15];
        uint32_t tinc = tv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up((int)tinc, o, 64);
            if (lane >= o) tinc += v;
        }
        const uint32_t total = (uint32_t)__shfl((int)tinc, 63, 64);
        const uint32_t texc = tinc - tv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = sc * SEL_SUPER + u * (SEL_THREADS * 8) + tid * 8;
            const uint32_t ex = (uint32_t)__shfl((int)texc, u * 16 + w, 64) + inc[u] - pk[u];
            uint32_t gt_before = gt_base + (ex & 0xffffu), eq_before = eq_base + (ex >> 16);
            if (pk[u]) {                                      // most vectors hold no winner
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t key = key_of(raw[u], e);
                    if (j0 + e < n) {
                        int pos = -1;
                        if (key > thr) { pos = (int)(gt_before + min(eq_before, quota)); gt_before++; }
                        else if (key == thr) { if (eq_before < quota) pos = (int)(gt_before + eq_before); eq_before++; }
                        if (pos >= 0) {
                            if (list_in_lds) { lidx[pos] = (uint32_t)(j0 + e); lkey[pos] = (uint16_t)key; }
                            else { out[pos] = (int64_t)(j0 + e); if (kout) kout[pos] = (uint16_t)key; }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);               // keep the four vectors' code apart (register pressure)
        }
        gt_base += total & 0xffffu;
        eq_base += total >> 16;
        __syncthreads();
    }
    if (list_in_lds) {
        for (int i = tid; i < k; i += SEL_THREADS) out[i] = (int64_t)lidx[i];
        if (kout) for (int i = tid; i < k; i += SEL_THREADS) kout[i] = lkey[i];
    }
    // pad the key list to a multiple of 8 with the smallest key (rank_partial reads whole vectors)
    if (kout) for (int i = k + tid; i < ((k + 7) & ~7); i += SEL_THREADS) kout[i] = 0;
}

// ------------------------------------------------------------------------------------------ split select
// The fused operator's selection: a row is cut into chunks of 2048 positions, one 256-thread workgroup each
// (grid = chunks x rows: 128 workgroups for the 8 rows of the 32k configuration instead of 8).  The 12-bit histogram
// of the row is complete when the kernel starts (score_finalize / tsp_rowsum built it), so every workgroup derives the
// threshold prefix by itself.  What a workgroup cannot know alone -- the low nibble of the k-th value and how many
// winners precede its chunk -- goes through a small table in the workspace:
//   phase 1  count the chunk's keys above the threshold prefix and the nibble histogram of those carrying it; publish the
//            17 counters as 8-byte {token, value} granules (one write-through store each: the data is the flag -- no drain,
//            no arrival counter; round 1 drained the stores and bumped a counter: two more memory round trips)
//   phase 2  sweep the granules of all chunks of the row until every tag is this call's token (all workgroups of the
//            launch are co-resident: the host takes this path only for <= SPL_MAX_WGS workgroups; the wait is bounded,
//            fk_device.h SpinCtl), sum the tables -> k-th value, quota of ties, and the number of winners before this
//            chunk; ordered compaction of the chunk straight to the output lists.
constexpr int SPL_THREADS = 256, SPL_CHUNK = SPL_THREADS * 8;     // (SPL_LINE, SPL_MAX_WGS: fk_host.h, they size the workspace's table areas)
static_assert(SELTAB_ONE_BYTES == (size_t)SPL_MAX_WGS * SPL_LINE * sizeof(uint64_t), "one granule line per workgroup of a table launch");

struct SplShared {
    uint32_t wtot[4];
    uint32_t bcast[4];
    uint32_t h4[16];
    uint32_t cg12;
    uint32_t tot[17], pre[17];
};

// WAITFREE (rows of at most SPL_WF_CHUNKS chunks -- every row of a 32k prompt, every shard of a sequence-sharded one): a
// workgroup does not ask the other chunks for their counters, it counts them itself -- it reads the whole row (64 KiB at 32k,
// straight out of the cache the scoring launch wrote it through) with all loads of a thread in flight at once.  No table,
// no token, no wait on a partner: one memory round trip instead of publish + poll + sweep, and nothing that needs the
// workgroups of the launch to be resident together.
constexpr int SPL_WF_CHUNKS = 16;

// The extra workgroups of a selection launch that carries the TSP row sums (fk_host.h TspFold): workgroup (chunk, b) sums the Hkv score
// rows of batch row b over the chunk's 2048 positions -- fp32, head order, one rounding: tsp_rowsum_kernel's arithmetic (score.hip;
// utils.py:127) -- eight positions per thread, every head's 16-B vector requested before the first is used.
__device__ __forceinline__ void tsp_fold_role(const uint16_t *__restrict__ c, int64_t c_row_stride, int n, int chunk, int b, const TspFold &f)
{
    __shared__ uint32_t s_thist[HIST12];
    const int tid = threadIdx.x;
    for (int i = tid; i < HIST12; i += SPL_THREADS) s_thist[i] = 0;
    __syncthreads();
    const int j0 = chunk * SPL_CHUNK + tid * 8;
    const int jl = ((n - 1) >> 3) << 3;
    const uint16_t *base = c + (size_t)b * f.Hkv * c_row_stride + (j0 < n ? j0 : jl);
    float a[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int g0 = 0; g0 < f.Hkv; g0 += 8) {
        uint4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const uint4 *>(base + (size_t)(g0 + u < f.Hkv ? g0 + u : g0) * c_row_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (g0 + u < f.Hkv) {
                const uint32_t wds[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = a[e] + h2f((uint16_t)((wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu));
            }
        }
    }
    uint16_t t16[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t16[e] = f2h_score(a[e]);
    if (j0 < n) {
        uint16_t *tp = f.t_out + (size_t)b * f.t_row_stride + j0;
        if (j0 + 7 < n && (reinterpret_cast<uintptr_t>(tp) & 15) == 0) {
            *reinterpret_cast<uint4 *>(tp) = make_uint4((uint32_t)t16[0] | ((uint32_t)t16[1] << 16), (uint32_t)t16[2] | ((uint32_t)t16[3] << 16),
                                                        (uint32_t)t16[4] | ((uint32_t)t16[5] << 16), (uint32_t)t16[6] | ((uint32_t)t16[7] << 16));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (j0 + e < n) tp[e] = t16[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) hist12_add(s_thist, mono16(t16[e]) >> 4, j0 + e < n, tid & 63);
    __syncthreads();
    uint32_t *gh = f.thist + (size_t)b * HIST12;
    for (int i = tid; i < HIST12; i += SPL_THREADS) { const uint32_t v = s_thist[i]; if (v) atomicAdd(&gh[i], v); }
}

template <bool WAITFREE>
__global__ void __launch_bounds__(SPL_THREADS) select_split_kernel(const uint16_t *__restrict__ scores, int64_t row_stride, int n, int k,
                                                                   int append, int64_t *__restrict__ idx_out, int64_t idx_row_stride,
                                                                   uint16_t *__restrict__ key_out, int64_t key_row_stride,
                                                                   const uint32_t *__restrict__ hist12,
                                                                   uint64_t *__restrict__ table, uint32_t *__restrict__ ctrl,
                                                                   uint32_t *__restrict__ host_flag, uint64_t spin_ticks, int rows, TspFold fold)
{
    if ((int)blockIdx.y >= rows) {                               // (the workgroups behind the selection's: the TSP row sums, fk_host.h TspFold)
        tsp_fold_role(scores, row_stride, n, blockIdx.x, (int)blockIdx.y - rows, fold);
        return;
    }
    __shared__ SplShared sh;
    __shared__ uint32_t s_abort;
    // bounded wait (fk_device.h SpinCtl): same token as the scoring launch of this operator call (the epoch advances in the
    // compaction kernel, after this one)
    const uint32_t token = WAITFREE ? 0u : handoff_token(ctrl[2]);
    const SpinCtl sp = WAITFREE ? SpinCtl{} : make_spin(ctrl, host_flag, token, spin_ticks);
    if (threadIdx.x == 0) s_abort = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chunk = blockIdx.x, nchunks = gridDim.x, rowi = blockIdx.y;
    const uint16_t *row = scores + (size_t)rowi * row_stride;
    int64_t *out = idx_out + (size_t)rowi * idx_row_stride;
    uint16_t *kout = key_out ? key_out + (size_t)rowi * key_row_stride : nullptr;
    uint64_t *tab = table + ((size_t)rowi * nchunks) * SPL_LINE;

    // this thread's 8 keys (rows of the fused operator are 16-B aligned with a padded stride: the vector that straddles n
    // is readable; a vector wholly past n re-reads the row's last one and is ignored)
    const int j0 = chunk * SPL_CHUNK + tid * 8;
    const int jl = ((n - 1) >> 3) << 3;
    const uint4 raw = *reinterpret_cast<const uint4 *>(row + (j0 < n ? j0 : jl));
    uint4 oth[WAITFREE ? SPL_WF_CHUNKS : 1];                             // this thread's slice of every chunk of the row
    if (WAITFREE) {
#pragma unroll
        for (int cc = 0; cc < SPL_WF_CHUNKS; ++cc) {
            const int jc = cc * SPL_CHUNK + tid * 8;
            oth[cc] = (cc < nchunks && cc != chunk) ? *reinterpret_cast<const uint4 *>(row + (jc < n ? jc : jl)) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // 12-bit histogram: thread t owns bins 4095-16t .. 4080-16t (descending)
    uint32_t c[16];
    {
        const uint4 *gh = reinterpret_cast<const uint4 *>(hist12 + (size_t)rowi * HIST12 + (HIST12 - 16 - 16 * tid));
        const uint4 a0 = gh[0], a1 = gh[1], a2 = gh[2], a3 = gh[3];
        c[15] = a0.x; c[14] = a0.y; c[13] = a0.z; c[12] = a0.w; c[11] = a1.x; c[10] = a1.y; c[9] = a1.z; c[8] = a1.w;
        c[7] = a2.x; c[6] = a2.y; c[5] = a2.z; c[4] = a2.w; c[3] = a3.x; c[2] = a3.y; c[1] = a3.z; c[0] = a3.w;
    }
    if (tid < 16) sh.h4[tid] = 0;
    if (tid == 16) sh.cg12 = 0;
    if (tid >= 32 && tid < 32 + 17) { sh.tot[tid - 32] = 0; sh.pre[tid - 32] = 0; }
    if (chunk == 0) {
        for (int a = tid; a < append; a += SPL_THREADS) out[k + a] = (int64_t)n + a;          // the window positions (utils.py:128-129)
        if (kout) for (int i = k + tid; i < ((k + 7) & ~7); i += SPL_THREADS) kout[i] = 0;   // rank_partial reads whole vectors
    }
    {
        uint32_t loc = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) loc += c[u];
        uint32_t inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up((int)inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) sh.wtot[w] = inc;
        __syncthreads();
        uint32_t above = inc - loc;
        for (int u = 0; u < w; ++u) above += sh.wtot[u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (above < (uint32_t)k && (uint32_t)k <= above + c[u]) { sh.bcast[0] = HIST12 - 1 - 16 * tid - u; sh.bcast[1] = above; }
            above += c[u];
        }
        __syncthreads();
    }
    const uint32_t thr12 = sh.bcast[0], above12 = sh.bcast[1];
    // ---------------- phase 1
    uint32_t keys[8];
    {
        uint32_t cg = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            keys[e] = key_of(raw, e);
            const bool ok = j0 + e < n;
            cg += (ok && (keys[e] >> 4) > thr12);
            if (ok && (keys[e] >> 4) == thr12) atomicAdd(&sh.h4[keys[e] & 15u], 1u);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cg += __shfl_xor((int)cg, o, 64);
        if (lane == 0 && cg) atomicAdd(&sh.cg12, cg);
    }
    __syncthreads();
    if (WAITFREE) {
        // totals over the row and over the chunks before this one, counted here: own chunk from the counters above, the others
        // from their keys
        if (tid < 17) {
            const uint32_t mine = tid < 16 ? sh.h4[tid] : sh.cg12;
            sh.tot[tid] = mine;                                            // (pre starts at zero: the own chunk is not before itself)
        }
        __syncthreads();
        uint32_t cg_all = 0, cg_pre = 0;
#pragma unroll
        for (int cc = 0; cc < SPL_WF_CHUNKS; ++cc) {
            if (cc < nchunks && cc != chunk) {
                const int jc = cc * SPL_CHUNK + tid * 8;
                const bool before = cc < chunk;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t key = key_of(oth[cc], e);
                    const bool ok = jc + e < n;
                    const bool gt = ok && (key >> 4) > thr12;
                    cg_all += gt;
                    cg_pre += gt && before;
                    if (ok && (key >> 4) == thr12) {
                        atomicAdd(&sh.tot[key & 15u], 1u);
                        if (before) atomicAdd(&sh.pre[key & 15u], 1u);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cg_all += __shfl_xor((int)cg_all, o, 64); cg_pre += __shfl_xor((int)cg_pre, o, 64); }
        if (lane == 0) { if (cg_all) atomicAdd(&sh.tot[16], cg_all); if (cg_pre) atomicAdd(&sh.pre[16], cg_pre); }
    } else {
    if (w == 0 && lane < 17)
        __hip_atomic_store(&tab[chunk * SPL_LINE + lane], ((uint64_t)token << 32) | (lane < 16 ? sh.h4[lane] : sh.cg12), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    // ---------------- phase 2: totals over the row and over the chunks before this one
    const int ng = nchunks * 17;
    if (ng <= 4 * SPL_THREADS) {
        // (rows of up to 60 chunks -- 122,880 positions: every prompt of the reference's drivers) the granules a thread polls stay
        // in its registers: the pass that finds every tag current IS the read -- one memory round trip less than poll, then read
        for (;;) {
            bool ok = true;
            uint64_t gr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + u * SPL_THREADS, ic = i < ng ? i : 0, cc = ic / 17, f = ic - cc * 17;
                gr[u] = __hip_atomic_load(&tab[cc * SPL_LINE + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (uint32_t)(gr[u] >> 32) == token;
            }
            if (__syncthreads_and(ok)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid + u * SPL_THREADS, cc = i / 17, f = i - cc * 17;
                    const uint32_t v = (uint32_t)gr[u];
                    if (i < ng && v) {
                        atomicAdd(&sh.tot[f], v);
                        if (cc < chunk) atomicAdd(&sh.pre[f], v);
                    }
                }
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            if (tid == 0 && spin_failed(sp)) s_abort = 1;                // chunks of this row never arrived: give up, loudly (host flag)
            __syncthreads();
            if (s_abort) return;
        }
    } else {
    for (;;) {                                                           // until every chunk's granules carry this call's token
        bool ok = true;
        for (int i = tid; i < ng; i += SPL_THREADS) {
            const int cc = i / 17, f = i - cc * 17;
            ok = ok && (uint32_t)(__hip_atomic_load(&tab[cc * SPL_LINE + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == token;
        }
        if (__syncthreads_and(ok)) break;
        __builtin_amdgcn_s_sleep(4);
        if (tid == 0 && spin_failed(sp)) s_abort = 1;                    // chunks of this row never arrived: give up, loudly (host flag)
        __syncthreads();
        if (s_abort) return;
    }
    for (int i = tid; i < ng; i += SPL_THREADS) {
        const int cc = i / 17, f = i - cc * 17;
        const uint32_t v = (uint32_t)__hip_atomic_load(&tab[cc * SPL_LINE + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) {
            atomicAdd(&sh.tot[f], v);
            if (cc < chunk) atomicAdd(&sh.pre[f], v);
        }
    }
    }
    }
    __syncthreads();
    if (w == 0) {
        const uint32_t cnt = lane < 16 ? sh.tot[15 - lane] : 0;          // lane l owns nibble 15-l (descending)
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t v = __shfl_up((int)inc, o, 64);
            if (lane >= o) inc += v;
        }
        const uint32_t above = inc - cnt, kk = (uint32_t)k - above12;
        if (lane < 16 && above < kk && kk <= above + cnt) { sh.bcast[2] = 15 - lane; sh.bcast[3] = above; }
    }
    __syncthreads();
    const uint32_t thr4 = sh.bcast[2];
    const uint32_t thr = (thr12 << 4) | thr4;
    const uint32_t quota = (uint32_t)k - (above12 + sh.bcast[3]);       // elements equal to thr that are kept
    uint32_t gt_base = sh.pre[16], eq_base = sh.pre[thr4];
    for (uint32_t nb = thr4 + 1; nb < 16; ++nb) gt_base += sh.pre[nb];
    // ordered compaction of the chunk
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = j0 + e < n;
        cg += (ok && keys[e] > thr);
        ce += (ok && keys[e] == thr);
    }
    const uint32_t pk = cg | (ce << 16);
    uint32_t inc = pk;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    __syncthreads();                                                     // wtot is reused
    if (lane == 63) sh.wtot[w] = inc;
    __syncthreads();
    uint32_t ex = inc - pk;
    for (int u = 0; u < w; ++u) ex += sh.wtot[u];
    uint32_t gt_before = gt_base + (ex & 0xffffu), eq_before = eq_base + (ex >> 16);
    if (pk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (j0 + e < n) {
                int pos = -1;
                if (keys[e] > thr) { pos = (int)(gt_before + min(eq_before, quota)); gt_before++; }
                else if (keys[e] == thr) { if (eq_before < quota) pos = (int)(gt_before + eq_before); eq_before++; }
                // (pos < k always holds on a consistent histogram; an abandoned scoring launch leaves anything in it)
                if (pos >= 0 && pos < k) { out[pos] = (int64_t)(j0 + e); if (kout) kout[pos] = (uint16_t)keys[e]; }
            }
        }
    }
}

// Stand-alone ORDER_SCORE: out[r, rank(p)] = idx_asc[r, p].  grid (ceil(k/16), rows), 256 threads = 16 winners x 16 lanes.
__global__ void __launch_bounds__(256) rank_scatter_kernel(const int64_t *__restrict__ idx_asc, int64_t asc_row_stride,
                                                           const uint16_t *__restrict__ keys, int64_t key_row_stride, int k,
                                                           int64_t *__restrict__ out, int64_t out_row_stride)
{
    const int rowi = blockIdx.y, sub = threadIdx.x & 15;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    const uint16_t *kr = keys + (size_t)rowi * key_row_stride;
    const int pc = p < k ? p : k - 1;
    uint32_t r = rank_partial(kr, k, pc, kr[pc], sub, 16);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) r += __shfl_xor((int)r, o, 64);
    if (p < k && sub == 0) out[(size_t)rowi * out_row_stride + r] = idx_asc[(size_t)rowi * asc_row_stride + p];
}

bool select_takes_split(const uint16_t *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, const uint32_t *hist12)
{
    const int64_t nchunks = (n + SPL_CHUNK - 1) / SPL_CHUNK;
    const bool vec = ((reinterpret_cast<uintptr_t>(scores) & 15) == 0) && (row_stride % 8 == 0) && (row_stride >= ((n + 7) & ~(int64_t)7));
    if (!(hist12 && k > 0 && nchunks >= 2 && rows >= 1 && rows <= 65000 && vec)) return false;
    const bool table_ok = abort_flag_device() && rows * nchunks <= SPL_MAX_WGS && !no_wait_mode();
    return table_ok || nchunks <= SPL_WF_CHUNKS;
}

hipError_t launch_select(const uint16_t *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, int append,
                         int64_t *idx_out, int64_t idx_row_stride, uint16_t *key_out, int64_t key_row_stride,
                         const uint32_t *hist12, uint32_t *arrive, uint32_t *table, hipStream_t st, uint32_t *ctrl, const TspFold *fold)
{
    if (rows == 0) return hipSuccess;
    const TspFold nofold = {0, 0, nullptr, 0, nullptr};
    const TspFold fd = fold ? *fold : nofold;
    const unsigned extra = fold ? (unsigned)fold->B : 0u;        // (the caller asked select_takes_split first: a fold never reaches select_topk)
    uint32_t *host_flag = ctrl ? abort_flag_device() : nullptr;
    const int64_t nchunks = (n + SPL_CHUNK - 1) / SPL_CHUNK;
    const bool vec = ((reinterpret_cast<uintptr_t>(scores) & 15) == 0) && (row_stride % 8 == 0) && (row_stride >= ((n + 7) & ~(int64_t)7));
    (void)arrive;                                                        // (round 1's arrival counters: the granules carry the signal now)
    // FASTKV_FUSED=0 is the library's "no in-launch waits" mode (processes sharing a GPU, include/fastkv_hip.h): the selection
    // then counts the row in every chunk instead of exchanging counters -- measured 18.4 us against 11.0 us per launch at 32k
    // (64 KiB of row per workgroup instead of one hand-off), which is why it is not the default
    const bool no_waits = no_wait_mode();
    const bool table_ok = hist12 && table && host_flag && k > 0 && nchunks >= 2 && rows * nchunks <= SPL_MAX_WGS && rows <= 65535 && vec;
    if (table_ok && !no_waits) {
        ProfScope ps_(K_SELECT_SPLIT, st);
        hipLaunchKernelGGL(select_split_kernel<false>, dim3((unsigned)nchunks, (unsigned)rows + extra), dim3(SPL_THREADS), 0, st, scores, row_stride,
                           (int)n, (int)k, append, idx_out, idx_row_stride, key_out, key_row_stride, hist12,
                           reinterpret_cast<uint64_t *>(table), ctrl, host_flag, spin_limit_ticks(), (int)rows, fd);
        return hipGetLastError();
    }
    if (hist12 && k > 0 && nchunks >= 2 && nchunks <= SPL_WF_CHUNKS && rows <= 65535 && vec) {
        // every chunk counts the row for itself: no table, no residency requirement, any number of rows (also what rows x
        // chunks beyond SPL_MAX_WGS get instead of one workgroup per row)
        ProfScope ps_(K_SELECT_SPLIT, st);
        hipLaunchKernelGGL(select_split_kernel<true>, dim3((unsigned)nchunks, (unsigned)rows + extra), dim3(SPL_THREADS), 0, st, scores, row_stride,
                           (int)n, (int)k, append, idx_out, idx_row_stride, key_out, key_row_stride, hist12,
                           reinterpret_cast<uint64_t *>(table), ctrl, host_flag, (uint64_t)0, (int)rows, fd);
        return hipGetLastError();
    }
    // (a caller that asked for the TSP row sums to ride along checked select_takes_split first; if the library's mode changed in between
    // -- another thread's fail-safe switch to the no-wait kernels on a row too long for the wait-free selection -- the row sums would
    // silently not be computed: refuse instead, loudly)
    if (fold) return hipErrorInvalidValue;
    const int64_t kal = (k + 7) & ~(int64_t)7;
    const int list_in_lds = kal <= 16384 ? 1 : 0;                      // 96 KiB of dynamic LDS at most
    const size_t dyn = list_in_lds ? (size_t)kal * 6 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(select_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  16384 * 6);
        attr_set = true;
    }
    ProfScope ps_(K_SELECT, st);
    hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)rows), dim3(SEL_THREADS), dyn, st, scores, row_stride, (int)n, (int)k,
                       append, idx_out, idx_row_stride, key_out, key_row_stride, hist12, list_in_lds);
    return hipGetLastError();
}

hipError_t launch_rank_scatter(const int64_t *idx_asc, int64_t asc_row_stride, const uint16_t *keys, int64_t key_row_stride,
                               int64_t rows, int64_t k, int64_t *out, int64_t out_row_stride, hipStream_t st)
{
    if (rows == 0 || k == 0) return hipSuccess;
    ProfScope ps_(K_RANK, st);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3((unsigned)((k + 15) / 16), (unsigned)rows), dim3(256), 0, st, idx_asc,
                       asc_row_stride, keys, key_row_stride, (int)k, out, out_row_stride);
    return hipGetLastError();
}

}  // namespace fk
