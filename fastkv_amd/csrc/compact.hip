// K/V gather + window append in ONE pass: replaces the two `gather`s and two `cat`s of
// FastKVCluster.update_kv (/root/reference/baselines/fastkv/utils.py:114-121; ops k12-k13 of
// SURVEY.md 2.3, which copy every selected row twice), and the TSP hidden-state / position-id
// gather of the decoder layer (/root/reference/baselines/fastkv/llama_model.py:254-257).
//
// Pure byte movement, HBM bound: every source row is read once with 16-B lane loads (a D=128
// fp16 row is 16 lanes x 16 B, so one wave instruction moves 4 rows), every destination byte is
// written once, fully coalesced.  Algorithmic bytes per layer at the 32k config:
// 2*B*Hkv*cap*D*2 read + the same written + B*Hkv*(cap-W)*8 index bytes.
#include "fk_device.h"
#include <cstdlib>
#include "fk_host.h"
#include <atomic>
#include "prof.h"
#include "rank.h"

namespace fk {

// The index-bounds debug mode SURVEY.md 5 plans in place of a GPU sanitizer: every gather of this file REPORTS an index outside
// [0, S) -- FASTKV_EBOUNDS from fastkv_last_status() / the next operator call -- besides clamping it as always.  (Not a trap: a
// trapped wave takes the process down with the GPU left unusable for minutes on this pool; the report is just as loud and the
// suite can run in this mode.)  A run-time switch (FASTKV_DEBUG_BOUNDS=1, read once); a build with -DFK_DEBUG_BOUNDS has it always on.
static bool debug_bounds()
{
#ifdef FK_DEBUG_BOUNDS
    return true;
#else
    static const bool on = []() { const char *e = getenv("FASTKV_DEBUG_BOUNDS"); return e && e[0] == '1'; }();
    return on;
#endif
}


// grid (ceil(cap / RPB), B*Hkv); LPR = lanes per row = D*2/16, RPB = 256/LPR rows per workgroup; a lane moves the
// same 16-B piece of the K row and of the V row.
// ONE row per thread group and many small workgroups: measured on MI355X (tools/probes/compact_probe.hip) this shape
// moves the 32k-config layer in 4.4 us and reaches 5.2 TB/s (65 % of the 8 TB/s HBM peak) at the 539 MB roofline
// shape, whereas 2-4 rows per thread (fewer, fatter workgroups) dropped to 0.9 TB/s at one layer: the gather is a
// two-step dependent chain (index -> row) and only occupancy hides it.
template <int LPR>
__global__ void __launch_bounds__(256) compact_kv_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                         const uint16_t *__restrict__ v, int64_t vs_b, int64_t vs_h, int64_t vs_s,
                                                         const int64_t *__restrict__ idx, const uint16_t *__restrict__ keys,
                                                         int64_t *__restrict__ idx_sorted, int Hkv, int S, int W, int cap,
                                                         int keys_in_lds, uint16_t *__restrict__ k_out,
                                                         uint16_t *__restrict__ v_out, uint32_t *__restrict__ epoch_bump,
                                                         int64_t os_b, int64_t os_h, int64_t os_r, const uint64_t *__restrict__ k_tab,
                                                         const uint64_t *__restrict__ v_tab, const uint64_t *__restrict__ ko_tab,
                                                         const uint64_t *__restrict__ vo_tab, uint32_t *__restrict__ bounds_flag)
{
    // last kernel of the operator: advance the workspace epoch after a fused score launch (fused.hip)
    if (epoch_bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const uint32_t e = *epoch_bump + (uint32_t)EPOCH_STRIDE;
        *epoch_bump = e ? e : (uint32_t)EPOCH_STRIDE;
    }
    constexpr int RPB = 256 / LPR;
    extern __shared__ __attribute__((aligned(16))) uint16_t s_keys[];   // winner keys for the ranking (dynamic: 0 B when unused)
    const int bg = blockIdx.y, b = bg / Hkv, g = bg % Hkv;
    // (per-entry base addresses instead of base + batch stride: fk_host.h PtrTables)
    const uint16_t *ksrc = (k_tab ? reinterpret_cast<const uint16_t *>(k_tab[b]) : k + b * ks_b) + (int64_t)g * ks_h;
    const uint16_t *vsrc = (v_tab ? reinterpret_cast<const uint16_t *>(v_tab[b]) : v + b * vs_b) + (int64_t)g * vs_h;
    // output [B,Hkv,cap,D] with element strides (os_b, os_h, os_r): contiguous, or a window of a larger cache slab
    uint16_t *kdst = (ko_tab ? reinterpret_cast<uint16_t *>(ko_tab[b]) : k_out + b * os_b) + g * os_h;
    uint16_t *vdst = (vo_tab ? reinterpret_cast<uint16_t *>(vo_tab[b]) : v_out + b * os_b) + g * os_h;
    const int kk = cap - W, n = S - W;
    const int sub = threadIdx.x % LPR;
    const int r = blockIdx.x * RPB + threadIdx.x / LPR;
    const int rc = r < cap ? r : cap - 1;
    // selected row, or one of the window rows appended after them (utils.py:118-121); K and V rows of the same
    // position are fetched together (two independent 16-B loads per lane in flight)
    // idx == nullptr: every candidate is kept in ascending position (capacity == S), the list is the identity: one
    // dependent memory round trip less
    int64_t srow = rc < kk ? (idx ? idx[(size_t)bg * kk + rc] : (int64_t)rc) : (int64_t)(n + (rc - kk));
    // one policy for every gather of the library (here, gather_rows, sp_compact): an index outside [0, S) never faults, it
    // reads a clamped row; FASTKV_DEBUG_BOUNDS=1 in the environment (or a build with -DFK_DEBUG_BOUNDS) makes it loud instead
    if (bounds_flag && (srow < 0 || srow >= S)) __hip_atomic_store(bounds_flag + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // FASTKV_EBOUNDS
    srow = srow < 0 ? 0 : (srow >= S ? (int64_t)S - 1 : srow);
    // Non-temporal on both sides: every source row is read once and every destination byte written once by this launch, nothing of
    // either is read again before the launch ends.  Round 4, tools/probes/compact_probe2.hip at the 541 MB roofline shape on one box:
    // plain 102.4 us (66 % of 8 TB/s), nt loads 97.8, nt stores 95.5, both 94.2 us (72 %) = a contiguous copy of the same bytes.
    // (Round 3 had tried the builtins on HIP's uint4 class, which they do not accept, and recorded "no gain".)
    typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
    const nt_u32x4 kval = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(ksrc + srow * ks_s + sub * 8));
    const nt_u32x4 vval = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(vsrc + srow * vs_s + sub * 8));
    int d = rc;
    if (keys && keys_in_lds == 2) {
        // ORDER_SCORE with the slots already known (rank_group_kernel ran first: many heads): one 2-byte load per row
        if (rc < kk) {
            d = keys[(size_t)bg * ((kk + 7) & ~7) + rc];
            if (idx_sorted && sub == 0) idx_sorted[(size_t)bg * kk + d] = srow;
        }
    } else if (keys) {
        // ORDER_SCORE: the winner at ascending-position slot r goes to slot rank(r) (value descending, ties by position);
        // the LPR lanes of the row share the comparison counting, which runs under the latency of the row loads above
        const int kal = (kk + 7) & ~7;
        const uint16_t *kr = keys + (size_t)bg * kal;
        const int pc = rc < kk ? rc : kk - 1;
        uint32_t rk;
        if (keys_in_lds) {                                    // one 16-B load per thread instead of kal/8/LPR dependent L2 trips
            const int kpad = (kal + 32 * LPR - 1) / (32 * LPR) * (32 * LPR);     // whole steps of rank_partial_padded; zeros never count
            for (int i = threadIdx.x * 8; i < kpad; i += 256 * 8)
                *reinterpret_cast<uint4 *>(s_keys + i) = i < kal ? *reinterpret_cast<const uint4 *>(kr + i) : make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            const uint32_t kp = s_keys[pc];
            rk = kp ? rank_partial_padded(s_keys, kpad, pc, kp, sub, LPR) : rank_partial(s_keys, kk, pc, kp, sub, LPR);
        } else {
            rk = rank_partial(kr, kk, pc, kr[pc], sub, LPR);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) rk += __shfl_xor((int)rk, o, 64);
        if (rc < kk) {
            d = (int)rk;
            if (idx_sorted && sub == 0) idx_sorted[(size_t)bg * kk + d] = srow;
        }
    }
    else if (idx_sorted && !idx && rc < kk && sub == 0) {
        idx_sorted[(size_t)bg * kk + rc] = srow;               // identity selection in ascending order: the list itself
    }
    if (r < cap) {
        __builtin_nontemporal_store(kval, reinterpret_cast<nt_u32x4 *>(kdst + (int64_t)d * os_r + sub * 8));
        __builtin_nontemporal_store(vval, reinterpret_cast<nt_u32x4 *>(vdst + (int64_t)d * os_r + sub * 8));
    }
}

// keys[row][0..kk) (order-preserving 16-bit keys of the winners in ascending position) -> the winners' ORDER_SCORE slots, IN
// PLACE: slot = number of winners with a larger key, or the same key at an earlier position.  One 1024-thread workgroup per
// row: the keys are grouped by value range (<= 4096 bins between the row's smallest and largest key, LDS histogram + suffix
// sums + atomic cursors), a winner's slot is the number of winners in higher bins plus a count over its own bin's few
// members.  O(k) per head where the counting inside compact_kv is O(k^2): with FEW heads that counting is spread over the
// chip's idle vector ALUs and costs less than this extra launch; with MANY heads (>= 64: batched prompts, the bench's roofline
// shape) it is what bounds the copy (174 us against 104 us in index order) and this kernel (all heads in parallel) takes
// over.  A row whose keys are all equal degenerates to one bin and k^2 / 1024 comparisons per thread (~10 us at k = 2040).
// BPT = bins per thread: 4096 bins for the budgets of a few thousand rows, fewer for longer winner lists so that bins + lists still fit the
// workgroup's LDS (round 5: the published recipe keeps 3276 rows per head at 32k and 13,107 at 128k -- beyond the 2688 winners the 4096-bin
// layout holds in 64 KiB, such calls used to fall back to the k^2 counting inside the copy kernel: 81 us instead of ~45 per 64-head launch).
// Measurement build (-DFK_STAMP): wall-clock stamps (100 MHz) of the grouping pass's stages, per wave (tools/stamp_rank_group.py)
#ifdef FK_STAMP
__device__ unsigned long long g_rstamps[4096 * 8];
#define FKR_STAMP(slot) do { if ((threadIdx.x & 63) == 0) g_rstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) % 4096 * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define FKR_STAMP(slot) do { } while (0)
#endif
template <int BPT>
__global__ void __launch_bounds__(1024) rank_group_kernel(uint16_t *__restrict__ keys, int kk, int kal)
{
    FKR_STAMP(0);
    constexpr int RG_BINS = 1024 * BPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char rg_smem[];
    uint32_t *s_start = reinterpret_cast<uint32_t *>(rg_smem);             // [RG_BINS] counts, then: winners in higher bins
    uint32_t *s_cnt = s_start + RG_BINS;                                    // [RG_BINS] members of the bin
    uint32_t *s_cur = s_cnt + RG_BINS;                                      // [RG_BINS] scatter cursors
    uint32_t *s_grp = s_cur + RG_BINS;                                      // [kal] composites grouped by bin
    uint16_t *s_key = reinterpret_cast<uint16_t *>(s_grp + kal);            // [kal]
    __shared__ uint32_t s_w[16], s_mn[16], s_mx[16];
    uint16_t *kr = keys + (size_t)blockIdx.x * kal;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t mn = 0xffffu, mx = 0;
    for (int i = threadIdx.x; i < kk; i += 1024) {
        const uint32_t x = kr[i];
        s_key[i] = (uint16_t)x;
        mn = min(mn, x);
        mx = max(mx, x);
    }
    for (int i = threadIdx.x; i < RG_BINS; i += 1024) { s_start[i] = 0; s_cur[i] = 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64)); }
    if (lane == 0) { s_mn[w] = mn; s_mx[w] = mx; }
    FKR_STAMP(1);
    __syncthreads();
    FKR_STAMP(2);
#pragma unroll
    for (int i = 0; i < 16; ++i) { mn = min(mn, s_mn[i]); mx = max(mx, s_mx[i]); }
    int sh = 0;
    while (((mx - mn) >> sh) >= (uint32_t)RG_BINS) ++sh;
    for (int i = threadIdx.x; i < kk; i += 1024) atomicAdd(&s_start[(s_key[i] - mn) >> sh], 1u);
    __syncthreads();
    FKR_STAMP(3);
    // suffix sums over the bins: thread t owns bins BPT t .. BPT t + BPT - 1
    {
        const int b0 = threadIdx.x * BPT;
        uint32_t cb[BPT], own = 0;
#pragma unroll
        for (int u = 0; u < BPT; ++u) { cb[u] = s_start[b0 + u]; own += cb[u]; }
        uint32_t v = own;                                                   // -> sum over this and the higher lanes of the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = (uint32_t)__shfl_down((int)v, o, 64); if (lane + o < 64) v += nb; }
        if (lane == 0) s_w[w] = v;
        __syncthreads();
        uint32_t higher = 0;
        for (int i = w + 1; i < 16; ++i) higher += s_w[i];
        uint32_t above = v - own + higher;                                  // winners in the bins of higher threads
#pragma unroll
        for (int u = BPT - 1; u >= 0; --u) { s_cnt[b0 + u] = cb[u]; s_start[b0 + u] = above; above += cb[u]; }
    }
    __syncthreads();
    FKR_STAMP(4);
    for (int i = threadIdx.x; i < kk; i += 1024) {
        const uint32_t key = s_key[i], bin = (key - mn) >> sh;
        s_grp[s_start[bin] + atomicAdd(&s_cur[bin], 1u)] = (key << 16) | (uint32_t)(65535 - i);
    }
    __syncthreads();
    FKR_STAMP(5);
    for (int i = threadIdx.x; i < kk; i += 1024) {
        const uint32_t key = s_key[i], bin = (key - mn) >> sh, me = (key << 16) | (uint32_t)(65535 - i);
        const uint32_t lo = s_start[bin], n = s_cnt[bin];
        uint32_t c = 0;
        for (uint32_t u = 0; u < n; ++u) c += s_grp[lo + u] > me ? 1u : 0u;   // larger key, or the same key at an earlier position
        kr[i] = (uint16_t)(lo + c);
    }
    FKR_STAMP(6);
}
#ifdef FK_STAMP
}  // namespace fk
extern "C" int fastkv_debug_read_rank_stamps(unsigned long long *host, size_t n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::g_rstamps), n * sizeof(unsigned long long));
}
namespace fk {
#endif

hipError_t launch_compact(const fastkv_problem &p, const void *k, const int64_t *ks, const void *v, const int64_t *vs,
                          const int64_t *idx, const uint16_t *keys, int64_t *idx_sorted_out, void *k_out, void *v_out,
                          hipStream_t st, uint32_t *epoch_bump, const int64_t *out_strides, const PtrTables *pt)
{
    const int lpr = p.D / 8;
    const int rows_per_block = 256 / lpr;
    dim3 grid((p.capacity + rows_per_block - 1) / rows_per_block, p.B * p.Hkv);
    const int64_t os_r = out_strides ? out_strides[2] : p.D, os_h = out_strides ? out_strides[1] : (int64_t)p.capacity * p.D;
    const int64_t os_b = out_strides ? out_strides[0] : (int64_t)p.Hkv * p.capacity * p.D;
    ProfScope ps_(K_COMPACT, st);
    const size_t kal = ((size_t)(p.capacity - p.window) + 7) & ~(size_t)7;
    int keys_in_lds = (keys && kal <= 16384) ? 1 : 0;                      // 32 KiB of LDS at most
    if (keys && (int64_t)p.B * p.Hkv >= 64 && kal <= 24000) {
        // many heads: every head's key list becomes its slot list (workspace memory of this call).  LDS: 12 bytes per bin + 6 per winner:
        // 4096 bins up to 2688 winners, 2048 up to 6656 (both within the default 64 KiB), 1024 bins up to 24,000 winners (156 KiB: the
        // published recipe keeps 13,107 rows per head at 128k -- counting those inside the copy kernel cost 0.5 ms per eight layers);
        // longer lists keep the counting inside the copy
        const int kk_ = p.capacity - p.window;
        const int bpt = kal <= 2688 ? 4 : kal <= 6656 ? 2 : 1;
        const size_t lds = (size_t)3 * 1024 * bpt * 4 + kal * 4 + kal * 2;
        // more than the default 64 KiB of dynamic LDS has to be asked for, per device, and may be refused (a part with less LDS per
        // workgroup): the grouping pass is then skipped and the copy kernel counts by itself, as it does beyond 24,000 winners
        bool lds_ok = true;
        if (lds > 64 * 1024 - 256) {
            static std::atomic<int> granted[16];                 // per device: 0 unknown, 1 granted, -1 refused
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) lds_ok = false;
            else {
                int g = granted[dev].load(std::memory_order_relaxed);
                if (!g) {
                    g = hipFuncSetAttribute(reinterpret_cast<const void *>(rank_group_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024 - 512) == hipSuccess ? 1 : -1;
                    if (g < 0) (void)hipGetLastError();
                    granted[dev].store(g, std::memory_order_relaxed);
                }
                lds_ok = g > 0;
            }
        }
        if (lds_ok) {
            const dim3 rg((unsigned)(p.B * p.Hkv));
            if (bpt == 4) hipLaunchKernelGGL(rank_group_kernel<4>, rg, dim3(1024), lds, st, const_cast<uint16_t *>(keys), kk_, (int)kal);
            else if (bpt == 2) hipLaunchKernelGGL(rank_group_kernel<2>, rg, dim3(1024), lds, st, const_cast<uint16_t *>(keys), kk_, (int)kal);
            else hipLaunchKernelGGL(rank_group_kernel<1>, rg, dim3(1024), lds, st, const_cast<uint16_t *>(keys), kk_, (int)kal);
            keys_in_lds = 2;
        }
    }
    const size_t kpad = (kal + 32 * (size_t)lpr - 1) / (32 * (size_t)lpr) * (32 * (size_t)lpr);   // (rank_partial_padded reads whole steps)
    const size_t dyn = keys_in_lds == 1 ? kpad * sizeof(uint16_t) : 0;
#define FK_COMPACT(LPRV)                                                                                                   \
    hipLaunchKernelGGL((compact_kv_kernel<LPRV>), grid, dim3(256), dyn, st, (const uint16_t *)k, ks[0], ks[1], ks[2],       \
                       (const uint16_t *)v, vs[0], vs[1], vs[2], idx, keys, idx_sorted_out, p.Hkv, p.S, p.window,           \
                       p.capacity, keys_in_lds, (uint16_t *)k_out, (uint16_t *)v_out, epoch_bump, os_b, os_h, os_r,               \
                       pt ? pt->k : nullptr, pt ? pt->v : nullptr, pt ? pt->k_out : nullptr, pt ? pt->v_out : nullptr, \
                       debug_bounds() ? abort_flag_device() : nullptr)
    if (lpr == 8) FK_COMPACT(8);
    else if (lpr == 16) FK_COMPACT(16);
    else FK_COMPACT(32);
#undef FK_COMPACT
    return hipGetLastError();
}

// keys[row, i] = order-preserving 16-bit key of scores[row, idx[row, i]] (zero padded to a multiple of 8): what the
// selection kernels hand to compact_kv inside the operator, for the stand-alone ORDER_SCORE compaction.
__global__ void __launch_bounds__(256) winner_keys_kernel(const uint16_t *__restrict__ scores, int64_t row_stride, int64_t n,
                                                          const int64_t *__restrict__ idx, int kk, int kal,
                                                          uint16_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (i >= kal) return;
    uint16_t key = 0;
    if (i < kk) {
        int64_t j = idx[(size_t)row * kk + i];
        j = (j >= 0 && j < n) ? j : 0;
        key = (uint16_t)mono16(scores[(size_t)row * row_stride + j]);
    }
    keys[(size_t)row * kal + i] = key;
}

hipError_t launch_winner_keys(const uint16_t *scores, int64_t row_stride, int64_t n, const int64_t *idx, int64_t rows, int kk,
                              uint16_t *keys, hipStream_t st)
{
    const int kal = (kk + 7) & ~7;
    hipLaunchKernelGGL(winner_keys_kernel, dim3((kal + 255) / 256, (unsigned)rows), dim3(256), 0, st, scores, row_stride, n, idx, kk, kal, keys);
    return hipGetLastError();
}

// Generic row gather, rows of `row_bytes` (multiple of 16).  `lpr` lanes (power of two <= 256) cooperate on a row,
// 256/lpr rows per workgroup, one 16-B piece per lane per iteration; grid (ceil(rows_out/rpb), batches).
// (pos_in / pos_out, optional: the position ids ride along -- pos_out[b, r] = pos_in[b, idx[b, r]] by the first lane of a row's group:
// llama_model.py:254 and :255-257 in ONE launch, round 6)
__global__ void __launch_bounds__(256) gather_rows_kernel(const unsigned char *__restrict__ src, int64_t sbs, int64_t srs,
                                                          const int64_t *__restrict__ idx, int64_t ibs, int64_t rows_out,
                                                          int64_t rows_in, int64_t row_bytes, int lpr_shift,
                                                          unsigned char *__restrict__ dst, uint32_t *__restrict__ bounds_flag,
                                                          const int64_t *__restrict__ pos_in, int64_t pbs, int64_t *__restrict__ pos_out)
{
    const int lpr = 1 << lpr_shift;
    const int sub = threadIdx.x & (lpr - 1);
    const int64_t b = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * (256 >> lpr_shift) + (threadIdx.x >> lpr_shift);
    if (r >= rows_out) return;
    int64_t s = idx[b * ibs + r];
    if (bounds_flag && (s < 0 || s >= rows_in)) __hip_atomic_store(bounds_flag + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // FASTKV_EBOUNDS
    s = s < 0 ? 0 : (s >= rows_in ? rows_in - 1 : s);          // out-of-range indices read a clamped row (never faults)
    if (pos_out && sub == 0) pos_out[b * rows_out + r] = pos_in[b * pbs + s];
    const unsigned char *sp = src + b * sbs + s * srs;
    unsigned char *dp = dst + (b * rows_out + r) * row_bytes;
    const int64_t pieces = row_bytes >> 4;
    // four pieces of a thread in flight before the first store (8 KiB hidden-state rows: both of a thread's pieces -- a "load, store,
    // load, store" loop pays a memory round trip per piece)
    // (non-temporal on both sides, as in compact_kv: every source row is read once, every destination byte written once by this launch)
    typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
    for (int64_t pc = sub; pc < pieces; pc += 4 * (int64_t)lpr) {
        nt_u32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = pc + u * (int64_t)lpr;
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(sp + (q < pieces ? q : pc) * 16));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t q = pc + u * (int64_t)lpr;
            if (q < pieces) __builtin_nontemporal_store(x[u], reinterpret_cast<nt_u32x4 *>(dp + q * 16));
        }
    }
}

hipError_t launch_gather_rows(const void *src, int64_t sbs, int64_t srs, const int64_t *idx, int64_t ibs, int64_t batches,
                              int64_t rows_out, int64_t rows_in, int64_t row_bytes, void *dst, hipStream_t st, const int64_t *pos_in,
                              int64_t pbs, int64_t *pos_out)
{
    if (rows_out == 0 || batches == 0) return hipSuccess;
    int lpr_shift = 0;
    // four 16-B pieces per lane where the row is long enough (all of them requested before the first store; with one piece per lane and
    // 256 lanes on an 8 KiB row half of a thread's four loads were clamped repeats)
    while ((1 << lpr_shift) < 256 && ((int64_t)64 << lpr_shift) < row_bytes) ++lpr_shift;
    const int rpb = 256 >> lpr_shift;
    dim3 grid((unsigned)((rows_out + rpb - 1) / rpb), (unsigned)batches);
    ProfScope ps_(K_GATHER, st);
    hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, st, (const unsigned char *)src, sbs, srs, idx, ibs, rows_out,
                       rows_in, row_bytes, lpr_shift, (unsigned char *)dst, debug_bounds() ? abort_flag_device() : nullptr, pos_in, pbs, pos_out);
    return hipGetLastError();
}

}  // namespace fk
