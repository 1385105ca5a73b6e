// Canonical top-k selection on fp16 score rows: replaces `attn_cache.topk(k).indices`
// (/root/reference/baselines/fastkv/utils.py:113) and the TSP selection + window union + sort
// (utils.py:127-130).
//
// One 1024-thread workgroup per score row (rows are <= a few hundred KiB and L2 resident; the
// stage is latency bound by design).  Exact radix select on the 16-bit order-preserving key:
//   pass 1  256-bin histogram of the high byte      -> threshold byte, #above
//   pass 2  256-bin histogram of the low byte among elements in the threshold byte -> k-th value
//   pass 3  ordered compaction (ballot-free block scan): every element above the k-th value plus
//           the first `quota` elements equal to it in ascending position  => canonical tie rule
//   ORDER_SCORE additionally sorts the k winners by descending value with a stable 4-pass LSD
//   radix sort (4-bit digits, wave-ballot ranking), so equal values keep ascending position.
// Histograms are privatised 8x by lane and a wave whose lanes agree issues one LDS atomic, so
// degenerate rows (all-equal scores, e.g. the all-ones benchmark prompt) do not serialise.
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"

namespace fk {

constexpr int SEL_LDS_LIST = 10240;   // winners kept in LDS for the sort (2 x (2+4) B each = 120 KiB)
constexpr int SEL_MAXIT_LDS = 16;     // radix-sort iterations whose counters fit the static LDS table

struct SelShared {
    uint32_t hist[256 * 8];
    uint32_t wtot[16];
    uint32_t bcast[4];
    uint32_t cnt[SEL_MAXIT_LDS * 256];
};

__device__ __forceinline__ void load8(const uint16_t *row, int j0, int n, bool vec, uint32_t key[8], bool ok[8])
{
    if (vec && j0 + 8 <= n) {
        uint4 raw = *reinterpret_cast<const uint4 *>(row + j0);
        uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { key[e] = mono16((wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu); ok[e] = true; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ok[e] = (j0 + e) < n;
            key[e] = ok[e] ? mono16(row[j0 + e]) : 0u;
        }
    }
}

// one histogram update per lane with wave-level agreement shortcut
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool active, int lane)
{
    uint64_t act = __ballot(active);
    if (act == 0) return;
    int leader = __builtin_ctzll(act);
    uint32_t first = __shfl((int)bin, leader, 64);
    uint64_t same = __ballot(active && bin == first);
    if (active) {
        if (bin == first) {
            if (lane == leader) atomicAdd(&hist[first * 8 + (lane & 7)], (uint32_t)__builtin_popcountll(same));
        } else {
            atomicAdd(&hist[bin * 8 + (lane & 7)], 1u);
        }
    }
}

// after a histogram pass: find the bin holding the kk-th largest element.  Executed by wave 0.
// out: bcast[0] = bin, bcast[1] = number of elements in bins above it
__device__ __forceinline__ void find_bin(SelShared &sh, uint32_t kk, int lane)
{
    // lane l owns bins 255-4l .. 252-4l (descending)
    uint32_t c[4], loc = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int bin = 255 - 4 * lane - u;
        uint32_t s = 0;
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) s += sh.hist[bin * 8 + cp];
        c[u] = s;
        loc += s;
    }
    uint32_t inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    uint32_t above = inc - loc;   // elements in bins owned by lower lanes (= higher bins)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (above < kk && kk <= above + c[u]) { sh.bcast[0] = 255 - 4 * lane - u; sh.bcast[1] = above; }
        above += c[u];
    }
}

// exclusive scan of arr[0..M) in place by the whole block (M <= 1024 * per, any M)
__device__ void block_exclusive_scan(uint32_t *arr, int M, SelShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (M + SEL_THREADS - 1) / SEL_THREADS;
    const int lo = tid * per, hi = min(lo + per, M);
    uint32_t loc = 0;
    for (int i = lo; i < hi; ++i) loc += arr[i];
    uint32_t inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) sh.wtot[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int u = 0; u < w; ++u) base += sh.wtot[u];
    uint32_t run = base + inc - loc;
    for (int i = lo; i < hi; ++i) { uint32_t v = arr[i]; arr[i] = run; run += v; }
    __syncthreads();
}

__global__ void __launch_bounds__(SEL_THREADS) select_topk_kernel(const uint16_t *__restrict__ scores, int64_t row_stride, int n,
                                                                  int k, int order, int append, int64_t *__restrict__ idx_out,
                                                                  uint32_t *__restrict__ g_idx, uint16_t *__restrict__ g_key,
                                                                  uint32_t *__restrict__ g_cnt, int list_in_lds)
{
    __shared__ SelShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rowi = blockIdx.x;
    const uint16_t *row = scores + (size_t)rowi * row_stride;
    int64_t *out = idx_out + (size_t)rowi * (size_t)(k + append);
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    const int nchunks = (n + 8191) / 8192;

    for (int a = tid; a < append; a += SEL_THREADS) out[k + a] = (int64_t)n + a;
    if (k == 0) return;

    // ---------------- pass 1: high byte
    for (int i = tid; i < 256 * 8; i += SEL_THREADS) sh.hist[i] = 0;
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        uint32_t key[8]; bool ok[8];
        load8(row, it * 8192 + tid * 8, n, vec, key, ok);
#pragma unroll
        for (int e = 0; e < 8; ++e) hist_add(sh.hist, key[e] >> 8, ok[e], lane);
    }
    __syncthreads();
    if (w == 0) find_bin(sh, (uint32_t)k, lane);
    __syncthreads();
    const uint32_t thr_hi = sh.bcast[0], above_hi = sh.bcast[1];
    __syncthreads();
    // ---------------- pass 2: low byte inside the threshold byte
    for (int i = tid; i < 256 * 8; i += SEL_THREADS) sh.hist[i] = 0;
    __syncthreads();
    for (int it = 0; it < nchunks; ++it) {
        uint32_t key[8]; bool ok[8];
        load8(row, it * 8192 + tid * 8, n, vec, key, ok);
#pragma unroll
        for (int e = 0; e < 8; ++e) hist_add(sh.hist, key[e] & 0xffu, ok[e] && (key[e] >> 8) == thr_hi, lane);
    }
    __syncthreads();
    if (w == 0) find_bin(sh, (uint32_t)k - above_hi, lane);
    __syncthreads();
    const uint32_t thr = (thr_hi << 8) | sh.bcast[0];
    const uint32_t quota = (uint32_t)k - (above_hi + sh.bcast[1]);   // elements equal to thr that are kept
    __syncthreads();

    // winner lists for ORDER_SCORE (ping-pong A/B)
    const int kal = (k + 7) & ~7;
    uint32_t *idxA, *idxB; uint16_t *keyA, *keyB;
    if (list_in_lds) {
        idxA = reinterpret_cast<uint32_t *>(dyn); idxB = idxA + kal;
        keyA = reinterpret_cast<uint16_t *>(idxB + kal); keyB = keyA + kal;
    } else {
        idxA = g_idx + (size_t)rowi * 2 * kal; idxB = idxA + kal;
        keyA = g_key + (size_t)rowi * 2 * kal; keyB = keyA + kal;
    }

    // ---------------- pass 3: ordered compaction
    uint32_t gt_base = 0, eq_base = 0;
    for (int it = 0; it < nchunks; ++it) {
        uint32_t key[8]; bool ok[8];
        const int j0 = it * 8192 + tid * 8;
        load8(row, j0, n, vec, key, ok);
        uint32_t cg = 0, ce = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { cg += (ok[e] && key[e] > thr); ce += (ok[e] && key[e] == thr); }
        uint32_t pk = cg | (ce << 16), inc = pk;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = __shfl_up((int)inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) sh.wtot[w] = inc;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int u = 0; u < 16; ++u) { uint32_t v = sh.wtot[u]; if (u < w) base += v; total += v; }
        uint32_t ex = base + inc - pk;
        uint32_t gt_before = gt_base + (ex & 0xffffu), eq_before = eq_base + (ex >> 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!ok[e]) continue;
            int pos = -1;
            if (key[e] > thr) { pos = (int)(gt_before + min(eq_before, quota)); gt_before++; }
            else if (key[e] == thr) { if (eq_before < quota) pos = (int)(gt_before + eq_before); eq_before++; }
            if (pos >= 0) {
                if (order == FASTKV_ORDER_INDEX) out[pos] = (int64_t)(j0 + e);
                else { idxA[pos] = (uint32_t)(j0 + e); keyA[pos] = (uint16_t)(0xffffu - key[e]); }
            }
        }
        gt_base += total & 0xffffu;
        eq_base += total >> 16;
        __syncthreads();
    }
    if (order == FASTKV_ORDER_INDEX) return;

    // ---------------- ORDER_SCORE: stable LSD radix sort of (inverted key, idx), ascending
    const int iters = (k + SEL_THREADS - 1) / SEL_THREADS;
    uint32_t *cnt = (iters <= SEL_MAXIT_LDS) ? sh.cnt : (g_cnt + (size_t)rowi * iters * 256);
    const int M = iters * 256;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 4;
        for (int i = tid; i < M; i += SEL_THREADS) cnt[i] = 0;
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const int item = it * SEL_THREADS + tid;
            const bool act = item < k;
            const uint32_t d = act ? ((keyA[item] >> shift) & 15u) : 16u;
            uint64_t mine = 0;
#pragma unroll
            for (uint32_t v = 0; v < 16; ++v) { uint64_t m = __ballot(d == v); if (d == v) mine = m; }
            if (act && (mine & lt_mask) == 0) cnt[(d * iters + it) * 16 + w] = (uint32_t)__builtin_popcountll(mine);
        }
        __syncthreads();
        block_exclusive_scan(cnt, M, sh);
        for (int it = 0; it < iters; ++it) {
            const int item = it * SEL_THREADS + tid;
            const bool act = item < k;
            const uint32_t kv = act ? keyA[item] : 0u;
            const uint32_t d = act ? ((kv >> shift) & 15u) : 16u;
            uint64_t mine = 0;
#pragma unroll
            for (uint32_t v = 0; v < 16; ++v) { uint64_t m = __ballot(d == v); if (d == v) mine = m; }
            if (act) {
                uint32_t pos = cnt[(d * iters + it) * 16 + w] + (uint32_t)__builtin_popcountll(mine & lt_mask);
                keyB[pos] = (uint16_t)kv;
                idxB[pos] = idxA[item];
            }
        }
        __syncthreads();
        uint32_t *ti = idxA; idxA = idxB; idxB = ti;
        uint16_t *tk = keyA; keyA = keyB; keyB = tk;
    }
    for (int i = tid; i < k; i += SEL_THREADS) out[i] = (int64_t)idxA[i];
}

hipError_t launch_select(const uint16_t *scores, int64_t rows, int64_t row_stride, int64_t n, int64_t k, int order,
                         int append, int64_t *idx_out, char *ws, hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    const int64_t kal = (k + 7) & ~(int64_t)7;
    const int list_in_lds = (order == FASTKV_ORDER_SCORE && kal <= SEL_LDS_LIST) ? 1 : 0;
    size_t dyn = list_in_lds ? (size_t)kal * 2 * (sizeof(uint32_t) + sizeof(uint16_t)) : 0;
    // global fallbacks (only touched when the lists / counters do not fit LDS)
    uint32_t *g_idx = reinterpret_cast<uint32_t *>(ws);
    uint16_t *g_key = reinterpret_cast<uint16_t *>(ws + align_up((size_t)rows * 2 * kal * sizeof(uint32_t), 256));
    uint32_t *g_cnt = reinterpret_cast<uint32_t *>(ws + align_up((size_t)rows * 2 * kal * sizeof(uint32_t), 256) +
                                                   align_up((size_t)rows * 2 * kal * sizeof(uint16_t), 256));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(select_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SEL_LDS_LIST * 2 * 6);
        attr_set = true;
    }
    ProfScope ps_(K_SELECT, st);
    hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)rows), dim3(SEL_THREADS), dyn, st, scores, row_stride, (int)n, (int)k,
                       order, append, idx_out, g_idx, g_key, g_cnt, list_in_lds);
    return hipGetLastError();
}

}  // namespace fk
