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
#include "fk_host.h"

namespace fk {

// grid (ceil(cap / ROWS_PER_BLOCK), B*Hkv, 2 {K,V}); LPR = lanes per row = D*2/16
template <int LPR>
__global__ void __launch_bounds__(256) compact_kv_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                         const uint16_t *__restrict__ v, int64_t vs_b, int64_t vs_h, int64_t vs_s,
                                                         const int64_t *__restrict__ idx, int Hkv, int S, int W, int cap,
                                                         uint16_t *__restrict__ k_out, uint16_t *__restrict__ v_out)
{
    constexpr int RPI = 256 / LPR;          // rows per block-wide instruction
    constexpr int UNROLL = 4;               // independent rows in flight per thread
    const int bg = blockIdx.y, b = bg / Hkv, g = bg % Hkv;
    const bool isv = blockIdx.z != 0;
    const uint16_t *src = isv ? v + b * vs_b + (int64_t)g * vs_h : k + b * ks_b + (int64_t)g * ks_h;
    const int64_t ss = isv ? vs_s : ks_s;
    uint16_t *dst = (isv ? v_out : k_out) + (size_t)bg * cap * (LPR * 8);
    const int kk = cap - W, n = S - W;
    const int64_t *ib = idx + (size_t)bg * kk;
    const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
    const int r0 = blockIdx.x * (RPI * UNROLL) + rloc;
    int64_t srow[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int r = r0 + u * RPI;
        srow[u] = r < kk ? ib[r] : (int64_t)(n + (r - kk));      // tail = window rows (utils.py:118-121)
    }
    uint4 val[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int r = r0 + u * RPI;
        if (r < cap) val[u] = *reinterpret_cast<const uint4 *>(src + srow[u] * ss + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int r = r0 + u * RPI;
        if (r < cap) *reinterpret_cast<uint4 *>(dst + (size_t)r * (LPR * 8) + sub * 8) = val[u];
    }
}

hipError_t launch_compact(const fastkv_problem &p, const void *k, const int64_t *ks, const void *v, const int64_t *vs,
                          const int64_t *idx, void *k_out, void *v_out, hipStream_t st)
{
    const int lpr = p.D / 8;
    const int rows_per_block = (256 / lpr) * 4;
    dim3 grid((p.capacity + rows_per_block - 1) / rows_per_block, p.B * p.Hkv, 2);
#define FK_COMPACT(LPRV)                                                                                                   \
    hipLaunchKernelGGL((compact_kv_kernel<LPRV>), grid, dim3(256), 0, st, (const uint16_t *)k, ks[0], ks[1], ks[2],         \
                       (const uint16_t *)v, vs[0], vs[1], vs[2], idx, p.Hkv, p.S, p.window, p.capacity, (uint16_t *)k_out,  \
                       (uint16_t *)v_out)
    if (lpr == 8) FK_COMPACT(8);
    else if (lpr == 16) FK_COMPACT(16);
    else FK_COMPACT(32);
#undef FK_COMPACT
    return hipGetLastError();
}

// Generic row gather, rows of `row_bytes` (multiple of 16).  grid (ceil(rows_out/RPB), batches); a wave
// moves whole rows: lanes stride over the 16-B pieces of a row, 4 rows in flight per wave.
__global__ void __launch_bounds__(256) gather_rows_kernel(const unsigned char *__restrict__ src, int64_t sbs, int64_t srs,
                                                          const int64_t *__restrict__ idx, int64_t ibs, int64_t rows_out,
                                                          int64_t rows_in, int64_t row_bytes, unsigned char *__restrict__ dst)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = blockIdx.y;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + w) * 4;       // 4 waves x 4 rows per block
    const unsigned char *sb = src + b * sbs;
    unsigned char *db = dst + b * rows_out * row_bytes;
    int64_t sr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int64_t r = r0 + u;
        int64_t s = r < rows_out ? idx[b * ibs + r] : 0;
        sr[u] = (s >= 0 && s < rows_in) ? s : 0;                // out-of-range indices read row 0 (never faults)
    }
    const int64_t pieces = row_bytes >> 4;
    for (int64_t pc = lane; pc < pieces; pc += 64) {
        uint4 val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u < rows_out) val[u] = *reinterpret_cast<const uint4 *>(sb + sr[u] * srs + pc * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u < rows_out) *reinterpret_cast<uint4 *>(db + (r0 + u) * row_bytes + pc * 16) = val[u];
    }
}

hipError_t launch_gather_rows(const void *src, int64_t sbs, int64_t srs, const int64_t *idx, int64_t ibs, int64_t batches,
                              int64_t rows_out, int64_t rows_in, int64_t row_bytes, void *dst, hipStream_t st)
{
    if (rows_out == 0 || batches == 0) return hipSuccess;
    dim3 grid((unsigned)((rows_out + 15) / 16), (unsigned)batches);
    hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, st, (const unsigned char *)src, sbs, srs, idx, ibs, rows_out,
                       rows_in, row_bytes, (unsigned char *)dst);
    return hipGetLastError();
}

}  // namespace fk
