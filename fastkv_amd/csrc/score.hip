// Window-attention scoring: replaces ops k1-k10 of FastKVCluster.update_kv
// (/root/reference/baselines/fastkv/utils.py:93-112, and the head sum of :127).
//
//   prep_q          (vector-ALU engine only) Q[b,h,S-W+r,:] fp16 -> fp32 scalar-operand blocks; the matrix-pipe
//                   kernel converts its A operand itself
//   score_logits    streams K ONCE (no repeat_kv materialisation) and writes the fp16-rounded raw logits
//                   L[b,h,r,j] = fp16( sum_d q[d]*k[d] ) with an fp32 fma chain in ascending d      utils.py:94 (matmul)
//                   Two engines with bit-identical results (both are the oracle's fmaf chain):
//                     * matrix pipe: v_mfma_f32_32x32x2_f32 -- the FP32-in/FP32-out MFMA, documented as "bit-for-bit a
//                       k-ordered f32 fmaf chain" at the same 157 TFLOP/s peak as the vector ALU; the 32 window-query
//                       rows of a KV head are exactly one 32-row A block held in registers (used when G*W >= 24);
//                     * vector ALU: v_pk_fma_f32 with the query values as wave-uniform scalar operands (small G*W).
//                   No reduced-precision MFMA anywhere: results must match the CPU oracle bit for bit.
//   row_stats       one workgroup per (b,h,r) row: scale by true division + window mask in place
//                   (utils.py:94-101), row max, sum_j exp(L - max) in 2^-40 fixed point
//                   (order-free, deterministic) -> gmax[row], rinv[row] = 1/sum                       utils.py:103
//   score_finalize  p = e*rinv -> fp16, sum over window rows -> fp16, pool -> fp16, sum over the
//                   G heads of the group -> fp16 = attn_cache c[b,g,j]                                utils.py:103-112
//   tsp_rowsum      t[b,j] = fp16(sum_g c[b,g,j])                                                     utils.py:127
//
// HBM traffic: K read once (B*Hkv*S*D*2 bytes); the logits (B*H*W*S*2 bytes, 16 MiB at the
// 32k config) round-trip through L2 / Infinity Cache.
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"
#include "mfma_tile.h"

namespace fk {


// Measurement build (-DFK_STAMP): per-wave wall-clock stamps (100 MHz) of the matrix-pipe kernel's stages.
#ifdef FK_STAMP
__device__ unsigned long long g_stamps[4096 * 8];
#define FK_STAMP_AT(slot) do { if (lane == 0) g_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 4096 * 8 + (slot)] = wall_clock64(); } while (0)
#define FK_CYC_AT(slot) do { if (lane == 0) g_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 4096 * 8 + (slot)] = clock64(); } while (0)
#else
#define FK_STAMP_AT(slot) do { } while (0)
#define FK_CYC_AT(slot) do { } while (0)
#endif

// A rank's view of a logits row.  Column x holds global prompt position pos0 + x.  One GPU: ncols = S, pos0 = 0,
// own = [0, S).  Sequence sharding (fastkv_amd/dist.py): the row holds the rank's positions plus `pad` halo columns on
// either side (needed by the pooling window); statistics and outputs cover the owned columns only.
struct ColWin {
    int ncols;            // columns present in the row
    int pos0;             // global position of column 0 (negative when the left halo precedes position 0)
    int own_lo, own_hi;   // owned columns
    int S_glob;           // global prompt length (window = its last W positions)
};

// ------------------------------------------------------------------------------------------ prep_q (vector-ALU layout)
// qf[bg][pass][d][RB] (row fastest), so the (row,row+1) operand pairs of v_pk_fma_f32 are adjacent scalars.
// grid (R_alloc, B*Hkv), D threads
__global__ void prep_q_kernel(const uint16_t *__restrict__ q, int64_t qs_b, int64_t qs_h, int64_t qs_s, int H, int Hkv, int S,
                              int D, int W, int R, int R_alloc, int RB, float *__restrict__ qf)
{
    const int row = blockIdx.x, bg = blockIdx.y, d = threadIdx.x;
    const int b = bg / Hkv, g = bg % Hkv, G = H / Hkv;
    float v = 0.0f;
    if (row < R) {
        const int i = row / W, r = row - i * W;
        v = h2f(q[b * qs_b + (int64_t)(g * G + i) * qs_h + (int64_t)(S - W + r) * qs_s + d]);
    }
    const int pass = row / RB, rr = row - pass * RB;
    qf[(((size_t)bg * (R_alloc / RB) + pass) * D + d) * RB + rr] = v;
}

// ------------------------------------------------------------------------------------------ score_logits, matrix pipe
// Persistent waves: grid.x = nblk * Hkv with blockIdx.x % Hkv = kv head (the Hkv workgroups that stream the same token
// range run together, one per XCD under round-robin placement, so an XCD's L2 keeps one head's query block: speed only),
// grid.y = B.  256 threads = 4 independent waves; a wave owns 64-key tiles (two 32-key B blocks) wt = wave id, wave id +
// nwaves, ...  The A operand (the head's 32 query rows, all D/2 k-steps) is loaded into registers once per wave; the
// K rows of the next phase/tile are fetched into registers while the matrix pipe works on the current LDS slab.
// Eight named 16-B registers (an indexed local array carried across the tile loop ends up in scratch memory).
template <int D, bool F16>
__global__ void __launch_bounds__(256, 2) score_logits_mfma_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h,
                                                                int64_t ks_s, const uint16_t *__restrict__ q, int64_t qs_b,
                                                                int64_t qs_h, int64_t qs_s, int q_row0, int H, int Hkv, int S,
                                                                int W, int R, int passes, int Sp, int col_off,
                                                                uint16_t *__restrict__ logits)
{
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][64 * ROWB];
    __shared__ __attribute__((aligned(16))) float As[(D / 2) * 64];   // A operand of every k-step, shared by the 4 waves (same head)
    constexpr int NPH = D / DH;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x % Hkv, blk = blockIdx.x / Hkv, nblk = gridDim.x / Hkv, b = blockIdx.y;
    const int G = H / Hkv;
    const int nwt = (S + 63) / 64;                       // 64-key wave tiles of this head
    const int wave_id = blk * 4 + w, nwaves = nblk * 4;
    const uint16_t *kb = k + b * ks_b + (int64_t)g * ks_h;
    unsigned char *my = slab[w];
    const int n31 = lane & 31, hi = lane >> 5, sh = hi * 16;

    FK_STAMP_AT(0);
    for (int pass = 0; pass < passes; ++pass) {
        // A image: As[s][l] = q[row = pass*32 + (l&31)][dim = 2s + (l>>5)] (lane l of k-step s holds A[i=l&31][k=l>>5]);
        // rows are the group's G heads x W window rows, row = i*W + r  <->  Q[b, g*G+i, S-W+r, :]   (utils.py:94).
        // The query rows are requested BEFORE the first K tile (loads return in order: waiting for Q must not wait for
        // the cold K rows), as unconditional 16-B vectors, all in flight together.
        constexpr int QV = 32 * (D / 8) / 256;                 // 16-B vectors per thread
        uint4 qv[QV];
#pragma unroll
        for (int u = 0; u < QV; ++u) {
            const int item = u * 256 + threadIdx.x, rowl = item / (D / 8), ch = item - rowl * (D / 8);
            const int row = pass * 32 + rowl, rc = row < R ? row : R - 1;
            const int i = rc / W, r = rc - i * W;
            qv[u] = *reinterpret_cast<const uint4 *>(q + b * qs_b + (int64_t)(g * G + i) * qs_h + (int64_t)(q_row0 + r) * qs_s + ch * 8);
        }
        // two register stages: the K rows of a phase are requested two phases (128 MFMAs) before they are committed to LDS
        KStage sA, sB;
        {
            const int first = (wave_id < nwt ? wave_id : 0) * 64;
            k_fetch(sA, kb, ks_s, first, S, 0, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pass) __syncthreads();
#pragma unroll
        for (int u = 0; u < QV; ++u) {
            const int item = u * 256 + threadIdx.x, rowl = item / (D / 8), ch = item - rowl * (D / 8);
            const bool live = pass * 32 + rowl < R;
            if (F16) {                                          // contract "mfma16": fp16 A-operand fragments (mfma_tile.h mfma_phase_f16)
                reinterpret_cast<uint4 *>(As)[(ch >> 1) * 64 + (ch & 1) * 32 + rowl] = live ? qv[u] : make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            const uint32_t wds[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {                       // fp16 pair (dims 2dp, 2dp+1), dp = ch*4 + e
                const uint32_t pr = live ? wds[e] : 0u;
                As[(ch * 4 + e) * 64 + rowl] = h2f((uint16_t)(pr & 0xffffu));
                As[(ch * 4 + e) * 64 + 32 + rowl] = h2f((uint16_t)(pr >> 16));
            }
        }
        __syncthreads();
        // the second stage is requested only now: at launch every wave of the chip asks for its first K rows at once
        // (16 MiB per stage at the 32k shape), and the query rows above queue behind whatever was requested before them
        {
            const int first = (wave_id < nwt ? wave_id : 0) * 64;
            if (NPH >= 2) k_fetch(sB, kb, ks_s, first, S, 1, lane);
            else if (wave_id + nwaves < nwt) k_fetch(sB, kb, ks_s, (wave_id + nwaves) * 64, S, 0, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        FK_STAMP_AT(1);
        uint16_t *lp = logits + ((size_t)(b * H + g * G) * W + (size_t)pass * 32) * Sp + col_off;
        int stamp_i = 2;
        (void)stamp_i;
        int tcount = 0;                                          // tiles done (NPH == 1: the stage alternates per tile)
        for (int wt = wave_id; wt < nwt; wt += nwaves, ++tcount) {
            const int key0 = wt * 64;
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                // commit the stage holding this phase, then re-use it for the phase two steps ahead.  The loads must be
                // issued HERE, ahead of the MFMAs that cover their latency: without the scheduling barriers the machine
                // scheduler sinks them to the end of the phase, right in front of the wait of the next commit.
                if (tcount == 0 && ph < 2) FK_CYC_AT(2 + ph * 3);
                const bool useA = NPH >= 2 ? ((ph & 1) == 0) : ((tcount & 1) == 0);
                int nkey, nph;                                   // phase (ph + 2) in tile order
                if (NPH == 1) { nkey = (wt + 2 * nwaves) * 64; nph = 0; }
                else if (ph + 2 < NPH) { nkey = key0; nph = ph + 2; }
                else { nkey = (wt + nwaves) * 64; nph = ph + 2 - NPH; }
                const bool more = nkey < nwt * 64;
                if (useA) { k_commit(sA, lane, my); if (more) k_fetch(sA, kb, ks_s, nkey, S, nph, lane); }
                else { k_commit(sB, lane, my); if (more) k_fetch(sB, kb, ks_s, nkey, S, nph, lane); }
                __builtin_amdgcn_sched_barrier(0);
                if (tcount == 0 && ph < 2) FK_CYC_AT(3 + ph * 3);
                if (F16) mfma_phase_f16(acc0, acc1, my, reinterpret_cast<const f16x8 *>(As) + ph * 4 * 64 + lane, n31, hi);
                else mfma_phase(acc0, acc1, my, As + ph * (DH / 2) * 64 + lane, n31, sh);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);              // ... and the next commit (which waits for them) stays behind the MFMAs
                if (tcount == 0 && ph < 2) FK_CYC_AT(4 + ph * 3);
            }
            // C/D map: register i of lane l is row (i&3) + 8*(i>>2) + 4*(l>>5), column l&31
            const int j0 = key0 + n31, j1 = key0 + 32 + n31;
            if (key0 + 64 <= S && pass * 32 + 32 <= R) {        // full tile: branch-free stores
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = (i & 3) + 8 * (i >> 2) + 4 * hi;
                    lp[(size_t)m * Sp + j0] = f2h(acc0[i]);
                    lp[(size_t)m * Sp + j1] = f2h(acc1[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = (i & 3) + 8 * (i >> 2) + 4 * hi;
                    if (pass * 32 + m < R) {
                        if (j0 < S) lp[(size_t)m * Sp + j0] = f2h(acc0[i]);
                        if (j1 < S) lp[(size_t)m * Sp + j1] = f2h(acc1[i]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ score_logits, vector ALU
// One key row per lane, the RB query rows of a pass are wave-uniform scalar operands (s_load), RB accumulators per lane.
template <int D, int RB>
__global__ void __launch_bounds__(256) score_logits_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                           const float *__restrict__ qf, int H, int Hkv, int S, int W, int R,
                                                           int passes, int Sp, int col_off, uint16_t *__restrict__ logits)
{
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][64 * ROWB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x % Hkv, tile = blockIdx.x / Hkv, b = blockIdx.y;
    const int G = H / Hkv;
    const int key0 = tile * TKA + w * 64;
    const int j = key0 + lane;
    const uint16_t *kb = k + b * ks_b + (int64_t)g * ks_h;
    const float *qg = qf + (size_t)(b * Hkv + g) * (size_t)(passes * RB) * D;
    unsigned char *my = slab[w];

    for (int pass = 0; pass < passes; ++pass) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
        const float *qp = qg + (size_t)pass * RB * D;
#pragma unroll 1
        for (int ph = 0; ph < D / DH; ++ph) {
            stage_k(kb, ks_s, key0, S, ph, lane, my);
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                uint4 kr = *reinterpret_cast<const uint4 *>(my + lane * ROWB + c * 16);
                float kf[8];
                kf[0] = h2f((uint16_t)(kr.x & 0xffff)); kf[1] = h2f((uint16_t)(kr.x >> 16));
                kf[2] = h2f((uint16_t)(kr.y & 0xffff)); kf[3] = h2f((uint16_t)(kr.y >> 16));
                kf[4] = h2f((uint16_t)(kr.z & 0xffff)); kf[5] = h2f((uint16_t)(kr.z >> 16));
                kf[6] = h2f((uint16_t)(kr.w & 0xffff)); kf[7] = h2f((uint16_t)(kr.w >> 16));
                const float *qc = qp + (size_t)(ph * DH + c * 8) * RB;      // [dd][row]
                // rows in groups of 8: bounds the number of live scalar query operands (64 SGPRs per group)
#pragma unroll
                for (int r0 = 0; r0 < RB; r0 += 8) {
#pragma unroll
                    for (int r = r0; r < r0 + 8; ++r) {
#pragma unroll
                        for (int dd = 0; dd < 8; ++dd) acc[r] = __builtin_fmaf(qc[dd * RB + r], kf[dd], acc[r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // epilogue: the matmul output rounded to fp16 (utils.py:94); scaling and masking happen in row_stats
        if (j < S) {
            uint16_t *lp = logits + ((size_t)(b * H + g * G) * W + (size_t)pass * RB) * Sp + col_off + j;
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (pass * RB + r < R) lp[(size_t)r * Sp] = f2h(acc[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------ row_stats
// grid (B*H*W) rows, 1024 threads.  A "super chunk" is 32768 positions: 4 x 16-B loads per thread, all issued before any
// is used.  Pass 1 rewrites the row in place with the scaled (+masked) logits and finds the row maximum; pass 2
// accumulates sum exp(x - max) in fixed point, re-using the registers of the last super chunk (the whole row when
// S <= 32768) and re-reading only what the same thread wrote for longer rows.
// Workgroup size by row length: 1024 threads for long rows, 256 for rows up to 8192 positions (post-TSP layers), where
// a 1024-thread group would be three quarters idle and pay 16-wave barriers.
// mode 0 (one GPU): scale+mask, max, sum, then the row is rewritten with the fp16 PROBABILITIES p = fp16(e * (1/sum))
// (utils.py:103) -- one exp per element: e stays in registers between the sum and the normalisation when the row fits
// one super chunk.  Sequence sharding splits it around the two all-reduces:
// mode 1: scale+mask, local max -> gmax[row];  mode 2: sum given the global max in gmax[row] -> sums[row] (2^-40 fixed
// point);  mode 3: probabilities in place from the global max (gmax[row]) and the global sum (sums[row], NaN flag gmax[rows+row]).
template <int RS_THREADS>
__global__ void __launch_bounds__(RS_THREADS) row_stats_kernel(uint16_t *__restrict__ logits, ColWin cw, int W, int Sp, float sqrtD,
                                                               float rsqrtD, int mode, float *__restrict__ gmax,
                                                               float *__restrict__ rinv, uint64_t *__restrict__ sums,
                                                               uint32_t *__restrict__ hist_zero, int hist_words)
{
    // zero the key histograms that score_finalize / tsp_rowsum (the next kernels on the stream) accumulate into
    if (hist_zero) {
        const int per = (hist_words + gridDim.x - 1) / gridDim.x;
        const int lo = blockIdx.x * per, hi = min(lo + per, hist_words);
        for (int i = lo + threadIdx.x; i < hi; i += RS_THREADS) hist_zero[i] = 0;
    }
    constexpr int RS_SUPER = RS_THREADS * 8 * 4;
    __shared__ float smax[RS_THREADS / 64];
    __shared__ uint64_t ssum[RS_THREADS / 64];
    __shared__ int snan[RS_THREADS / 64];
    __shared__ float s_rinv;
    const int row = blockIdx.x, rw = row % W, n = cw.S_glob - W, S = cw.ncols;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint16_t *p = logits + (size_t)row * Sp;
    const int nsc = (S + RS_SUPER - 1) / RS_SUPER;
    const int jlast = ((S - 1) >> 3) << 3;
    uint4 keep[4];
    float m = -INFINITY;
    int sawnan = 0;
    for (int sc = 0; sc < (mode >= 2 ? 0 : nsc); ++sc) {
        const int base = sc * RS_SUPER + threadIdx.x * 8;
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // unconditional loads (a predicated load costs a branch and serialises the batch): out-of-row
            // vectors re-read the row's last vector and are ignored below
            const int j0 = base + u * (RS_THREADS * 8);
            raw[u] = *reinterpret_cast<const uint4 *>(p + (j0 < S ? j0 : jlast));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = base + u * (RS_THREADS * 8);
            const uint32_t wds[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            uint32_t outw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {                                         // two columns per packed instruction
                const f32x2 sc = scale_div2((f32x2){h2f((uint16_t)(wds[e >> 1] & 0xffffu)), h2f((uint16_t)(wds[e >> 1] >> 16))}, sqrtD, rsqrtD);
                uint16_t s2[2] = {f2h(sc.x), f2h(sc.y)};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = j0 + e + h, jg = cw.pos0 + j;                      // column, global position
                    if (jg >= n && (jg - n) > rw) s2[h] = f2h(h2f(s2[h]) + (-65504.0f));    // utils.py:95-101
                    if (j >= cw.own_lo && j < cw.own_hi) { const float xs = h2f(s2[h]); m = fmaxf(m, xs); sawnan |= (xs != xs); }
                }
                outw[e >> 1] = (uint32_t)s2[0] | ((uint32_t)s2[1] << 16);
            }
            keep[u] = make_uint4(outw[0], outw[1], outw[2], outw[3]);
            // (one GPU, row in one super chunk: the scaled values stay in registers; only the probabilities are stored)
            if (j0 < S && !(mode == 0 && nsc == 1)) *reinterpret_cast<uint4 *>(p + j0) = keep[u];
        }
    }
    m = wave_max(m);
    if (lane == 0) smax[w] = m;
    __syncthreads();
    m = smax[0];
#pragma unroll
    for (int u = 1; u < RS_THREADS / 64; ++u) m = fmaxf(m, smax[u]);
    if (mode == 1) {                                           // gmax[rows + row] = 1 if the owned columns hold a NaN
        sawnan = __syncthreads_or(sawnan);
        if (threadIdx.x == 0) { gmax[row] = m; gmax[gridDim.x + row] = sawnan ? 1.0f : 0.0f; }
        return;
    }
    if (mode >= 2) m = gmax[row];

    uint64_t ahi = 0, alo = 0;
    int nan = 0;
    float ev[32];                                              // e of the last super chunk processed below (sc == 0)
    for (int sc = (mode == 3 ? -1 : nsc - 1); sc >= 0; --sc) {
        const int base = sc * RS_SUPER + threadIdx.x * 8;
        if (sc != nsc - 1 || mode == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j0 = base + u * (RS_THREADS * 8);
                keep[u] = *reinterpret_cast<const uint4 *>(p + (j0 < S ? j0 : jlast));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = base + u * (RS_THREADS * 8);
            const uint32_t wds[4] = {keep[u].x, keep[u].y, keep[u].z, keep[u].w};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 ex = det_expf2((f32x2){h2f((uint16_t)(wds[e >> 1] & 0xffffu)), h2f((uint16_t)(wds[e >> 1] >> 16))} - splat2(m));
                ev[u * 8 + e] = ex.x;
                ev[u * 8 + e + 1] = ex.y;
                uint32_t h0, l0, h1, l1;
                exp_to_fix2(ex, h0, l0, h1, l1);
                if (j0 + e >= cw.own_lo && j0 + e < cw.own_hi) {
                    if (ex.x != ex.x) nan = 1;
                    else { ahi += h0; alo += l0; }
                }
                if (j0 + e + 1 >= cw.own_lo && j0 + e + 1 < cw.own_hi) {
                    if (ex.y != ex.y) nan = 1;
                    else { ahi += h1; alo += l1; }
                }
            }
        }
    }
    uint64_t tot = wave_sum_u64((ahi << 24) + alo);
    nan = __any(nan);
    if (lane == 0) { ssum[w] = tot; snan[w] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        int bad = 0;
#pragma unroll
        for (int u = 0; u < RS_THREADS / 64; ++u) { s += ssum[u]; bad |= snan[u]; }
        if (mode == 2) {
            sums[row] = bad ? FK_SUM_POISON : s;
        } else if (mode == 0) {
            const float ri = bad ? __builtin_nanf("") : 1.0f / fix_to_f32(s);
            gmax[row] = m;
            rinv[row] = ri;
            s_rinv = ri;
        } else {                                               // mode 3: globally reduced sum, NaN flag after the maxima
            const int64_t gs = (int64_t)sums[row];
            s_rinv = (gmax[gridDim.x + row] != 0.0f || gs < 0) ? __builtin_nanf("") : 1.0f / fix_to_f32((uint64_t)gs);
        }
    }
    if (mode == 2) return;
    __syncthreads();
    // ---- pass 3: probabilities in place (every column, halo columns included: score_finalize pools over them)
    const float ri = s_rinv;
    for (int sc = 0; sc < nsc; ++sc) {
        const int base = sc * RS_SUPER + threadIdx.x * 8;
        const bool have_e = (mode == 0 && sc == 0);            // pass 2 ended on super chunk 0 with its e in registers
        if (!have_e) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j0 = base + u * (RS_THREADS * 8);
                keep[u] = *reinterpret_cast<const uint4 *>(p + (j0 < S ? j0 : jlast));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j0 = base + u * (RS_THREADS * 8);
            const uint32_t wds[4] = {keep[u].x, keep[u].y, keep[u].z, keep[u].w};
            uint32_t outw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ex = have_e ? ev[u * 8 + e]
                                        : det_expf(h2f((uint16_t)((wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu)) - m);
                outw[e >> 1] |= (uint32_t)f2h(ex * ri) << ((e & 1) * 16);
            }
            if (j0 < S) *reinterpret_cast<uint4 *>(p + j0) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------ score_finalize
// grid (tilesC, Hkv, B), 256 threads; thread t <-> position tile*TP - pad + t, TP = 256 - 2*pad outputs per block.
// Reads `probs` (wr rows per query head: the W fp16 probability rows that row_stats left, or the single row of window-row
// sums that the fused kernel left): sum over the window rows -> fp16 (utils.py:104), pool -> fp16 (utils.py:105-108),
// sum over the G heads of the group -> fp16 (utils.py:112).  The values of a position are fetched in batches of up to 32
// independent loads before any arithmetic; heads go through the LDS tile four at a time (one barrier per four heads).
constexpr int FIN_HB = 4;
__global__ void __launch_bounds__(256) score_finalize_kernel(const uint16_t *__restrict__ probs, int H, int Hkv, ColWin cw, int W,
                                                             int wr, int Sp, int ksize, int pooling, uint16_t *__restrict__ c_out,
                                                             int64_t c_row_stride, uint32_t *__restrict__ hist12,
                                                             int64_t *__restrict__ all_idx, uint16_t *__restrict__ all_keys,
                                                             int64_t all_key_stride)
{
    __shared__ float s_tile[2][FIN_HB][256];
    __shared__ uint32_t s_hist[HIST12];
    const int g = blockIdx.y, b = blockIdx.z;
    // wr = rows of `probs` per query head: W (probabilities, summed here) or 1 (the fused kernel already summed them)
    const int G = H / Hkv, n = cw.S_glob - W, pad = ksize / 2, TP = 256 - 2 * pad, R = G * wr;
    const int t = threadIdx.x;
    const int j = cw.own_lo + blockIdx.x * TP - pad + t;            // column; global candidate position pos0 + j
    const bool inrange = (j >= 0) && (j < cw.ncols) && (cw.pos0 + j >= 0) && (cw.pos0 + j < n);
    const bool is_out = (t >= pad) && (t < pad + TP) && inrange && (j >= cw.own_lo) && (j < cw.own_hi);
    const size_t row0 = (size_t)(b * H + g * G) * wr;
    const uint16_t *lp = probs + row0 * Sp + (inrange ? j : 0);
    const bool want_hist = hist12 && !all_idx;
    if (want_hist) for (int i = t; i < HIST12; i += 256) s_hist[i] = 0;
    const float padv = pooling == FASTKV_POOL_AVG ? 0.0f : -INFINITY;            // padding, utils.py:106,108
    float gsum = 0.0f, a = 0.0f;
    int head = 0, rw = 0;
    for (int rb = 0; rb < R; rb += 32) {
        uint16_t x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u] = lp[(size_t)(rb + u < R ? rb + u : R - 1) * Sp];     // unconditional (clamped) loads
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int rr = rb + u;                                  // uniform
            if (rr < R) {
                a = a + h2f(x[u]);                                  // sum over the window rows (utils.py:104)
                if (++rw == wr) {                                   // last window row of head `head`
                    rw = 0;
                    s_tile[(head / FIN_HB) & 1][head % FIN_HB][t] = inrange ? h2f(f2h(a)) : padv;
                    a = 0.0f;
                    ++head;
                    if (head % FIN_HB == 0 || head == G) {          // a batch of heads is in the tile: pool them
                        const int hb = (head - 1) / FIN_HB, nh = head - hb * FIN_HB;
                        __syncthreads();
                        if (is_out) {
                            for (int hh = 0; hh < nh; ++hh) {
                                const float pv = pool_taps(s_tile[hb & 1][hh], t, pad, ksize, pooling == FASTKV_POOL_AVG);
                                gsum = gsum + h2f(f2h(pv));          // sum over the heads of the group (utils.py:112)
                            }
                        }
                        // the tile is double buffered: batch hb+2 rewrites this half only after the barrier of batch hb+1,
                        // which every thread reaches after its reads above
                    }
                }
            }
        }
    }
    const uint16_t c16 = f2h_score(gsum);
    if (is_out) c_out[(size_t)(b * Hkv + g) * c_row_stride + (j - cw.own_lo)] = c16;
    if (all_idx) {
        // k == n (post-TSP layers in constant mode): every candidate is selected, so the "selection" is the
        // identity list in ascending position plus the keys the ranking needs -- no selection kernel is launched
        const int nloc = n;                                           // one GPU only (cw.pos0 == 0, own == all)
        if (is_out) {
            all_idx[(size_t)(b * Hkv + g) * nloc + j] = (int64_t)j;
            if (all_keys) all_keys[(size_t)(b * Hkv + g) * all_key_stride + j] = (uint16_t)mono16(c16);
        }
        if (all_keys && blockIdx.x == 0 && t < (int)(all_key_stride - nloc)) all_keys[(size_t)(b * Hkv + g) * all_key_stride + nloc + t] = 0;
        return;
    }
    // high-12-bit key histogram of this row for the selection kernel: block-local first, then one global atomic per
    // non-empty bin (integer atomics: the counts do not depend on arrival order).  (Letting the thread that opens a bin
    // flush it -- no scan over the 4096 bins -- was measured slower: returning LDS atomics on the few hot bins serialise.)
    if (!want_hist) return;
    hist12_add(s_hist, mono16(c16) >> 4, is_out, t & 63);
    __syncthreads();
    uint32_t *gh = hist12 + (size_t)(b * Hkv + g) * HIST12;
    for (int i = t; i < HIST12; i += 256) { const uint32_t v = s_hist[i]; if (v) atomicAdd(&gh[i], v); }
}

// ------------------------------------------------------------------------------------------ tsp_rowsum
__global__ void __launch_bounds__(256) tsp_rowsum_kernel(const uint16_t *__restrict__ c, int64_t c_row_stride, int Hkv, int n,
                                                         uint16_t *__restrict__ t_out, int64_t t_row_stride,
                                                         uint32_t *__restrict__ thist)
{
    __shared__ uint32_t s_hist[HIST12];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < HIST12; i += 256) s_hist[i] = 0;
    __syncthreads();
    float a = 0.0f;
    const int jc = j < n ? j : n - 1;
    for (int g = 0; g < Hkv; ++g) a = a + h2f(c[(size_t)(b * Hkv + g) * c_row_stride + jc]);
    const uint16_t t16 = f2h_score(a);
    if (j < n) t_out[(size_t)b * t_row_stride + j] = t16;
    if (thist) {
        hist12_add(s_hist, mono16(t16) >> 4, j < n, threadIdx.x & 63);
        __syncthreads();
        uint32_t *gh = thist + (size_t)b * HIST12;
        for (int i = threadIdx.x; i < HIST12; i += 256) { const uint32_t v = s_hist[i]; if (v) atomicAdd(&gh[i], v); }
    }
}

// ------------------------------------------------------------------------------------------ pool_rows
// out[row, j] = pool(in[row, j - pad .. j + pad]) over fp16 rows, padding 0 (avg, count_include_pad as F.avg_pool1d) or -inf
// (max): the pooling of utils.py:105-108 as a stage of its own -- the GemFilter rule pools AFTER its head sum
// (/root/reference/baselines/gemfilter/utils.py:31-33).  fp32 taps in tap order, /kernel, -> fp16; NaN scores canonical.
__global__ void __launch_bounds__(256) pool_rows_kernel(const uint16_t *__restrict__ in, int64_t in_stride, int n, int ksize, int pooling,
                                                        uint16_t *__restrict__ out, int64_t out_stride)
{
    const int row = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int pad = ksize / 2;
    const uint16_t *r = in + (size_t)row * in_stride;
    const bool avg = pooling == FASTKV_POOL_AVG;
    float pv = avg ? 0.0f : -INFINITY;
    for (int t = j - pad; t <= j + pad; ++t) {
        if (t < 0 || t >= n) continue;                          // avg: the padding adds 0; max: -inf never wins
        const float x = h2f(r[t]);
        if (avg) pv = pv + x;
        else if (x > pv || x != x) pv = x;
    }
    if (avg) pv = pv / (float)ksize;
    out[(size_t)row * out_stride + j] = f2h_score(pv);
}

hipError_t launch_pool_rows(const uint16_t *in, int64_t in_stride, int64_t rows, int64_t n, int ksize, int pooling, uint16_t *out,
                            int64_t out_stride, hipStream_t st)
{
    if (rows == 0 || n == 0) return hipSuccess;
    hipLaunchKernelGGL(pool_rows_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, st, in, in_stride, (int)n, ksize,
                       pooling, out, out_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ launchers
// K streaming + contraction: logits[b,h,r, col_off + j] = fp16(q_r . k_j) for the p.S keys of `k` (raw, unscaled).
// q rows are q[b, h, q_row0 + r, :].  `qf` = fp32 scratch of the vector-ALU engine (unused by the matrix-pipe engine).
static hipError_t launch_logits(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, int q_row0,
                                const void *k, const int64_t *ks, float *qf, uint16_t *logits, int Sp, int col_off, hipStream_t st)
{
    const uint16_t *kp = (const uint16_t *)k;
    hipError_t e;
    if (L.engine == ENGINE_MFMA || L.engine == ENGINE_MFMA16) {
        const bool f16 = L.engine == ENGINE_MFMA16;
        ProfScope ps_(K_LOGITS, st);
        // persistent grid: 2 workgroups per CU (8 waves/CU at <=256 VGPRs), balanced over the 64-key wave tiles
        const int nwt = (p.S + 63) / 64;
        int nblk = (2 * 256) / (p.Hkv * p.B);
        if (nblk < 1) nblk = 1;
        if (nblk > (nwt + 3) / 4) nblk = (nwt + 3) / 4;
        const int per = (nwt + nblk * 4 - 1) / (nblk * 4);           // tiles per wave
        nblk = (nwt + per * 4 - 1) / (per * 4);                      // fewest workgroups with that many tiles per wave
        dim3 gridM(nblk * p.Hkv, p.B);
#define FK_LAUNCH_MFMA1(DV, FV)                                                                                                     \
    hipLaunchKernelGGL((score_logits_mfma_kernel<DV, FV>), gridM, dim3(256), 0, st, kp, ks[0], ks[1], ks[2], (const uint16_t *)q, \
                       qs[0], qs[1], qs[2], q_row0, p.H, p.Hkv, p.S, p.window, L.R, L.passes, Sp, col_off, logits)
#define FK_LAUNCH_MFMA(DV) do { if (f16) FK_LAUNCH_MFMA1(DV, true); else FK_LAUNCH_MFMA1(DV, false); } while (0)
        if (p.D == 64) FK_LAUNCH_MFMA(64);
        else if (p.D == 128) FK_LAUNCH_MFMA(128);
        else FK_LAUNCH_MFMA(256);
#undef FK_LAUNCH_MFMA
#undef FK_LAUNCH_MFMA1
        return hipGetLastError();
    }
    {
        ProfScope ps_(K_PREP_Q, st);
        // prep_q indexes the query rows as (S - W + r): pass S = q_row0 + W
        hipLaunchKernelGGL(prep_q_kernel, dim3(L.R_alloc, p.B * p.Hkv), dim3(p.D), 0, st, (const uint16_t *)q, qs[0], qs[1], qs[2],
                           p.H, p.Hkv, q_row0 + p.window, p.D, p.window, L.R, L.R_alloc, L.RB, qf);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    ProfScope ps_(K_LOGITS, st);
    dim3 gridA(((p.S + TKA - 1) / TKA) * p.Hkv, p.B);
#define FK_LAUNCH_VALU(DV, RBV)                                                                                              \
    hipLaunchKernelGGL((score_logits_kernel<DV, RBV>), gridA, dim3(256), 0, st, kp, ks[0], ks[1], ks[2], qf, p.H, p.Hkv, p.S, \
                       p.window, L.R, L.passes, Sp, col_off, logits)
#define FK_LAUNCH_VALU_D(DV)                                                                \
    switch (L.RB) {                                                                         \
    case 8: FK_LAUNCH_VALU(DV, 8); break;                                                   \
    case 16: FK_LAUNCH_VALU(DV, 16); break;                                                 \
    case 32: FK_LAUNCH_VALU(DV, 32); break;                                                 \
    default: FK_LAUNCH_VALU(DV, 64); break;                                                 \
    }
    if (p.D == 64) { FK_LAUNCH_VALU_D(64) }
    else if (p.D == 128) { FK_LAUNCH_VALU_D(128) }
    else { FK_LAUNCH_VALU_D(256) }
#undef FK_LAUNCH_VALU_D
#undef FK_LAUNCH_VALU
    return hipGetLastError();
}

// After a fused score launch the workspace epoch must advance before the next one (fused.hip: the hand-off token).  In the
// whole operator the compaction kernel does it; the stand-alone scoring entry point launches this single thread.
__global__ void epoch_bump_kernel(uint32_t *epoch)
{
    const uint32_t e = *epoch + (uint32_t)EPOCH_STRIDE;
    *epoch = e ? e : (uint32_t)EPOCH_STRIDE;
}

hipError_t launch_epoch_bump(uint32_t *epoch, hipStream_t st)
{
    hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, st, epoch);
    return hipGetLastError();
}

hipError_t launch_score(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, const void *k,
                        const int64_t *ks, uint16_t *c_out, int64_t c_row_stride, uint16_t *t_out, int64_t t_row_stride,
                        char *ws, hipStream_t st, int64_t *all_idx, uint16_t *all_keys, int64_t all_key_stride,
                        uint32_t **epoch_bump_later, const PtrTables *pt)
{
    float *qf = reinterpret_cast<float *>(ws + L.off_qf);
    uint16_t *logits = reinterpret_cast<uint16_t *>(ws + L.off_logits);
    float *gmax = reinterpret_cast<float *>(ws + L.off_gmax);
    float *rinv = reinterpret_cast<float *>(ws + L.off_rinv);
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws + L.off_hist);      // [B*Hkv + B][HIST12], TSP rows last
    const float sqrtD = (float)sqrt((double)p.D);
    const float rsqrtD = 1.0f / sqrtD;
    const ColWin cw = {p.S, 0, 0, p.S, p.S};
    hipError_t e;

    if (epoch_bump_later) *epoch_bump_later = nullptr;
    // one launch for the common geometry (the `logits` area then only holds the workgroups' halo granules) ...
    const bool fused = launch_score_fused(p, L, q, qs, k, ks, c_out, c_row_stride, all_idx, all_keys, all_key_stride,
                                          ws, st, &e, pt);
    if (fused) {
        if (e != hipSuccess) return e;
    } else if (pt) {
        return hipErrorNotSupported;                             // per-entry base addresses exist on the fused path only
    } else {
        // ... or logits -> row statistics / probabilities -> window-row sum + pooling + head sum
        if ((e = launch_logits(p, L, q, qs, p.S - p.window, k, ks, qf, logits, L.Sp, 0, st)) != hipSuccess) return e;
        {
            ProfScope ps_(K_ROWSTATS, st);
            if (p.S <= 8192)
                hipLaunchKernelGGL(row_stats_kernel<256>, dim3(p.B * p.H * p.window), dim3(256), 0, st, logits, cw, p.window, L.Sp,
                                   sqrtD, rsqrtD, 0, gmax, rinv, (uint64_t *)nullptr, hist, L.zero_words);
            else
                hipLaunchKernelGGL(row_stats_kernel<1024>, dim3(p.B * p.H * p.window), dim3(1024), 0, st, logits, cw, p.window, L.Sp,
                                   sqrtD, rsqrtD, 0, gmax, rinv, (uint64_t *)nullptr, hist, L.zero_words);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        ProfScope ps_(K_FINALIZE, st);
        const int pad = p.kernel / 2, TP = 256 - 2 * pad;
        dim3 gridC((L.n + TP - 1) / TP, p.Hkv, p.B);
        hipLaunchKernelGGL(score_finalize_kernel, gridC, dim3(256), 0, st, logits, p.H, p.Hkv, cw, p.window, p.window, L.Sp, p.kernel,
                           p.pooling, c_out, c_row_stride, hist, all_idx, all_keys, all_key_stride);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (t_out) {
        ProfScope ps_(K_TSP_ROWSUM, st);
        hipLaunchKernelGGL(tsp_rowsum_kernel, dim3((L.n + 255) / 256, p.B), dim3(256), 0, st, c_out, c_row_stride, p.Hkv, L.n, t_out,
                           t_row_stride, hist + (size_t)p.B * p.Hkv * HIST12);
        if ((e = hipGetLastError()) != hipSuccess) {
            if (fused) (void)launch_epoch_bump(reinterpret_cast<uint32_t *>(ws) + 2, st);     // the token of the fused launch is spent
            return e;
        }
    }
    if (fused) {
        uint32_t *epoch = reinterpret_cast<uint32_t *>(ws) + 2;
        if (epoch_bump_later) *epoch_bump_later = epoch;
        else hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, st, epoch);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_head_sum(const uint16_t *c, int64_t B, int64_t R, int64_t n, uint16_t *t_out, hipStream_t st)
{
    if (B == 0 || n == 0) return hipSuccess;
    ProfScope ps_(K_TSP_ROWSUM, st);
    hipLaunchKernelGGL(tsp_rowsum_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, c, n, (int)R, (int)n, t_out,
                       n, (uint32_t *)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ sequence-sharded stages
// (fastkv_amd/dist.py; include/fastkv_hip.h "fastkv_sp_*").  `p` describes the rank-local call: p.S = keys in `k`.
hipError_t launch_sp_logits(const fastkv_problem &p, const void *q_win, const int64_t *qs, const void *k, const int64_t *ks,
                            uint16_t *logits, int Sp, int col_off, float *qf_scratch, hipStream_t st)
{
    fastkv_problem pp = p;
    pp.capacity = pp.S;
    const Layout L = make_layout(pp);
    return launch_logits(pp, L, q_win, qs, 0, k, ks, qf_scratch, logits, Sp, col_off, st);
}

hipError_t launch_sp_rowstats(const fastkv_problem &p, uint16_t *logits, const fastkv_sp_window &w, int mode, float *gmax,
                              int64_t *sums, hipStream_t st)
{
    const float sqrtD = (float)sqrt((double)p.D);
    const ColWin cw = {w.ncols, w.pos0, w.own_lo, w.own_hi, w.S_glob};
    ProfScope ps_(K_ROWSTATS, st);
    hipLaunchKernelGGL(row_stats_kernel<1024>, dim3(p.B * p.H * p.window), dim3(1024), 0, st, logits, cw, p.window, w.Sp, sqrtD,
                       1.0f / sqrtD, mode, gmax, (float *)nullptr, (uint64_t *)sums, (uint32_t *)nullptr, 0);
    return hipGetLastError();
}

hipError_t launch_sp_scores(const fastkv_problem &p, uint16_t *logits, const fastkv_sp_window &w, const float *gmax,
                            const int64_t *sums, uint16_t *c_out, int64_t c_row_stride, uint16_t *t_out,
                            int64_t t_row_stride, int n_own, hipStream_t st)
{
    const ColWin cw = {w.ncols, w.pos0, w.own_lo, w.own_hi, w.S_glob};
    const float sqrtD = (float)sqrt((double)p.D);
    hipError_t e;
    {
        ProfScope ps_(K_ROWSTATS, st);   // probabilities in place from the globally reduced max / sum
        hipLaunchKernelGGL(row_stats_kernel<1024>, dim3(p.B * p.H * p.window), dim3(1024), 0, st, logits, cw, p.window, w.Sp, sqrtD,
                           1.0f / sqrtD, 3, const_cast<float *>(gmax), (float *)nullptr, (uint64_t *)sums, (uint32_t *)nullptr, 0);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (n_own <= 0) return hipSuccess;
    {
        ProfScope ps_(K_FINALIZE, st);
        const int pad = p.kernel / 2, TP = 256 - 2 * pad;
        dim3 gridC((n_own + TP - 1) / TP, p.Hkv, p.B);
        hipLaunchKernelGGL(score_finalize_kernel, gridC, dim3(256), 0, st, logits, p.H, p.Hkv, cw, p.window, p.window, w.Sp, p.kernel,
                           p.pooling, c_out, c_row_stride, (uint32_t *)nullptr, (int64_t *)nullptr, (uint16_t *)nullptr, (int64_t)0);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (t_out) {
        ProfScope ps_(K_TSP_ROWSUM, st);
        hipLaunchKernelGGL(tsp_rowsum_kernel, dim3((n_own + 255) / 256, p.B), dim3(256), 0, st, c_out, c_row_stride, p.Hkv, n_own,
                           t_out, t_row_stride, (uint32_t *)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

#ifdef FK_STAMP
}  // namespace fk
extern "C" int fastkv_debug_read_stamps(unsigned long long *host, size_t n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::g_stamps), n * sizeof(unsigned long long));
}
namespace fk {
#endif
}  // namespace fk
