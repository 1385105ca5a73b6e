// Weight-streaming GEMV for the decode step over the compressed cache (the loop being served is
// /root/reference/benchmark/e2e.py:72-93: one token per step, so every projection of the model is y = W x with ONE x row per
// batch element and the step is bound by reading the 16 GB of fp16 weights once).  One launch covers what the stock modules
// spread over 3-4:
//   * up to three matrices that share the input (q_proj / k_proj / v_proj), rows concatenated in the output;
//   * optionally RMSNorm of the input first (LlamaRMSNorm arithmetic: fp32 mean of squares, x * rsqrt(var + eps) -> fp16,
//     times the fp16 weight -> fp16), recomputed by every workgroup from the 8 KB row (L2 resident) -- cheaper than a launch;
//   * optionally the gated-MLP epilogue out = fp16(fp16(silu(gate)) * up) over a (gate, up) pair of matrices;
//   * optionally the residual add out = fp16(fp16(acc) + residual).
// Rounding points are those of the stock fp16 modules (a Linear rounds its fp32 accumulator to fp16 before anything else
// touches it); the accumulation ORDER differs from hipBLASLt's, so results agree to fp16 tolerance, not bit for bit
// (tests/test_decode_gpu.py).
//
// HBM bound.  A wave owns NR weight rows at a time and streams them with 16-B non-temporal loads, lane l taking bytes
// [16 l, 16 l + 16) of every 1 KB row segment: U segments x NR rows = 16 loads per lane in flight, 16 waves per CU.  The input
// row sits in LDS as fp16 (one ds_read_b128 per segment, shared by the NR rows); products accumulate in fp32 through
// v_dot2c_f32_f16.  Algorithmic bytes: N*K*2 per launch (+ K*2 per workgroup for the input row, from L2).
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"
#include <cstdlib>

namespace fk {

typedef _Float16 gv_h2 __attribute__((ext_vector_type(2)));
typedef uint32_t gv_u4 __attribute__((ext_vector_type(4)));

struct GemvArgs {
    const uint16_t *x; int64_t x_row;              // [BB][K]
    const uint16_t *nw; float eps;                 // RMSNorm weight [K] or nullptr
    const uint16_t *w0, *w1, *w2; int n0, n1, n2;  // matrices [n_i][K], contiguous rows
    int glu;                                       // out[b][n] = silu(w0 x)[n] * (w1 x)[n]   (n0 == n1, w2 unused)
    const uint16_t *res; int64_t res_row;          // residual [BB][N] or nullptr
    uint16_t *out; int64_t out_row;                // [BB][N]
    int K;
};

// (the elements are copied to scalars first: __builtin_bit_cast applied directly to a vector element expression `w.y` reads
// element 0 with this compiler)
__device__ __forceinline__ float gv_dot8(gv_u4 wv, gv_u4 xv, float acc)
{
    const uint32_t w0 = wv[0], w1 = wv[1], w2 = wv[2], w3 = wv[3], x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, w0), __builtin_bit_cast(gv_h2, x0), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, w1), __builtin_bit_cast(gv_h2, x1), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, w2), __builtin_bit_cast(gv_h2, x2), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, w3), __builtin_bit_cast(gv_h2, x3), acc, false);
    return acc;
}

// BB = batch rows (1, 2, 4); NR = weight rows in flight per wave (with glu: NR / 2 gate rows + their NR / 2 up rows)
template <int BB, int NR, int U>
__global__ void __launch_bounds__(256) decode_gemv_kernel(GemvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t s_x[];          // [BB][K]
    __shared__ float s_part[BB][4];
    const int K = a.K, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int OR = NR;                                                   // rows handled per wave and iteration
    const int glu = a.glu;
    const int nout = glu ? a.n0 : a.n0 + a.n1 + a.n2;                        // output columns
    const int orows = glu ? NR / 2 : NR;                                     // output columns per wave and iteration
    const uint16_t *wr[OR];
    auto set_rows = [&](int r0) {
#pragma unroll
        for (int i = 0; i < OR; ++i) {
            int r = r0 + (glu ? (i >> 1) : i);                               // glu: rows 2j / 2j+1 = gate / up of column r0 + j
            r = r < nout ? r : nout - 1;
            const uint16_t *base;
            if (glu) base = (i & 1) ? a.w1 : a.w0;
            else if (r < a.n0) base = a.w0;
            else if (r < a.n0 + a.n1) { base = a.w1; r -= a.n0; }
            else { base = a.w2; r -= a.n0 + a.n1; }
            wr[i] = base + (int64_t)r * K + lane * 8;
        }
    };
    // The weights do not depend on the input: the first U segments of the wave's first rows are requested BEFORE the input row
    // is staged (and normalised), so the launch pays one memory round trip, not two.
    const int r_first = (blockIdx.x * 4 + w) * orows;
    const bool pre = r_first < nout && K >= 512 * U;
    gv_u4 wv[OR][U];
    if (pre) {
        set_rows(r_first);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < OR; ++i) wv[i][u] = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(wr[i] + u * 512));
    }
    // ---- the input row(s) -> LDS, normalised on the way if asked
    if (a.nw) {
        float ss[BB];
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            ss[b] = 0.0f;
            for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
                const uint4 v = *reinterpret_cast<const uint4 *>(a.x + b * a.x_row + i);
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = h2f((uint16_t)(wds[e] & 0xffffu)), q = h2f((uint16_t)(wds[e] >> 16));
                    ss[b] = __builtin_fmaf(p, p, ss[b]);
                    ss[b] = __builtin_fmaf(q, q, ss[b]);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss[b] += __shfl_xor(ss[b], o, 64);
            if (lane == 0) s_part[b][w] = ss[b];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            const float var = (s_part[b][0] + s_part[b][1] + s_part[b][2] + s_part[b][3]) / (float)K;
            const float r = 1.0f / __builtin_sqrtf(var + a.eps);
            for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
                const uint4 v = *reinterpret_cast<const uint4 *>(a.x + b * a.x_row + i), g = *reinterpret_cast<const uint4 *>(a.nw + i);
                const uint32_t wx[4] = {v.x, v.y, v.z, v.w}, wg[4] = {g.x, g.y, g.z, g.w};
                uint32_t o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t n0 = f2h(h2f((uint16_t)(wx[e] & 0xffffu)) * r), n1 = f2h(h2f((uint16_t)(wx[e] >> 16)) * r);
                    const uint16_t y0 = f2h(h2f((uint16_t)(wg[e] & 0xffffu)) * h2f(n0)), y1 = f2h(h2f((uint16_t)(wg[e] >> 16)) * h2f(n1));
                    o4[e] = (uint32_t)y0 | ((uint32_t)y1 << 16);
                }
                *reinterpret_cast<uint4 *>(s_x + b * K + i) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < BB; ++b)
            for (int i = threadIdx.x * 8; i < K; i += 256 * 8)
                *reinterpret_cast<uint4 *>(s_x + b * K + i) = *reinterpret_cast<const uint4 *>(a.x + b * a.x_row + i);
    }
    __syncthreads();

    for (int r0 = r_first; r0 < nout; r0 += gridDim.x * 4 * orows) {
        float acc[OR][BB];
#pragma unroll
        for (int i = 0; i < OR; ++i)
#pragma unroll
            for (int b = 0; b < BB; ++b) acc[i][b] = 0.0f;
        // K is a multiple of 512 (host check): whole 1 KB segments only, U at a time, then the remaining ones singly
        int c = 0;
        bool loaded = pre && r0 == r_first;                                  // the first batch is already on its way
        if (!loaded) set_rows(r0);
        for (; c + 512 * U <= K; c += 512 * U) {
            if (!loaded) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < OR; ++i) wv[i][u] = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(wr[i] + c + u * 512));
            }
            loaded = false;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int b = 0; b < BB; ++b) {
                    const gv_u4 xv = *reinterpret_cast<const gv_u4 *>(s_x + b * K + c + u * 512 + lane * 8);
#pragma unroll
                    for (int i = 0; i < OR; ++i) acc[i][b] = gv_dot8(wv[i][u], xv, acc[i][b]);
                }
        }
        // (what is left of K: 4 segments at a time where U is 8 -- K = 14336 is three batches of 8 and one of 4 --, then singly)
        if (U >= 8) {
            for (; c + 512 * 4 <= K; c += 512 * 4) {
                gv_u4 wq[OR][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < OR; ++i) wq[i][u] = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(wr[i] + c + u * 512));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int b = 0; b < BB; ++b) {
                        const gv_u4 xv = *reinterpret_cast<const gv_u4 *>(s_x + b * K + c + u * 512 + lane * 8);
#pragma unroll
                        for (int i = 0; i < OR; ++i) acc[i][b] = gv_dot8(wq[i][u], xv, acc[i][b]);
                    }
            }
        }
        for (; c < K; c += 512) {
            gv_u4 wt[OR];
#pragma unroll
            for (int i = 0; i < OR; ++i) wt[i] = __builtin_nontemporal_load(reinterpret_cast<const gv_u4 *>(wr[i] + c));
#pragma unroll
            for (int b = 0; b < BB; ++b) {
                const gv_u4 xv = *reinterpret_cast<const gv_u4 *>(s_x + b * K + c + lane * 8);
#pragma unroll
                for (int i = 0; i < OR; ++i) acc[i][b] = gv_dot8(wt[i], xv, acc[i][b]);
            }
        }
#pragma unroll
        for (int i = 0; i < OR; ++i)
#pragma unroll
            for (int b = 0; b < BB; ++b)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc[i][b] += __shfl_xor(acc[i][b], o, 64);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < BB; ++b) {
                if (glu) {
#pragma unroll
                    for (int j = 0; j < OR / 2; ++j) {
                        const int r = r0 + j;
                        if (r >= nout) continue;
                        const float g = h2f(f2h(acc[2 * j][b]));               // gate_proj / up_proj outputs are fp16 tensors
                        const uint16_t s16 = f2h(g / (1.0f + __expf(-g)));
                        a.out[b * a.out_row + r] = f2h(h2f(s16) * h2f(f2h(acc[2 * j + 1][b])));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < OR; ++i) {
                        const int r = r0 + i;
                        if (r >= nout) continue;
                        uint16_t y = f2h(acc[i][b]);
                        if (a.res) y = f2h(h2f(a.res[b * a.res_row + r]) + h2f(y));
                        a.out[b * a.out_row + r] = y;
                    }
                }
            }
        }
    }
}

}  // namespace fk

using namespace fk;

extern "C" int fastkv_decode_gemv_f16(int32_t B, int32_t K, const void *x, int64_t x_row_stride, const void *norm_weight, float eps,
                                      int32_t n_mats, const void *const *weights, const int32_t *rows, int32_t glu,
                                      const void *residual, int64_t res_row_stride, void *out, int64_t out_row_stride, void *stream)
{
    if (B < 1 || K < 512 || (K & 511) || !x || !weights || !rows || !out || n_mats < 1 || n_mats > 3 || (x_row_stride & 7)) return FASTKV_EINVAL;
    if (B != 1 && B != 2 && B != 4) return FASTKV_EUNSUPPORTED;
    if ((size_t)B * K * 2 > 64 * 1024 - 256) return FASTKV_EUNSUPPORTED;           // the input rows live in LDS (+ a few static words)
    if (glu && (n_mats != 2 || rows[0] != rows[1] || residual)) return FASTKV_EINVAL;
    GemvArgs a = {};
    a.x = (const uint16_t *)x; a.x_row = x_row_stride;
    a.nw = (const uint16_t *)norm_weight; a.eps = eps;
    const uint16_t *wp[3] = {nullptr, nullptr, nullptr};
    int nn[3] = {0, 0, 0};
    uintptr_t al = (uintptr_t)x | (uintptr_t)norm_weight;
    for (int i = 0; i < n_mats; ++i) {
        if (!weights[i] || rows[i] < 1) return FASTKV_EINVAL;
        wp[i] = (const uint16_t *)weights[i];
        nn[i] = rows[i];
        al |= (uintptr_t)weights[i];
    }
    if (al & 15) return FASTKV_EINVAL;
    a.w0 = wp[0]; a.w1 = wp[1]; a.w2 = wp[2]; a.n0 = nn[0]; a.n1 = nn[1]; a.n2 = nn[2];
    a.glu = glu ? 1 : 0;
    a.res = (const uint16_t *)residual; a.res_row = res_row_stride;
    a.out = (uint16_t *)out; a.out_row = out_row_stride;
    a.K = K;
    const int64_t nout = glu ? nn[0] : (int64_t)nn[0] + nn[1] + nn[2];
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps_(K_DECODE, st);
    // rows in flight per wave: 4 x 4 segments for wide outputs, 2 x 8 for narrow ones (more workgroups; either way 16 loads
    // of 16 B per lane are outstanding).  Workgroups are sized so that every one runs the same number of iterations.
    static const int force_cfg = []() { const char *e = getenv("FASTKV_GEMV_CFG"); return e ? atoi(e) : 0; }();   // measurement aid: 1 narrow, 2 wide, 3 = 4 rows x 8 segments
    const bool wide = force_cfg ? force_cfg >= 2 : (glu || nout >= 16384);
    const int nr = wide ? 4 : 2;
    const int orows = glu ? nr / 2 : nr;
    const int64_t iters = (nout + 4 * orows - 1) / (4 * orows);
    const int64_t per_wg = (iters + 2047) / 2048;
    const int64_t wgs = (iters + per_wg - 1) / per_wg;
    const size_t lds = (size_t)B * K * 2;
#define FK_GEMV(BBV)                                                                                             \
    do {                                                                                                         \
        if (force_cfg == 3) hipLaunchKernelGGL((decode_gemv_kernel<BBV, 4, 8>), dim3((unsigned)wgs), dim3(256), lds, st, a); \
        else if (wide) hipLaunchKernelGGL((decode_gemv_kernel<BBV, 4, 4>), dim3((unsigned)wgs), dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((decode_gemv_kernel<BBV, 2, 8>), dim3((unsigned)wgs), dim3(256), lds, st, a);      \
    } while (0)
    if (B == 1) FK_GEMV(1);
    else if (B == 2) FK_GEMV(2);
    else FK_GEMV(4);
#undef FK_GEMV
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}
