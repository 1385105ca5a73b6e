// Window-attention scoring: replaces ops k1-k10 of FastKVCluster.update_kv
// (/root/reference/baselines/fastkv/utils.py:93-112, and the head sum of :127).
//
//   prep_q          Q[b,h,S-W+r,:] fp16 (strided) -> qf[b,g,i*W+r,:] fp32, zero padded to R_alloc rows
//   score_logits    streams K ONCE (no repeat_kv materialisation): one key row per lane, the G*W
//                   window-query vectors are wave-uniform scalar operands; fp32 fma chain over d;
//                   writes scaled+masked fp16 logits L[b,h,r,j] and per-tile row maxima
//   score_sumexp    sum_j exp(L - rowmax) in 2^-40 fixed point (order-free, deterministic)
//   score_finalize  p = e/sum -> fp16, sum over window rows -> fp16, pool -> fp16, sum over the
//                   G heads of the group -> fp16 = attn_cache c[b,g,j]
//   tsp_rowsum      t[b,j] = fp16(sum_g c[b,g,j])   (utils.py:127)
//
// HBM traffic: K read once (B*Hkv*S*D*2 bytes); the logits (B*H*W*S*2 bytes, 16 MiB at the
// 32k config) round-trip through L2 / Infinity Cache.  No MFMA: the contraction is done with
// v_fma_f32 / v_pk_fma_f32 so that the result is bit-identical to the CPU oracle.
#include "fk_device.h"
#include "fk_host.h"

namespace fk {

// ------------------------------------------------------------------------------------------ prep_q
__global__ void __launch_bounds__(256) prep_q_kernel(const uint16_t *__restrict__ q, int64_t qs_b, int64_t qs_h, int64_t qs_s,
                                                     int H, int Hkv, int S, int D, int W, int R, int R_alloc,
                                                     float *__restrict__ qf)
{
    const int bg = blockIdx.x;
    const int b = bg / Hkv, g = bg % Hkv, G = H / Hkv;
    float *dst = qf + (size_t)bg * R_alloc * D;
    for (int e = threadIdx.x; e < R_alloc * D; e += blockDim.x) {
        int row = e / D, d = e - row * D;
        float v = 0.0f;
        if (row < R) {
            int i = row / W, r = row - i * W;
            v = h2f(q[b * qs_b + (int64_t)(g * G + i) * qs_h + (int64_t)(S - W + r) * qs_s + d]);
        }
        dst[e] = v;
    }
}

// ------------------------------------------------------------------------------------------ score_logits
// grid.x = ntA * Hkv with blockIdx.x % Hkv = kv head (the 8 workgroups that stream the same token range run
// together, one per XCD under round-robin placement, so an XCD's L2 keeps one head's query block: speed only),
// grid.y = B.  256 threads = 4 independent waves, each owning 64 consecutive keys (one key row per lane).
// Per phase a wave stages 64 dims of its 64 rows through a private LDS slab (coalesced 16-B global loads,
// 144-B padded rows -> conflict-free ds_read_b128 with one row per lane).
template <int D, int RB>
__global__ void __launch_bounds__(256) score_logits_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                           const float *__restrict__ qf, int H, int Hkv, int S, int W, int R,
                                                           int passes, int Sp, int ntA, float sqrtD,
                                                           uint16_t *__restrict__ logits, float *__restrict__ pm)
{
    constexpr int DH = 64;
    constexpr int ROWB = DH * 2 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][64 * ROWB];
    __shared__ float red[4][RB];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x % Hkv, tile = blockIdx.x / Hkv, b = blockIdx.y;
    const int G = H / Hkv, n = S - W;
    const int j = tile * TKA + w * 64 + lane;
    const bool valid = j < S;
    const uint16_t *kb = k + b * ks_b + (int64_t)g * ks_h;
    const float *qg = qf + (size_t)(b * Hkv + g) * (size_t)(passes * RB) * D;
    unsigned char *my = slab[w];

    const int lrow = lane >> 3, lchunk = lane & 7;
    for (int pass = 0; pass < passes; ++pass) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
        const float *qp = qg + (size_t)pass * RB * D;
#pragma unroll 1
        for (int ph = 0; ph < D / DH; ++ph) {
            uint4 st[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int jj = tile * TKA + w * 64 + i * 8 + lrow;
                jj = jj < S ? jj : S - 1;
                st[i] = *reinterpret_cast<const uint4 *>(kb + (int64_t)jj * ks_s + ph * DH + lchunk * 8);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint4 *>(my + (i * 8 + lrow) * ROWB + lchunk * 16) = st[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                uint4 kr = *reinterpret_cast<const uint4 *>(my + lane * ROWB + c * 16);
                float kf[8];
                kf[0] = h2f((uint16_t)(kr.x & 0xffff)); kf[1] = h2f((uint16_t)(kr.x >> 16));
                kf[2] = h2f((uint16_t)(kr.y & 0xffff)); kf[3] = h2f((uint16_t)(kr.y >> 16));
                kf[4] = h2f((uint16_t)(kr.z & 0xffff)); kf[5] = h2f((uint16_t)(kr.z >> 16));
                kf[6] = h2f((uint16_t)(kr.w & 0xffff)); kf[7] = h2f((uint16_t)(kr.w >> 16));
                const float *qc = qp + ph * DH + c * 8;
#pragma unroll
                for (int r = 0; r < RB; ++r) {
#pragma unroll
                    for (int dd = 0; dd < 8; ++dd) acc[r] = __builtin_fmaf(qc[r * D + dd], kf[dd], acc[r]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // epilogue: round, scale by true division, round, window mask (utils.py:94-101), row maxima
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = pass * RB + r;            // wave-uniform
            float mval = -INFINITY;
            if (row < R) {
                const int i = row / W, rw = row - i * W;
                uint16_t l16 = f2h(acc[r]);
                uint16_t s16 = f2h(h2f(l16) / sqrtD);
                if (j >= n && (j - n) > rw) s16 = f2h(h2f(s16) + (-65504.0f));
                if (valid) {
                    logits[((size_t)(b * H + g * G + i) * W + rw) * Sp + j] = s16;
                    mval = h2f(s16);
                }
            }
            mval = wave_max(mval);
            if (lane == 0) red[w][r] = mval;
        }
        __syncthreads();
        if (threadIdx.x < RB) {
            const int row = pass * RB + threadIdx.x;
            if (row < R) {
                const int i = row / W, rw = row - i * W;
                float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
                pm[((size_t)(b * H + g * G + i) * W + rw) * ntA + tile] = m;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ score_sumexp
// grid (nchB, B*H); each workgroup: all W rows of one head over 2048 positions.
__global__ void __launch_bounds__(256) score_sumexp_kernel(const uint16_t *__restrict__ logits, const float *__restrict__ pm,
                                                           int S, int W, int Sp, int ntA, int nchB, uint64_t *__restrict__ ps)
{
    __shared__ float gmax_s[64];
    __shared__ uint64_t wsum[4];
    __shared__ int wnan[4];
    const int bh = blockIdx.y, ch = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // row maxima: wave w reduces rows w, w+4, ...
    for (int r = w; r < W; r += 4) {
        const float *p = pm + ((size_t)bh * W + r) * ntA;
        float m = -INFINITY;
        for (int t = lane; t < ntA; t += 64) m = fmaxf(m, p[t]);
        m = wave_max(m);
        if (lane == 0) gmax_s[r] = m;
    }
    __syncthreads();
    const int j0 = ch * CHB + threadIdx.x * 8;
    for (int r = 0; r < W; ++r) {
        const float m = gmax_s[r];
        uint32_t ahi = 0, alo = 0;
        int nan = 0;
        if (j0 < S) {
            uint4 raw = *reinterpret_cast<const uint4 *>(logits + ((size_t)bh * W + r) * Sp + j0);
            uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                if (j0 + e8 < S) {
                    uint16_t hb = (uint16_t)((wds[e8 >> 1] >> ((e8 & 1) * 16)) & 0xffff);
                    float e = det_expf(h2f(hb) - m);
                    if (e != e) { nan = 1; }
                    else { uint32_t hi, lo; exp_to_fix(e, hi, lo); ahi += hi; alo += lo; }
                }
            }
        }
        uint64_t tot = ((uint64_t)wave_sum_u32(ahi) << 24) + wave_sum_u64((uint64_t)alo);   // hi <= 64*8*2^16 fits u32; lo needs 64 bits
        nan = __any(nan);
        if (lane == 0) { wsum[w] = tot; wnan[w] = nan; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (wnan[0] | wnan[1] | wnan[2] | wnan[3]) s = FK_SUM_POISON;
            ps[((size_t)bh * W + r) * nchB + ch] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ score_finalize
// grid (tilesC, Hkv, B), 256 threads; thread t <-> position tile*TP - pad + t, TP = 256 - 2*pad outputs per block.
__global__ void __launch_bounds__(256) score_finalize_kernel(const uint16_t *__restrict__ logits, const float *__restrict__ pm,
                                                             const uint64_t *__restrict__ ps, int H, int Hkv, int S, int W,
                                                             int Sp, int ntA, int nchB, int ksize, int pooling,
                                                             uint16_t *__restrict__ c_out, int64_t c_row_stride)
{
    extern __shared__ float dyn[];                   // gmax[G*W], rinv[G*W]
    __shared__ float s_tile[2][256];
    const int g = blockIdx.y, b = blockIdx.z;
    const int G = H / Hkv, n = S - W, pad = ksize / 2, TP = 256 - 2 * pad;
    const int R = G * W;
    float *gmax = dyn, *rinv = dyn + R;
    // row statistics: 8 lanes per row
    {
        const int sub = threadIdx.x & 7;
        for (int row = threadIdx.x >> 3; row < R; row += 32) {
            const int i = row / W, rw = row - i * W;
            const size_t rid = (size_t)(b * H + g * G + i) * W + rw;
            float m = -INFINITY;
            for (int t = sub; t < ntA; t += 8) m = fmaxf(m, pm[rid * ntA + t]);
            uint64_t s = 0;
            int poison = 0;
            for (int t = sub; t < nchB; t += 8) { uint64_t v = ps[rid * nchB + t]; if (v == FK_SUM_POISON) poison = 1; else s += v; }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                m = fmaxf(m, __shfl_xor(m, o, 64));
                s += __shfl_xor(s, o, 64);
                poison |= __shfl_xor(poison, o, 64);
            }
            if (sub == 0) { gmax[row] = m; rinv[row] = poison ? __builtin_nanf("") : 1.0f / fix_to_f32(s); }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    const int j = blockIdx.x * TP - pad + t;
    const bool inrange = (j >= 0) && (j < n);
    const bool is_out = (t >= pad) && (t < pad + TP) && inrange;
    float gsum = 0.0f;
    for (int i = 0; i < G; ++i) {
        float a;
        if (inrange) {
            a = 0.0f;
            const uint16_t *lp = logits + (size_t)(b * H + g * G + i) * W * Sp + j;
            for (int r = 0; r < W; ++r) {
                float e = det_expf(h2f(lp[(size_t)r * Sp]) - gmax[i * W + r]);
                a = a + h2f(f2h(e * rinv[i * W + r]));
            }
            a = h2f(f2h(a));
        } else {
            a = pooling == FASTKV_POOL_AVG ? 0.0f : -INFINITY;     // zero / -inf padding (utils.py:106,108)
        }
        float *st = s_tile[i & 1];
        st[t] = a;
        __syncthreads();
        if (is_out) {
            float pv;
            if (pooling == FASTKV_POOL_AVG) {
                pv = 0.0f;
                for (int u = -pad; u <= pad; ++u) pv = pv + st[t + u];
                pv = pv / (float)ksize;
            } else {
                pv = -INFINITY;
                for (int u = -pad; u <= pad; ++u) { float x = st[t + u]; if (x > pv || x != x) pv = x; }
            }
            gsum = gsum + h2f(f2h(pv));
        }
        // s_tile is double buffered: head i+2 rewrites this half only after the barrier of head i+1,
        // which every thread reaches after its reads above
    }
    if (is_out) c_out[(size_t)(b * Hkv + g) * c_row_stride + j] = f2h(gsum);
}

// ------------------------------------------------------------------------------------------ tsp_rowsum
__global__ void __launch_bounds__(256) tsp_rowsum_kernel(const uint16_t *__restrict__ c, int64_t c_row_stride, int Hkv, int n,
                                                         uint16_t *__restrict__ t_out, int64_t t_row_stride)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float a = 0.0f;
    for (int g = 0; g < Hkv; ++g) a = a + h2f(c[(size_t)(b * Hkv + g) * c_row_stride + j]);
    t_out[(size_t)b * t_row_stride + j] = f2h(a);
}

// ------------------------------------------------------------------------------------------ launcher
template <int D>
static hipError_t launch_logits_d(int RB, dim3 grid, hipStream_t st, const uint16_t *k, const int64_t *ks, const float *qf,
                                  const fastkv_problem &p, const Layout &L, float sqrtD, uint16_t *logits, float *pm)
{
#define FK_LAUNCH_RB(RBV)                                                                                                  \
    hipLaunchKernelGGL((score_logits_kernel<D, RBV>), grid, dim3(256), 0, st, k, ks[0], ks[1], ks[2], qf, p.H, p.Hkv, p.S, \
                       p.window, L.R, L.passes, L.Sp, L.ntA, sqrtD, logits, pm)
    switch (RB) {
    case 8: FK_LAUNCH_RB(8); break;
    case 16: FK_LAUNCH_RB(16); break;
    case 32: FK_LAUNCH_RB(32); break;
    default: FK_LAUNCH_RB(64); break;
    }
#undef FK_LAUNCH_RB
    return hipGetLastError();
}

hipError_t launch_score(const fastkv_problem &p, const Layout &L, const void *q, const int64_t *qs, const void *k,
                        const int64_t *ks, uint16_t *c_out, int64_t c_row_stride, uint16_t *t_out, int64_t t_row_stride,
                        char *ws, hipStream_t st)
{
    float *qf = reinterpret_cast<float *>(ws + L.off_qf);
    uint16_t *logits = reinterpret_cast<uint16_t *>(ws + L.off_logits);
    float *pm = reinterpret_cast<float *>(ws + L.off_pm);
    uint64_t *ps = reinterpret_cast<uint64_t *>(ws + L.off_ps);
    const float sqrtD = (float)sqrt((double)p.D);
    hipError_t e;

    hipLaunchKernelGGL(prep_q_kernel, dim3(p.B * p.Hkv), dim3(256), 0, st, (const uint16_t *)q, qs[0], qs[1], qs[2], p.H, p.Hkv,
                       p.S, p.D, p.window, L.R, L.R_alloc, qf);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    dim3 gridA(L.ntA * p.Hkv, p.B);
    if (p.D == 64) e = launch_logits_d<64>(L.RB, gridA, st, (const uint16_t *)k, ks, qf, p, L, sqrtD, logits, pm);
    else if (p.D == 128) e = launch_logits_d<128>(L.RB, gridA, st, (const uint16_t *)k, ks, qf, p, L, sqrtD, logits, pm);
    else e = launch_logits_d<256>(L.RB, gridA, st, (const uint16_t *)k, ks, qf, p, L, sqrtD, logits, pm);
    if (e != hipSuccess) return e;

    hipLaunchKernelGGL(score_sumexp_kernel, dim3(L.nchB, p.B * p.H), dim3(256), 0, st, logits, pm, p.S, p.window, L.Sp, L.ntA,
                       L.nchB, ps);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    const int pad = p.kernel / 2, TP = 256 - 2 * pad;
    dim3 gridC((L.n + TP - 1) / TP, p.Hkv, p.B);
    hipLaunchKernelGGL(score_finalize_kernel, gridC, dim3(256), (size_t)2 * L.R * sizeof(float), st, logits, pm, ps, p.H, p.Hkv,
                       p.S, p.window, L.Sp, L.ntA, L.nchB, p.kernel, p.pooling, c_out, c_row_stride);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    if (t_out) {
        hipLaunchKernelGGL(tsp_rowsum_kernel, dim3((L.n + 255) / 256, p.B), dim3(256), 0, st, c_out, c_row_stride, p.Hkv, L.n,
                           t_out, t_row_stride);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace fk
