// Decode step over the compressed per-layer cache (/root/reference/baselines/fastkv/llama_model.py:143-145
// appends the new K/V row with `past_key_value.update` -- a torch.cat of the whole layer -- and attends with flash-attn,
// /root/reference/benchmark/e2e.py:72-93 times it).  Here the cache is a pre-sized slab [B,Hkv,rows,D] per layer
// (fastkv_amd/cache.py) whose current length lives in DEVICE memory, so the whole decode step has static shapes and can be
// replayed from a HIP graph:
//   decode_append    the step's K/V row -> slab row `len`                        (one 16-B piece per thread)
//   decode_partial   GQA attention of the G query heads of a KV head over a slice of the len+1 rows: scores with one cache
//                    row per lane (the row's 2*D bytes in 16-B loads), online softmax per wave, P.V with the lanes across
//                    head_dim (coalesced 4-B loads of V rows, probabilities from LDS) -> {max, sum, o[D]} per slice
//   decode_combine   merges the slices, writes fp16 [B,1,H*D], advances `len`
// HBM bound by construction (every cache byte is read once: 2*Hkv*(len+1)*D*2 bytes per layer, 9.4 MB at budget 2048 + 256
// decoded tokens), in practice latency bound at these sizes: the slices spread a layer over 8*nsplit workgroups.
// fp32 softmax; the result is compared with PyTorch SDPA to fp16 tolerance (tests/test_decode_gpu.py), not bit for bit.
// These are the step's SEPARATE launches (eager callers, tests); the captured step of benchmark/e2e.py runs the whole attention
// part as one launch: decode_step.hip.
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"

namespace fk {

constexpr int DEC_THREADS = 256;

template <int LPR>
__global__ void __launch_bounds__(64) decode_append_kernel(const uint16_t *__restrict__ k_new, int64_t kn_b, int64_t kn_h,
                                                          const uint16_t *__restrict__ v_new, int64_t vn_b, int64_t vn_h,
                                                          uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab, int64_t s_b,
                                                          int64_t s_h, int64_t s_r, int rows, const int32_t *__restrict__ len_dev)
{
    const int h = blockIdx.x, b = blockIdx.y, sub = threadIdx.x;
    int len = *len_dev;
    if (sub >= LPR) return;
    if (len >= rows) len = rows - 1;                             // a full slab overwrites its last row instead of its neighbour (the host sizes it)
    const uint4 kv = *reinterpret_cast<const uint4 *>(k_new + b * kn_b + h * kn_h + sub * 8);
    const uint4 vv = *reinterpret_cast<const uint4 *>(v_new + b * vn_b + h * vn_h + sub * 8);
    *reinterpret_cast<uint4 *>(kslab + b * s_b + h * s_h + (int64_t)len * s_r + sub * 8) = kv;
    *reinterpret_cast<uint4 *>(vslab + b * s_b + h * s_h + (int64_t)len * s_r + sub * 8) = vv;
}

// part[b][h][c] = {m, l, o[D]} (fp32).  grid (nsplit, Hkv, B), 256 threads.
template <int D, int G>
__global__ void __launch_bounds__(DEC_THREADS) decode_partial_kernel(const uint16_t *__restrict__ q, int64_t q_b, int64_t q_h,
                                                                   const uint16_t *__restrict__ kslab,
                                                                   const uint16_t *__restrict__ vslab, int64_t s_b, int64_t s_h,
                                                                   int64_t s_r, int rows, const int32_t *__restrict__ len_dev,
                                                                   float scaling, float *__restrict__ part, int nsplit)
{
    constexpr int DPL = D / 64;                                  // head-dim elements per lane in the P.V product
    __shared__ float s_q[G][D];
    __shared__ float s_p[4][G][64];
    __shared__ float s_m[4][G], s_l[4][G];
    __shared__ float s_o[4][G][D];
    const int c = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int RB = DPL <= 2 ? 64 : 32;                       // V rows in flight per batch
    const uint16_t *kb = kslab + b * s_b + hk * s_h, *vb = vslab + b * s_b + hk * s_h;
    uint4 kv[D / 8];
    uint32_t vv[RB][(DPL + 1) / 2];
    int len = *len_dev + 1;                                      // the step's own row is row *len_dev (decode_append wrote it)
    if (len > rows) len = rows;
    const int chunk = ((len + nsplit - 1) / nsplit + 63) / 64 * 64;
    const int lo = c * chunk, hi = min(len, lo + chunk);
    for (int i = threadIdx.x; i < G * D; i += DEC_THREADS) {
        const int g = i / D, d = i - g * D;
        s_q[g][d] = h2f(q[b * q_b + (int64_t)(hk * G + g) * q_h + d]) * scaling;
    }
    __syncthreads();
    float m[G], l[G], o[G][DPL];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.0f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[g][e] = 0.0f;
    }
    for (int t0 = lo + w * 64; t0 < hi; t0 += 4 * 64) {
        // ---- scores: lane = cache row t0 + lane
        const int j = t0 + lane;
        const bool valid = j < hi;
        const uint16_t *kr = kb + (int64_t)(valid ? j : hi - 1) * s_r;
        float sc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) sc[g] = 0.0f;
        // V rows of the tile (lane = head-dim elements lane*DPL ..., one coalesced row load per row) are requested TOGETHER with
        // the K rows: they do not depend on the scores, so a tile costs one memory round trip (head_dim 256: the second
        // half of the V rows follows in a second batch, registers)
        const int nrow = min(64, hi - t0);
        auto load_v = [&](int r0) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int jr = min(t0 + r0 + u, hi - 1);
                const uint16_t *vr = vb + (int64_t)jr * s_r + lane * DPL;
                if (DPL == 1) vv[u][0] = *vr;
                else if (DPL == 2) vv[u][0] = *reinterpret_cast<const uint32_t *>(vr);
                else { const uint2 x = *reinterpret_cast<const uint2 *>(vr); vv[u][0] = x.x; vv[u][1] = x.y; }
            }
        };
#pragma unroll
        for (int u = 0; u < D / 8; ++u) kv[u] = *reinterpret_cast<const uint4 *>(kr + u * 8);
        load_v(0);
#pragma unroll
        for (int u = 0; u < D / 8; ++u) {
            const uint32_t wds[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float k0 = h2f((uint16_t)(wds[e] & 0xffffu)), k1 = h2f((uint16_t)(wds[e] >> 16));
                const int d = u * 8 + e * 2;
#pragma unroll
                for (int g = 0; g < G; ++g) sc[g] = __builtin_fmaf(s_q[g][d + 1], k1, __builtin_fmaf(s_q[g][d], k0, sc[g]));
            }
        }
        // ---- online softmax of the tile (wave-wide), probabilities to LDS
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float s = valid ? sc[g] : -INFINITY;
            const float tm = wave_max(s);
            const float mn = fmaxf(m[g], tm);
            const float p = valid ? __expf(s - mn) : 0.0f;
            float ps = p;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off, 64);
            const float corr = __expf(m[g] - mn);                // m = -inf on the first tile: exp(-inf) = 0
            l[g] = l[g] * corr + ps;
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[g][e] *= corr;
            m[g] = mn;
            s_p[w][g][lane] = p;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- P.V
        for (int r0 = 0; r0 < nrow; r0 += RB) {
            if (r0 > 0) load_v(r0);
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (r0 + u < nrow) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float p = s_p[w][g][r0 + u];
#pragma unroll
                        for (int e = 0; e < DPL; ++e) {
                            const uint32_t wd = vv[u][e / 2];
                            o[g][e] = __builtin_fmaf(p, h2f((uint16_t)((e & 1) ? wd >> 16 : wd & 0xffffu)), o[g][e]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the four waves' partials -> one record of the slice
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (lane == 0) { s_m[w][g] = m[g]; s_l[w][g] = l[g]; }
#pragma unroll
        for (int e = 0; e < DPL; ++e) s_o[w][g][lane * DPL + e] = o[g][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G * D; i += DEC_THREADS) {
        const int g = i / D, d = i - g * D;
        const float M = fmaxf(fmaxf(s_m[0][g], s_m[1][g]), fmaxf(s_m[2][g], s_m[3][g]));
        float L = 0.0f, O = 0.0f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float f = s_m[ww][g] == -INFINITY ? 0.0f : __expf(s_m[ww][g] - M);
            L += s_l[ww][g] * f;
            O += s_o[ww][g][d] * f;
        }
        float *rec = part + (((size_t)b * gridDim.y * G + hk * G + g) * nsplit + c) * (D + 2);
        rec[2 + d] = O;
        if (d == 0) { rec[0] = M; rec[1] = L; }
    }
}

// grid (H, B), D threads; advances *len_dev (block 0, thread 0) -- every reader of the old length has finished by now
template <int D>
__global__ void __launch_bounds__(D) decode_combine_kernel(const float *__restrict__ part, int nsplit, uint16_t *__restrict__ out, int H,
                                                           int32_t *__restrict__ len_dev, int rows, uint32_t *__restrict__ host_flag)
{
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float *rec = part + ((size_t)b * H + h) * nsplit * (D + 2);
    float M = -INFINITY;
    for (int c = 0; c < nsplit; ++c) M = fmaxf(M, rec[c * (D + 2)]);
    float L = 0.0f, O = 0.0f;
    for (int c = 0; c < nsplit; ++c) {
        const float mc = rec[c * (D + 2)];
        const float f = mc == -INFINITY ? 0.0f : __expf(mc - M);
        L += rec[c * (D + 2) + 1] * f;
        O += rec[c * (D + 2) + 2 + d] * f;
    }
    out[((size_t)b * H + h) * D + d] = f2h(O / L);
    if (h == 0 && b == 0 && d == 0 && len_dev) {
        const int n = *len_dev + 1;
        *len_dev = n < rows ? n : rows;
        if (n > rows && host_flag) __hip_atomic_store(host_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // slab overrun (see the step kernel)
    }
}

// ---- the step's small operators, one launch each (the stock modules run 7 / 8 / 2 elementwise launches for them) ----
// RMSNorm as transformers' LlamaRMSNorm / MistralRMSNorm compute it: fp32 mean of squares, x * rsqrt(var + eps) -> fp16,
// times the fp16 weight -> fp16.  One 256-thread workgroup per row.
__global__ void __launch_bounds__(256) decode_rmsnorm_kernel(const uint16_t *__restrict__ x, int64_t x_row, const uint16_t *__restrict__ wgt,
                                                            uint16_t *__restrict__ out, int hidden, float eps)
{
    __shared__ float s_part[4];
    const uint16_t *xr = x + (int64_t)blockIdx.x * x_row;
    float ss = 0.0f;
    for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4 *>(xr + i);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = h2f((uint16_t)(wds[e] & 0xffffu)), b = h2f((uint16_t)(wds[e] >> 16));
            ss = __builtin_fmaf(a, a, ss);
            ss = __builtin_fmaf(b, b, ss);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float var = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)hidden;
    const float r = 1.0f / __builtin_sqrtf(var + eps);
    for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
        const uint4 v = *reinterpret_cast<const uint4 *>(xr + i), g = *reinterpret_cast<const uint4 *>(wgt + i);
        const uint32_t wx[4] = {v.x, v.y, v.z, v.w}, wg[4] = {g.x, g.y, g.z, g.w};
        uint32_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t n0 = f2h(h2f((uint16_t)(wx[e] & 0xffffu)) * r), n1 = f2h(h2f((uint16_t)(wx[e] >> 16)) * r);
            const uint16_t y0 = f2h(h2f((uint16_t)(wg[e] & 0xffffu)) * h2f(n0)), y1 = f2h(h2f((uint16_t)(wg[e] >> 16)) * h2f(n1));
            o4[e] = (uint32_t)y0 | ((uint32_t)y1 << 16);
        }
        *reinterpret_cast<uint4 *>(out + (int64_t)blockIdx.x * hidden + i) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
}

// apply_rotary_pos_emb on the step's q [B,H,1,D] and k [B,Hkv,1,D] (both rewritten in place), cos / sin [B,1,D]:
// x*cos -> fp16, rotate_half(x)*sin -> fp16, sum -> fp16 -- the stock fp16 sequence, rounding for rounding.
__global__ void __launch_bounds__(128) decode_rope_kernel(uint16_t *__restrict__ q, int64_t q_b, int64_t q_h, int H,
                                                         uint16_t *__restrict__ k, int64_t k_b, int64_t k_h, int Hkv,
                                                         const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                                         int64_t cs_b, int D)
{
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    if (d >= D / 2) return;
    uint16_t *x = h < H ? q + b * q_b + (int64_t)h * q_h : k + b * k_b + (int64_t)(h - H) * k_h;
    const float x1 = h2f(x[d]), x2 = h2f(x[d + D / 2]);
    const float c1 = h2f(cosv[b * cs_b + d]), c2 = h2f(cosv[b * cs_b + d + D / 2]);
    const float s1 = h2f(sinv[b * cs_b + d]), s2 = h2f(sinv[b * cs_b + d + D / 2]);
    // rotate_half(x) = cat(-x2, x1)
    const uint16_t o1 = f2h(h2f(f2h(x1 * c1)) + h2f(f2h(-x2 * s1)));
    const uint16_t o2 = f2h(h2f(f2h(x2 * c2)) + h2f(f2h(x1 * s2)));
    x[d] = o1;
    x[d + D / 2] = o2;
}

// act_fn(gate) * up of the MLP (SiLU): silu in fp32 -> fp16, times up -> fp16
__global__ void __launch_bounds__(256) decode_silu_mul_kernel(const uint16_t *__restrict__ gate, const uint16_t *__restrict__ up,
                                                             uint16_t *__restrict__ out, int64_t n)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 g = *reinterpret_cast<const uint4 *>(gate + i), u = *reinterpret_cast<const uint4 *>(up + i);
    const uint32_t wg[4] = {g.x, g.y, g.z, g.w}, wu[4] = {u.x, u.y, u.z, u.w};
    uint32_t o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = h2f((uint16_t)(wg[e] & 0xffffu)), g1 = h2f((uint16_t)(wg[e] >> 16));
        const uint16_t a0 = f2h(g0 / (1.0f + __expf(-g0))), a1 = f2h(g1 / (1.0f + __expf(-g1)));
        const uint16_t y0 = f2h(h2f(a0) * h2f((uint16_t)(wu[e] & 0xffffu))), y1 = f2h(h2f(a1) * h2f((uint16_t)(wu[e] >> 16)));
        o4[e] = (uint32_t)y0 | ((uint32_t)y1 << 16);
    }
    *reinterpret_cast<uint4 *>(out + i) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
}


// Greedy sampling of a step, one launch: argmax over the vocabulary row of every batch entry (torch.argmax's rule: the FIRST maximal
// value; NaN counts as maximal), the new token written to tok[b] (the next step's input), appended to the log and the positions
// advanced -- what `logits.argmax(-1)` + four bookkeeping kernels do in benchmark/e2e.py's captured step.  Every workgroup folds its
// share into a 64-bit key {order-preserving value : 16, ~index : 32} and does ONE atomic max per batch entry; the last workgroup
// to arrive writes the results and leaves the scratch words as it found them (graph-replayable).
// grid (nwg, B), 256 threads; scratch: B + 1 uint64 words, zero between launches.
__global__ void __launch_bounds__(256) decode_greedy_kernel(const uint16_t *__restrict__ logits, int64_t row_stride, int V,
                                                           unsigned long long *__restrict__ scratch, int64_t *__restrict__ tok,
                                                           int64_t *__restrict__ pos, int64_t *__restrict__ log, int64_t *__restrict__ log_index,
                                                           int log_cap)
{
    __shared__ unsigned long long s_best[4];
    const int b = blockIdx.y, B = gridDim.y, nwg = gridDim.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint16_t *row = logits + (int64_t)b * row_stride;
    unsigned long long best = 0;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 8; i < V; i += nwg * 256 * 8) {
        const uint4 x = *reinterpret_cast<const uint4 *>(row + i);          // (V is a multiple of 8, rows 16-B aligned: checked by the host)
        const uint32_t wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t h = (wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            if (h == 0x8000u) h = 0;                                                       // -0.0 == +0.0 for torch.argmax: the first zero wins
            const uint32_t key = (h & 0x7fffu) > 0x7c00u ? 0xffffu : mono16(h);           // NaN: maximal
            const unsigned long long k = ((unsigned long long)key << 32) | (uint32_t)~(uint32_t)(i + e);
            best = k > best ? k : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = (unsigned long long)__shfl_xor((long long)best, o, 64); best = other > best ? other : best; }
    if (lane == 0) s_best[w] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = s_best[i] > best ? s_best[i] : best;
        __hip_atomic_fetch_max(scratch + 1 + b, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned long long arrived = __hip_atomic_fetch_add(scratch, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (unsigned long long)nwg * B - 1) {                   // the last workgroup of the launch: every maximum is in
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            int64_t li = log_index ? *log_index : 0;
            for (int bb = 0; bb < B; ++bb) {
                const unsigned long long k = __hip_atomic_load(scratch + 1 + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t t = (int64_t)(uint32_t)~(uint32_t)k;
                tok[bb] = t;
                if (pos) pos[bb] += 1;
                __hip_atomic_store(scratch + 1 + bb, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bb == 0 && log && li < log_cap) log[li] = t;            // (the log follows batch entry 0, as the driver's token list does)
            }
            if (log_index) *log_index = li + 1;
            __hip_atomic_store(scratch, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// Rotary tables of a one-token step, one launch: cos / sin [B,1,D] fp16 = LlamaRotaryEmbedding.forward / MistralRotaryEmbedding.forward
// for ONE position per batch entry (transformers: freqs = inv_freq (fp32) x position (as fp32) -- one fp32 product --, emb = cat(freqs,
// freqs), cos(emb) * attention_scaling -> fp16; the same device cosf / sinf).  grid B, D/2 threads.
__global__ void __launch_bounds__(128) decode_rotary_kernel(const float *__restrict__ inv_freq, const int64_t *__restrict__ pos, float scaling,
                                                           uint16_t *__restrict__ cosv, uint16_t *__restrict__ sinv, int half)
{
    const int b = blockIdx.x, d = threadIdx.x;
    if (d >= half) return;
    const float f = inv_freq[d] * (float)pos[b];
    const uint16_t c = f2h(cosf(f) * scaling), sn = f2h(sinf(f) * scaling);
    cosv[(size_t)b * 2 * half + d] = c;  cosv[(size_t)b * 2 * half + half + d] = c;
    sinv[(size_t)b * 2 * half + d] = sn; sinv[(size_t)b * 2 * half + half + d] = sn;
}

}  // namespace fk

using namespace fk;

extern "C" {

int fastkv_decode_append_f16(int32_t B, int32_t Hkv, int32_t D, const void *k_new, const int64_t kn_strides[2], const void *v_new,
                             const int64_t vn_strides[2], void *kslab, void *vslab, const int64_t slab_strides[3], int32_t rows,
                             const int32_t *len_dev, void *stream)
{
    if (B < 1 || Hkv < 1 || rows < 1 || !k_new || !v_new || !kslab || !vslab || !len_dev || !kn_strides || !vn_strides || !slab_strides)
        return FASTKV_EINVAL;
    if (D != 64 && D != 128 && D != 256) return FASTKV_EUNSUPPORTED;
    if (((uintptr_t)k_new | (uintptr_t)v_new | (uintptr_t)kslab | (uintptr_t)vslab) & 15) return FASTKV_EINVAL;
    for (int i = 0; i < 2; ++i) if ((kn_strides[i] & 7) || (vn_strides[i] & 7)) return FASTKV_EINVAL;
    for (int i = 0; i < 3; ++i) if (slab_strides[i] & 7) return FASTKV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps_(K_DECODE, st);
    dim3 grid(Hkv, B);
#define FK_APP(LPRV)                                                                                                                \
    hipLaunchKernelGGL((decode_append_kernel<LPRV>), grid, dim3(64), 0, st, (const uint16_t *)k_new, kn_strides[0], kn_strides[1],  \
                       (const uint16_t *)v_new, vn_strides[0], vn_strides[1], (uint16_t *)kslab, (uint16_t *)vslab, slab_strides[0], \
                       slab_strides[1], slab_strides[2], rows, len_dev)
    if (D == 64) FK_APP(8);
    else if (D == 128) FK_APP(16);
    else FK_APP(32);
#undef FK_APP
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_attention_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, const void *q, const int64_t q_strides[2],
                                const void *kslab, const void *vslab, const int64_t slab_strides[3], int32_t rows, int32_t *len_dev,
                                float scaling, int32_t nsplit, void *out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (B < 1 || Hkv < 1 || H < Hkv || (H % Hkv) || rows < 1 || nsplit < 1 || nsplit > 256) return FASTKV_EINVAL;
    if (!q || !kslab || !vslab || !len_dev || !out || !workspace || !q_strides || !slab_strides) return FASTKV_EINVAL;
    if (D != 64 && D != 128 && D != 256) return FASTKV_EUNSUPPORTED;
    const int G = H / Hkv;
    if (G != 1 && G != 2 && G != 4 && G != 8) return FASTKV_EUNSUPPORTED;
    if (((uintptr_t)kslab | (uintptr_t)vslab) & 15) return FASTKV_EINVAL;
    for (int i = 0; i < 3; ++i) if (slab_strides[i] & 7) return FASTKV_EINVAL;
    if (workspace_bytes < fastkv_decode_workspace_bytes(B, H, D, nsplit)) return FASTKV_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    dim3 grid(nsplit, Hkv, B);
    {
        ProfScope ps_(K_DECODE, st);
#define FK_PART(DV, GV)                                                                                                              \
    hipLaunchKernelGGL((decode_partial_kernel<DV, GV>), grid, dim3(DEC_THREADS), 0, st, (const uint16_t *)q, q_strides[0], q_strides[1], \
                       (const uint16_t *)kslab, (const uint16_t *)vslab, slab_strides[0], slab_strides[1], slab_strides[2], rows, \
                       len_dev, scaling, part, nsplit)
#define FK_PART_G(DV)                                                                                      \
    do {                                                                                                   \
        if (G == 1) FK_PART(DV, 1); else if (G == 2) FK_PART(DV, 2); else if (G == 4) FK_PART(DV, 4); else FK_PART(DV, 8); \
    } while (0)
        if (D == 64) FK_PART_G(64);
        else if (D == 128) FK_PART_G(128);
        else FK_PART_G(256);
#undef FK_PART_G
#undef FK_PART
    }
    if (hipGetLastError() != hipSuccess) return FASTKV_ELAUNCH;
    ProfScope ps2_(K_DECODE, st);
    if (D == 64) hipLaunchKernelGGL((decode_combine_kernel<64>), dim3(H, B), dim3(64), 0, st, part, nsplit, (uint16_t *)out, H, len_dev, rows, abort_flag_device());
    else if (D == 128) hipLaunchKernelGGL((decode_combine_kernel<128>), dim3(H, B), dim3(128), 0, st, part, nsplit, (uint16_t *)out, H, len_dev, rows, abort_flag_device());
    else hipLaunchKernelGGL((decode_combine_kernel<256>), dim3(H, B), dim3(256), 0, st, part, nsplit, (uint16_t *)out, H, len_dev, rows, abort_flag_device());
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_rmsnorm_f16(const void *x, int64_t rows, int64_t x_row_stride, int32_t hidden, const void *weight, float eps, void *out,
                              void *stream)
{
    if (!x || !weight || !out || rows < 0 || hidden < 8 || (hidden & 7) || (x_row_stride & 7) || x_row_stride < hidden) return FASTKV_EINVAL;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)out) & 15) return FASTKV_EINVAL;
    if (rows == 0) return FASTKV_OK;
    ProfScope ps_(K_DECODE, (hipStream_t)stream);
    hipLaunchKernelGGL(decode_rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, x_row_stride,
                       (const uint16_t *)weight, (uint16_t *)out, hidden, eps);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_rope_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, void *q, const int64_t q_strides[2], void *k,
                           const int64_t k_strides[2], const void *cosv, const void *sinv, int64_t cs_batch_stride, void *stream)
{
    if (B < 1 || H < 1 || Hkv < 1 || D < 2 || (D & 1) || D > 256 || !q || !k || !cosv || !sinv || !q_strides || !k_strides) return FASTKV_EINVAL;
    ProfScope ps_(K_DECODE, (hipStream_t)stream);
    hipLaunchKernelGGL(decode_rope_kernel, dim3(H + Hkv, B), dim3(128), 0, (hipStream_t)stream, (uint16_t *)q, q_strides[0], q_strides[1], H,
                       (uint16_t *)k, k_strides[0], k_strides[1], Hkv, (const uint16_t *)cosv, (const uint16_t *)sinv, cs_batch_stride, D);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_rotary_f16(int32_t B, int32_t D, const float *inv_freq, const int64_t *pos, float scaling, void *cosv, void *sinv, void *stream)
{
    if (B < 1 || D < 2 || (D & 1) || D > 256 || !inv_freq || !pos || !cosv || !sinv) return FASTKV_EINVAL;
    ProfScope ps_(K_DECODE, (hipStream_t)stream);
    hipLaunchKernelGGL(decode_rotary_kernel, dim3((unsigned)B), dim3(128), 0, (hipStream_t)stream, inv_freq, pos, scaling, (uint16_t *)cosv, (uint16_t *)sinv, D / 2);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_greedy_f16(int32_t B, int32_t V, const void *logits, int64_t row_stride, void *scratch, int64_t *tok, int64_t *pos,
                             int64_t *log, int64_t *log_index, int32_t log_cap, void *stream)
{
    if (B < 1 || B > 1024 || V < 8 || (V & 7) || !logits || !scratch || !tok || (row_stride & 7) || row_stride < V) return FASTKV_EINVAL;
    if (((uintptr_t)logits & 15) || ((uintptr_t)scratch & 7) || (log && (!log_index || log_cap < 1))) return FASTKV_EINVAL;
    int nwg = (V / 8 + 255) / 256;                               // one 16-B piece per thread where the vocabulary allows: 63 workgroups at 128k
    if (nwg > 256) nwg = 256;
    ProfScope ps_(K_DECODE, (hipStream_t)stream);
    hipLaunchKernelGGL(decode_greedy_kernel, dim3((unsigned)nwg, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)logits, row_stride, V,
                       (unsigned long long *)scratch, tok, pos, log, log_index, log_cap);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_decode_silu_mul_f16(const void *gate, const void *up, int64_t n, void *out, void *stream)
{
    if (!gate || !up || !out || n < 0 || (n & 7) || (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)out) & 15)) return FASTKV_EINVAL;
    if (n == 0) return FASTKV_OK;
    ProfScope ps_(K_DECODE, (hipStream_t)stream);
    hipLaunchKernelGGL(decode_silu_mul_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)gate,
                       (const uint16_t *)up, (uint16_t *)out, n);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

}  // extern "C"
