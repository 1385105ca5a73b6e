// Device-side arithmetic contract shared by all kernels (gfx950 only).
//
// Every function here has a bit-identical twin in oracle/fastkv_oracle.c; the parity tests
// compare the two on the GPU.  Only IEEE fp32 fma / mul / add / div, RNE conversions and
// integer arithmetic are used (no v_exp_f32, no v_rcp_f32, no v_dot2: their results are not
// reproducible on a CPU -- measured, see DESIGN.md "dot2 probe").
// Compile with -ffp-contract=off so that nothing below is re-associated or fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FK_WAVE 64

#include "fk_hunt.h"   // measurement switches of the round-3 hunt: every macro is empty unless the build defines FK_HUNT

namespace fk {

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// fp32 -> fp16, round to nearest even.  The empty asm makes the fp32 value opaque: without it the compiler folds
// `(_Float16)(a * b)` into v_fma_mixlo_f16, which rounds the exact product ONCE to fp16, whereas the reference
// (fp32 tensor, then .to(fp16)) and the oracle round twice -- a 1-ulp difference in ~1e-4 of the elements.
__device__ __forceinline__ uint16_t f2h(float f)
{
    asm("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}
// A FINAL score (the c and t rows the selection ranks): every NaN becomes the canonical quiet NaN 0x7e00.  Sign and payload of a
// generated NaN differ between this GPU and the oracle's x86 (positive vs the negative "default NaN") and mono16 orders NaNs
// by their bits; 0x7e00 ranks above +inf, where torch.topk puts a NaN (utils.py:109, :115).  oracle/fastkv_oracle.c: f2h_score.
__device__ __forceinline__ uint16_t f2h_score(float f)
{
    const uint16_t h = f2h(f);
    return (h & 0x7fffu) > 0x7c00u ? (uint16_t)0x7e00u : h;
}
__device__ __forceinline__ float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// fp16 bit pattern -> uint16 key with the same total order as the values (utils.py:113 compares values)
__device__ __forceinline__ uint32_t mono16(uint32_t h) { return (h & 0x8000u) ? (~h & 0xffffu) : (h | 0x8000u); }

// x / c for an fp16-valued x, bit-identical to the IEEE fp32 division the reference performs (utils.py:94
// `/ math.sqrt(head_dim)`): reciprocal multiply + one Newton correction, which is exact for every one of the
// 63488 finite fp16 inputs (exhaustively checked on the GPU against the oracle's true division, tests/
// test_hip_parity.py::test_arithmetic_contract_on_gpu); zeros and infinities take the plain product.
__device__ __forceinline__ float scale_div(float x, float c, float rc)
{
    float q0 = x * rc;
    float r = __builtin_fmaf(-q0, c, x);
    float q1 = __builtin_fmaf(r, rc, q0);
    return (x == 0.0f || __builtin_isinf(x)) ? q0 : q1;
}

// exp(d) for d <= 0: Cody-Waite reduction + degree-6 polynomial, fma only.  d < -87 -> 0.
__device__ __forceinline__ float det_expf(float d)
{
    if (!(d >= -87.0f)) return (d != d) ? d : 0.0f;
    if (d > 0.0f) d = 0.0f;
    const float LOG2E = 1.44269504088896341f;
    const float LN2_HI = 0.693359375f;
    const float LN2_LO = -2.12194440e-4f;
    float n = __builtin_rintf(d * LOG2E);
    float r = __builtin_fmaf(n, -LN2_HI, d);
    r = __builtin_fmaf(n, -LN2_LO, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    int32_t ni = (int32_t)n;
    return bits_f32((uint32_t)((int32_t)f32_bits(p) + ni * (1 << 23)));
}

// e in [0,1] -> two 32-bit addends of a 2^-40 fixed-point value (hi * 2^24 + lo)
__device__ __forceinline__ void exp_to_fix(float e, uint32_t &hi, uint32_t &lo)
{
    float a = e * 65536.0f;
    float hf = __builtin_truncf(a);
    float rem = a - hf;
    float lf = __builtin_rintf(rem * 16777216.0f);
    hi = (uint32_t)hf;
    lo = (uint32_t)lf;
}

// 2^-40 fixed point -> fp32, round to nearest even, integer arithmetic only
__device__ __forceinline__ float fix_to_f32(uint64_t s)
{
    if (s == 0) return 0.0f;
    int msb = 63 - __builtin_clzll(s);
    uint32_t mant;
    int exp2 = msb;
    if (msb <= 23) {
        mant = (uint32_t)(s << (23 - msb));
    } else {
        int sh = msb - 23;
        uint64_t q = s >> sh;
        uint64_t rem = s & ((1ull << sh) - 1);
        uint64_t half = 1ull << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q++;
        if (q == (1ull << 24)) { q >>= 1; exp2++; }
        mant = (uint32_t)q;
    }
    return bits_f32(((uint32_t)(exp2 - 40 + 127) << 23) | (mant & 0x7fffffu));
}

// two fp32 -> one packed fp16 pair, round to nearest even per component: v_cvt_pk_f16_f32 (gfx950), ONE instruction where
// f2h twice + a pack take three.  Bit-identical to f2h per component (checked element-wise on the GPU over every fp16
// rounding boundary and its neighbours: fastkv_debug_contract op 12, tests/test_hip_parity.py).  The empty asm keeps the
// producer of the pair opaque, as in f2h (no single-rounding mix instruction may swallow a preceding multiply).
typedef _Float16 fk_h16x2 __attribute__((ext_vector_type(2)));
typedef float fk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2h2(float lo, float hi)
{
    fk_f32x2 v = {lo, hi};
    asm("" : "+v"(v));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fk_h16x2));
}

// ---- v_fma_mix_f32: an fp32 fma whose operands may be fp16 values read straight from one half of a register -- the conversion
// v_cvt_f32_f16 in front of an fp32 add / fma is exact, so "convert, then operate" and the mixed instruction give the same bits with
// one instruction less.  h2 = a packed pair of fp16 values; _h0 / _h1 pick the low / high half.
//   mix_add(h, c)     = float(h) + c            (fma(float(h), 1.0, c): the product is exact)
//   mix_mul(h, b)     = float(h) * b            (fma(float(h), b, -0.0): adding -0 changes no product, not even a zero's sign)
//   mix_fma(h, b, c)  = fma(float(h), b, c)
//   mix_fnma(a, b, h) = fma(-a, b, float(h))
__device__ __forceinline__ float mix_add_h0(uint32_t h2, float c) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c)); return d; }
__device__ __forceinline__ float mix_add_h1(uint32_t h2, float c) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c)); return d; }
__device__ __forceinline__ float mix_mul_h0(uint32_t h2, float b) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(-0.0f)); return d; }
__device__ __forceinline__ float mix_mul_h1(uint32_t h2, float b) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(-0.0f)); return d; }
__device__ __forceinline__ float mix_fma_h0(uint32_t h2, float b, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float mix_fma_h1(uint32_t h2, float b, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float mix_fnma_h0(float a, float b, uint32_t h2) { float d; asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(h2)); return d; }
__device__ __forceinline__ float mix_fnma_h1(float a, float b, uint32_t h2) { float d; asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(h2)); return d; }

// ---- two elements per instruction: v_pk_mul/fma/add_f32 are IEEE per component, so every component below is bit-identical
// to the scalar function above (same operations, same order); only rint / convert / select / integer steps stay scalar
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float x) { return (f32x2){x, x}; }

// NONPOS: the caller guarantees d <= 0 or NaN (softmax arguments x - max), which makes det_expf's clamp of positive
// arguments dead code
template <bool NONPOS = false> __device__ __forceinline__ f32x2 det_expf2(f32x2 d)
{
    const bool ok0 = d.x >= -87.0f, ok1 = d.y >= -87.0f;
    f32x2 dc;
    dc.x = ok0 ? (!NONPOS && d.x > 0.0f ? 0.0f : d.x) : -1.0f;
    dc.y = ok1 ? (!NONPOS && d.y > 0.0f ? 0.0f : d.y) : -1.0f;
    f32x2 n = dc * splat2(1.44269504088896341f);
    n.x = __builtin_rintf(n.x);
    n.y = __builtin_rintf(n.y);
    f32x2 r = fma2(n, splat2(-0.693359375f), dc);
    r = fma2(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = fma2(p, r, splat2(1.3981999507e-3f));
    p = fma2(p, r, splat2(8.3334519073e-3f));
    p = fma2(p, r, splat2(4.1665795894e-2f));
    p = fma2(p, r, splat2(1.6666665459e-1f));
    p = fma2(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    p = fma2(p, r2, r);
    p = p + splat2(1.0f);
    f32x2 out;
    out.x = bits_f32((uint32_t)((int32_t)f32_bits(p.x) + (int32_t)n.x * (1 << 23)));
    out.y = bits_f32((uint32_t)((int32_t)f32_bits(p.y) + (int32_t)n.y * (1 << 23)));
    if (!ok0) out.x = (d.x != d.x) ? d.x : 0.0f;
    if (!ok1) out.y = (d.y != d.y) ? d.y : 0.0f;
    return out;
}

// det_expf2 for arguments the caller knows to be <= 0 and not NaN: the range test becomes a clamp (v_max_f32) and nothing
// is selected afterwards.  For d >= -87 the result is det_expf(d) bit for bit; for d < -87 (and -inf) it is exp(-87) ~ 1.6e-38
// instead of 0 -- indistinguishable downstream: below 2^-41 the fixed-point addend is 0 (exp_to_fix) and the probability
// e * (1/sum) with sum >= 1 rounds to the fp16 zero either way (fused.hip phase B explains when a tile may take this path).
__device__ __forceinline__ f32x2 det_expf2_clamped(f32x2 d)
{
    f32x2 dc = {fmaxf(d.x, -87.0f), fmaxf(d.y, -87.0f)};
    f32x2 n = dc * splat2(1.44269504088896341f);
    n.x = __builtin_rintf(n.x);
    n.y = __builtin_rintf(n.y);
    f32x2 r = fma2(n, splat2(-0.693359375f), dc);
    r = fma2(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = fma2(p, r, splat2(1.3981999507e-3f));
    p = fma2(p, r, splat2(8.3334519073e-3f));
    p = fma2(p, r, splat2(4.1665795894e-2f));
    p = fma2(p, r, splat2(1.6666665459e-1f));
    p = fma2(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    p = fma2(p, r2, r);
    p = p + splat2(1.0f);
    f32x2 out;
    out.x = bits_f32((uint32_t)((int32_t)f32_bits(p.x) + (int32_t)n.x * (1 << 23)));
    out.y = bits_f32((uint32_t)((int32_t)f32_bits(p.y) + (int32_t)n.y * (1 << 23)));
    return out;
}

// exp_to_fix of both components; NaN components must be excluded by the caller (as with exp_to_fix)
__device__ __forceinline__ void exp_to_fix2(f32x2 e, uint32_t &hi0, uint32_t &lo0, uint32_t &hi1, uint32_t &lo1)
{
    // (a >= 0: a - trunc(a) is v_fract_f32, exact, and the conversion to an integer truncates by itself -- two instructions where
    // trunc, subtract, convert were three; the same bits)
    const f32x2 a = e * splat2(65536.0f);
    const f32x2 rem = {__builtin_amdgcn_fractf(a.x), __builtin_amdgcn_fractf(a.y)};
    const f32x2 l = rem * splat2(16777216.0f);
    hi0 = (uint32_t)a.x; hi1 = (uint32_t)a.y;
    lo0 = (uint32_t)__builtin_rintf(l.x); lo1 = (uint32_t)__builtin_rintf(l.y);
}

// The same 2^-40 fixed-point value in FOUR operations per component (round 6), split 20 / 20 with a SIGNED low part:
//   hi = RNE(e * 2^20),   lo = RNE((e * 2^20 - hi) * 2^20) in [-2^19, 2^19],   hi * 2^20 + lo = RNE(e * 2^40)
// -- which is what exp_to_fix defines (its trunc / RNE pair is RNE(e * 2^40) as well: hi * 2^24 is an even integer, so rounding the
// rest to nearest even rounds the whole).  Both roundings are done by the fp32 adder against the magic constant 1.5 * 2^23: the sum
// t = fma(e, 2^20, M) holds RNE(e * 2^20) in its low mantissa bits (e <= 1: t stays in M's binade, one ulp = 1), h = t - M and
// r = fma(e, 2^20, -h) are exact, u = fma(r, 2^20, M) holds the signed low part.  What is returned are the RAW BITS of t and u: the
// caller adds them up as integers and takes count * FIX_MAGIC_BITS off at the end (32-bit wrap-around cancels).  e must not be NaN.
constexpr uint32_t FIX_MAGIC_BITS = 0x4B400000u;               // 1.5 * 2^23
__device__ __forceinline__ void exp_to_fix2_magic(f32x2 e, uint32_t &th0, uint32_t &tl0, uint32_t &th1, uint32_t &tl1)
{
    const float M = 12582912.0f, S = 1048576.0f;
    const float t0 = __builtin_fmaf(e.x, S, M), t1 = __builtin_fmaf(e.y, S, M);
    const float h0 = t0 - M, h1 = t1 - M;
    const float r0 = __builtin_fmaf(e.x, S, -h0), r1 = __builtin_fmaf(e.y, S, -h1);
    const float u0 = __builtin_fmaf(r0, S, M), u1 = __builtin_fmaf(r1, S, M);
    th0 = f32_bits(t0); th1 = f32_bits(t1);
    tl0 = f32_bits(u0); tl1 = f32_bits(u1);
}

// scale_div of both components
__device__ __forceinline__ f32x2 scale_div2(f32x2 x, float c, float rc)
{
    const f32x2 q0 = x * splat2(rc);
    const f32x2 r = fma2(-q0, splat2(c), x);
    f32x2 q1 = fma2(r, splat2(rc), q0);
    if (x.x == 0.0f || __builtin_isinf(x.x)) q1.x = q0.x;
    if (x.y == 0.0f || __builtin_isinf(x.y)) q1.y = q0.y;
    return q1;
}

// scale_div2 for FINITE x whose zero sign does not matter: the plain Newton pair, nothing selected.  x = +-inf would give NaN
// (the caller sends such tiles through scale_div2); x = -0 gives +0 where the division gives -0 -- indistinguishable for a
// softmax LOGIT: exp(x - max) and max itself treat the two zeros alike (fused.hip, phase A).
__device__ __forceinline__ f32x2 scale_div2_finite(f32x2 x, float c, float rc)
{
    const f32x2 q0 = x * splat2(rc);
    const f32x2 r = fma2(-q0, splat2(c), x);
    return fma2(r, splat2(rc), q0);
}

// x / c for a packed pair of FINITE fp16 values in TWO operations per component, read by the mixed fma (no conversions in front):
//   q = fma(x, rc, x * rc_lo),   rc = fp32(1 / c),   rc_lo = fma(-rc, c, 1) * rc   (what rc leaves of 1 / c)
// -- the product x * rc is exact inside the fma (11 x 24 bits), the correction is good to 2^-49 of q, and an fp16 numerator over one of
// the three divisors sqrt(64), sqrt(128), sqrt(256) never brings the quotient that close to an fp32 rounding boundary: bit-identical to
// the IEEE fp32 division for EVERY finite fp16 x, zeros with their signs (exhaustive on the GPU: tests/test_hip_parity.py
// test_arithmetic_contract_on_gpu, op 13).  Other divisors (one wrong zero sign each among 40 tried) and +-inf (NaN) stay with
// scale_div2.
__device__ __forceinline__ float recip_lo(float c, float rc) { return __builtin_fmaf(-rc, c, 1.0f) * rc; }
__device__ __forceinline__ f32x2 scale_div2_finite_h2(uint32_t h2, float rc, float rc_lo)
{
    const float t0 = mix_mul_h0(h2, rc_lo), t1 = mix_mul_h1(h2, rc_lo);
    return (f32x2){mix_fma_h0(h2, rc, t0), mix_fma_h1(h2, rc, t1)};
}

// Reduction of 16 per-lane values over the 32 lanes of a half wave in 16 exchanges instead of 80: every step halves the
// number of values a lane carries (it keeps the half selected by one bit of its lane id and sends the other half to the
// partner).  Returns the fully reduced value of index halfwave_red_index(lane); lanes 2t and 2t+1 hold the same one.
__device__ __forceinline__ int halfwave_red_index(int lane)
{
    return ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}
template <typename T, typename Op> __device__ __forceinline__ T halfwave_reduce16(const T (&v)[16], int lane, Op op)
{
    T a[8], b[4], c[2], d;
    {
        const bool up = (lane & 16) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const T keep = up ? v[8 + i] : v[i], send = up ? v[i] : v[8 + i]; a[i] = op(keep, __shfl_xor(send, 16, 64)); }
    }
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const T keep = up ? a[4 + i] : a[i], send = up ? a[i] : a[4 + i]; b[i] = op(keep, __shfl_xor(send, 8, 64)); }
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const T keep = up ? b[2 + i] : b[i], send = up ? b[i] : b[2 + i]; c[i] = op(keep, __shfl_xor(send, 4, 64)); }
    }
    {
        const bool up = (lane & 2) != 0;
        const T keep = up ? c[1] : c[0], send = up ? c[0] : c[1];
        d = op(keep, __shfl_xor(send, 2, 64));
    }
    return op(d, __shfl_xor(d, 1, 64));
}

// pool1d over `ksize` taps centred on st[t] (utils.py:105-108): avg = sequential fp32 sum / ksize, max with NaN propagation.
// The caller stores the padding value (0 for avg, -inf for max) at out-of-range positions.
// The taps are read into registers first (KMAX independent LDS reads) and combined afterwards in tap order: a rolled
// "read, add, read, add" loop is a chain of LDS round trips, which two waves per SIMD cannot hide (measured in the fused
// score kernel: 5.4 us for 12 pools per thread).
// T = float, or uint16_t holding fp16 bits (the pooled values ARE fp16 values: h2f is exact)
__device__ __forceinline__ float pool_elem(float x) { return x; }
__device__ __forceinline__ float pool_elem(uint16_t x) { return h2f(x); }
template <int KMAX, typename T> __device__ __forceinline__ float pool_taps_n(const T *st, int t, int pad, int ksize, bool avg)
{
    float v[KMAX];
#pragma unroll
    for (int o = 0; o < KMAX; ++o) v[o] = pool_elem(st[t - pad + (o < ksize ? o : ksize - 1)]);
    float pv;
    if (avg) {
        pv = 0.0f;
#pragma unroll
        for (int o = 0; o < KMAX; ++o) if (o < ksize) pv = pv + v[o];
        pv = pv / (float)ksize;
    } else {
        pv = -INFINITY;
#pragma unroll
        for (int o = 0; o < KMAX; ++o) if (o < ksize && (v[o] > pv || v[o] != v[o])) pv = v[o];
    }
    return pv;
}
template <typename T> __device__ __forceinline__ float pool_taps(const T *st, int t, int pad, int ksize, bool avg)
{
    if (ksize <= 7) return pool_taps_n<7>(st, t, pad, ksize, avg);
    if (ksize <= 15) return pool_taps_n<15>(st, t, pad, ksize, avg);
    float pv;
    if (avg) {
        pv = 0.0f;
        for (int o = -pad; o <= pad; ++o) pv = pv + pool_elem(st[t + o]);
        pv = pv / (float)ksize;
    } else {
        pv = -INFINITY;
        for (int o = -pad; o <= pad; ++o) { const float xv = pool_elem(st[t + o]); if (xv > pv || xv != xv) pv = xv; }
    }
    return pv;
}

// one LDS histogram update per lane; the lanes that agree with lane 0 are folded into a single atomic
__device__ __forceinline__ void hist12_add(uint32_t *hist, uint32_t bin, bool active, int lane)
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

// ---- bounded waits of the in-launch hand-offs (fused.hip, select.hip) ----------------------------------------------
// A workgroup that waits for a partner can only be served if the partner is resident.  When another kernel holds compute
// units (another stream, another process) the partner may be scheduled late -- the wait then simply takes longer -- or, if
// two such launches overlap, never.  Every wait is therefore bounded by wall-clock time (s_memrealtime, 100 MHz); a
// workgroup that gives up publishes the launch's token in the workspace's abort word (so that every other workgroup of the
// launch stops waiting at its next poll) and raises the process-wide flag in pinned host memory that the next API call
// reports as FASTKV_EABORTED.  Nothing traps, nothing hangs, the HIP context stays usable.
struct SpinCtl {
    uint32_t *abort_word;     // control block word 3 of the operator workspace
    uint32_t *host_flag;      // pinned, host-coherent
    uint32_t token;           // this launch's hand-off token
    uint64_t deadline;        // s_memrealtime tick after which a wait gives up
};
__device__ __forceinline__ SpinCtl make_spin(uint32_t *ctrl, uint32_t *host_flag, uint32_t token, uint64_t limit_ticks)
{
    SpinCtl sp;
    sp.abort_word = ctrl + 3;
    sp.host_flag = host_flag;
    sp.token = token;
    sp.deadline = __builtin_amdgcn_s_memrealtime() + limit_ticks;
    return sp;
}
// one call per poll iteration of a waiting lane: true = stop waiting, the launch is abandoned
__device__ __forceinline__ bool spin_failed(const SpinCtl &sp)
{
    if (__hip_atomic_load(sp.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sp.token) return true;
    if (__builtin_amdgcn_s_memrealtime() > sp.deadline) {
        __hip_atomic_store(sp.abort_word, sp.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(sp.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (reasons combine: capi.hip take_abort_status)
        return true;
    }
    return false;
}
// the hand-off token of the launch that follows epoch `e` (a bijective mix, never 0: memory this library never wrote --
// zeros, small integers, fp16 data -- does not look like a current granule)
__device__ __forceinline__ uint32_t handoff_token(uint32_t epoch)
{
    const uint32_t t = (epoch + 1u) * 0x9E3779B1u ^ 0x7F4A7C15u;
    return t ? t : 0x6B43A9B5u;
}

#define FK_SUM_POISON 0xffffffffffffffffull   // a NaN was seen in the row (oracle: rinv = NaN)

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace fk
