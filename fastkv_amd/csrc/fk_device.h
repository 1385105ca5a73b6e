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
