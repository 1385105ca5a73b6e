// Destination slot of a winner under ORDER_SCORE by comparison counting (no sort).
#pragma once
#include "fk_device.h"

namespace fk {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// per 16-bit half: 1 if the half of w is strictly greater than the half of t, else 0 (v_pk_sub_u16 clamp + v_pk_min_u16)
__device__ __forceinline__ uint32_t pk_gt(uint32_t w, uint32_t t)
{
    const us2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, w), __builtin_bit_cast(us2, t));
    const us2 one = {1, 1};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(d, one));
}

// rank(p) = #{j : key_j > key_p} + #{j < p : key_j == key_p}: value descending, equal values in ascending position
// (the winner list is in ascending position) -- the order of `topk(sorted=True)` (utils.py:113) with the canonical tie
// rule.  The lanes `sub` of a group of `nsub` split the ceil(k/8) key vectors; the caller adds the partial counts.
// keys: 16-B aligned list of k order-preserving 16-bit keys, zero padded to a multiple of 8.
__device__ __forceinline__ uint32_t rank_partial(const uint16_t *__restrict__ keys, int k, int p, uint32_t kp, int sub, int nsub)
{
    const int nv = (k + 7) >> 3, pv = p >> 3, pe = p & 7;
    if (kp == 0) {                                                // degenerate smallest key (a NaN score): plain counting
        uint32_t cnt = 0;
        for (int j = sub; j < k; j += nsub) { const uint32_t kj = keys[j]; cnt += (kj > kp) || (kj == kp && j < p); }
        return cnt;
    }
    // branch-free packed counting: per 16-bit half, 1 if key_j > threshold_j with
    //   threshold_j = kp - 1 for j < p (key_j >= kp), kp for j >= p (key_j > kp; j == p itself never counts),
    //   0xffff for vectors past the list (nothing counts; the zero padding inside the last vector never exceeds kp >= 1)
    const uint32_t t_hi = kp * 0x00010001u, t_lo = (kp - 1u) * 0x00010001u;
    uint32_t tm[4];                                               // thresholds of the vector that holds p
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (2 * i < pe) ? kp - 1u : kp, hi = (2 * i + 1 < pe) ? kp - 1u : kp;
        tm[i] = lo | (hi << 16);
    }
    uint32_t acc = 0;
    // 4 key vectors per step, loaded together (a one-vector loop is a chain of dependent round trips)
    for (int v0 = sub; v0 < nv; v0 += 4 * nsub) {
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = v0 + u * nsub;
            x[u] = *reinterpret_cast<const uint4 *>(keys + (v < nv ? v : nv - 1) * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = v0 + u * nsub;
            const uint32_t tside = v >= nv ? 0xffffffffu : (v < pv ? t_lo : t_hi);
            const bool mid = (v == pv);
            acc += pk_gt(x[u].x, mid ? tm[0] : tside) + pk_gt(x[u].y, mid ? tm[1] : tside) +
                   pk_gt(x[u].z, mid ? tm[2] : tside) + pk_gt(x[u].w, mid ? tm[3] : tside);      // halves stay < 65536: k <= 131064
        }
    }
    return (acc & 0xffffu) + (acc >> 16);
}

// The same count over a key list in LDS that is ZERO PADDED TO A MULTIPLE OF 8 * 4 * nsub keys (the copy kernel pads it): no
// vector needs a bounds test, and the vector that holds p is counted with the side threshold like every other one and corrected
// afterwards by its owner -- 14 instead of ~20 instructions per vector (the counting is the copy launch's vector-ALU bill: k^2
// packed compares per head, 4.7 us of issue time per 32k layer).  kp >= 1.
__device__ __forceinline__ uint32_t rank_partial_padded(const uint16_t *__restrict__ keys, int kpad, int p, uint32_t kp, int sub, int nsub)
{
    const int nvp = kpad >> 3, pv = p >> 3, pe = p & 7;
    const uint32_t t_hi = kp * 0x00010001u, t_lo = (kp - 1u) * 0x00010001u;
    uint32_t acc = 0;
    for (int v0 = sub; v0 < nvp; v0 += 4 * nsub) {
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const uint4 *>(keys + (v0 + u * nsub) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t ts = (v0 + u * nsub) < pv ? t_lo : t_hi;       // vector pv itself: t_hi, corrected below
            acc += pk_gt(x[u].x, ts) + pk_gt(x[u].y, ts) + pk_gt(x[u].z, ts) + pk_gt(x[u].w, ts);
        }
    }
    uint32_t cnt = (acc & 0xffffu) + (acc >> 16);
    if ((pv % nsub) == sub) {
        // elements of vector pv in front of p count when key >= kp, not only when key > kp: add those equal to kp
        const uint16_t *kv = keys + pv * 8;
        for (int e = 0; e < pe; ++e) cnt += kv[e] == kp;
    }
    return cnt;
}

}  // namespace fk
