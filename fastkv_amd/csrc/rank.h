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
    const int nv = (k + 7) >> 3, pv = p >> 3;
    const uint32_t t_hi = kp * 0x00010001u;                       // j > p:  key_j >  kp
    const uint32_t t_lo = (kp - 1u) * 0x00010001u;                // j < p:  key_j >= kp  <=>  key_j > kp - 1   (kp > 0)
    uint32_t acc = 0, extra = 0;
    for (int v = sub; v < nv; v += nsub) {
        const uint4 x = *reinterpret_cast<const uint4 *>(keys + v * 8);
        if (v == pv || kp == 0) {                                 // the vector that holds p (or the degenerate key 0): element-wise
            const uint32_t wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t kj = (wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                const int j = v * 8 + e;
                extra += (j < k) && ((kj > kp) || (kj == kp && j < p));
            }
        } else {
            const uint32_t t = v < pv ? t_lo : t_hi;
            acc += pk_gt(x.x, t) + pk_gt(x.y, t) + pk_gt(x.z, t) + pk_gt(x.w, t);      // halves stay < 65536: k <= 131064
        }
    }
    return (acc & 0xffffu) + (acc >> 16) + extra;
}

}  // namespace fk
