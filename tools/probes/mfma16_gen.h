/* Shared by probe_mfma16_chain.hip (GPU) and mfma16_model_search.c (CPU): deterministic operands of the chained-tile check, so that the
 * dump only has to hold the GPU's outputs.  Tile t: A[32][128], Bt[32][128] fp16 bit patterns. */
#ifndef MFMA16_GEN_H
#define MFMA16_GEN_H
#include <stdint.h>
#include <string.h>
static inline uint64_t mg_next(uint64_t *st) { *st = *st * 6364136223846793005ULL + 1442695040888963407ULL; return *st >> 33; }
static inline uint16_t mg_f2h(float f)
{   /* round-to-nearest-even float -> half, finite inputs of moderate size only */
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000;
    int e = (int)((u >> 23) & 255) - 127 + 15;
    uint32_t m = u & 0x7fffff;
    if (e >= 31) return (uint16_t)(s | 0x7bff);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000;
        const int sh = 14 - e;
        uint32_t q = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q++;
        return (uint16_t)(s | q);
    }
    uint32_t q = m >> 13, rem = m & 0x1fff;
    if (rem > 0x1000 || (rem == 0x1000 && (q & 1))) q++;
    return (uint16_t)((s | ((uint32_t)e << 10)) + q);
}
static inline float mg_gauss(uint64_t *st) { float s = 0; for (int i = 0; i < 12; i++) s += (float)(mg_next(st) & 0xffff) / 65536.0f; return s - 6.0f; }
/* kinds: 0 N(0,1) x N(0,1) (the scoring kernel's operands); 1 N(0,1) x N(0,4) with a planted heavy row; 2 exponents over the whole fp16
 * range; 3 mostly tiny values incl. fp16 subnormals; 4 special values sprinkled in (inf / nan / -0 / max) */
static inline void mg_tile(int t, uint16_t *A, uint16_t *Bt)
{
    uint64_t st = 0x9E3779B97F4A7C15ULL * (uint64_t)(t + 1);
    const int kind = t % 5;
    for (int i = 0; i < 32 * 128; i++) {
        uint16_t a, b;
        if (kind == 0) { a = mg_f2h(mg_gauss(&st)); b = mg_f2h(mg_gauss(&st)); }
        else if (kind == 1) { a = mg_f2h(mg_gauss(&st) * ((i / 128) % 7 == 0 ? 6.0f : 1.0f)); b = mg_f2h(mg_gauss(&st) * 2.0f); }
        else if (kind == 2) { const uint32_t r = (uint32_t)mg_next(&st), q = (uint32_t)mg_next(&st);
                              a = (uint16_t)((r & 0x8000) | ((1 + (r >> 16) % 30) << 10) | (r & 0x3ff)); b = (uint16_t)((q & 0x8000) | ((8 + (q >> 16) % 12) << 10) | (q & 0x3ff)); }
        else if (kind == 3) { const uint32_t r = (uint32_t)mg_next(&st), q = (uint32_t)mg_next(&st);
                              a = (uint16_t)((r & 0x8000) | (((r >> 16) % 6) << 10) | (r & 0x3ff)); b = (uint16_t)((q & 0x8000) | ((10 + (q >> 16) % 8) << 10) | (q & 0x3ff)); }
        else { a = mg_f2h(mg_gauss(&st)); b = mg_f2h(mg_gauss(&st));
               const uint32_t r = (uint32_t)mg_next(&st);
               if ((r & 0x3ff) == 0) { static const uint16_t sp[8] = {0x7c00, 0xfc00, 0x7e00, 0x8000, 0x7bff, 0xfbff, 0x0001, 0xfe00}; if (r & 0x400) a = sp[(r >> 12) & 7]; else b = sp[(r >> 12) & 7]; } }
        A[i] = a; Bt[i] = b;
    }
}
#endif
