/* Fast twin of tests/gen_inputs.py:normal_f16 (scale 1.0): splitmix64 -> twelve 16-bit uniforms -> Irwin-Hall N(0,1) -> fp16.
 * Test infrastructure only (input generation); bit-identical to the numpy version (checked by tests/test_oracle_golden.py).
 * Build: gcc -O3 -mf16c -fopenmp -shared -fPIC */
#include <stdint.h>
#include <immintrin.h>

static inline uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void gen_normal_f16(uint64_t base, int64_t count, uint16_t *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) {
        int64_t acc = 0;
        for (int c = 0; c < 3; ++c) {
            const uint64_t w = mix(base + (uint64_t)(3 * i + c + 1) * 0x9E3779B97F4A7C15ull);
            acc += (int64_t)(w & 0xFFFF) + (int64_t)((w >> 16) & 0xFFFF) + (int64_t)((w >> 32) & 0xFFFF) + (int64_t)(w >> 48);
        }
        /* (acc - 6*65536) / 65536 has at most 20 significant bits: exact in fp32, so fp32 -> fp16 is the one rounding */
        const float x = (float)(acc - 6 * 65536) / 65536.0f;
        out[i] = _cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    }
}
