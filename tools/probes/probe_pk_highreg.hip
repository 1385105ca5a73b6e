// Probe: packed-fp32 arithmetic on HIGH-numbered registers (v[192:255] of a wave that owns all 256) beside the partner's matrix phase
// (mfma_tile.h, CUT 4) -- the standalone form of docs/HISTORY.md's pair once more, this time with the register numbers the kernel's
// softmax phases really use.  B computes a chain of v_pk_fma_f32 in inline assembly on v[200:215] and the same chain with v_fma_f32 on low
// registers; they must agree bit for bit.
#include "../../fastkv_amd/csrc/mfma_tile.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace fk;
constexpr int LDS_BYTES = 80896;
#define CLOB "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v255"
__global__ void __launch_bounds__(256, 2) probe(uint32_t *out, const uint16_t *kbuf, int S, int a_mode, int gap_ticks)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const uint32_t wg = blockIdx.x, tix = threadIdx.x;
    const int lane = tix & 63, w = tix >> 6, n31 = lane & 31, hi = lane >> 5;
    const uint64_t t_end = wall_clock64() + 15000;
    if (wg < 256) {
        float *As = reinterpret_cast<float *>(smem + 4 * 64 * ROWB);
        for (int i = tix; i < 64 * 64; i += 256) As[i] = 0.001f * (i & 255);
        __syncthreads();
        unsigned char *my = smem + w * (64 * ROWB);
        f16x8 pm0, pm1;
        perm_operands(lane, pm0, pm1);
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        KStage sA;
        int key0 = ((wg * 4 + w) * 64) % (S - 64);
        while (wall_clock64() < t_end) {
            if (a_mode == 0) { __builtin_amdgcn_s_sleep(8); continue; }
            for (int ph = 0; ph < 2; ++ph) {
                k_fetch<2>(sA, kbuf, 128, key0, S, ph, lane);
                k_commit<2>(sA, lane, my);
                mfma_phase_mx<2, 4>(acc0, acc1, my, As + ph * 32 * 64 + lane, n31, hi, pm0, pm1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            key0 = (key0 + 4096) % (S - 64);
            if (gap_ticks) { const uint64_t t1 = wall_clock64() + gap_ticks; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(8); }
        }
        float s = 0.0f;
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        if (s == 123.456f) out[0] = 1;
    } else {
        uint32_t bad = 0;
        for (int it = 0; it < 6000; ++it) {
            const float x0 = 0.5f + (float)((lane * 37 + it * 11) % 1000) * 1e-3f, x1 = 0.25f + (float)((lane * 53 + it * 7) % 1000) * 2e-3f;
            const float c0 = 0.99f, c1 = 1.01f, d0 = 0.125f, d1 = -0.0625f;
            float r0, r1;
            // v[200:201] = x; 24 times: x = x * c + d (packed, in place), c in v[202:203], d in v[204:205]
            asm volatile("v_mov_b32 v200, %2\n v_mov_b32 v201, %3\n v_mov_b32 v202, %4\n v_mov_b32 v203, %5\n v_mov_b32 v204, %6\n v_mov_b32 v205, %7\n"
                         "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[204:205]\n v_pk_fma_f32 v[208:209], v[206:207], v[202:203], v[204:205]\n"
                         "v_pk_fma_f32 v[210:211], v[208:209], v[202:203], v[204:205]\n v_pk_fma_f32 v[212:213], v[210:211], v[202:203], v[204:205]\n"
                         "v_pk_fma_f32 v[214:215], v[212:213], v[202:203], v[204:205]\n v_pk_fma_f32 v[200:201], v[214:215], v[202:203], v[204:205]\n"
                         "v_pk_mul_f32 v[206:207], v[200:201], v[202:203]\n v_pk_add_f32 v[208:209], v[206:207], v[204:205]\n"
                         "v_pk_fma_f32 v[210:211], v[208:209], v[202:203], v[204:205]\n v_pk_fma_f32 v[212:213], v[210:211], v[202:203], v[204:205]\n"
                         "v_pk_fma_f32 v[214:215], v[212:213], v[202:203], v[204:205]\n v_pk_fma_f32 v[200:201], v[214:215], v[202:203], v[204:205]\n"
                         "s_nop 4\n v_mov_b32 %0, v200\n v_mov_b32 %1, v201\n"
                         : "=v"(r0), "=v"(r1) : "v"(x0), "v"(x1), "v"(c0), "v"(c1), "v"(d0), "v"(d1) : CLOB);
            float y0 = x0, y1 = x1;
            for (int k = 0; k < 6; ++k) { y0 = __builtin_fmaf(y0, c0, d0); y1 = __builtin_fmaf(y1, c1, d1); }
            y0 = y0 * c0; y1 = y1 * c1; y0 = y0 + d0; y1 = y1 + d1;
            for (int k = 0; k < 4; ++k) { y0 = __builtin_fmaf(y0, c0, d0); y1 = __builtin_fmaf(y1, c1, d1); }
            bad += (f32_bits(r0) != f32_bits(y0)) + (f32_bits(r1) != f32_bits(y1));
        }
        bad = __reduce_add_sync(~0ull, bad);
        if (lane == 0) atomicAdd(&out[wg], bad);
    }
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20, S = 32768;
    uint32_t *d;
    uint16_t *kb;
    static uint32_t h[512];
    CK(hipMalloc(&d, sizeof(h)));
    CK(hipMalloc(&kb, (size_t)S * 128 * 2));
    CK(hipMemset(kb, 0x3c, (size_t)S * 128 * 2));
    for (int a_mode = 0; a_mode <= 1; ++a_mode)
        for (int gap = 0; gap <= 1200; gap += 600) {
            long bad = 0;
            for (int r = 0; r < reps; ++r) {
                CK(hipMemset(d, 0, sizeof(h)));
                hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, kb, S, a_mode, gap);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                for (int i = 256; i < 512; ++i) bad += h[i];
            }
            printf("A %s, %d us between its tiles: %ld packed results differ from the scalar chain in %d launches\n", a_mode ? "in its matrix phase" : "asleep", gap / 100, bad, reps);
        }
    return 0;
}
