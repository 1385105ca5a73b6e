// Probe: the pair the hunt of docs/HISTORY.md ended at, outside the kernel.  Workgroups 0..255 ("A") run the fused kernel's matrix phase
// (mfma_tile.h: K from LDS into fp16 MFMAs, their results -- CUT 4: copied by v_mov_b32 -- into fp32 MFMAs) in bursts with sleeps between;
// workgroups 256..511 ("B", their partners on the compute units; both own 80 KiB of LDS and 256 registers) run chains of PACKED fp32
// arithmetic (the softmax phases' exponential) on inputs that depend on (lane, iteration) only and keep an XOR checksum per thread.
// Reference = the same launch with A asleep.  b_mode 1 = the same chains with scalar instructions.
#include "../../fastkv_amd/csrc/mfma_tile.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace fk;
constexpr int LDS_BYTES = 80896;
__device__ __forceinline__ float expo1(float d)
{
    float dc = fmaxf(d, -87.0f);
    float n = __builtin_rintf(dc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, dc);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f); p = __builtin_fmaf(p, r, 8.3334519073e-3f); p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f); p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p, r * r, r);
    p = p + 1.0f;
    return __builtin_bit_cast(float, (uint32_t)((int32_t)__builtin_bit_cast(uint32_t, p) + (int32_t)n * (1 << 23)));
}
template <int CUT>
__global__ void __launch_bounds__(256, 2) probe(uint32_t *out, const uint16_t *kbuf, int S, int a_mode, int b_mode, int gap_ticks)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const uint32_t wg = blockIdx.x, tix = threadIdx.x;
    const int lane = tix & 63, w = tix >> 6, n31 = lane & 31, hi = lane >> 5;
    const uint64_t t_end = wall_clock64() + 15000;                // 150 us
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
                mfma_phase_mx<2, CUT>(acc0, acc1, my, As + ph * 32 * 64 + lane, n31, hi, pm0, pm1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            key0 = (key0 + 4096) % (S - 64);
            if (gap_ticks) { const uint64_t t1 = wall_clock64() + gap_ticks; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(8); }
        }
        float s = 0.0f;
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        if (s == 123.456f) out[0] = 1;
        out[wg * 256 + tix] = 0;
    } else {
        for (int i = tix; i < LDS_BYTES / 4; i += 256) reinterpret_cast<uint32_t *>(smem)[i] = i;
        __syncthreads();
        uint32_t cs = 0;
        for (int it = 0; it < 12000; ++it) {                     // ~100 us of packed arithmetic
            const float x0 = -((float)((lane * 37 + it * 11) % 1000)) * 0.02f, x1 = -((float)((lane * 53 + it * 7) % 1000)) * 0.03f;
            if (b_mode == 0) {
                const f32x2 e = det_expf2_clamped((f32x2){x0, x1});          // the kernel's own packed exponential
                const f32x2 pr = e * (f32x2){0.37f, 0.11f};
                f32x2 s2 = splat2(0.0f);
                s2 = s2 + pr; s2 = s2 + e;
                cs ^= f32_bits(e.x) * 3u ^ f32_bits(e.y) * 5u ^ f32_bits(s2.x) * 7u ^ f32_bits(s2.y) * 11u;
            } else {
                const float e0 = expo1(x0), e1 = expo1(x1);
                const float s0 = (0.0f + e0 * 0.37f) + e0, s1 = (0.0f + e1 * 0.11f) + e1;
                cs ^= f32_bits(e0) * 3u ^ f32_bits(e1) * 5u ^ f32_bits(s0) * 7u ^ f32_bits(s1) * 11u;
            }
            cs = (cs << 1) | (cs >> 31);
        }
        out[wg * 256 + tix] = cs;
    }
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20, S = 32768;
    uint32_t *d;
    uint16_t *kb;
    static uint32_t ref[512 * 256], cur[512 * 256];
    CK(hipMalloc(&d, sizeof(ref)));
    CK(hipMalloc(&kb, (size_t)S * 128 * 2));
    CK(hipMemset(kb, 0x3c, (size_t)S * 128 * 2));
    for (int b_mode = 0; b_mode <= 1; ++b_mode) {
        for (int cut = 0; cut <= 4; cut += 4)
            for (int gap = 0; gap <= 1200; gap += 600) {
                long bad = 0;
                for (int r = 0; r < reps + 1; ++r) {
                    const int a_mode = r == 0 ? 0 : 1;            // launch 0: A asleep = the reference
                    if (cut == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), 0, 0, d, kb, S, a_mode, b_mode, gap);
                    else hipLaunchKernelGGL(probe<4>, dim3(512), dim3(256), 0, 0, d, kb, S, a_mode, b_mode, gap);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(r == 0 ? ref : cur, d, sizeof(ref), hipMemcpyDeviceToHost));
                    if (r) for (int i = 256 * 256; i < 512 * 256; ++i) bad += cur[i] != ref[i];
                }
                printf("B %s, A matrix phase cut %d, %d us between A's tiles: %ld checksums differ in %d launches\n", b_mode ? "scalar" : "packed", cut, gap / 100, bad, reps);
            }
    }
    return 0;
}
