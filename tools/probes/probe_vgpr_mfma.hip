// Probe: do the registers of a wave survive while the OTHER wave of its SIMD (another workgroup on the same compute unit) issues MFMAs,
// when BOTH kernels own 256 VGPRs (two waves fill the SIMD's register file, as the fused scoring kernel's do)?
// 512 workgroups x 256 threads, 80 KiB of LDS each (two per compute unit, workgroup i beside i + 256).
//   A (0..255):   bursts of v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16 whose accumulators sit in v[192:255], with sleeps between
//                 the bursts (a_mode 1) or without (a_mode 2); a_mode 0 = asleep
//   B (256..511): keeps 160 values in registers (v[64:223]), and for ~150 us re-checks them against their closed form, exchanging them
//                 through ds_bpermute (b_mode 1) or not (b_mode 0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int LDS_BYTES = 80896, NR = 160;
__global__ void __launch_bounds__(256, 2) probe(uint32_t *out, int a_mode, int b_mode)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const uint32_t wg = blockIdx.x, tix = threadIdx.x, lane = tix & 63;
    for (int i = tix; i < LDS_BYTES; i += 256) smem[i] = (unsigned char)i;
    __syncthreads();
    if (smem[(tix * 977 + a_mode) % LDS_BYTES] == 255 && a_mode == 77) out[0] = 1;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");               // the kernel owns all 256 registers
    const uint64_t t_end = wall_clock64() + 15000;                // 150 us
    if (wg < 256) {
        float a = 1.0f + lane * 1e-3f, b = 0.5f;
        f16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * lane); }
        asm volatile("v_mov_b32 v192, 0\n v_mov_b32 v208, 0\n v_mov_b32 v224, 0\n v_mov_b32 v240, 0" ::: "v192", "v208", "v224", "v240");
        while (wall_clock64() < t_end) {
            if (a_mode == 0) { __builtin_amdgcn_s_sleep(8); continue; }
            for (int r = 0; r < 32; ++r) {
                asm volatile("v_mfma_f32_32x32x2_f32 v[192:207], %0, %1, v[192:207]\n"
                             "v_mfma_f32_32x32x2_f32 v[208:223], %1, %0, v[208:223]\n"
                             "v_mfma_f32_32x32x16_f16 v[224:239], %2, %3, v[224:239]\n"
                             "v_mfma_f32_32x32x2_f32 v[240:255], %0, %0, v[240:255]\n"
                             :: "v"(a), "v"(b), "v"(ha), "v"(hb)
                             : "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207",
                               "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223",
                               "v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239",
                               "v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255");
            }
            if (a_mode == 1) { const uint64_t t1 = wall_clock64() + 300; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(4); }
        }
        out[wg] = 0;
    } else {
        uint32_t r[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { r[i] = (lane * 2654435761u) ^ (i * 40503u + wg); asm volatile("" : "+v"(r[i])); }
        uint32_t bad = 0;
        int it = 0;
        while (wall_clock64() < t_end) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                asm volatile("" : "+v"(r[i]));
                uint32_t v = r[i];
                if (b_mode == 1 && (i & 7) == (it & 7)) {
                    v = (uint32_t)__shfl_xor((int)v, 32, 64);
                    bad += v != ((((lane ^ 32) * 2654435761u) ^ (i * 40503u + wg)));
                } else bad += v != ((lane * 2654435761u) ^ (i * 40503u + wg));
            }
            ++it;
        }
        bad = __reduce_add_sync(~0ull, bad);
        if (lane == 0) atomicAdd(&out[wg], bad);
    }
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    uint32_t *d;
    static uint32_t h[512];
    CK(hipMalloc(&d, sizeof(h)));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(probe)));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, 0));
    printf("registers per thread %d, LDS %zu B, %d workgroups per compute unit\n", fa.numRegs, fa.sharedSizeBytes, occ);
    for (int b_mode = 0; b_mode <= 1; ++b_mode)
        for (int a_mode = 0; a_mode <= 2; ++a_mode) {
            long bad = 0;
            for (int r = 0; r < reps; ++r) {
                CK(hipMemset(d, 0, sizeof(h)));
                hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, a_mode, b_mode);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                for (int i = 256; i < 512; ++i) bad += h[i];
            }
            printf("b_mode %d a_mode %d: %ld register values wrong in %d launches\n", b_mode, a_mode, bad, reps);
        }
    return 0;
}
