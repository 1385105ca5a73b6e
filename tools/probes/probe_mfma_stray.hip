// Probe: does an MFMA of one wave write into the SAME-NUMBERED registers of the other wave of its SIMD?  Both waves own 256 VGPRs (the
// kernel touches v255).  Workgroups 0..255 ("A") issue bursts of v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16 into v[192:255] with
// pauses between the bursts; workgroups 256..511 ("B", their partners on the compute units) park a pattern in THEIR v[192:255], idle or
// do vector work in low registers for ~150 us, and read v[192:255] back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int LDS_BYTES = 80896;
#define CLOB64 "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207", \
               "v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223", \
               "v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239", \
               "v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
#define W4(n0) "v_add_u32 v" #n0 ", %0, %1\n"
#define SET16(b) "v_add_u32 v" #b ", " "%0, %1\n"
__global__ void __launch_bounds__(256, 2) probe(uint32_t *out, int a_mode, int b_mode)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const uint32_t wg = blockIdx.x, tix = threadIdx.x, lane = tix & 63;
    for (int i = tix; i < LDS_BYTES; i += 256) smem[i] = (unsigned char)i;
    __syncthreads();
    if (smem[(tix * 977 + a_mode) % LDS_BYTES] == 255 && a_mode == 77) out[0] = 1;
    const uint64_t t_end = wall_clock64() + 15000;                // 150 us
    if (wg < 256) {
        float a = 1.0f + lane * 1e-3f, b = 0.5f;
        f16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * lane); }
        asm volatile("v_mov_b32 v192, 0\n v_mov_b32 v208, 0\n v_mov_b32 v224, 0\n v_mov_b32 v240, 0" ::: CLOB64);
        while (wall_clock64() < t_end) {
            if (a_mode == 0) { __builtin_amdgcn_s_sleep(8); continue; }
            for (int r = 0; r < 32; ++r)
                asm volatile("v_mfma_f32_32x32x2_f32 v[192:207], %0, %1, v[192:207]\n"
                             "v_mfma_f32_32x32x2_f32 v[208:223], %1, %0, v[208:223]\n"
                             "v_mfma_f32_32x32x16_f16 v[224:239], %2, %3, v[224:239]\n"
                             "v_mfma_f32_32x32x2_f32 v[240:255], %0, %0, v[240:255]\n"
                             :: "v"(a), "v"(b), "v"(ha), "v"(hb) : CLOB64);
            if (a_mode == 1) { const uint64_t t1 = wall_clock64() + 300; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(4); }
        }
        out[wg] = 0;
    } else {
        const uint32_t base = lane * 2654435761u + wg * 40503u;
        // v[192 + k] = base + k (k = 0..63), parked for the whole run
        asm volatile(
            "v_add_u32 v192, %0, 0\n v_add_u32 v193, %0, 1\n v_add_u32 v194, %0, 2\n v_add_u32 v195, %0, 3\n v_add_u32 v196, %0, 4\n v_add_u32 v197, %0, 5\n v_add_u32 v198, %0, 6\n v_add_u32 v199, %0, 7\n"
            "v_add_u32 v200, %0, 8\n v_add_u32 v201, %0, 9\n v_add_u32 v202, %0, 10\n v_add_u32 v203, %0, 11\n v_add_u32 v204, %0, 12\n v_add_u32 v205, %0, 13\n v_add_u32 v206, %0, 14\n v_add_u32 v207, %0, 15\n"
            "v_add_u32 v208, %0, 16\n v_add_u32 v209, %0, 17\n v_add_u32 v210, %0, 18\n v_add_u32 v211, %0, 19\n v_add_u32 v212, %0, 20\n v_add_u32 v213, %0, 21\n v_add_u32 v214, %0, 22\n v_add_u32 v215, %0, 23\n"
            "v_add_u32 v216, %0, 24\n v_add_u32 v217, %0, 25\n v_add_u32 v218, %0, 26\n v_add_u32 v219, %0, 27\n v_add_u32 v220, %0, 28\n v_add_u32 v221, %0, 29\n v_add_u32 v222, %0, 30\n v_add_u32 v223, %0, 31\n"
            "v_add_u32 v224, %0, 32\n v_add_u32 v225, %0, 33\n v_add_u32 v226, %0, 34\n v_add_u32 v227, %0, 35\n v_add_u32 v228, %0, 36\n v_add_u32 v229, %0, 37\n v_add_u32 v230, %0, 38\n v_add_u32 v231, %0, 39\n"
            "v_add_u32 v232, %0, 40\n v_add_u32 v233, %0, 41\n v_add_u32 v234, %0, 42\n v_add_u32 v235, %0, 43\n v_add_u32 v236, %0, 44\n v_add_u32 v237, %0, 45\n v_add_u32 v238, %0, 46\n v_add_u32 v239, %0, 47\n"
            "v_add_u32 v240, %0, 48\n v_add_u32 v241, %0, 49\n v_add_u32 v242, %0, 50\n v_add_u32 v243, %0, 51\n v_add_u32 v244, %0, 52\n v_add_u32 v245, %0, 53\n v_add_u32 v246, %0, 54\n v_add_u32 v247, %0, 55\n"
            "v_add_u32 v248, %0, 56\n v_add_u32 v249, %0, 57\n v_add_u32 v250, %0, 58\n v_add_u32 v251, %0, 59\n v_add_u32 v252, %0, 60\n v_add_u32 v253, %0, 61\n v_add_u32 v254, %0, 62\n v_add_u32 v255, %0, 63\n"
            :: "v"(base) : CLOB64);
        // meanwhile: nothing, or vector work in low registers
        float f = (float)lane;
        while (wall_clock64() < t_end) {
            if (b_mode == 0) __builtin_amdgcn_s_sleep(4);
            else for (int q = 0; q < 64; ++q) f = __builtin_fmaf(f, 1.0000001f, 1e-30f);
        }
        // read back: xor of (v[192 + k] - base - k) over k must be 0, and count the registers that differ
        uint32_t bad = 0;
#define RD(k) { uint32_t t; asm volatile("v_mov_b32 %0, v" #k : "=v"(t) :: ); bad += t != base + (k - 192); }
        RD(192) RD(193) RD(194) RD(195) RD(196) RD(197) RD(198) RD(199) RD(200) RD(201) RD(202) RD(203) RD(204) RD(205) RD(206) RD(207)
        RD(208) RD(209) RD(210) RD(211) RD(212) RD(213) RD(214) RD(215) RD(216) RD(217) RD(218) RD(219) RD(220) RD(221) RD(222) RD(223)
        RD(224) RD(225) RD(226) RD(227) RD(228) RD(229) RD(230) RD(231) RD(232) RD(233) RD(234) RD(235) RD(236) RD(237) RD(238) RD(239)
        RD(240) RD(241) RD(242) RD(243) RD(244) RD(245) RD(246) RD(247) RD(248) RD(249) RD(250) RD(251) RD(252) RD(253) RD(254) RD(255)
        if (f == 123.456f) bad += 1000;
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
            printf("b_mode %d a_mode %d: %ld parked register values changed in %d launches\n", b_mode, a_mode, bad, reps);
        }
    return 0;
}
