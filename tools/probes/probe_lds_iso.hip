// Probe: are the LDS blocks of two workgroups that share a compute unit isolated from each other when each block is ~80 KiB
// (two of them fill the unit's 160 KiB)?  Workgroups 0..255 ("A") write a recognisable pattern into the first 36 KiB of their
// block with ds_write_b128 in bursts separated by sleeps; workgroups 256..511 ("B", one per compute unit beside an A) fill
// their whole block with their own pattern and keep re-reading it for ~200 us, counting words that are not theirs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int LDS_BYTES = 80896, NV = LDS_BYTES / 16;
__device__ __forceinline__ uint32_t pat(uint32_t wg, uint32_t i) { return 0xB0000000u | (wg << 16) | (i & 0xffffu); }
__global__ void __launch_bounds__(256) probe(uint32_t *out, int a_bytes, int mode)
{
    __shared__ __attribute__((aligned(16))) uint4 buf[NV];
    const uint32_t wg = blockIdx.x, tix = threadIdx.x;
    const uint64_t t_end = wall_clock64() + 20000;                // 200 us
    uint32_t bad = 0, first_i = 0, first_v = 0;
    if (wg < 256) {
        const uint4 v = make_uint4(0xA0000000u | wg, 0xA1000000u | wg, 0xA2000000u | wg, 0xA3000000u | wg);
        while (wall_clock64() < t_end) {
            if (mode & 1) { const uint64_t t1 = wall_clock64() + 500; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(8); }
            for (int i = tix; i < a_bytes / 16; i += 256) buf[i] = v;
            __syncthreads();
        }
    } else {
        for (int i = tix; i < NV; i += 256) buf[i] = make_uint4(pat(wg, 4 * i), pat(wg, 4 * i + 1), pat(wg, 4 * i + 2), pat(wg, 4 * i + 3));
        __syncthreads();
        while (wall_clock64() < t_end) {
            for (int i = tix; i < NV; i += 256) {
                const uint4 r = buf[i];
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
                for (int e = 0; e < 4; ++e)
                    if (rr[e] != pat(wg, 4 * i + e)) { if (!bad) { first_i = 4 * i + e; first_v = rr[e]; } ++bad; }
            }
            __syncthreads();
            if (mode & 2) {                                       // rewrite the block (a workgroup in its later phases writes too)
                for (int i = tix; i < NV; i += 256) buf[i] = make_uint4(pat(wg, 4 * i), pat(wg, 4 * i + 1), pat(wg, 4 * i + 2), pat(wg, 4 * i + 3));
                __syncthreads();
            }
        }
    }
    // per workgroup: [0] mismatching words, [1] first index, [2] first value, [3] HW_ID, [4] XCC_ID
    atomicAdd(&out[wg * 8 + 0], bad);
    if (bad) { atomicMax(&out[wg * 8 + 1], first_i); atomicMax(&out[wg * 8 + 2], first_v); }
    if (tix == 0) { out[wg * 8 + 3] = __builtin_amdgcn_s_getreg(63492); out[wg * 8 + 4] = __builtin_amdgcn_s_getreg(63508); }
}
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 3, a_bytes = argc > 2 ? atoi(argv[2]) : 36864, reps = argc > 3 ? atoi(argv[3]) : 20;
    uint32_t *d, h[512 * 8];
    CK(hipMalloc(&d, sizeof(h)));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, 0));
    printf("occupancy %d workgroups per compute unit\n", occ);
    long total_bad = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(d, 0, sizeof(h)));
        hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, a_bytes, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        int shared = 0;
        for (int b = 256; b < 512; ++b)
            for (int a = 0; a < 256; ++a)
                if (((h[a * 8 + 3] >> 8) & 0xff) == ((h[b * 8 + 3] >> 8) & 0xff) && ((h[a * 8 + 3] >> 13) & 7) == ((h[b * 8 + 3] >> 13) & 7) &&
                    (h[a * 8 + 4] & 15) == (h[b * 8 + 4] & 15)) { ++shared; break; }
        for (int b = 0; b < 512; ++b)
            if (h[b * 8]) {
                total_bad += h[b * 8];
                if (total_bad < 100000) printf("rep %d wg %d: %u foreign words, first at word %u value %08x\n", r, b, h[b * 8], h[b * 8 + 1], h[b * 8 + 2]);
            }
        if (r == 0) printf("B workgroups sharing a compute unit with an A workgroup: %d of 256\n", shared);
    }
    printf("mode %d a_bytes %d: %ld foreign words in %d launches\n", mode, a_bytes, total_bad, reps);
    return 0;
}
