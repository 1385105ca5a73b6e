// Probe: LDS isolation of two 80-KiB workgroups on one compute unit by ACCESS WIDTH.  The second workgroup's block lies above 80 KiB of the
// unit's 160 KiB: its offsets >= 48 KiB are physical addresses >= 128 KiB.  B (workgroups 256..511) fills and re-checks its block with
// 1-, 2-, 4-, 8- or 16-byte LDS accesses (or LDS atomics) while A (0..255) keeps rewriting ITS first 36 KiB with ds_write_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int LDS_BYTES = 80896;
template <typename T> __device__ __forceinline__ T pat(uint32_t wg, uint32_t i) { return (T)(0x5a5a5a5a5a5a5a5aull ^ ((uint64_t)(wg * 2654435761u + i * 40503u) * 0x9E3779B97F4A7C15ull)); }
template <typename T> __device__ uint32_t run_b(unsigned char *smem, uint32_t wg, uint32_t tix, uint64_t t_end, uint32_t &first)
{
    volatile T *buf = reinterpret_cast<volatile T *>(smem);
    const int n = LDS_BYTES / sizeof(T);
    uint32_t bad = 0;
    for (int i = tix; i < n; i += 256) buf[i] = pat<T>(wg, i);
    __syncthreads();
    while (wall_clock64() < t_end) {
        for (int i = tix; i < n; i += 256) {
            const T r = buf[i];
            if (r != pat<T>(wg, i)) { if (!bad) first = i * sizeof(T); ++bad; }
        }
        __syncthreads();
        for (int i = tix; i < n; i += 256) buf[i] = pat<T>(wg, i);
        __syncthreads();
    }
    return bad;
}
__global__ void __launch_bounds__(256) probe(uint32_t *out, int width, int a_mode)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const uint32_t wg = blockIdx.x, tix = threadIdx.x;
    const uint64_t t_end = wall_clock64() + 20000;
    uint32_t bad = 0, first = 0;
    if (wg < 256) {
        uint4 *buf = reinterpret_cast<uint4 *>(smem);
        const uint4 v = make_uint4(0xA0000000u | wg, 0xA1000000u | wg, 0xA2000000u | wg, 0xA3000000u | wg);
        for (int i = tix; i < LDS_BYTES / 16; i += 256) buf[i] = v;
        __syncthreads();
        while (wall_clock64() < t_end) {
            if (a_mode) { const uint64_t t1 = wall_clock64() + 300; while (wall_clock64() < t1) __builtin_amdgcn_s_sleep(4); }
            for (int i = tix; i < 36864 / 16; i += 256) buf[i] = v;
            __syncthreads();
            // A checks its own block too (the part it does not rewrite is the interesting one: nobody of A touches it again)
            for (int i = 36864 / 16 + tix; i < LDS_BYTES / 16; i += 256) {
                const uint4 r = buf[i];
                if (r.x != v.x || r.y != v.y || r.z != v.z || r.w != v.w) { if (!bad) first = i * 16; ++bad; }
            }
            for (int i = tix; i < 36864 / 16; i += 256) {
                const uint4 r = buf[i];
                if (r.x != v.x || r.y != v.y || r.z != v.z || r.w != v.w) { if (!bad) first = i * 16; ++bad; }
            }
            __syncthreads();
        }
    } else {
        if (width == 1) bad = run_b<uint8_t>(smem, wg, tix, t_end, first);
        else if (width == 2) bad = run_b<uint16_t>(smem, wg, tix, t_end, first);
        else if (width == 4) bad = run_b<uint32_t>(smem, wg, tix, t_end, first);
        else if (width == 8) bad = run_b<uint64_t>(smem, wg, tix, t_end, first);
        else {                                                    // LDS atomics: every word is incremented in place and must come back by one
            uint32_t *buf = reinterpret_cast<uint32_t *>(smem);
            for (int i = tix; i < LDS_BYTES / 4; i += 256) buf[i] = i;
            __syncthreads();
            uint32_t round = 0;
            while (wall_clock64() < t_end) {
                for (int i = tix; i < LDS_BYTES / 4; i += 256) {
                    const uint32_t old = atomicAdd(&buf[i], 1u);
                    if (old != (uint32_t)i + round) { if (!bad) first = i * 4; ++bad; }
                }
                ++round;
                __syncthreads();
            }
        }
    }
    atomicAdd(&out[wg * 2], bad);
    if (bad) atomicMax(&out[wg * 2 + 1], first);
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    uint32_t *d;
    static uint32_t h[1024];
    CK(hipMalloc(&d, sizeof(h)));
    const int widths[6] = {1, 2, 4, 8, 16, 0};
    for (int a_mode = 0; a_mode <= 1; ++a_mode)
        for (int wi = 0; wi < 6; ++wi) {
            const int width = widths[wi] == 16 ? 4 : widths[wi];
            if (widths[wi] == 16) continue;
            long bad_a = 0, bad_b = 0;
            uint32_t fa = 0, fb = 0;
            for (int r = 0; r < reps; ++r) {
                CK(hipMemset(d, 0, sizeof(h)));
                hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, width, a_mode);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                for (int i = 0; i < 256; ++i) { bad_a += h[i * 2]; if (h[i * 2] && !fa) fa = h[i * 2 + 1]; }
                for (int i = 256; i < 512; ++i) { bad_b += h[i * 2]; if (h[i * 2] && !fb) fb = h[i * 2 + 1]; }
            }
            printf("a_mode %d, B accesses of %d bytes%s: A saw %ld foreign words (first at byte %u), B saw %ld (first at byte %u)\n", a_mode, width,
                   width == 0 ? " (atomics)" : "", bad_a, fa, bad_b, fb);
        }
    return 0;
}
