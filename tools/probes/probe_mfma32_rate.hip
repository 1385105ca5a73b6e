// Probe: what does v_mfma_f32_32x32x2_f32 sustain chip-wide, and what does each ingredient of the score_logits loop
// (LDS A operand, LDS K operand + fp16 unpack/convert, global prefetch) take away from it?
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma32_rate probe_mfma32_rate.hip && ./probe_mfma32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// variant 0: registers only.  1: A operand from LDS per MFMA pair.  2: + B operands from an LDS fp16 slab with unpack+cvt
// NACC accumulators in rotation.
template <int VAR, int NACC>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters, const uint16_t *gk)
{
    __shared__ float As[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][64 * 144];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) As[i] = (float)(i % 13) * 0.25f;
    for (int i = threadIdx.x; i < 4 * 64 * 144 / 4; i += 256) reinterpret_cast<uint32_t *>(&slab[0][0])[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    float a = (float)lane, b0 = 1.0f + lane, b1 = 2.0f + lane;
    const int n31 = lane & 31, sh = (lane >> 5) * 16;
    unsigned char *my = slab[w];
    for (int it = 0; it < iters; ++it) {       // one iteration = 64 MFMAs (one 64-dim phase of a 64-key tile)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t w0[4], w1[4];
            if (VAR >= 2) {
                const uint4 k0 = *reinterpret_cast<const uint4 *>(my + n31 * 144 + c * 16);
                const uint4 k1 = *reinterpret_cast<const uint4 *>(my + (32 + n31) * 144 + c * 16);
                w0[0] = k0.x; w0[1] = k0.y; w0[2] = k0.z; w0[3] = k0.w; w1[0] = k1.x; w1[1] = k1.y; w1[2] = k1.z; w1[3] = k1.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (VAR >= 1) a = As[((it & 1) * 32 + c * 4 + u) * 64 + lane];
                if (VAR >= 2) {
                    b0 = (float)__builtin_bit_cast(_Float16, (uint16_t)((w0[u] >> sh) & 0xffffu));
                    b1 = (float)__builtin_bit_cast(_Float16, (uint16_t)((w1[u] >> sh) & 0xffffu));
                }
                acc[(2 * u) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[(2 * u) % NACC], 0, 0, 0);
                acc[(2 * u + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[(2 * u + 1) % NACC], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int a2 = 0; a2 < NACC; ++a2) for (int i = 0; i < 16; ++i) s += acc[a2][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// a dependent VALU chain: calibrates the shader clock (v_fma_f32 dependent issue = known cycles) under the same load shape
__global__ void __launch_bounds__(256) k_valu(float *out, int iters)
{
    float x = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    }
    if (x == 12345.678f) out[threadIdx.x] = x;
}

template <typename F> float time_it(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3f;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4096));
    uint16_t *gk; CK(hipMalloc(&gk, 1 << 20));
    const int iters = 64;                      // 64 x 64 = 4096 MFMAs per wave
    auto report = [&](const char *name, float us, int grid, int waves_per_wg) {
        // MFMAs per SIMD = waves per SIMD x 4096; waves per SIMD = grid*waves_per_wg / 1024
        const double per_simd = (double)grid * waves_per_wg / 1024.0 * iters * 64;
        printf("%-44s grid=%4d  %8.1f us  %6.1f ns/MFMA/SIMD = %5.1f cyc @2.4GHz   %6.1f TFLOP/s\n", name, grid, us, us * 1e3 / per_simd,
               us * 1e3 / per_simd * 2.4, (double)grid * waves_per_wg * iters * 64 * 4096.0 / (us * 1e-6) / 1e12);
    };
    for (int grid : {256, 512, 1024}) {
        report("regs only, 2 acc", time_it([&] { k_rate<0, 2><<<grid, 256>>>(out, iters, gk); }, 5), grid, 4);
        report("regs only, 4 acc", time_it([&] { k_rate<0, 4><<<grid, 256>>>(out, iters, gk); }, 5), grid, 4);
        report("A from LDS, 2 acc", time_it([&] { k_rate<1, 2><<<grid, 256>>>(out, iters, gk); }, 5), grid, 4);
        report("A + K from LDS (unpack+cvt), 2 acc", time_it([&] { k_rate<2, 2><<<grid, 256>>>(out, iters, gk); }, 5), grid, 4);
        report("A + K from LDS (unpack+cvt), 4 acc", time_it([&] { k_rate<2, 4><<<grid, 256>>>(out, iters, gk); }, 5), grid, 4);
    }
    // clock calibration: 4096 dependent v_fma per wave
    for (int grid : {1, 256, 1024}) {
        float us = time_it([&] { k_valu<<<grid, 256>>>(out, iters); }, 5);
        printf("dependent v_fma chain grid=%4d: %8.1f us = %6.2f ns per fma (1 wave/SIMD: N cycles dependent issue / clock)\n", grid, us,
               us * 1e3 / (iters * 64));
    }
    // short launches like the real kernel: 128 MFMAs x 2 tiles per wave
    for (int it2 : {2, 4, 8, 16}) {
        float us = time_it([&] { k_rate<2, 2><<<512, 256>>>(out, it2, gk); }, 20);
        printf("A+K LDS, grid 512, %3d MFMAs/wave: %7.2f us (ideal %5.2f us @2.4GHz 64cyc)\n", it2 * 64, us, it2 * 64 * 2 * 64 / 2.4e3);
    }
    return 0;
}
