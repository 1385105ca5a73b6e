// Probe: does vector-ALU work that is INDEPENDENT of the fp32 MFMAs hide behind them?
// (probe_phase.hip priced DEPENDENT conversions: a solo wave ran at 85 instead of 66 cycles per MFMA.)
// A wave runs `iters` blocks of 16 v_mfma_f32_32x32x2_f32 (two accumulator chains, loop-invariant operands) and
// F filler instructions per MFMA that touch other registers only.  Modes:
//   0  interleaved      MFMA, F fillers, MFMA, F fillers, ...                (software-pipelined phases inside one wave)
//   1  back to back     16 MFMAs, then 16*F fillers                           (what a lock-step phase structure does)
//   2  staggered roles  as 1, but waves 4-7 of the 512-thread workgroup run the filler block FIRST
//                       (SIMD partners out of phase: one in its matrix block while the other is in its vector block)
// Filler kinds: 0 v_fma_f32, 1 v_pk_fma_f32, 2 the mix of the softmax phases (pk_fma / rndne / cvt / integer).
// Geometry: one 512-thread workgroup per CU (2 waves per SIMD, partners = wave w and w+4) or one 256-thread workgroup
// per CU (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND> __device__ __forceinline__ void filler(float (&f)[8], f32x2 (&p)[4], int (&n)[4], int j)
{
    if (KIND == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j & 7]) : "v"(f[(j + 3) & 7]), "v"(1.0f));
    } else if (KIND == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3]));
    } else {
        switch (j & 7) {
        case 0: case 3: case 5: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3])); break;
        case 1: asm volatile("v_rndne_f32 %0, %1" : "=v"(f[1]) : "v"(f[2])); break;
        case 2: asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[0]) : "v"(f[3])); break;
        case 4: asm volatile("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(n[1]) : "v"(n[2]), "v"(n[3])); break;
        case 6: asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(f[4]) : "v"(f[5])); break;
        default: asm volatile("v_rndne_f32 %0, %1" : "=v"(f[6]) : "v"(f[7])); break;
        }
    }
}

template <int MODE, int KIND, int F, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float *out, long long *st, int iters)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x16 a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 0; }
    float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    float f[8]; f32x2 p[4]; int n[4];
    for (int i = 0; i < 8; ++i) f[i] = 0.25f + i * 0.01f + lane * 1e-4f;
    for (int i = 0; i < 4; ++i) { p[i] = (f32x2){0.5f + i * 0.01f, 0.3f}; n[i] = i + lane; }
    const bool second = (MODE == 2) && (w >= THREADS / 128);
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m & 1) MFMA(a1); else MFMA(a0);
#pragma unroll
                for (int j = 0; j < F; ++j) filler<KIND>(f, p, n, m * F + j);
            }
        } else {
            if (!second) {
#pragma unroll
                for (int m = 0; m < 16; ++m) { if (m & 1) MFMA(a1); else MFMA(a0); }
            }
#pragma unroll
            for (int j = 0; j < 16 * F; ++j) filler<KIND>(f, p, n, j);
            if (second) {
#pragma unroll
                for (int m = 0; m < 16; ++m) { if (m & 1) MFMA(a1); else MFMA(a0); }
            }
        }
    }
    const long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y + (float)n[i];
    if (s == 1234.5f) out[0] = s;
    if (lane == 0 && blockIdx.x == 7) st[w] = c1 - c0;
}

template <int MODE, int KIND, int F, int THREADS> void run(float *out, long long *st)
{
    const int iters = 64;
    long long h[8];
    for (int rep = 0; rep < 2; ++rep) { k<MODE, KIND, F, THREADS><<<256, THREADS>>>(out, st, iters); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 10; ++rep) k<MODE, KIND, F, THREADS><<<256, THREADS>>>(out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int wps = THREADS / 256;
    const char *modes[3] = {"interleaved", "back-to-back", "staggered"}, *kinds[3] = {"fma", "pk_fma", "mix"};
    printf("%-13s %-6s F=%2d %d wave/SIMD: wave0 %6.1f, last wave %6.1f cycles per MFMA; kernel %7.1f us = %5.1f ns per MFMA per SIMD\n",
           modes[MODE], kinds[KIND], F, wps, (double)h[0] / (iters * 16), (double)h[THREADS / 64 - 1] / (iters * 16), ms * 100,
           ms * 100 * 1e3 / (iters * 16 * wps));
}

template <int KIND, int F> void sweep(float *out, long long *st)
{
    run<0, KIND, F, 256>(out, st);
    run<1, KIND, F, 256>(out, st);
    run<0, KIND, F, 512>(out, st);
    run<1, KIND, F, 512>(out, st);
    run<2, KIND, F, 512>(out, st);
}

int main()
{
    float *out; long long *st; CK(hipMalloc(&out, 64)); CK(hipMalloc(&st, 64));
    run<0, 0, 0, 256>(out, st);
    run<0, 0, 0, 512>(out, st);
    sweep<0, 4>(out, st); sweep<0, 8>(out, st); sweep<0, 12>(out, st); sweep<0, 16>(out, st);
    sweep<1, 4>(out, st); sweep<1, 8>(out, st); sweep<1, 16>(out, st);
    sweep<2, 8>(out, st); sweep<2, 12>(out, st); sweep<2, 16>(out, st); sweep<2, 20>(out, st);
    return 0;
}
