// Probe: shader clock under a short MFMA kernel: clock64() (s_memtime) vs wall_clock64() (100 MHz) around 128 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__global__ void __launch_bounds__(256) k(float *out, long long *st, int iters)
{
    f32x16 a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 0; }
    float a = threadIdx.x, b = 1.0f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a1, 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    long long c1 = clock64(), w1 = wall_clock64();
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { st[0] = c1 - c0; st[1] = w1 - w0; }
}
int main()
{
    float *out; long long *st, h[2]; CK(hipMalloc(&out, 64)); CK(hipMalloc(&st, 64));
    for (int grid : {64, 256, 512})
        for (int iters : {2, 4, 64}) {
            for (int rep = 0; rep < 3; ++rep) {
                k<<<grid, 256>>>(out, st, iters); CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
                printf("grid %3d iters %3d (%4d MFMAs/wave): clock64 delta %8lld  wall %6lld (x10ns)  -> clock64 rate %.1f MHz; %.1f wall-ns per MFMA\n", grid, iters, iters * 64, h[0], h[1],
                       h[0] / (h[1] * 10e-9) / 1e6, h[1] * 10.0 / (iters * 64));
            }
        }
    return 0;
}
