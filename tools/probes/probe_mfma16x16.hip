// Is v_mfma_f32_16x16x4_f32 (16 query rows x 16 keys, FOUR k-steps per instruction) the ascending-d fmaf chain the arithmetic
// contract prescribes, like v_mfma_f32_32x32x2_f32 is?  It would let a 16-row block (2 query heads x window 8: G = 2 models)
// use the fused scoring kernel.  Compares the instruction with CPU models on operands whose products span a wide exponent
// range (so that the order of the four additions shows).
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma16x16 probe_mfma16x16.hip && ./probe_mfma16x16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D, int tiles)
{
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        // A [16][4], B [4][16], C/D [16][16] per tile
        const float a = A[t * 64 + (lane % 16) * 4 + lane / 16];
        const float b = B[t * 64 + (lane / 16) * 16 + lane % 16];
        f32x4 c;
        for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + ((lane / 16) * 4 + r) * 16 + lane % 16];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[t * 256 + ((lane / 16) * 4 + r) * 16 + lane % 16] = c[r];
    }
}
static float h(float x) { return (float)(_Float16)x; }
int main()
{
    const int tiles = 4096;
    std::vector<float> A(tiles * 64), B(tiles * 64), C(tiles * 256), D(tiles * 256);
    srand(7);
    auto rnd = [&]() { const float m = (rand() / (float)RAND_MAX) * 2 - 1; const int e = rand() % 14 - 7; return h(ldexpf(m, e)); };
    for (auto &x : A) x = rnd();
    for (auto &x : B) x = rnd();
    for (auto &x : C) x = ldexpf((rand() / (float)RAND_MAX) * 2 - 1, rand() % 20 - 10);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, tiles);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad_seq = 0, bad_rev = 0, bad_pair = 0, bad_exact = 0, n = 0;
    for (int t = 0; t < tiles; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const float *a = &A[t * 64 + i * 4];
                float b[4];
                for (int kk = 0; kk < 4; ++kk) b[kk] = B[t * 64 + kk * 16 + j];
                const float c = C[t * 256 + i * 16 + j], d = D[t * 256 + i * 16 + j];
                float s = c;
                for (int kk = 0; kk < 4; ++kk) s = fmaf(a[kk], b[kk], s);
                float r = c;
                for (int kk = 3; kk >= 0; --kk) r = fmaf(a[kk], b[kk], r);
                const float p = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]) + c;
                const float e = (float)((double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2] + (double)a[3] * b[3] + (double)c);
                ++n;
                bad_seq += d != s; bad_rev += d != r; bad_pair += d != p; bad_exact += d != e;
            }
    printf("v_mfma_f32_16x16x4_f32 vs CPU models over %ld outputs: ascending fmaf chain %ld mismatches, descending chain %ld, pairwise %ld, "
           "exactly rounded sum %ld\n", n, bad_seq, bad_rev, bad_pair, bad_exact);
    return 0;
}
