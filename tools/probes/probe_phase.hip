// Probe: cycles of one score_logits MFMA phase (64 MFMAs) for a solo wave per SIMD, by ingredient.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int ROWB = 144;
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
struct KGroup { uint4 k0, k1; float a0, a1, a2, a3; };
template <int VAR>
__device__ __forceinline__ KGroup read_group(const unsigned char *my, const float *Ap, int n31, int c)
{
    KGroup g;
    if (VAR & 1) {   // no LDS: synthesize
        g.k0 = make_uint4(0x3c003c00u + c, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); g.k1 = g.k0;
        g.a0 = 1.0f + c; g.a1 = 2.0f; g.a2 = 3.0f; g.a3 = 4.0f;
        asm volatile("" : "+v"(g.k0.x), "+v"(g.k0.y), "+v"(g.k0.z), "+v"(g.k0.w));
        asm volatile("" : "+v"(g.k1.x), "+v"(g.k1.y), "+v"(g.k1.z), "+v"(g.k1.w));
        return g;
    }
    g.k0 = *reinterpret_cast<const uint4 *>(my + n31 * ROWB + c * 16);
    g.k1 = *reinterpret_cast<const uint4 *>(my + (32 + n31) * ROWB + c * 16);
    g.a0 = Ap[(c * 4 + 0) * 64]; g.a1 = Ap[(c * 4 + 1) * 64]; g.a2 = Ap[(c * 4 + 2) * 64]; g.a3 = Ap[(c * 4 + 3) * 64];
    return g;
}
struct BGroup { float b0[4], b1[4], a[4]; };
template <int VAR>
__device__ __forceinline__ float cvt_lo_hi(uint32_t wd, int sh)
{
    if (VAR & 2) return __builtin_bit_cast(float, wd);      // no conversion
    return h2f((uint16_t)((wd >> sh) & 0xffffu));
}
template <int VAR>
__device__ __forceinline__ void mfma_phase(f32x16 &acc0, f32x16 &acc1, const unsigned char *my, const float *Ap, int n31, int sh)
{
    KGroup r1 = read_group<VAR>(my, Ap, n31, 0), r2 = read_group<VAR>(my, Ap, n31, 1);
    BGroup cur;
    {
        const uint32_t w0[4] = {r1.k0.x, r1.k0.y, r1.k0.z, r1.k0.w}, w1[4] = {r1.k1.x, r1.k1.y, r1.k1.z, r1.k1.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) { cur.b0[u] = cvt_lo_hi<VAR>(w0[u], sh); cur.b1[u] = cvt_lo_hi<VAR>(w1[u], sh); }
        cur.a[0] = r1.a0; cur.a[1] = r1.a1; cur.a[2] = r1.a2; cur.a[3] = r1.a3;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        KGroup r3 = r2;
        if (c + 2 < 8) r3 = read_group<VAR>(my, Ap, n31, c + 2);
        BGroup nxt = cur;
        const uint32_t w0[4] = {r2.k0.x, r2.k0.y, r2.k0.z, r2.k0.w}, w1[4] = {r2.k1.x, r2.k1.y, r2.k1.z, r2.k1.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b0[u], acc0, 0, 0, 0);
            if (!(VAR & 4)) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b1[u], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b1[u], acc0, 0, 0, 0);    // single chain
            if (c + 1 < 8) { nxt.b0[u] = cvt_lo_hi<VAR>(w0[u], sh); nxt.b1[u] = cvt_lo_hi<VAR>(w1[u], sh); }
        }
        if (c + 1 < 8) { nxt.a[0] = r2.a0; nxt.a[1] = r2.a1; nxt.a[2] = r2.a2; nxt.a[3] = r2.a3; }
        if (VAR & 16) {    // keep this group's operands alive to the end of the group: the conversions get other registers
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" :: "v"(cur.b0[u]), "v"(cur.b1[u]));
        }
        cur = nxt;
        r2 = r3;
        if (!(VAR & 8)) {
            if (c + 2 < 8 && !(VAR & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (c + 1 < 8 && !(VAR & 2)) __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            }
        }
    }
}
// matrix-pipe conversion: per 32-dim chunk and 32-key block two v_mfma_f32_32x32x16_f16 with a permutation A operand turn
// the fp16 K piece (the lane's ds_read_b128) into the 16 fp32 B operands of the chunk's k-steps.  No VALU in the loop.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_phase_mx(f32x16 &acc0, f32x16 &acc1, const unsigned char *my, const float *Ap, int n31, int hi,
                                              f16x8 p0, f16x8 p1)
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    // chunk ch (32 dims = 64 B per key row): pieces 4ch .. 4ch+3; slice t uses piece 4ch + 2t + hi
    auto conv = [&](int blk, int ch) {
        const unsigned char *rowp = my + (blk * 32 + n31) * ROWB + ch * 64 + hi * 16;
        const f16x8 s0 = *reinterpret_cast<const f16x8 *>(rowp), s1 = *reinterpret_cast<const f16x8 *>(rowp + 32);
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, s0, z, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, s1, d, 0, 0, 0);
    };
    f32x16 b0 = conv(0, 0), b1 = conv(1, 0);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        f32x16 n0 = b0, n1 = b1;
        float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] = Ap[(ch * 16 + i) * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1[i], acc1, 0, 0, 0);
            if (ch == 0 && i == 7) { n0 = conv(0, 1); n1 = conv(1, 1); }
        }
        b0 = n0; b1 = n1;
    }
}
template <int VAR>
__global__ void __launch_bounds__(256, 2) k(float *out, long long *st, int phases)
{
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][64 * ROWB];
    __shared__ float As[64 * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) As[i] = (float)(i % 13) * 0.25f;
    for (int i = threadIdx.x; i < 4 * 64 * ROWB / 4; i += 256) reinterpret_cast<uint32_t *>(&slab[0][0])[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    f32x16 a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = 0; a1[i] = 0; }
    const int n31 = lane & 31, sh = (lane >> 5) * 16;
    long long c0 = clock64();
    f16x8 p0, p1;
    for (int e = 0; e < 8; ++e) { p0[e] = (_Float16)((lane + e) % 7 == 0 ? 1.0f : 0.0f); p1[e] = (_Float16)((lane + e) % 5 == 0 ? 1.0f : 0.0f); }
    for (int p = 0; p < phases; ++p) {
        if (VAR == 32) mfma_phase_mx(a0, a1, slab[w], As + (p & 1) * 32 * 64 + lane, n31, lane >> 5, p0, p1);
        else mfma_phase<VAR>(a0, a1, slab[w], As + (p & 1) * 32 * 64 + lane, n31, sh);
        __builtin_amdgcn_sched_barrier(0);
    }
    long long c1 = clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) st[0] = c1 - c0;
}
template <int VAR> void run(const char *name, float *out, long long *st)
{
    for (int grid : {256, 512}) {
        long long h;
        for (int rep = 0; rep < 2; ++rep) { k<VAR><<<grid, 256>>>(out, st, 16); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 10; ++rep) k<VAR><<<grid, 256>>>(out, st, 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s grid %3d: %6.1f cycles per MFMA (wave 0 view, %d waves/SIMD); kernel %6.1f us = %5.1f ns per MFMA per SIMD\n", name, grid,
               (double)h / (16 * 64), grid / 256, ms * 100, ms * 100 * 1e3 / (16 * 64 * (grid / 256)));
    }
}
int main()
{
    float *out; long long *st; CK(hipMalloc(&out, 64)); CK(hipMalloc(&st, 64));
    run<0>("as in the kernel", out, st);
    run<1>("no LDS reads", out, st);
    run<2>("no conversion", out, st);
    run<3>("no LDS reads, no conversion", out, st);
    run<4>("single accumulator chain", out, st);
    run<16>("operands kept alive (distinct conversion regs)", out, st);
    run<32>("conversion on the matrix pipe (fp16 MFMA x permutation)", out, st);
    run<8>("no sched_group_barrier (compiler order)", out, st);
    run<9>("no LDS, compiler order", out, st);
    return 0;
}
