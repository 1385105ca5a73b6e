// Probe: does a wave's packed-fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) stay exact while the OTHER wave of its
// SIMD (another workgroup on the same compute unit) issues MFMAs?  512 workgroups of 256 threads with 80 KiB of LDS each: two per
// compute unit, workgroup i and i + 256 side by side (checked through HW_ID).  Workgroups 0..255 ("A") run `a_mode` for ~150 us:
//   0 sleep   1 v_mfma_f32_32x32x2_f32 back to back   2 v_mfma_f32_32x32x16_f16   3 both interleaved with ds_read_b128
// Workgroups 256..511 ("B") run `b_mode` on inputs that depend on (lane, iteration) only and keep an XOR checksum per thread:
//   0 packed-fp32 exponential chain (det_expf2_clamped of the product)   1 the same chain with scalar v_fma_f32
// The checksums of a launch with A asleep are the reference for the launches with A busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 expo2(f32x2 d)
{
    f32x2 dc = {fmaxf(d.x, -87.0f), fmaxf(d.y, -87.0f)};
    f32x2 n = dc * splat2(1.44269504088896341f);
    n.x = __builtin_rintf(n.x); n.y = __builtin_rintf(n.y);
    f32x2 r = fma2(n, splat2(-0.693359375f), dc);
    r = fma2(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = fma2(p, r, splat2(1.3981999507e-3f)); p = fma2(p, r, splat2(8.3334519073e-3f)); p = fma2(p, r, splat2(4.1665795894e-2f));
    p = fma2(p, r, splat2(1.6666665459e-1f)); p = fma2(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    p = fma2(p, r2, r);
    p = p + splat2(1.0f);
    f32x2 out;
    out.x = __builtin_bit_cast(float, (uint32_t)((int32_t)__builtin_bit_cast(uint32_t, p.x) + (int32_t)n.x * (1 << 23)));
    out.y = __builtin_bit_cast(float, (uint32_t)((int32_t)__builtin_bit_cast(uint32_t, p.y) + (int32_t)n.y * (1 << 23)));
    return out;
}
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
constexpr int LDS_BYTES = 80896;
__global__ void __launch_bounds__(256, 2) probe(uint32_t *out, uint32_t *hw, int a_mode, int b_mode, int iters)
{
    asm volatile("v_mov_b32 v255, 0" ::: "v255");               // the kernel owns all 256 registers: the second wave of a SIMD starts at 256

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    const uint32_t wg = blockIdx.x, tix = threadIdx.x, lane = tix & 63;
    for (int i = tix; i < LDS_BYTES / 4; i += 256) reinterpret_cast<uint32_t *>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    if (tix == 0) { hw[wg * 2] = __builtin_amdgcn_s_getreg(63492); hw[wg * 2 + 1] = __builtin_amdgcn_s_getreg(63508); }
    if (wg < 256) {
        const uint64_t t_end = wall_clock64() + 15000;            // 150 us
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        float a = 1.0f + lane * 1e-3f, b = 0.5f;
        f16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * lane); }
        while (wall_clock64() < t_end) {
            if (a_mode == 0) __builtin_amdgcn_s_sleep(8);
            else
                for (int r = 0; r < 64; ++r) {
                    if (a_mode == 1 || a_mode == 3) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                    }
                    if (a_mode == 2 || a_mode == 3) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
                    if (a_mode == 4) {
                        *reinterpret_cast<uint4 *>(smem + (((tix >> 6) * 9216 + (lane >> 3) * 144 + (lane & 7) * 16 + (r & 7) * 8 * 144)) ) = make_uint4(r, lane, r, lane);
                        const uint4 v = *reinterpret_cast<const uint4 *>(smem + ((tix >> 6) * 9216 + (lane & 31) * 144 + (r & 7) * 16));
                        b += __builtin_bit_cast(float, v.x) * 1e-30f;
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                    }
                    if (a_mode == 3) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(smem + ((lane * 144 + r * 16) % (LDS_BYTES - 16) & ~15));
                        b += __builtin_bit_cast(float, v.x) * 1e-30f;
                    }
                }
        }
        float s = 0.0f;
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        out[wg * 256 + tix] = __builtin_bit_cast(uint32_t, s);
    } else {
        uint32_t cs = 0;
        for (int it = 0; it < iters; ++it) {
            const float x0 = -((float)((lane * 37 + it * 11) % 1000)) * 0.02f, x1 = -((float)((lane * 53 + it * 7) % 1000)) * 0.03f;
            if (b_mode == 2) {
                const uint32_t v = lane * 2654435761u + it * 40503u, want = (lane ^ 32) * 2654435761u + it * 40503u;
                const uint32_t r = (uint32_t)__shfl_xor((int)v, 32, 64);
                const uint32_t v2 = (lane ^ 16) * 97u + it, r2 = (uint32_t)__shfl_xor((int)(lane * 97u + it), 16, 64);
                cs += (r != want) + (r2 != v2);
                continue;
            }
            if (b_mode == 3) {                                   // phase C shape: packed math, then the half-wave exchange, then packed math
                f32x2 e = expo2((f32x2){x0, x1});
                f32x2 a2 = splat2(0.0f);
                a2 = a2 + e; a2 = a2 + e * splat2(0.5f);
                f32x2 c2 = {__shfl_xor(a2.x, 32, 64), __shfl_xor(a2.y, 32, 64)};
                c2 = c2 + e;
                cs ^= __builtin_bit_cast(uint32_t, c2.x) * 3u ^ __builtin_bit_cast(uint32_t, c2.y) * 5u;
                cs = (cs << 1) | (cs >> 31);
                continue;
            }
            if (b_mode == 0) {
                f32x2 e = expo2((f32x2){x0, x1});
                const f32x2 pr = e * (f32x2){0.37f, 0.11f};           // v_pk_mul_f32 (phase C)
                f32x2 s2 = splat2(0.0f);
                s2 = s2 + pr; s2 = s2 + e;                            // v_pk_add_f32
                cs ^= __builtin_bit_cast(uint32_t, e.x) * 3u ^ __builtin_bit_cast(uint32_t, e.y) * 5u ^ __builtin_bit_cast(uint32_t, s2.x) * 7u ^
                      __builtin_bit_cast(uint32_t, s2.y) * 11u;
            } else {
                const float e0 = expo1(x0), e1 = expo1(x1);
                const float s0 = (0.0f + e0 * 0.37f) + e0, s1 = (0.0f + e1 * 0.11f) + e1;
                cs ^= __builtin_bit_cast(uint32_t, e0) * 3u ^ __builtin_bit_cast(uint32_t, e1) * 5u ^ __builtin_bit_cast(uint32_t, s0) * 7u ^
                      __builtin_bit_cast(uint32_t, s1) * 11u;
            }
            cs = (cs << 1) | (cs >> 31);
        }
        out[wg * 256 + tix] = cs;
    }
}
int main(int argc, char **argv)
{
    const int b_mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 20000, reps = argc > 3 ? atoi(argv[3]) : 10;
    uint32_t *d, *dh;
    static uint32_t ref[512 * 256], cur[512 * 256], hw[1024];
    CK(hipMalloc(&d, sizeof(ref))); CK(hipMalloc(&dh, sizeof(hw)));
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, dh, 0, b_mode, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref, d, sizeof(ref), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hw, dh, sizeof(hw), hipMemcpyDeviceToHost));
    int shared = 0;
    for (int b = 256; b < 512; ++b)
        if (((hw[(b - 256) * 2] >> 8) & 0xff) == ((hw[b * 2] >> 8) & 0xff) && ((hw[(b - 256) * 2] >> 13) & 7) == ((hw[b * 2] >> 13) & 7) &&
            (hw[(b - 256) * 2 + 1] & 15) == (hw[b * 2 + 1] & 15)) ++shared;
    printf("workgroup i + 256 shares its compute unit with workgroup i: %d of 256\n", shared);
    // lanes of one wave see the same inputs in every workgroup: all B workgroups must agree with each other, too
    for (int a_mode = 0; a_mode <= 4; ++a_mode) {
        long bad = 0;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, 0, d, dh, a_mode, b_mode, iters);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(cur, d, sizeof(cur), hipMemcpyDeviceToHost));
            for (int i = 256 * 256; i < 512 * 256; ++i) if (cur[i] != ref[i]) { if (bad < 5) printf("   a_mode %d rep %d: wg %d thread %d %08x != %08x\n", a_mode, r, i / 256, i % 256, cur[i], ref[i]); ++bad; }
        }
        printf("b_mode %d, a_mode %d: %ld checksums differ in %d launches\n", b_mode, a_mode, bad, reps);
    }
    return 0;
}
