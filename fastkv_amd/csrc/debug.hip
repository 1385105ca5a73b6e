// Test hook: evaluates the primitives of the arithmetic contract (fk_device.h) element-wise on the GPU so that
// tests can compare them bit-for-bit with their twins in oracle/fastkv_oracle.c.  Not used by the product path.
#include "fk_device.h"
#include "fk_host.h"

namespace fk {
__global__ void debug_contract_kernel(int op, const float *a, const float *b, float *out, uint64_t *out64, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    uint64_t r64 = 0;
    switch (op) {
    case 0: r = det_expf(x); break;
    case 1: r = x / y; break;
    case 2: r = h2f(f2h(x)); break;
    case 3: { uint32_t hi, lo; exp_to_fix(x, hi, lo); r64 = ((uint64_t)hi << 24) + lo; r = fix_to_f32(r64); break; }
    case 4: r = __builtin_fmaf(x, y, out[i]); break;
    case 5: r = fix_to_f32(((uint64_t)f32_bits(x) << 32) | f32_bits(y)); break;
    case 6: r = x * y; break;
    case 7: r = x + y; break;
    case 8: r = scale_div(x, y, 1.0f / y); break;
    // packed twins (v_pk_*_f32): element i is paired with its neighbour i^1 so that both components are exercised
    case 9: { const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i]; const f32x2 e = det_expf2((i & 1) ? (f32x2){x2, x} : (f32x2){x, x2}); r = (i & 1) ? e.y : e.x; break; }
    case 10: {
        const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i];
        uint32_t h0, l0, h1, l1;
        exp_to_fix2((i & 1) ? (f32x2){x2, x} : (f32x2){x, x2}, h0, l0, h1, l1);
        r64 = (i & 1) ? ((uint64_t)h1 << 24) + l1 : ((uint64_t)h0 << 24) + l0;
        r = fix_to_f32(r64);
        break;
    }
    case 14: {                                                   // the four-operation twin of the fused kernel's phase B (20 / 20 split, signed low part)
        const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i];
        uint32_t h0, l0, h1, l1;
        exp_to_fix2_magic((i & 1) ? (f32x2){x2, x} : (f32x2){x, x2}, h0, l0, h1, l1);
        const uint32_t h = ((i & 1) ? h1 : h0) - FIX_MAGIC_BITS, l = ((i & 1) ? l1 : l0) - FIX_MAGIC_BITS;
        r64 = ((uint64_t)h << 20) + (uint64_t)(int64_t)(int32_t)l;
        r = fix_to_f32(r64);
        break;
    }
    case 12: { const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i]; const uint32_t w = (i & 1) ? f2h2(x2, x) : f2h2(x, x2); r = h2f((uint16_t)((i & 1) ? w >> 16 : w & 0xffffu)); break; }
    case 13: {                                                   // the two-operation quotient of the fused kernel's epilogue (x: an fp16 value)
        const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i];
        const uint32_t w = (i & 1) ? f2h2(x2, x) : f2h2(x, x2);
        const f32x2 q = scale_div2_finite_h2(w, 1.0f / y, recip_lo(y, 1.0f / y));
        r = (i & 1) ? q.y : q.x;
        break;
    }
    case 11: { const float x2 = a[(i ^ 1) < n ? (i ^ 1) : i]; const f32x2 e = scale_div2((i & 1) ? (f32x2){x2, x} : (f32x2){x, x2}, y, 1.0f / y); r = (i & 1) ? e.y : e.x; break; }
    }
    out[i] = r;
    if (out64) out64[i] = r64;
}
// Test hook of the "mfma16" contract: the raw v_mfma_f32_32x32x16_f16 on caller-supplied tiles, so that tests can hold the oracle's
// restatement of the instruction (oracle/fastkv_oracle.c mfma16_block) against the chip they run on.  One wave per tile: A [32][dd],
// Bt [32][dd] fp16 row-major (lane l holds d = 16 c + 8 (l / 32) + j of chunk c, as the scoring kernels do), C [32][32] fp32 or null
// (+0), out [32][32]: dd / 16 chained instructions.
typedef _Float16 dbg_f16x8 __attribute__((ext_vector_type(8)));
typedef float dbg_f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(64) debug_mfma16_kernel(const _Float16 *A, const _Float16 *Bt, const float *C, float *out, int dd)
{
    const size_t t = blockIdx.x;
    A += t * 32 * dd; Bt += t * 32 * dd; out += t * 1024;
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    dbg_f32x16 acc;
    for (int i = 0; i < 16; i++) { const int m = (i & 3) + 8 * (i >> 2) + 4 * h; acc[i] = C ? C[t * 1024 + m * 32 + r] : 0.0f; }
    for (int c = 0; c < dd / 16; c++) {
        dbg_f16x8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = A[r * dd + 16 * c + 8 * h + j]; b[j] = Bt[r * dd + 16 * c + 8 * h + j]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; i++) { const int m = (i & 3) + 8 * (i >> 2) + 4 * h; out[m * 32 + r] = acc[i]; }
}
// Test hook: `wgs` workgroups that each hold `lds_bytes` of LDS and spin for `usec` microseconds of wall clock -- "another
// kernel is holding compute units" for the residency tests.
__global__ void __launch_bounds__(256) debug_occupy_kernel(uint64_t ticks, uint32_t *sink)
{
    extern __shared__ uint32_t hold[];
    hold[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (hold[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;
}
}  // namespace fk

extern "C" int fastkv_debug_occupy(int wgs, int lds_bytes, int64_t usec, void *stream)
{
    if (wgs < 1 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || usec < 0) return FASTKV_EINVAL;
    static uint32_t *sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return FASTKV_ELAUNCH;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fk::debug_occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(fk::debug_occupy_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (uint64_t)usec * 100u, sink);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

extern "C" int fastkv_debug_mfma16(const void *a, const void *bt, const float *c, float *out, int ntiles, int dd, void *stream)
{
    if (!a || !bt || !out || ntiles < 0 || dd < 16 || (dd & 15) || dd > 1024) return FASTKV_EINVAL;
    if (!ntiles) return FASTKV_OK;
    hipLaunchKernelGGL(fk::debug_mfma16_kernel, dim3(ntiles), dim3(64), 0, (hipStream_t)stream, (const _Float16 *)a, (const _Float16 *)bt, c, out, dd);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

extern "C" int fastkv_debug_contract(int op, const float *a, const float *b, float *out, uint64_t *out64, int n, void *stream)
{
    if (!a || !out || n < 0 || op < 0 || op > 14) return FASTKV_EINVAL;
    hipLaunchKernelGGL(fk::debug_contract_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, a, b, out, out64, n);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

// Test hook (host only): where a problem's token-tagged hand-off areas lie in its workspace -- {fused score records, head-sum chain,
// split-selection tables} -- and the workspace size.  tests/test_host_logic.py: the offsets do not depend on the problem's shape.
extern "C" int fastkv_debug_granule_areas(const fastkv_problem *p, size_t out[4])
{
    if (!p || !out) return FASTKV_EINVAL;
    const fk::Layout L = fk::make_layout(*p);
    out[0] = L.off_fpart; out[1] = L.off_fchain; out[2] = L.off_seltab; out[3] = L.total;
    return FASTKV_OK;
}

