// Round 4 probe: variants of the K/V gather + compact kernel at the ROOFLINE SHAPE of bench.py (32 layers stacked: 256 (batch x KV head)
// rows, S = 32768, D = 128, cap = 2048: 541 MB of algorithmic traffic, three rotated source sets of 4 GiB each so that nothing is served
// by the 256 MiB Infinity Cache).  Index lists are sorted ascending per head (what select_split hands over).
// hipcc --offload-arch=gfx950 -O3 -o compact_probe2 compact_probe2.hip ; ./compact_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

struct Args { const uint16_t *k, *v; int64_t ss, sh, sb; const int64_t *idx; const uint16_t *slot; int Hkv, S, W, cap; uint16_t *ko, *vo; };

__device__ __forceinline__ uint4 ldg(const uint16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldg_nt(const uint16_t *p) { const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); return make_uint4(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ void stg(uint16_t *p, uint4 x) { *reinterpret_cast<uint4 *>(p) = x; }
__device__ __forceinline__ void stg_nt(uint16_t *p, uint4 x) { u32x4 y = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(y, reinterpret_cast<u32x4 *>(p)); }

// U rows per 16-lane group; K and V of a row by the same lane (the product kernel is U = 1).  MODE bit 0: nontemporal loads, bit 1:
// nontemporal stores, bit 2: ranked destination (score order: slot list), bit 3: no stores (read side only), bit 4: no loads (write only)
template <int U, int MODE>
__global__ void __launch_bounds__(256) k_rows(Args a)
{
    const int bg = blockIdx.y, b = bg / a.Hkv, g = bg % a.Hkv;
    const uint16_t *ks = a.k + b * a.sb + g * a.sh, *vs = a.v + b * a.sb + g * a.sh;
    uint16_t *kd = a.ko + (size_t)bg * a.cap * 128, *vd = a.vo + (size_t)bg * a.cap * 128;
    const int kk = a.cap - a.W, n = a.S - a.W, sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    int64_t srow[U];
    int d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int r = (blockIdx.x * U + u) * 16 + rl, rc = r < a.cap ? r : a.cap - 1;
        srow[u] = rc < kk ? a.idx[(size_t)bg * kk + rc] : (int64_t)(n + rc - kk);
        d[u] = rc;
        if (MODE & 4) d[u] = rc < kk ? a.slot[(size_t)bg * kk + rc] : rc;
    }
    uint4 kv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (MODE & 16) { kv[u] = make_uint4(srow[u], 1, 2, 3); vv[u] = kv[u]; }
        else if (MODE & 1) { kv[u] = ldg_nt(ks + srow[u] * a.ss + sub * 8); vv[u] = ldg_nt(vs + srow[u] * a.ss + sub * 8); }
        else { kv[u] = ldg(ks + srow[u] * a.ss + sub * 8); vv[u] = ldg(vs + srow[u] * a.ss + sub * 8); }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int r = (blockIdx.x * U + u) * 16 + rl;
        if (MODE & 8) { if ((kv[u].x ^ vv[u].y) == 0x12345u && r < a.cap) stg(kd, kv[u]); continue; }
        if (r < a.cap) {
            if (MODE & 2) { stg_nt(kd + (size_t)d[u] * 128 + sub * 8, kv[u]); stg_nt(vd + (size_t)d[u] * 128 + sub * 8, vv[u]); }
            else { stg(kd + (size_t)d[u] * 128 + sub * 8, kv[u]); stg(vd + (size_t)d[u] * 128 + sub * 8, vv[u]); }
        }
    }
}

// K and V by different workgroups (blockIdx.z): one stream per workgroup, twice the workgroups
template <int U>
__global__ void __launch_bounds__(256) k_split(Args a)
{
    const int bg = blockIdx.y, b = bg / a.Hkv, g = bg % a.Hkv;
    const uint16_t *s = (blockIdx.z ? a.v : a.k) + b * a.sb + g * a.sh;
    uint16_t *dd = (blockIdx.z ? a.vo : a.ko) + (size_t)bg * a.cap * 128;
    const int kk = a.cap - a.W, n = a.S - a.W, sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    int64_t srow[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int r = (blockIdx.x * U + u) * 16 + rl, rc = r < a.cap ? r : a.cap - 1;
        srow[u] = rc < kk ? a.idx[(size_t)bg * kk + rc] : (int64_t)(n + rc - kk);
    }
    uint4 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = ldg(s + srow[u] * a.ss + sub * 8);
#pragma unroll
    for (int u = 0; u < U; u++) { const int r = (blockIdx.x * U + u) * 16 + rl; if (r < a.cap) stg(dd + (size_t)r * 128 + sub * 8, x[u]); }
}

// persistent: gridDim.x workgroups walk the (head, 16-row chunk) list; the next chunk's indices and rows are requested before this
// chunk's rows are stored (two chunks in flight per workgroup)
template <int U>
__global__ void __launch_bounds__(256) k_persist(Args a, int heads)
{
    const int kk = a.cap - a.W, n = a.S - a.W, sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int cpr = (a.cap + 16 * U - 1) / (16 * U), total = heads * cpr;
    int64_t srow[U];
    uint4 kv[U], vv[U];
    int c = blockIdx.x;
    auto fetch_idx = [&](int cc) {
        const int bg = cc / cpr, x = cc % cpr;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = (x * U + u) * 16 + rl, rc = r < a.cap ? r : a.cap - 1;
            srow[u] = rc < kk ? a.idx[(size_t)bg * kk + rc] : (int64_t)(n + rc - kk);
        }
    };
    auto fetch_rows = [&](int cc) {
        const int bg = cc / cpr, b = bg / a.Hkv, g = bg % a.Hkv;
        const uint16_t *ks = a.k + b * a.sb + g * a.sh, *vs = a.v + b * a.sb + g * a.sh;
#pragma unroll
        for (int u = 0; u < U; u++) { kv[u] = ldg(ks + srow[u] * a.ss + sub * 8); vv[u] = ldg(vs + srow[u] * a.ss + sub * 8); }
    };
    if (c >= total) return;
    fetch_idx(c);
    fetch_rows(c);
    for (; c < total; c += gridDim.x) {
        const int nx = c + gridDim.x;
        uint4 k0[U], v0[U];
#pragma unroll
        for (int u = 0; u < U; u++) { k0[u] = kv[u]; v0[u] = vv[u]; }
        if (nx < total) { fetch_idx(nx); fetch_rows(nx); }
        const int bg = c / cpr, x = c % cpr;
        uint16_t *kd = a.ko + (size_t)bg * a.cap * 128, *vd = a.vo + (size_t)bg * a.cap * 128;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = (x * U + u) * 16 + rl;
            if (r < a.cap) { stg(kd + (size_t)r * 128 + sub * 8, k0[u]); stg(vd + (size_t)r * 128 + sub * 8, v0[u]); }
        }
    }
}

__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}

int main(int argc, char **argv)
{
    const int S = 32768, B = argc > 1 ? atoi(argv[1]) : 32, cap = 2048, W = 8, Hkv = 8, D = 128, kk = cap - W, n = S - W, NL = 3, HB = Hkv * B;
    std::vector<uint16_t *> K(NL), V(NL), KO(NL), VO(NL);
    const size_t kb = (size_t)B * S * Hkv * D * 2, ob = (size_t)HB * cap * D * 2;
    for (int i = 0; i < NL; i++) {
        CK(hipMalloc(&K[i], kb)); CK(hipMalloc(&V[i], kb)); CK(hipMalloc(&KO[i], ob)); CK(hipMalloc(&VO[i], ob));
        CK(hipMemset(K[i], 1 + i, kb)); CK(hipMemset(V[i], 5 + i, kb));
    }
    std::mt19937_64 rng(1);
    std::vector<int64_t> idxs((size_t)HB * kk);
    std::vector<uint16_t> slot((size_t)HB * kk);
    for (int g = 0; g < HB; g++) {
        // winners as the selection leaves them: a third of them in runs of 7 neighbours (maxpool plateaus), the rest scattered
        std::vector<char> take(n, 0);
        int cnt = 0;
        while (cnt < kk) {
            const int p = rng() % n, run = (rng() % 3 == 0) ? 7 : 1;
            for (int u = 0; u < run && cnt < kk; u++) if (p + u < n && !take[p + u]) { take[p + u] = 1; cnt++; }
        }
        int o = 0;
        for (int p = 0; p < n; p++) if (take[p]) idxs[(size_t)g * kk + o++] = p;
        std::vector<uint16_t> perm(kk);
        for (int i = 0; i < kk; i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < kk; i++) slot[(size_t)g * kk + i] = perm[i];
    }
    int64_t *ds; uint16_t *dslot;
    CK(hipMalloc(&ds, idxs.size() * 8)); CK(hipMalloc(&dslot, slot.size() * 2));
    CK(hipMemcpy(ds, idxs.data(), idxs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dslot, slot.data(), slot.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto args = [&](int i) { Args a = {K[i], V[i], (int64_t)Hkv * D, (int64_t)D, (int64_t)S * Hkv * D, ds, dslot, Hkv, S, W, cap, KO[i], VO[i]}; return a; };
    const double bytes = 2.0 * 2 * HB * cap * D * 2 + (double)HB * kk * 8;
    auto run = [&](const char *name, auto launch, double by) {
        for (int i = 0; i < 6; i++) launch(i % NL);
        CK(hipDeviceSynchronize());
        std::vector<float> t;
        for (int rep = 0; rep < 5; rep++) {
            const int N = 9;
            CK(hipEventRecord(e0));
            for (int i = 0; i < N; i++) launch(i % NL);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f / N);
        }
        std::sort(t.begin(), t.end());
        printf("%-52s %8.2f us (min %7.2f max %7.2f)  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, t[2], t[0], t[4], by / t[2] / 1e3, by / t[2] / 1e3 / 80.0);
        fflush(stdout);
    };
    for (int pass = 0; pass < 2; pass++) {
        printf("---- pass %d (B = %d: %d heads, %.0f MB)\n", pass, B, HB, bytes / 1e6);
        run("U1 (product shape) grid (128, heads)", [&](int i) { hipLaunchKernelGGL((k_rows<1, 0>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U2", [&](int i) { hipLaunchKernelGGL((k_rows<2, 0>), dim3(cap / 32, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U4", [&](int i) { hipLaunchKernelGGL((k_rows<4, 0>), dim3(cap / 64, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U1 nt loads", [&](int i) { hipLaunchKernelGGL((k_rows<1, 1>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U1 nt stores", [&](int i) { hipLaunchKernelGGL((k_rows<1, 2>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U2 nt loads + stores", [&](int i) { hipLaunchKernelGGL((k_rows<2, 3>), dim3(cap / 32, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U1 ranked destination (score order)", [&](int i) { hipLaunchKernelGGL((k_rows<1, 4>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U2 ranked destination (score order)", [&](int i) { hipLaunchKernelGGL((k_rows<2, 4>), dim3(cap / 32, HB), dim3(256), 0, 0, args(i)); }, bytes);
        run("U1 read side only", [&](int i) { hipLaunchKernelGGL((k_rows<1, 8>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes / 2);
        run("U1 write side only", [&](int i) { hipLaunchKernelGGL((k_rows<1, 16>), dim3(cap / 16, HB), dim3(256), 0, 0, args(i)); }, bytes / 2);
        run("split K / V workgroups U1", [&](int i) { hipLaunchKernelGGL((k_split<1>), dim3(cap / 16, HB, 2), dim3(256), 0, 0, args(i)); }, bytes);
        run("split K / V workgroups U2", [&](int i) { hipLaunchKernelGGL((k_split<2>), dim3(cap / 32, HB, 2), dim3(256), 0, 0, args(i)); }, bytes);
        run("split K / V workgroups U4", [&](int i) { hipLaunchKernelGGL((k_split<4>), dim3(cap / 64, HB, 2), dim3(256), 0, 0, args(i)); }, bytes);
        run("persistent 2048 WGs U1", [&](int i) { hipLaunchKernelGGL((k_persist<1>), dim3(2048), dim3(256), 0, 0, args(i), HB); }, bytes);
        run("persistent 2048 WGs U2", [&](int i) { hipLaunchKernelGGL((k_persist<2>), dim3(2048), dim3(256), 0, 0, args(i), HB); }, bytes);
        run("persistent 1024 WGs U2", [&](int i) { hipLaunchKernelGGL((k_persist<2>), dim3(1024), dim3(256), 0, 0, args(i), HB); }, bytes);
        run("persistent 4096 WGs U1", [&](int i) { hipLaunchKernelGGL((k_persist<1>), dim3(4096), dim3(256), 0, 0, args(i), HB); }, bytes);
        run("plain copy of the output bytes (contiguous)", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4 *)K[i], (uint4 *)KO[i], ob / 16);
                                                                       hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4 *)V[i], (uint4 *)VO[i], ob / 16); }, bytes);
    }
    return 0;
}
