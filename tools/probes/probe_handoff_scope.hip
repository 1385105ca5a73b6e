// Which loads see another compute unit's store, and what does one poll cost?  (gfx950, MI355X)
// 16 workgroups: consumer b < 8 and its producer b + 8 share an XCD under round-robin placement (checked: XCC_ID is printed);
// with PAIR=1 the producer of consumer b is b + 9 (the next XCD).  The consumer loads the word first (it is in its L1 and L2 then), tells the
// producer to go (agent-scope flag), the producer waits 3 us and stores a new value; the consumer polls with one of the load variants and
// records: polls until fresh (or "never" after 200 us) and the average time per poll.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_handoff_scope probe_handoff_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t ld(const uint32_t *p, int variant, uint32_t zero)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p), 0, 0x7fffffff, 0x00020000);
    switch (variant) {
    case 0: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);             // plain
    case 1: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 1);             // sc0
    case 2: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 2);             // nt
    case 3: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 3);             // sc0 nt
    case 4: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 16);            // sc1
    case 5: return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 17);            // sc0 sc1
    case 6: return __hip_atomic_fetch_add(const_cast<uint32_t *>(p), zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // RMW, returning
    case 7: asm volatile("buffer_inv sc0" ::: "memory"); return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
    case 8: asm volatile("buffer_inv sc1" ::: "memory"); return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
    default: asm volatile("buffer_inv sc0 sc1" ::: "memory"); return __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
    }
}

// out[b] = {polls until fresh (0 = never), ticks per poll x 16, xcc consumer, xcc producer}
__global__ void probe(uint32_t *words, uint32_t *go, uint32_t *out, int load_variant, int store_variant, int pair, uint32_t gen, uint32_t *xcc, uint32_t zero)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg(63508);
    if (b < 8) {
        uint32_t *wd = words + b * 1024;                         // 4 KB apart: own lines
        if (threadIdx.x == 0) {
            uint32_t seen = ld(wd, 0, zero);                     // the line is in this CU's L1 and this XCD's L2 now
            seen += ld(wd, load_variant, zero);
            __hip_atomic_store(go + b * 64, gen + (seen & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t t0 = wall_clock64();
            uint32_t polls = 0, fresh = 0;
            while (wall_clock64() - t0 < 20000) {                // 200 us
                const uint32_t v = ld(wd, load_variant, zero);
                ++polls;
                if (v == gen) { fresh = polls; break; }
            }
            const uint64_t t1 = wall_clock64();
            out[b * 4] = fresh;
            out[b * 4 + 1] = (uint32_t)((t1 - t0) * 16 / (polls ? polls : 1));
            out[b * 4 + 2] = (uint32_t)(t1 - t0);
        }
    } else {
        const int cb = pair ? (b - 8 + 7) % 8 : b - 8;          // pair=0: same XCD as the consumer; pair=1: consumer on the previous XCD
        uint32_t *wd = words + cb * 1024;
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(go + cb * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(2);
            const uint64_t t0 = wall_clock64();
            while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(2);    // 3 us
            if (store_variant == 0) *reinterpret_cast<volatile uint32_t *>(wd) = gen;
            else if (store_variant == 1) __hip_atomic_store(wd, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (store_variant == 2) __hip_atomic_store(wd, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(wd, gen - (gen - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the word holds gen - 1
        }
    }
}

int main()
{
    uint32_t *words, *go, *out, *xcc;
    CK(hipMalloc(&words, 8 * 4096)); CK(hipMalloc(&go, 8 * 256)); CK(hipMalloc(&out, 8 * 16)); CK(hipMalloc(&xcc, 64));
    CK(hipMemset(words, 0, 8 * 4096)); CK(hipMemset(go, 0, 8 * 256));
    const char *ln[] = {"plain", "sc0", "nt", "sc0 nt", "sc1", "sc0 sc1", "atomic add 0 (returning)", "buffer_inv sc0 + plain", "buffer_inv sc1 + plain", "buffer_inv sc0 sc1 + plain"};
    const char *sn[] = {"plain store", "sc0 store", "sc1 store", "atomic add"};
    uint32_t gen = 0;
    for (int pair = 0; pair < 2; ++pair) {
        printf("==== producer on %s\n", pair ? "the NEXT XCD" : "the SAME XCD");
        for (int sv = 0; sv < 4; ++sv) {
            for (int lv = 0; lv < 10; ++lv) {
                int fresh_n = 0; double polls_fresh = 0, us_poll = 0, us_total = 0; int mism = 0;
                const int reps = 6;
                for (int rep = 0; rep < reps; ++rep) {
                    ++gen;
                    std::vector<uint32_t> init(8 * 1024, gen - 1);
                    CK(hipMemcpy(words, init.data(), 8 * 4096, hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(probe, dim3(16), dim3(64), 0, 0, words, go, out, lv, sv, pair, gen, xcc, 0u);
                    CK(hipDeviceSynchronize());
                    uint32_t h[32], hx[16];
                    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
                    for (int b = 0; b < 8; ++b) {
                        const int pb = pair ? 8 + (b + 1) % 8 : b + 8;
                        if ((hx[b] == hx[pb]) != (pair == 0)) ++mism;
                        if (h[b * 4]) { ++fresh_n; polls_fresh += h[b * 4]; us_total += h[b * 4 + 2] / 100.0; }
                        us_poll += h[b * 4 + 1] / 16.0 / 100.0;
                    }
                }
                printf("%-12s | %-28s | fresh %2d/%2d", sn[sv], ln[lv], fresh_n, 8 * reps);
                if (fresh_n) printf(" after %6.1f polls, %6.2f us", polls_fresh / fresh_n, us_total / fresh_n);
                else printf("  (never in 200 us)          ");
                printf(" | %5.2f us per poll%s\n", us_poll / (8 * reps), mism ? "  [placement not as assumed]" : "");
            }
        }
    }
    return 0;
}
