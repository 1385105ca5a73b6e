// The attention part of a one-token decode step in ONE launch (/root/reference/baselines/fastkv/llama_model.py:143-145
// appends the step's K/V row with `past_key_value.update` and attends with flash-attn, /root/reference/benchmark/e2e.py:72-93 times it).
// RoPE of q and of the new K row, append of the K/V row to the layer's slab, GQA attention over the slab, merge of the slices,
// advance of the device-side length.  Built for LATENCY: at budget 2048 a layer's cache is 9.4 MB -- 1.5 us of HBM time -- so what
// the launch costs is its chain of dependent steps, and every one of them is spread as wide as it goes:
//   * a wave owns a 32-row tile of the slab (2304 rows x 8 KV heads = 576 waves on 144 workgroups) and requests its K and V rows
//     (16 loads of 16 B per lane) BEFORE it knows the cache length: the slices partition the slab's capacity;
//   * q.K^T on the matrix pipe: K rows are the A operand of v_mfma_f32_16x16x32_f16 straight from the global loads (lane = row
//     l & 15, dims 8 (l >> 4) ... of a 32-dim chunk), the G query heads are columns 0 .. G-1 of the B operand;
//   * the probabilities never move: lane (g, r) of the accumulator layout holds p[g][rows 4r .. 4r+3 of a 16-row block], the 16
//     lanes of row group r load exactly those V rows (16 B per lane across head_dim) and take p from lane g of their own group
//     with a DPP row broadcast -- no LDS round trip, no shuffle between the scores and P.V;
//   * no maximum is shared inside a wave: each of its four row groups keeps its own (max, sum, o[D]); the 16 partial sets of a
//     workgroup (+ the step's own row: its K from LDS, not from the slab it has just been written to) are merged through LDS;
//   * slice records are 8-byte {token, fp32} granules written with one write-through store each (the data is the flag: no drain,
//     no arrival counter); one extra workgroup per KV head -- the LAST of the head in dispatch order, so everything it waits for
//     has been dispatched -- polls them, merges and writes the fp16 output;
//   * the length is advanced by the workgroup that is the last to have READ it: every workgroup reports with one atomic as soon as
//     its read has returned, and the answer is looked at when the workgroup ends (off the critical path).
// fp32 softmax; compared with PyTorch SDPA to fp16 tolerance (tests/test_decode_gpu.py), not bit for bit.
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"
#include <cstdlib>

namespace fk {

constexpr int ST_THREADS = 256;
// slab rows per wave tile: KB 16-row blocks of the matrix instruction (template parameter: 2; 1 is a measurement variant, see step_tile_rows)
constexpr int ST_PASS = 512;                                     // (query head, head_dim) values merged per pass through LDS

// Measurement build (-DFK_STAMP): per-wave wall-clock stamps (100 MHz) of the step kernel's stages, read by tools/stamp_decode_step.py
#ifdef FK_STAMP
__device__ unsigned long long g_dstamps[2048 * 16];
#define FKD_STAMP(slot) do { if (lane == 0) g_dstamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 2048 * 16 + (slot)] = wall_clock64(); } while (0)
#define FKD_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define FKD_STAMP(slot) do { } while (0)
#define FKD_DRAIN() do { } while (0)
#endif

typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t st_u32x2 __attribute__((ext_vector_type(2)));
constexpr int ST_BATCH = 24;                                     // slices whose records a merger thread has in flight at once

struct StepArgs {
    const uint16_t *k_new; int64_t kn_b, kn_h;                   // [B,Hkv,1,D] raw k_proj output of the step
    const uint16_t *v_new; int64_t vn_b, vn_h;
    const uint16_t *cosv, *sinv; int64_t cs_b;                   // [B,1,D]
    uint32_t *counters;                                          // [0] arrivals (zero between launches), [1] launch epoch of the records
    uint16_t *out; int H, Hkv;                                   // [B,1,H*D]
    uint32_t *host_flag;                                         // pinned status words of the process (capi.hip): [0] abandoned wait, [1] slab overrun
};

// lane N of every 16-lane row to all lanes of the row (row_newbcast)
template <int N> __device__ __forceinline__ float row_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}
template <int G, int DL, int g = 0> __device__ __forceinline__ void pv_row(float (&o)[G][DL], float p, const float (&vf)[DL])
{
    if constexpr (g < G) {
        const float pb = row_bcast<g>(p);
#pragma unroll
        for (int e = 0; e < DL; ++e) o[g][e] = __builtin_fmaf(pb, vf[e], o[g][e]);
        pv_row<G, DL, g + 1>(o, p, vf);
    }
}
template <int G, int DL, int g = 0> __device__ __forceinline__ void scale_rows(float (&o)[G][DL], float corr)
{
    if constexpr (g < G) {
        const float cb = row_bcast<g>(corr);
#pragma unroll
        for (int e = 0; e < DL; ++e) o[g][e] *= cb;
        scale_rows<G, DL, g + 1>(o, corr);
    }
}

__device__ __forceinline__ uint64_t step_granule(uint32_t token, float v) { return ((uint64_t)token << 32) | f32_bits(v); }
// One 16-B load of two adjacent granules; SC = 16: agent coherence (sc1).  Measured on this chip (tools/probes/probe_handoff_scope.hip): plain and
// sc0 loads are served by the compute unit's L1 and NEVER see another unit's store to a line the unit has read before; nt and sc1
// loads do, same XCD or not, at 0.2 us per dependent load.
template <int SC> __device__ __forceinline__ st_u32x4 load_granules(__amdgpu_buffer_rsrc_t rs, int byte_off)
{
    // (for the compiler this is an ordinary read: a poll loop needs a memory clobber per iteration, or the read is hoisted out of it)
    return __builtin_bit_cast(st_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, SC));
}

// grid (B*Hkv, nsplit + (nsplit > 1)), 256 threads; chunk = slab rows per slice (a multiple of the workgroup tile).  The KV head is the
// FAST grid index: workgroups go to the eight XCDs round-robin in dispatch order, so with a multiple of eight heads all workgroups
// of a head run on ONE XCD (speed only: a store is seen ~0.5 us sooner there than on another XCD).
template <int D, int G, int KB>
__global__ void __launch_bounds__(ST_THREADS) decode_step_kernel(const uint16_t *__restrict__ q, int64_t q_b, int64_t q_h,
                                                                uint16_t *__restrict__ kslab, uint16_t *__restrict__ vslab, int64_t s_b,
                                                                int64_t s_h, int64_t s_r, int rows, int32_t *__restrict__ len_dev,
                                                                float scaling, uint64_t *__restrict__ rec, int nsplit, int chunk,
                                                                StepArgs sa, uint64_t limit_ticks)
{
    constexpr int ST_TILE = 16 * KB, ST_WG_ROWS = 4 * ST_TILE;   // slab rows per wave tile / per workgroup tile
    constexpr int DL = D / 16;                                   // head-dim elements per lane in P.V
    constexpr int NC = D / 32;                                   // 32-dim chunks of the contraction
    constexpr int GP = (ST_PASS / D) < G ? (ST_PASS / D) : G;    // query heads per merge pass
    constexpr int NPASS = G / GP;
    constexpr int PD = GP * D;                                   // values per pass (<= 512), two per thread
    constexpr int DB = D / 16;                                   // 8-element blocks in half a head row
    constexpr int NQ = G * DB, NR = NQ + DB;                     // RoPE work items: the G query rows + the new K row
    constexpr int NG = D >= 128 ? 1 : 128 / D;                   // query heads a wave of the merger works on (its 128 values of a pass)
    __shared__ __attribute__((aligned(16))) uint16_t s_qh[G * D];
    __shared__ __attribute__((aligned(16))) uint16_t s_knew[D], s_vnew[D];
    __shared__ float s_m[5 * G], s_l[4 * G];                     // the four waves' sets (+ the step's own row: its score)
    __shared__ __attribute__((aligned(16))) float s_o[4 * G * D];
    const int hb = blockIdx.x, c = blockIdx.y, b = hb / sa.Hkv, hk = hb - b * sa.Hkv;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, x = lane & 15, qd = lane >> 4;
    const bool merger = nsplit > 1 && c == nsplit;
    const uint16_t *kb = kslab + b * s_b + hk * s_h, *vb = vslab + b * s_b + hk * s_h;
    const int lo = c * chunk;
    FKD_STAMP(0);
#ifdef FK_STAMP
    if (lane == 0) { g_dstamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 2048 * 16 + 14] = 0; g_dstamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 2048 * 16 + 15] = 0; }
    if (lane == 0) g_dstamps[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 2048 * 16 + 13] = __builtin_amdgcn_s_getreg(63508) | ((uint64_t)__builtin_amdgcn_s_getreg(63492) << 32);
#endif

    // ---- the length is advanced by the workgroup that is the LAST to report that it has read it (and the records' epoch with it)
    auto report_length_read = [&](int len_old, uint32_t epoch) {
        if (tid == 0) {
            uint32_t inc = 1;
            asm volatile("" : "+v"(inc) : "v"(len_old), "v"(epoch));  // (the increment depends on both loads: the atomic cannot overtake them)
            const uint32_t arrive = __hip_atomic_fetch_add(sa.counters, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrive == gridDim.x * gridDim.y - 1) {
                __hip_atomic_store(sa.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sa.counters[1] = epoch + 1u;
                *len_dev = len_old + 1 < rows ? len_old + 1 : rows;
                // a step into a FULL slab has overwritten the last cached row and the length stays where it is: wrong tokens from
                // here on.  Reported, not silent: fastkv_last_status() / the next operator call return FASTKV_EOVERFLOW.
                if (len_old >= rows && sa.host_flag) __hip_atomic_store(sa.host_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    uint64_t *rec_head = rec + (size_t)hb * nsplit * G * (D + 2);
    if (!merger) {
    // ---- everything the step needs from memory is requested NOW, in one round trip and in straight-line code (the compiler's wait
    //      counts are exact only then: the rotation below waits for ITS loads, not for the tile): the raw q rows / new K row with
    //      their rotary factors (16-B pieces: thread = 8 dims of the lower half of a row and the 8 dims they rotate with; threads
    //      beyond the work items repeat the last one), the new V row, the length, the records' epoch, and the wave's first K / V
    //      tile (rows at or beyond the length, or the slab, are clamped / in bounds, loaded and never used)
    const bool rope_item = tid < NR, v_item = tid >= NR && tid < NR + D / 8;
    const int rt = tid < NR ? tid : NR - 1;
    const int ritem = rt < NQ ? rt / DB : G, d0 = (rt < NQ ? rt - ritem * DB : rt - NQ) * 8;
    const uint16_t *xr = rt < NQ ? q + b * q_b + (int64_t)(hk * G + ritem) * q_h : sa.k_new + b * sa.kn_b + (int64_t)hk * sa.kn_h;
    const uint16_t *cs = sa.cosv + b * sa.cs_b, *sn = sa.sinv + b * sa.cs_b;
    const uint4 x1 = *reinterpret_cast<const uint4 *>(xr + d0), x2 = *reinterpret_cast<const uint4 *>(xr + d0 + D / 2);
    const uint4 c1 = *reinterpret_cast<const uint4 *>(cs + d0), c2 = *reinterpret_cast<const uint4 *>(cs + d0 + D / 2);
    const uint4 s1 = *reinterpret_cast<const uint4 *>(sn + d0), s2 = *reinterpret_cast<const uint4 *>(sn + d0 + D / 2);
    const uint4 vn = *reinterpret_cast<const uint4 *>(sa.v_new + b * sa.vn_b + (int64_t)hk * sa.vn_h + ((tid >= NR ? tid - NR : tid) % (D / 8)) * 8);
    const int len_old = *len_dev;
    const uint32_t epoch = sa.counters[1];
    uint4 kf[KB][NC];
    uint32_t vv[KB][4][DL / 2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int k2 = 0; k2 < KB; ++k2) {
            const uint16_t *kr = kb + (int64_t)min(t + k2 * 16 + x, rows - 1) * s_r + qd * 8;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) kf[k2][cc] = *reinterpret_cast<const uint4 *>(kr + cc * 32);
        }
#pragma unroll
        for (int k2 = 0; k2 < KB; ++k2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint16_t *vr = vb + (int64_t)min(t + k2 * 16 + 4 * qd + i, rows - 1) * s_r + x * DL;
                if (DL == 4) {
                    const uint2 u = *reinterpret_cast<const uint2 *>(vr);
                    vv[k2][i][0] = u.x; vv[k2][i][1] = u.y;
                } else {
#pragma unroll
                    for (int e = 0; e < DL / 8; ++e) {
                        const uint4 u = *reinterpret_cast<const uint4 *>(vr + e * 8);
                        vv[k2][i][e * 4] = u.x; vv[k2][i][e * 4 + 1] = u.y; vv[k2][i][e * 4 + 2] = u.z; vv[k2][i][e * 4 + 3] = u.w;
                    }
                }
            }
        }
    };
    __builtin_amdgcn_sched_barrier(0);                           // (the scheduler would issue the tile's loads first: in-order return, the rotation would wait for them)
    load_tile(lo + w * ST_TILE);
    bool loaded = true;
    FKD_STAMP(1);
    const uint32_t token = handoff_token(epoch);
    const int jnew = min(len_old, rows - 1);                     // the step's own row (a full slab overwrites its last row: reported below)
    const bool owner = jnew >= lo && jnew < lo + chunk;
    const bool live = lo <= jnew;

    if (live) {
        // rotate-half RoPE, the stock fp16 sequence rounding for rounding (decode.hip decode_rope_kernel): x*cos -> fp16,
        // rotate_half(x)*sin -> fp16, sum -> fp16
        if (rope_item) {
            const uint32_t a1[4] = {x1.x, x1.y, x1.z, x1.w}, a2[4] = {x2.x, x2.y, x2.z, x2.w}, k1[4] = {c1.x, c1.y, c1.z, c1.w},
                           k2[4] = {c2.x, c2.y, c2.z, c2.w}, n1[4] = {s1.x, s1.y, s1.z, s1.w}, n2[4] = {s2.x, s2.y, s2.z, s2.w};
            uint32_t o1[4], o2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint16_t r1[2], r2[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int sh = hh * 16;
                    const float xa = h2f((uint16_t)(a1[e] >> sh)), xb = h2f((uint16_t)(a2[e] >> sh));
                    r1[hh] = f2h(h2f(f2h(xa * h2f((uint16_t)(k1[e] >> sh)))) + h2f(f2h(-xb * h2f((uint16_t)(n1[e] >> sh)))));
                    r2[hh] = f2h(h2f(f2h(xb * h2f((uint16_t)(k2[e] >> sh)))) + h2f(f2h(xa * h2f((uint16_t)(n2[e] >> sh)))));
                }
                o1[e] = (uint32_t)r1[0] | ((uint32_t)r1[1] << 16);
                o2[e] = (uint32_t)r2[0] | ((uint32_t)r2[1] << 16);
            }
            uint16_t *dst = tid < NQ ? s_qh + ritem * D : s_knew;
            *reinterpret_cast<uint4 *>(dst + d0) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
            *reinterpret_cast<uint4 *>(dst + d0 + D / 2) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
        }
        if (v_item) *reinterpret_cast<uint4 *>(s_vnew + (tid - NR) * 8) = vn;
        __syncthreads();
        if (owner && tid < D / 8) {                              // the row goes to the slab for the steps to come
            *reinterpret_cast<uint4 *>(kslab + b * s_b + hk * s_h + (int64_t)jnew * s_r + tid * 8) = *reinterpret_cast<const uint4 *>(s_knew + tid * 8);
            *reinterpret_cast<uint4 *>(vslab + b * s_b + hk * s_h + (int64_t)jnew * s_r + tid * 8) = *reinterpret_cast<const uint4 *>(s_vnew + tid * 8);
        }
        FKD_STAMP(2);
        // B operand: query head x (columns G .. 15 are zero), dims cc*32 + qd*8 ...
        uint4 qf[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
            qf[cc] = x < G ? *reinterpret_cast<const uint4 *>(s_qh + x * D + cc * 32 + qd * 8) : make_uint4(0u, 0u, 0u, 0u);

        const int hi = min(jnew, lo + chunk);                    // the slice's rows that are in the cache
        float m = -INFINITY, l = 0.0f, o[G][DL];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < DL; ++e) o[g][e] = 0.0f;
        bool first = true;
        for (int t0 = lo + w * ST_TILE; t0 < hi; t0 += ST_WG_ROWS) {
            if (!loaded) load_tile(t0);
            loaded = false;
            FKD_DRAIN();
            FKD_STAMP(3);
            st_f32x4 acc[KB];
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2) {
                acc[k2] = (st_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int cc = 0; cc < NC; ++cc)
                    acc[k2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(st_f16x8, kf[k2][cc]), __builtin_bit_cast(st_f16x8, qf[cc]),
                                                                    acc[k2], 0, 0, 0);
            }
            // lane (x, qd): scores of query head x against rows t0 + k2*16 + 4*qd + i
            float s[KB][4], tm = -INFINITY;
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[k2][i] = t0 + k2 * 16 + 4 * qd + i < hi ? acc[k2][i] * scaling : -INFINITY;
                    tm = fmaxf(tm, s[k2][i]);
                }
            const float mn = fmaxf(m, tm), mref = mn == -INFINITY ? 0.0f : mn;
            const float corr = __expf(m - mref);                 // m = -inf: 0
            float ps = 0.0f;
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2)
#pragma unroll
                for (int i = 0; i < 4; ++i) { s[k2][i] = __expf(s[k2][i] - mref); ps += s[k2][i]; }
            l = l * corr + ps;
            m = mn;
            if (!first) scale_rows<G, DL>(o, corr);
            first = false;
            const bool ragged = t0 + ST_TILE > hi;               // rows beyond the cache may hold anything (0 * NaN): zero them
#pragma unroll
            for (int k2 = 0; k2 < KB; ++k2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vf[DL];
                    const bool dropped = ragged && t0 + k2 * 16 + 4 * qd + i >= hi;
#pragma unroll
                    for (int e = 0; e < DL / 2; ++e) {
                        const uint32_t wd = dropped ? 0u : vv[k2][i][e];
                        vf[2 * e] = h2f((uint16_t)(wd & 0xffffu));
                        vf[2 * e + 1] = h2f((uint16_t)(wd >> 16));
                    }
                    pv_row<G, DL>(o, s[k2][i], vf);
                }
        }
        FKD_STAMP(4);
        // ---- the wave's four row groups -> ONE set per wave, in registers: the common maximum (two swap-and-max steps), every group's
        //      values scaled to it, then a transposing reduction -- v_permlane32_swap / v_permlane16_swap exchange half of one register
        //      with half of another, so one swap + one add folds TWO registers into one: G*DL registers of per-group partial sums
        //      become G*DL/4 registers of wave totals, row r of register i holding what register 4 i + {0, 2, 1, 3}[r] held.
        //      (A wave alone on its SIMD issues an instruction every 4-8 cycles: 16 sets through LDS cost 1.3 us, this costs 0.x.)
        constexpr int N = G * DL;
        float mw = m;
        { const st_u32x2 r = __builtin_amdgcn_permlane16_swap(f32_bits(mw), f32_bits(mw), false, false); mw = fmaxf(bits_f32(r[0]), bits_f32(r[1])); }
        { const st_u32x2 r = __builtin_amdgcn_permlane32_swap(f32_bits(mw), f32_bits(mw), false, false); mw = fmaxf(bits_f32(r[0]), bits_f32(r[1])); }
        const float fsc = m == -INFINITY ? 0.0f : __expf(m - mw);
        float lw = l * fsc;
        { const st_u32x2 r = __builtin_amdgcn_permlane16_swap(f32_bits(lw), f32_bits(lw), false, false); lw = bits_f32(r[0]) + bits_f32(r[1]); }
        { const st_u32x2 r = __builtin_amdgcn_permlane32_swap(f32_bits(lw), f32_bits(lw), false, false); lw = bits_f32(r[0]) + bits_f32(r[1]); }
        scale_rows<G, DL>(o, fsc);
        float S[N / 2], T[N / 4];
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            const st_u32x2 r = __builtin_amdgcn_permlane32_swap(f32_bits(o[(2 * j) / DL][(2 * j) % DL]), f32_bits(o[(2 * j + 1) / DL][(2 * j + 1) % DL]), false, false);
            S[j] = bits_f32(r[0]) + bits_f32(r[1]);
        }
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const st_u32x2 r = __builtin_amdgcn_permlane16_swap(f32_bits(S[2 * i]), f32_bits(S[2 * i + 1]), false, false);
            T[i] = bits_f32(r[0]) + bits_f32(r[1]);
        }
        const int pr = qd == 1 ? 2 : qd == 2 ? 1 : qd;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const int n = 4 * i + pr, g = n / DL, e = n - g * DL;
            s_o[w * (G * D) + g * D + x * DL + e] = T[i];
        }
        if (x < G && qd == 0) { s_m[w * G + x] = mw; s_l[w * G + x] = lw; }
        if (owner) {                                             // the step's own row: a fifth set {its score, 1, its V row}; wave w scores heads w, w + 4
            for (int g = w; g < G; g += 4) {
                float a = 0.0f;
                for (int d = lane; d < D; d += 64) a = __builtin_fmaf(h2f(s_qh[g * D + d]), h2f(s_knew[d]), a);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                if (lane == 0) s_m[4 * G + g] = a * scaling;
            }
        }
        __syncthreads();
        // ---- the four waves' sets (+ the step's own row) -> one record of the slice
        for (int i2 = tid * 2; i2 < G * D; i2 += 2 * ST_THREADS) {
            const int g = i2 / D, d = i2 - g * D;
            float M = owner ? s_m[4 * G + g] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) M = fmaxf(M, s_m[u * G + g]);
            float L = 0.0f, O0 = 0.0f, O1 = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float mu = s_m[u * G + g], f = mu == -INFINITY ? 0.0f : __expf(mu - M);
                const float2 ov = *reinterpret_cast<const float2 *>(s_o + u * (G * D) + g * D + d);
                L += s_l[u * G + g] * f;
                O0 += ov.x * f;
                O1 += ov.y * f;
            }
            if (owner) {
                const float f = __expf(s_m[4 * G + g] - M);
                L += f;
                O0 += h2f(s_vnew[d]) * f;
                O1 += h2f(s_vnew[d + 1]) * f;
            }
            if (nsplit == 1) {
                *reinterpret_cast<uint32_t *>(sa.out + ((size_t)b * sa.H + hk * G + g) * D + d) = (uint32_t)f2h(O0 / L) | ((uint32_t)f2h(O1 / L) << 16);
            } else {
                // agent-scope stores: through the XCD's L2 (where a merger on the same XCD finds them) to memory (where any merger does)
                uint64_t *r = rec_head + ((size_t)c * G + g) * (D + 2);
                __hip_atomic_store(r + 2 + d, step_granule(token, O0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(r + 3 + d, step_granule(token, O1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d == 0) {
                    __hip_atomic_store(r, step_granule(token, M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(r + 1, step_granule(token, L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        FKD_STAMP(5);
    }
    // "I have read the length": a producer says so when it is done (the answer costs a trip to memory that nothing here should wait for)
    report_length_read(len_old, epoch);
    FKD_STAMP(12);
    } else {
        const int len_old = *len_dev;
        const uint32_t epoch = sa.counters[1];
        report_length_read(len_old, epoch);                      // (the merger has nothing else to do yet)
        const uint32_t token = handoff_token(epoch);
        const int jnew = min(len_old, rows - 1);
        // ---- the head's records.  A poll IS the read: every thread requests its share of ALL records of a batch of slices at once (16-B
        //      agent-scope loads of two granules: past this compute unit's L1, which another unit's stores never refresh) and repeats the
        //      whole request until every token is this launch's -- one round trip after the last record has landed.  Wave w owns values
        //      w*128 .. w*128+127 of a pass; lane u of a wave also fetches the {max, sum} pair of slice u and hands the slice's weight
        //      to the other lanes through v_readlane.
        const int nlive = jnew / chunk + 1;
        const uint64_t deadline = __builtin_amdgcn_s_memrealtime() + limit_ticks;
        bool dead = false;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec_head, 0, 0x7fffffff, 0x00020000);
        constexpr int CB = G * (D + 2) * 8;                      // bytes between the records of consecutive slices
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ++ps) {
            const int i2 = tid * 2, gg = (i2 < PD ? i2 : 0) / D, d = (i2 < PD ? i2 : 0) - gg * D, g = ps * GP + gg;
            const int gsel = NG == 1 ? 0 : lane / (64 / NG), msl = NG == 1 ? lane : lane % (64 / NG);   // the {max, sum} pair this lane fetches
            const int mg = ps * GP + ((w * 128) / D) % GP + gsel;
            float M = -INFINITY, L = 0.0f, O0 = 0.0f, O1 = 0.0f;
#pragma unroll 1
            for (int c0 = 0; c0 < nlive; c0 += ST_BATCH) {
                const int nb = min(ST_BATCH, nlive - c0);
                const int off_o = ((c0 * G + g) * (D + 2) + 2 + d) * 8, off_m = (((c0 + min(msl, nb - 1)) * G + min(mg, G - 1)) * (D + 2)) * 8;
                st_u32x4 ov[ST_BATCH], mlv;
                for (;;) {
                    asm volatile("" ::: "memory");               // every iteration reads memory again
                    mlv = load_granules<16>(rs, off_m);
#pragma unroll
                    for (int u = 0; u < ST_BATCH; ++u) ov[u] = load_granules<16>(rs, off_o + min(u, nb - 1) * CB);
                    bool ok = mlv[1] == token && mlv[3] == token;
#pragma unroll
                    for (int u = 0; u < ST_BATCH; ++u) ok = ok && ov[u][1] == token && ov[u][3] == token;
                    if (ok) break;
                    if (__builtin_amdgcn_s_memrealtime() > deadline) { dead = true; break; }
                }
                FKD_STAMP(6 + (c0 / ST_BATCH < 3 ? c0 / ST_BATCH : 3));
                // This lane's slice: its maximum; then, the batch maximum known, its weight and weighted sum.  Slices beyond the batch
                // weigh nothing (their loads repeat the last slice): the sums below run over all ST_BATCH slots without a branch --
                // a wave alone on its SIMD issues an instruction every 4-8 cycles, what this costs is its instruction count.
                const float mym = msl < nb ? bits_f32(mlv[0]) : -INFINITY;
                float bm = mym;                                  // maximum over the batch, per query head of the wave (NG halves of the wave)
#pragma unroll
                for (int off = 1; off < 64 / NG; off <<= 1) bm = fmaxf(bm, __shfl_xor(bm, off, 64));
                const float Mn = fmaxf(M, bm);                   // (a lane's pair and its values belong to the same head: its half of the wave)
                const float wgt = mym == -INFINITY ? 0.0f : __expf(mym - Mn);
                float wl = msl < nb ? bits_f32(mlv[2]) * wgt : 0.0f;
#pragma unroll
                for (int off = 1; off < 64 / NG; off <<= 1) wl += __shfl_xor(wl, off, 64);
                const float corr = M == -INFINITY ? 0.0f : __expf(M - Mn);
                L = L * corr + wl; O0 *= corr; O1 *= corr;
#pragma unroll
                for (int u = 0; u < ST_BATCH; ++u) {
                    float f;
                    if (NG == 1) f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wgt), u));
                    else f = __shfl(wgt, (lane & ~(64 / NG - 1)) + u, 64);
                    O0 = __builtin_fmaf(bits_f32(ov[u][0]), f, O0);
                    O1 = __builtin_fmaf(bits_f32(ov[u][2]), f, O1);
                }
                M = Mn;
            }
            if (i2 < PD) *reinterpret_cast<uint32_t *>(sa.out + ((size_t)b * sa.H + hk * G + g) * D + d) = (uint32_t)f2h(O0 / L) | ((uint32_t)f2h(O1 / L) << 16);
        }
        FKD_STAMP(10);
        // a record that never came (a producer died): reported like every abandoned wait, FASTKV_EABORTED from the next call
        if (dead && sa.host_flag) __hip_atomic_fetch_or(sa.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

}

}  // namespace fk

#ifdef FK_STAMP
extern "C" int fastkv_debug_read_step_stamps(unsigned long long *host, size_t n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fk::g_dstamps), n * sizeof(unsigned long long));
}
#endif

using namespace fk;

// Workgroup tile: 128 slab rows (four 32-row wave tiles).  64-row tiles (FASTKV_STEP_TILE=64: twice the workgroups, every compute unit
// busy at budget 2048) were measured SLOWER, 13.5 against 11.3 us per layer: the cache arrives no sooner (it is the memory's latency
// under 9.4 MB of requests, not the compute units' appetite, that sets the 4.5 us), and the merger reads twice the records.
static int step_tile_rows(int B, int Hkv, int rows)
{
    static const int forced = []() { const char *e = getenv("FASTKV_STEP_TILE"); return e ? atoi(e) : 0; }();     // measurements: 64 | 128
    (void)B; (void)Hkv; (void)rows;
    return forced == 64 ? 64 : 128;
}
// slices per KV head when the caller leaves the choice to the library: one workgroup tile per slice while the grid stays within
// ~2.5 workgroups per compute unit, at most 64
static int step_auto_nsplit(int B, int Hkv, int rows)
{
    const int wg_rows = step_tile_rows(B, Hkv, rows);
    int cap = 640 / (B * Hkv);
    cap = cap < 1 ? 1 : cap > 64 ? 64 : cap;
    int n = (rows + wg_rows - 1) / wg_rows;
    return n < 1 ? 1 : n > cap ? cap : n;
}

extern "C" {

size_t fastkv_decode_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t nsplit)
{
    if (B < 1 || H < 1 || D < 1) return 0;
    if (nsplit < 1) nsplit = 64;                                 // "the library chooses": room for every choice
    return align_up((size_t)B * H * nsplit * (D + 2) * sizeof(uint64_t), 256);
}

int fastkv_decode_step_attention_f16(int32_t B, int32_t H, int32_t Hkv, int32_t D, const void *q, const int64_t q_strides[2],
                                     const void *k_new, const int64_t kn_strides[2], const void *v_new, const int64_t vn_strides[2],
                                     const void *cosv, const void *sinv, int64_t cs_batch_stride, void *kslab, void *vslab,
                                     const int64_t slab_strides[3], int32_t rows, int32_t *len_dev, float scaling, int32_t nsplit,
                                     void *out, void *workspace, size_t workspace_bytes, void *counters, void *stream)
{
    if (B < 1 || Hkv < 1 || H < Hkv || (H % Hkv) || rows < 1 || nsplit > 64 || (size_t)B * Hkv > 65535) return FASTKV_EINVAL;
    if (!q || !k_new || !v_new || !cosv || !sinv || !kslab || !vslab || !len_dev || !out || !workspace || !counters || !q_strides ||
        !kn_strides || !vn_strides || !slab_strides)
        return FASTKV_EINVAL;
    if (D != 64 && D != 128 && D != 256) return FASTKV_EUNSUPPORTED;
    const int G = H / Hkv;
    if (G != 1 && G != 2 && G != 4 && G != 8) return FASTKV_EUNSUPPORTED;
    if (((uintptr_t)kslab | (uintptr_t)vslab | (uintptr_t)out | (uintptr_t)workspace) & 15) return FASTKV_EINVAL;
    for (int i = 0; i < 3; ++i) if (slab_strides[i] & 7) return FASTKV_EINVAL;
    if (nsplit < 1) nsplit = step_auto_nsplit(B, Hkv, rows);
    if (workspace_bytes < fastkv_decode_workspace_bytes(B, H, D, nsplit)) return FASTKV_EWORKSPACE;
    const int wg_rows = step_tile_rows(B, Hkv, rows);
    const int chunk = ((rows + nsplit - 1) / nsplit + wg_rows - 1) / wg_rows * wg_rows;
    hipStream_t st = (hipStream_t)stream;
    StepArgs sa;
    sa.k_new = (const uint16_t *)k_new; sa.kn_b = kn_strides[0]; sa.kn_h = kn_strides[1];
    sa.v_new = (const uint16_t *)v_new; sa.vn_b = vn_strides[0]; sa.vn_h = vn_strides[1];
    sa.cosv = (const uint16_t *)cosv; sa.sinv = (const uint16_t *)sinv; sa.cs_b = cs_batch_stride;
    sa.counters = (uint32_t *)counters; sa.out = (uint16_t *)out; sa.H = H; sa.Hkv = Hkv; sa.host_flag = abort_flag_device();
    dim3 grid(B * Hkv, nsplit + (nsplit > 1 ? 1 : 0));
    ProfScope ps_(K_DECODE, st);
#define FK_STEP(DV, GV)                                                                                                                 \
    if (wg_rows == 64)                                                                                                                  \
        hipLaunchKernelGGL((decode_step_kernel<DV, GV, 1>), grid, dim3(ST_THREADS), 0, st, (const uint16_t *)q, q_strides[0], q_strides[1], \
                       (uint16_t *)kslab, (uint16_t *)vslab, slab_strides[0], slab_strides[1], slab_strides[2], rows, len_dev, scaling, \
                       (uint64_t *)workspace, nsplit, chunk, sa, spin_limit_ticks());                                                   \
    else                                                                                                                                \
    hipLaunchKernelGGL((decode_step_kernel<DV, GV, 2>), grid, dim3(ST_THREADS), 0, st, (const uint16_t *)q, q_strides[0], q_strides[1],    \
                       (uint16_t *)kslab, (uint16_t *)vslab, slab_strides[0], slab_strides[1], slab_strides[2], rows, len_dev, scaling, \
                       (uint64_t *)workspace, nsplit, chunk, sa, spin_limit_ticks())
#define FK_STEP_G(DV)                                                                                      \
    do {                                                                                                   \
        if (G == 1) { FK_STEP(DV, 1); } else if (G == 2) { FK_STEP(DV, 2); } else if (G == 4) { FK_STEP(DV, 4); } else { FK_STEP(DV, 8); } \
    } while (0)
    if (D == 64) FK_STEP_G(64);
    else if (D == 128) FK_STEP_G(128);
    else FK_STEP_G(256);
#undef FK_STEP_G
#undef FK_STEP
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

}  // extern "C"
