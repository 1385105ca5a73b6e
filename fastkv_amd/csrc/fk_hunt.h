// Measurement switches of the round-3 co-residency hunt (docs/HISTORY.md, "a silent wrong answer ..."; tools/probes/README.md has the
// erratum note and the reproducer commands).  The PRODUCT build never defines FK_HUNT: every macro below is then empty and
// csrc/fused.hip compiles to exactly the code it would have without them.  A hunt build is
//     FASTKV_BUILD_DIR=build_x_hunt FASTKV_CXXFLAGS="-DFK_HUNT -DFK_DBG_DELAY=4 ..." python fastkv_amd/_build.py
// (measurement builds never go in-tree: fastkv_amd/_build.py).  Switches, all inert without -DFK_HUNT:
//   FK_OLD_NUMBERING        (entry, unit, span) from the launch's linear order; no placement check
//   FK_DBG_DELAY=4|5|6|8|9  who (FK_DBG_WHO, default: entry 0) is delayed and where: 4 s_sleep per tile of phase A, 5 a dependent-FMA loop
//                           there, 6 a hashed quarter of ALL waves, 8 asleep at the start, 9 asleep BEHIND phase A
//   FK_DBG_MATE_NOLDS / _NOMFMA / _VALUCVT / _CUT=n / _REGMFMA   what the delayed workgroups' matrix phase consists of
//   FK_DBG_VICTIM_WAIT      everybody else sits out 150 us between phase A and the first hand-off
//   FK_DBG_NOSLEEP, FK_DBG_SYNCAND, FK_DBG_NO_HIST, FK_DBG_VALU_SHFL   single constructs of the victims' side replaced / removed
// The macros expand INSIDE score_fused_body and use its locals (yb, w, t, s, lane, my, As, acc0, acc1, ...).
#pragma once

#ifndef FK_HUNT
#if defined(FK_DBG_DELAY) || defined(FK_OLD_NUMBERING) || defined(FK_DBG_VALU_SHFL) || defined(FK_DBG_SYNCAND) || defined(FK_DBG_NO_HIST) || defined(FK_DBG_NOSLEEP)
#error "the FK_DBG_* / FK_OLD_NUMBERING switches need -DFK_HUNT (csrc/fk_hunt.h)"
#endif
#define FKH_GLOBALS
#define FKH_POLL_SLEEP() __builtin_amdgcn_s_sleep(8)
#define FKH_SHARED
#define FKH_OLD_NUMBERING 0
#define FKH_DELAY_AT_START()
#define FKH_MATE_NO_STAGING
#define FKH_MATE_PHASE
#define FKH_DELAY_AFTER_TILE()
#define FKH_BEFORE_FIRST_HANDOFF()
#define FKH_SYNC_AND(ok) __syncthreads_and(ok)
#define FKH_HIST 1
#else

#ifdef FK_DBG_VALU_SHFL
// every __shfl_xor of the translation unit without the LDS crossbar -- v_permlane32_swap / v_permlane16_swap (gfx950) for the masks
// 32 / 16, DPP for 8 / 4 / 2 / 1
namespace fk {
__device__ __forceinline__ uint32_t dbg_xor_u32(uint32_t v, int mask)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (mask == 32) { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
    if (mask == 16) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
    if (mask == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);                 // row_ror:8
    if (mask == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);               // row_shr:4: lane i <- i - 4
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);               // row_shl:4: lane i <- i + 4
        return (lane & 4) ? a : b;
    }
    if (mask == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, true);                  // quad_perm [2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);                                 // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dbg_shfl_xor(float v, int m, int) { return __builtin_bit_cast(float, dbg_xor_u32(__builtin_bit_cast(uint32_t, v), m)); }
__device__ __forceinline__ int dbg_shfl_xor(int v, int m, int) { return (int)dbg_xor_u32((uint32_t)v, m); }
__device__ __forceinline__ uint32_t dbg_shfl_xor(uint32_t v, int m, int) { return dbg_xor_u32(v, m); }
__device__ __forceinline__ uint64_t dbg_shfl_xor(uint64_t v, int m, int)
{
    return ((uint64_t)dbg_xor_u32((uint32_t)(v >> 32), m) << 32) | dbg_xor_u32((uint32_t)v, m);
}
}  // namespace fk
#define __shfl_xor(v, m, w) fk::dbg_shfl_xor((v), (m), (w))
#endif

#ifdef FK_DBG_DELAY
#define FKH_GLOBALS __device__ int g_dbg_delay_ticks = 1500;     /* 100 MHz ticks per delay (fastkv_debug_set_delay) */
#ifndef FK_DBG_WHO
#define FK_DBG_WHO (yb == 0)
#endif
#define FKH_SLEEP_TICKS(ticks) do { const uint64_t t_end_ = wall_clock64() + (ticks); while (wall_clock64() < t_end_) __builtin_amdgcn_s_sleep(8); } while (0)
#else
#define FKH_GLOBALS
#endif

#ifdef FK_DBG_NOSLEEP
#define FKH_POLL_SLEEP()
#else
#define FKH_POLL_SLEEP() __builtin_amdgcn_s_sleep(8)
#endif

#ifdef FK_DBG_SYNCAND
#define FKH_SHARED __shared__ uint32_t s_dbg_flag;
// __syncthreads_and replaced by a flag word in LDS
#define FKH_SYNC_AND(ok) ([&]() { if (tix == 0) s_dbg_flag = 0; __syncthreads(); if (!(ok)) s_dbg_flag = 1; __syncthreads(); return !s_dbg_flag; }())
#else
#define FKH_SHARED
#define FKH_SYNC_AND(ok) __syncthreads_and(ok)
#endif

#ifdef FK_OLD_NUMBERING
#define FKH_OLD_NUMBERING 1
#else
#define FKH_OLD_NUMBERING 0
#endif

#if defined(FK_DBG_DELAY) && FK_DBG_DELAY == 8
#define FKH_DELAY_AT_START() do { if (FK_DBG_WHO) FKH_SLEEP_TICKS(g_dbg_delay_ticks); } while (0)
#else
#define FKH_DELAY_AT_START()
#endif

// the delayed workgroups stage no K (no LDS writes, no loads): `FKH_MATE_NO_STAGING <the staging statement>`
#if defined(FK_DBG_DELAY) && defined(FK_DBG_MATE_NOLDS)
#define FKH_MATE_NO_STAGING if (FK_DBG_WHO) { } else
#else
#define FKH_MATE_NO_STAGING
#endif

// what the delayed workgroups run instead of the kernel's matrix phase: `FKH_MATE_PHASE <the matrix phase statement>`
#if defined(FK_DBG_DELAY) && defined(FK_DBG_MATE_NOMFMA)         // no MFMA (and no LDS read)
#define FKH_MATE_PHASE if (!(FK_DBG_WHO))
#elif defined(FK_DBG_DELAY) && defined(FK_DBG_MATE_VALUCVT)      // K converted on the vector ALU: fp32 MFMAs only, none of the fp16 ones
#define FKH_MATE_PHASE if (FK_DBG_WHO) mfma_phase<NB>(acc0, acc1, my, As + s * AS_FLOATS + ph * (DH / 2) * 64 + lane, n31, sh); else
#elif defined(FK_DBG_DELAY) && defined(FK_DBG_MATE_CUT)          // one of the cuts of mfma_phase_mx (mfma_tile.h)
#define FKH_MATE_PHASE if (FK_DBG_WHO) mfma_phase_mx<NB, FK_DBG_MATE_CUT>(acc0, acc1, my, As + s * AS_FLOATS + ph * (DH / 2) * 64 + lane, n31, hi, pm0, pm1); else
#elif defined(FK_DBG_DELAY) && defined(FK_DBG_MATE_REGMFMA)      // the same MFMAs on register operands: no LDS read in the matrix phase
#define FKH_MATE_PHASE                                                                                     \
    if (FK_DBG_WHO) {                                                                                      \
        f32x16 z9;                                                                                         \
        for (int i9 = 0; i9 < 16; ++i9) z9[i9] = 0.0f;                                                     \
        for (int r9 = 0; r9 < 4; ++r9) {                                                                   \
            z9 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pm0, pm1, z9, 0, 0, 0);                            \
            z9 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pm1, pm0, z9, 0, 0, 0);                            \
        }                                                                                                  \
        for (int r9 = 0; r9 < 32; ++r9) {                                                                  \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(z9[r9 & 15], 1.0f, acc0, 0, 0, 0);                 \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, z9[r9 & 15], acc1, 0, 0, 0);                 \
        }                                                                                                  \
    } else
#else
#define FKH_MATE_PHASE
#endif

// some workgroups are slow in phase A (no NaN involved): behind every tile
#if defined(FK_DBG_DELAY) && FK_DBG_DELAY < 8
#if FK_DBG_DELAY == 6
// every workgroup, but only SOME of its waves (a hash of workgroup, wave and tile picks them): the waves of a workgroup reach the end of
// phase A far apart -- does the kernel depend on its waves running in step?
#define FKH_DELAY_AFTER_TILE() do { if (FK_DBG_WHO) { if ((((blockIdx.y * gridDim.x + blockIdx.x) * 2654435761u + w * 40503u + t * 977u) >> 7 & 3u) == 0u) FKH_SLEEP_TICKS(g_dbg_delay_ticks); } } while (0)
#elif FK_DBG_DELAY == 4
#define FKH_DELAY_AFTER_TILE() do { if (FK_DBG_WHO) FKH_SLEEP_TICKS(g_dbg_delay_ticks); } while (0)
#else
#define FKH_DELAY_AFTER_TILE() do { if (FK_DBG_WHO) { const uint64_t t_end = wall_clock64() + g_dbg_delay_ticks; float zz = acc0[0]; \
        while (wall_clock64() < t_end) { for (int q9 = 0; q9 < 64; ++q9) zz = __builtin_fmaf(zz, 1.0000001f, 1e-30f); }                 \
        if (zz == 123.456f) acc0[0] = zz; } } while (0)
#endif
#else
#define FKH_DELAY_AFTER_TILE()
#endif

#if defined(FK_DBG_DELAY) && defined(FK_DBG_VICTIM_WAIT)         // everybody ELSE sits out 150 us between phase A and the first hand-off
#define FKH_BEFORE_FIRST_HANDOFF() do { if (!(FK_DBG_WHO)) FKH_SLEEP_TICKS(15000); } while (0)
#elif defined(FK_DBG_DELAY) && FK_DBG_DELAY == 9
#define FKH_BEFORE_FIRST_HANDOFF() do { if (FK_DBG_WHO) FKH_SLEEP_TICKS(g_dbg_delay_ticks); } while (0)
#else
#define FKH_BEFORE_FIRST_HANDOFF()
#endif

#ifdef FK_DBG_NO_HIST
#define FKH_HIST 0
#else
#define FKH_HIST 1
#endif

#endif  // FK_HUNT
