// K-tile staging and the fp32-MFMA phase shared by the score kernels (score.hip, fused.hip).  gfx950 only.
#pragma once
#include "fk_device.h"

namespace fk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------ K staging shared by both engines
// A wave stages 64 dims of its 64 consecutive key rows through a private LDS slab: coalesced 16-B global loads (8 lanes
// per 128-B row segment), 144-B padded rows so that one-row-per-lane ds_read_b128 is bank-conflict free.
constexpr int DH = 64;
constexpr int ROWB = DH * 2 + 16;

__device__ __forceinline__ void stage_k(const uint16_t *__restrict__ kb, int64_t ks_s, int key0, int S, int ph, int lane,
                                        unsigned char *my)
{
    const int lrow = lane >> 3, lchunk = lane & 7;
    uint4 st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int jj = key0 + i * 8 + lrow;
        jj = jj < S ? jj : S - 1;
        st[i] = *reinterpret_cast<const uint4 *>(kb + (int64_t)jj * ks_s + ph * DH + lchunk * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(my + (i * 8 + lrow) * ROWB + lchunk * 16) = st[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct KStage { u32x4 r0, r1, r2, r3, r4, r5, r6, r7; };

// One phase of a 64-key tile: 8 x 16 B per lane (lane l: key row i*8 + (l>>3), bytes (l&7)*16 of the 128-B phase slice).
// Full tiles are buffer loads: ONE per-lane 32-bit offset, the tile base in a wave-uniform descriptor and the row block
// in the scalar offset; only the ragged last tile clamps the row per lane.  Per-load 64-bit vector addresses for two stages cost 64 VGPRs and spilled.
// NB = 32-key column blocks per tile: 2 (64-key tiles, the default) or 1 (32-key tiles: twice the waves on short prompts;
// only registers r0-r3 of a stage are used).
template <int NB = 2>
__device__ __forceinline__ void k_fetch(KStage &st, const uint16_t *__restrict__ kb, int64_t ks_s, int key0, int S, int ph, int lane)
{
    if (key0 + 32 * NB <= S) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t *>(kb + (int64_t)key0 * ks_s + ph * DH), 0, 0x7fffffff, 0x00020000);        // wave-uniform
        const int loff = ((lane >> 3) * (int)ks_s + (lane & 7) * 8) * 2;
        const int step = (int)ks_s * 16;                                                                     // 8 rows, bytes
#define FK_KLD(i) __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, loff, (i) * step, 0))
        st.r0 = FK_KLD(0); st.r1 = FK_KLD(1); st.r2 = FK_KLD(2); st.r3 = FK_KLD(3);
        if (NB == 2) { st.r4 = FK_KLD(4); st.r5 = FK_KLD(5); st.r6 = FK_KLD(6); st.r7 = FK_KLD(7); }
#undef FK_KLD
    } else {
        u32x4 t[4 * NB];
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) {
            int jj = key0 + i * 8 + (lane >> 3);
            jj = jj < S ? jj : S - 1;
            t[i] = *reinterpret_cast<const u32x4 *>(kb + (int64_t)jj * ks_s + ph * DH + (lane & 7) * 8);
        }
        st.r0 = t[0]; st.r1 = t[1]; st.r2 = t[2]; st.r3 = t[3];
        if (NB == 2) { st.r4 = t[4 % (4 * NB)]; st.r5 = t[5 % (4 * NB)]; st.r6 = t[6 % (4 * NB)]; st.r7 = t[7 % (4 * NB)]; }
    }
}
template <int NB = 2> __device__ __forceinline__ void k_commit(const KStage &st, int lane, unsigned char *my)
{
    unsigned char *base = my + (lane >> 3) * ROWB + (lane & 7) * 16;
    *reinterpret_cast<u32x4 *>(base + 0 * 8 * ROWB) = st.r0; *reinterpret_cast<u32x4 *>(base + 1 * 8 * ROWB) = st.r1;
    *reinterpret_cast<u32x4 *>(base + 2 * 8 * ROWB) = st.r2; *reinterpret_cast<u32x4 *>(base + 3 * 8 * ROWB) = st.r3;
    if (NB == 2) {
        *reinterpret_cast<u32x4 *>(base + 4 * 8 * ROWB) = st.r4; *reinterpret_cast<u32x4 *>(base + 5 * 8 * ROWB) = st.r5;
        *reinterpret_cast<u32x4 *>(base + 6 * 8 * ROWB) = st.r6; *reinterpret_cast<u32x4 *>(base + 7 * 8 * ROWB) = st.r7;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One 64-dim phase of a 64-key tile: 64 k-steps x 2 column blocks = 128 MFMAs, in 8 groups of 4 k-steps.
// Software pipeline, two groups deep: the LDS reads of group c+2 (2 x 16 B of K per lane, 4 A values) and the fp16->fp32
// conversions of group c+1 are written between the MFMAs of group c (the compiler places them as it likes; measured:
// tools/probes/probe_phase.hip).  A wave's vector instructions do NOT overlap its own fp32 MFMAs -- 2 conversions per
// MFMA make a solo wave run at 85 cycles per MFMA instead of 66 -- so the loop carries nothing else and a SIMD hosts two
// waves, each filling the other's conversion slots (34 ns per MFMA and SIMD against 29 ns of pure issue).
struct KGroup { uint4 k0, k1; float a0, a1, a2, a3; };
template <int NB = 2> __device__ __forceinline__ KGroup read_group(const unsigned char *my, const float *Ap, int n31, int c)
{
    KGroup g;
    g.k0 = *reinterpret_cast<const uint4 *>(my + n31 * ROWB + c * 16);
    g.k1 = NB == 2 ? *reinterpret_cast<const uint4 *>(my + (32 + n31) * ROWB + c * 16) : g.k0;
    g.a0 = Ap[(c * 4 + 0) * 64]; g.a1 = Ap[(c * 4 + 1) * 64]; g.a2 = Ap[(c * 4 + 2) * 64]; g.a3 = Ap[(c * 4 + 3) * 64];
    return g;
}
struct BGroup { float b0[4], b1[4], a[4]; };
__device__ __forceinline__ float cvt_lo_hi(uint32_t wd, int sh) { return h2f((uint16_t)((wd >> sh) & 0xffffu)); }
template <int NB = 2>
__device__ __forceinline__ void mfma_phase(f32x16 &acc0, f32x16 &acc1, const unsigned char *my, const float *Ap, int n31, int sh)
{
    KGroup r1 = read_group<NB>(my, Ap, n31, 0), r2 = read_group<NB>(my, Ap, n31, 1);
    BGroup cur;
    {
        const uint32_t w0[4] = {r1.k0.x, r1.k0.y, r1.k0.z, r1.k0.w}, w1[4] = {r1.k1.x, r1.k1.y, r1.k1.z, r1.k1.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) { cur.b0[u] = cvt_lo_hi(w0[u], sh); cur.b1[u] = cvt_lo_hi(w1[u], sh); }
        cur.a[0] = r1.a0; cur.a[1] = r1.a1; cur.a[2] = r1.a2; cur.a[3] = r1.a3;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // r2 = raw operands of group c+1 (already requested); request group c+2
        KGroup r3 = r2;
        if (c + 2 < 8) r3 = read_group<NB>(my, Ap, n31, c + 2);
        BGroup nxt = cur;
        const uint32_t w0[4] = {r2.k0.x, r2.k0.y, r2.k0.z, r2.k0.w}, w1[4] = {r2.k1.x, r2.k1.y, r2.k1.z, r2.k1.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {                            // k-step 4c+u: dims 2s (lanes 0-31), 2s+1 (lanes 32-63)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b0[u], acc0, 0, 0, 0);
            if (NB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u], cur.b1[u], acc1, 0, 0, 0);
            if (c + 1 < 8) { nxt.b0[u] = cvt_lo_hi(w0[u], sh); nxt.b1[u] = cvt_lo_hi(w1[u], sh); }
        }
        if (c + 1 < 8) { nxt.a[0] = r2.a0; nxt.a[1] = r2.a1; nxt.a[2] = r2.a2; nxt.a[3] = r2.a3; }
        cur = nxt;
        r2 = r3;
    }
}


// The same phase with the fp16 -> fp32 conversion of K done by the matrix pipe itself: per 32-dim chunk and 32-key block two
// v_mfma_f32_32x32x16_f16 multiply the K piece (the lane's ds_read_b128, already in B-operand layout) by a 0/1 permutation
// matrix, which lands K[j][2i + hi] in accumulator register i of lane (j, hi) -- exactly the B operand of k-step i of the
// fp32 MFMA.  Exact for finite K (one product 1*x, the rest 0*x = 0); a non-finite K element turns its whole chunk into NaN
// (0 * inf), which the caller detects in the results and redoes with mfma_phase.  8 extra short MFMAs per phase instead of
// 128 vector instructions that the fp32 MFMAs of the SIMD do not hide.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void perm_operands(int lane, f16x8 &p0, f16x8 &p1)
{
    const int r = lane & 31, h = lane >> 5;
    const int i = (r & 3) + 4 * (r >> 3), delta = 2 * i + ((r >> 2) & 1);      // output row r <-> dim delta of the chunk
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p0[e] = (_Float16)(delta == 8 * h + e ? 1.0f : 0.0f);
        p1[e] = (_Float16)(delta == 16 + 8 * h + e ? 1.0f : 0.0f);
    }
}
// (CUT: measurement builds of the hunt in docs/HISTORY.md -- 0 the kernel's phase; 1 the fp16 MFMAs take their K operand from
// registers instead of LDS; 2 the fp32 MFMAs take a constant B operand instead of the fp16 MFMAs' results; 3 the fp32 MFMAs take their A
// operand from registers instead of LDS; 4 the kernel's phase with the fp16 MFMAs' results copied by v_mov_b32 before the fp32 MFMAs read them.  Results are garbage for CUT != 0: only the delayed workgroups run those.)
template <int NB = 2, int CUT = 0>
__device__ __forceinline__ void mfma_phase_mx(f32x16 &acc0, f32x16 &acc1, const unsigned char *my, const float *Ap, int n31, int hi,
                                              f16x8 p0, f16x8 p1)
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    auto conv = [&](int blk, int ch) {
        const unsigned char *rowp = my + (blk * 32 + n31) * ROWB + ch * 64 + hi * 16;
        const f16x8 s0 = CUT == 1 ? p1 : *reinterpret_cast<const f16x8 *>(rowp), s1 = CUT == 1 ? p0 : *reinterpret_cast<const f16x8 *>(rowp + 32);
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, s0, z, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, s1, d, 0, 0, 0);
        if (CUT == 4) {                                          // the fp32 MFMAs read copies made by the vector ALU, not the matrix pipe's own result registers
#pragma unroll
            for (int i = 0; i < 16; ++i) { float o; asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(d[i])); d[i] = o; }
        }
        return d;
    };
    f32x16 b0 = conv(0, 0), b1 = NB == 2 ? conv(1, 0) : z;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        f32x16 n0 = b0, n1 = b1;
        float av[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] = CUT == 3 ? (float)(n31 + i) : Ap[(ch * 16 + i) * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], CUT == 2 ? 1.0f : b0[i], acc0, 0, 0, 0);
            if (NB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], CUT == 2 ? 2.0f : b1[i], acc1, 0, 0, 0);
            if (ch == 0 && i == 7) { n0 = conv(0, 1); if (NB == 2) n1 = conv(1, 1); }
        }
        if (CUT == 2) { acc0[0] += b0[0] * 1e-30f + b1[0] * 1e-30f; }
        b0 = n0; b1 = n1;
    }
}


// ------------------------------------------------------------------------------------------ the fp16 matrix instruction itself
// Contract "mfma16" (round 4; oracle/fastkv_oracle.c FK_CONTRACT_MFMA16): the contraction IS v_mfma_f32_32x32x16_f16 chained over the
// head dimension in ascending chunks of 16, accumulator from +0 -- no conversion of K, no fp32 matrix instruction.  One 64-dim phase of
// a tile = 4 instructions per 32-key block instead of 64 fp32 ones + 4 permutation products (1/16 of the matrix time).  The instruction's
// arithmetic (blocks of eight products, aligned truncating adds) is restated bit for bit by the oracle; Inf / NaN operands follow
// IEEE in both, so there is no redo path.
// Operands: A = the query block as fp16 fragments in LDS, Qf[c][lane] = dims 16c + 8 (lane / 32) .. + 7 of query row lane % 32 (16 B per
// lane, consecutive lanes: conflict-free ds_read_b128); B = the lane's K row piece straight from the wave's slab (same reads as the
// permutation products of mfma_phase_mx).  `qf` points at chunk 4 * ph of the stream's fragments, + lane.
template <int NB = 2>
__device__ __forceinline__ void mfma_phase_f16(f32x16 &acc0, f32x16 &acc1, const unsigned char *my, const f16x8 *qf, int n31, int hi)
{
    const unsigned char *r0 = my + n31 * ROWB + hi * 16, *r1 = my + (32 + n31) * ROWB + hi * 16;
    f16x8 ka[4], kb[4], qa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                               // chunk c of the phase: dims 16c .. 16c + 15
        qa[c] = qf[c * 64];
        ka[c] = *reinterpret_cast<const f16x8 *>(r0 + c * 32);
        if (NB == 2) kb[c] = *reinterpret_cast<const f16x8 *>(r1 + c * 32);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[c], ka[c], acc0, 0, 0, 0);
        if (NB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[c], kb[c], acc1, 0, 0, 0);
    }
}

// mfma_phase_mx is exact for finite K only: a NaN among a tile's results (non-finite K or Q) sends the wave back over the
// tile with the vector-ALU conversion (synchronous staging through the wave's slab: rare), whose results are the fmaf chain
// on any input.  `Ap` = the wave's A-operand pointer for phase 0 (As + lane).
// Returns whether the tile's results held a NaN (wave-uniform): such a tile keeps the general softmax path later on.
template <int NPH, int NB = 2>
__device__ __forceinline__ bool redo_tile_if_nan(f32x16 &acc0, f32x16 &acc1, const uint16_t *__restrict__ kb, int64_t ks_s, int key0,
                                                 int S, int lane, unsigned char *my, const float *Ap, int n31, int sh)
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) bad = bad || (NB == 2 ? __builtin_isunordered(acc0[i], acc1[i]) : acc0[i] != acc0[i]);   // one compare per pair
    if (!__any(bad)) return false;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
#pragma unroll 1
        for (int i = 0; i < 4 * NB; ++i) {
            int jj = key0 + i * 8 + (lane >> 3);
            jj = jj < S ? jj : S - 1;
            *reinterpret_cast<u32x4 *>(my + (i * 8 + (lane >> 3)) * ROWB + (lane & 7) * 16) =
                *reinterpret_cast<const u32x4 *>(kb + (int64_t)jj * ks_s + ph * DH + (lane & 7) * 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        mfma_phase<NB>(acc0, acc1, my, Ap + ph * (DH / 2) * 64, n31, sh);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return true;
}

}  // namespace fk
