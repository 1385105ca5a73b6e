// Sequence-sharded operator (fastkv_amd/dist.py: sp_update_kv), the stages around the collectives.  The reference has no
// multi-GPU path; the contract is "same bits as FastKVCluster.update_kv on one device"
// (/root/reference/baselines/fastkv/utils.py:113-130).
//
//   sp_pack     local canonical winners of a rank -> fixed-size candidate records {fp16 score bits << 32 | global position},
//               padded with {-inf, 0xffffffff}: the payload of the ONE candidate all-gather (kv rows and the TSP row share it)
//   sp_unpack   gathered records [P][...] -> fp16 score rows [rows, P*k] for the final canonical selection
//   sp_pick     final winners (indices into the P*k candidates) -> global positions (+ the window positions of the TSP index)
//   sp_compact  K/V rows of the global winners THIS rank owns (+ the window rows on the last rank) into [B,Hkv,cap,D];
//               rows owned by other ranks are written as zeros, so the ranks' outputs add up to the single-device result
// All of it is index arithmetic and byte movement: one thread per record / one 16-lane group per row, 16-B accesses.
#include "fk_device.h"
#include "fk_host.h"
#include "prof.h"

namespace fk {

constexpr uint64_t SP_PAD = ((uint64_t)0xFC00u << 32) | 0xFFFFFFFFull;     // fp16 -inf, no position

__global__ void __launch_bounds__(256) sp_pack_kernel(const uint16_t *__restrict__ scores, int64_t row_stride,
                                                      const int64_t *__restrict__ idx_local, int64_t kl, int64_t k, int64_t pos0,
                                                      uint64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (i >= k) return;
    uint64_t rec = SP_PAD;
    if (i < kl) {
        const int64_t j = idx_local[row * kl + i];
        rec = ((uint64_t)scores[row * row_stride + j] << 32) | (uint64_t)(uint32_t)(j + pos0);
    }
    out[row * k + i] = rec;
}

__global__ void __launch_bounds__(256) sp_unpack_kernel(const uint64_t *__restrict__ allc, int64_t rank_stride, int64_t off, int P,
                                                        int64_t k, uint16_t *__restrict__ scores_out, int64_t out_stride)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (c >= P * k) return;
    const int64_t r = c / k, i = c - r * k;
    scores_out[row * out_stride + c] = (uint16_t)(allc[r * rank_stride + off + row * k + i] >> 32);
}

__global__ void __launch_bounds__(256) sp_pick_kernel(const uint64_t *__restrict__ allc, int64_t rank_stride, int64_t off, int64_t k,
                                                      const int64_t *__restrict__ sel, int64_t kout, int64_t append, int64_t n_glob,
                                                      int64_t *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (j >= kout + append) return;
    int64_t pos;
    if (j < kout) {
        const int64_t c = sel[row * kout + j], r = c / k, i = c - r * k;
        pos = (int64_t)(uint32_t)allc[r * rank_stride + off + row * k + i];
    } else {
        pos = n_glob + (j - kout);                              // utils.py:128-129
    }
    out[row * (kout + append) + j] = pos;
}

// grid (ceil(cap / RPB), B*Hkv); one row per LPR-lane group as in compact_kv (compact.hip)
template <int LPR>
__global__ void __launch_bounds__(256) sp_compact_kernel(const uint16_t *__restrict__ k, int64_t ks_b, int64_t ks_h, int64_t ks_s,
                                                         const uint16_t *__restrict__ v, int64_t vs_b, int64_t vs_h, int64_t vs_s,
                                                         const int64_t *__restrict__ kv_idx, int Hkv, int S_r, int64_t pos0, int W,
                                                         int cap, int window_owner, uint16_t *__restrict__ k_out,
                                                         uint16_t *__restrict__ v_out)
{
    constexpr int RPB = 256 / LPR;
    const int bg = blockIdx.y, b = bg / Hkv, g = bg % Hkv;
    const int kk = cap - W;
    const int sub = threadIdx.x % LPR;
    const int r = blockIdx.x * RPB + threadIdx.x / LPR;
    if (r >= cap) return;
    int64_t lrow;
    bool own;
    if (r < kk) {
        const int64_t gp = kv_idx[(size_t)bg * kk + r];
        own = gp >= pos0 && gp < pos0 + S_r;
        lrow = gp - pos0;
    } else {
        own = window_owner != 0;                                 // the prompt's last W positions live on the last rank
        lrow = (int64_t)S_r - W + (r - kk);
    }
    if (!own || lrow < 0) { own = false; lrow = 0; }             // unconditional loads from a valid row, zeros selected afterwards
    uint4 kval = *reinterpret_cast<const uint4 *>(k + b * ks_b + (int64_t)g * ks_h + lrow * ks_s + sub * 8);
    uint4 vval = *reinterpret_cast<const uint4 *>(v + b * vs_b + (int64_t)g * vs_h + lrow * vs_s + sub * 8);
    if (!own) { kval = make_uint4(0, 0, 0, 0); vval = kval; }
    const size_t o = ((size_t)bg * cap + r) * (LPR * 8) + sub * 8;
    *reinterpret_cast<uint4 *>(k_out + o) = kval;
    *reinterpret_cast<uint4 *>(v_out + o) = vval;
}

}  // namespace fk

using namespace fk;

extern "C" {

int fastkv_sp_pack_f16(const void *scores, int64_t rows, int64_t row_stride, const int64_t *idx_local, int64_t kl, int64_t k,
                       int64_t pos0, int64_t *records_out, void *stream)
{
    if (rows < 0 || kl < 0 || k < 0 || kl > k || pos0 < 0 || !records_out || (kl > 0 && (!scores || !idx_local))) return FASTKV_EINVAL;
    if (rows > 65535 || pos0 >= (1ll << 31)) return FASTKV_EUNSUPPORTED;
    if (rows == 0 || k == 0) return FASTKV_OK;
    ProfScope ps_(K_SP_AUX, (hipStream_t)stream);
    hipLaunchKernelGGL(sp_pack_kernel, dim3((unsigned)((k + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)scores, row_stride, idx_local, kl, k, pos0, (uint64_t *)records_out);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_sp_unpack_f16(const int64_t *records, int64_t rank_stride, int64_t offset, int32_t P, int64_t rows, int64_t k,
                         void *scores_out, int64_t out_stride, void *stream)
{
    if (!records || !scores_out || P < 1 || rows < 0 || k < 0 || offset < 0 || rank_stride < offset + rows * k || out_stride < P * k)
        return FASTKV_EINVAL;
    if (rows > 65535) return FASTKV_EUNSUPPORTED;
    if (rows == 0 || k == 0) return FASTKV_OK;
    ProfScope ps_(K_SP_AUX, (hipStream_t)stream);
    hipLaunchKernelGGL(sp_unpack_kernel, dim3((unsigned)((P * k + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       (const uint64_t *)records, rank_stride, offset, (int)P, k, (uint16_t *)scores_out, out_stride);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_sp_pick(const int64_t *records, int64_t rank_stride, int64_t offset, int64_t rows, int64_t k, const int64_t *sel,
                   int64_t kout, int64_t append, int64_t n_glob, int64_t *idx_out, void *stream)
{
    if (!records || !idx_out || rows < 0 || k < 1 || kout < 0 || append < 0 || offset < 0 || (kout > 0 && !sel)) return FASTKV_EINVAL;
    if (rows > 65535) return FASTKV_EUNSUPPORTED;
    if (rows == 0 || kout + append == 0) return FASTKV_OK;
    ProfScope ps_(K_SP_AUX, (hipStream_t)stream);
    hipLaunchKernelGGL(sp_pick_kernel, dim3((unsigned)((kout + append + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       (const uint64_t *)records, rank_stride, offset, k, sel, kout, append, n_glob, idx_out);
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

int fastkv_sp_compact_f16(int32_t B, int32_t Hkv, int32_t S_r, int32_t D, int32_t window, int32_t capacity, const void *k,
                          const int64_t k_strides[4], const void *v, const int64_t v_strides[4], const int64_t *kv_idx, int64_t pos0,
                          int32_t window_owner, void *k_out, void *v_out, void *stream)
{
    if (B < 1 || Hkv < 1 || S_r < 1 || window < 0 || capacity <= window || pos0 < 0) return FASTKV_EINVAL;
    if (D != 64 && D != 128 && D != 256) return FASTKV_EUNSUPPORTED;
    if (window_owner && S_r < window) return FASTKV_EINVAL;
    if (!k || !v || !kv_idx || !k_out || !v_out || !k_strides || !v_strides || k_strides[3] != 1 || v_strides[3] != 1) return FASTKV_EINVAL;
    if (((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_out | (uintptr_t)v_out) & 15) return FASTKV_EINVAL;
    for (int i = 0; i < 3; ++i) if ((k_strides[i] & 7) || (v_strides[i] & 7)) return FASTKV_EINVAL;
    if ((int64_t)B * Hkv > 65535) return FASTKV_EUNSUPPORTED;
    const int lpr = D / 8, rpb = 256 / lpr;
    dim3 grid((capacity + rpb - 1) / rpb, B * Hkv);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps_(K_COMPACT, st);
#define FK_SPC(LPRV)                                                                                                             \
    hipLaunchKernelGGL((sp_compact_kernel<LPRV>), grid, dim3(256), 0, st, (const uint16_t *)k, k_strides[0], k_strides[1],        \
                       k_strides[2], (const uint16_t *)v, v_strides[0], v_strides[1], v_strides[2], kv_idx, Hkv, S_r, pos0,       \
                       window, capacity, window_owner, (uint16_t *)k_out, (uint16_t *)v_out)
    if (lpr == 8) FK_SPC(8);
    else if (lpr == 16) FK_SPC(16);
    else FK_SPC(32);
#undef FK_SPC
    return hipGetLastError() == hipSuccess ? FASTKV_OK : FASTKV_ELAUNCH;
}

}  // extern "C"
