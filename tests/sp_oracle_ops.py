"""Oracle-backed rank-local stages for the CPU (gloo) tests of fastkv_amd/dist.py -- test infrastructure only.

Implements the `LocalOps` interface of fastkv_amd.dist with the sequence-sharded twins in oracle/fastkv_oracle.c, so the
distributed control flow (halo exchange, MAX / SUM all-reduces, candidate all-gather, ownership masks) can be checked on
machines without a GPU against the single-process oracle."""
import ctypes

import torch

from oracle import fastkv_oracle as O


def _i64x4(t):
    return (ctypes.c_int64 * 4)(*t.stride())


class OracleLocalOps:
    def __init__(self):
        L = O.lib()
        vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.fastkv_oracle_sp_logits.argtypes = [vp, i64p, vp, i64p] + [ci] * 6 + [vp, i64, i64]
        L.fastkv_oracle_sp_rowmax.argtypes = [vp] + [ci] * 8 + [i64, vp]
        L.fastkv_oracle_sp_rowsum.argtypes = [vp, ci, ci, ci, i64, vp, vp]
        L.fastkv_oracle_sp_scores.argtypes = [vp] + [ci] * 11 + [i64, vp, vp, vp, vp]
        self.L = L

    def logits(self, q_win, k, logits_ext, col_off, window, kernel_size, pooling):
        B, H, W, D = q_win.shape
        q_win, k = q_win.contiguous(), k if k.stride(3) == 1 else k.contiguous()
        rc = self.L.fastkv_oracle_sp_logits(q_win.data_ptr(), _i64x4(q_win), k.data_ptr(), _i64x4(k), B, H, k.shape[1], k.shape[2], D,
                                            W, logits_ext.data_ptr(), logits_ext.shape[-1], col_off)
        assert rc == 0

    def rowmax(self, logits_ext, win, Hkv, D, window, kernel_size, pooling):
        B, H, W, Sp = logits_ext.shape
        out = torch.empty(2 * B * H * W, dtype=torch.float32)
        assert self.L.fastkv_oracle_sp_rowmax(logits_ext.data_ptr(), B * H * W, W, D, win[0], win[1], win[2], win[3], win[4], Sp,
                                              out.data_ptr()) == 0
        return out

    def rowsum(self, logits_ext, win, gmax, Hkv, D, window, kernel_size, pooling):
        B, H, W, Sp = logits_ext.shape
        out = torch.empty(B * H * W, dtype=torch.int64)
        assert self.L.fastkv_oracle_sp_rowsum(logits_ext.data_ptr(), B * H * W, win[2], win[3], Sp, gmax.data_ptr(), out.data_ptr()) == 0
        return out

    def scores(self, logits_ext, win, gmax, gsum, n_own, want_tsp, Hkv, D, window, kernel_size, pooling):
        B, H, W, Sp = logits_ext.shape
        c = torch.empty(B, Hkv, max(n_own, 0), dtype=torch.float16)
        t = torch.empty(B, max(n_own, 0), dtype=torch.float16) if want_tsp else None
        rc = self.L.fastkv_oracle_sp_scores(logits_ext.data_ptr(), B, H, Hkv, W, kernel_size, O.POOLING[pooling], win[0], win[1], win[2],
                                            win[3], win[4], Sp, gmax.data_ptr(), gsum.data_ptr(), c.data_ptr(),
                                            t.data_ptr() if t is not None else None)
        assert rc == 0
        return c, t

    def select(self, rows2d, k, order="index"):
        rows2d = rows2d.contiguous()
        return torch.stack([O.canonical_topk(rows2d[i], k, order) for i in range(rows2d.shape[0])]) if rows2d.shape[0] else \
            torch.empty(0, k, dtype=torch.int64)

    def compact(self, k, v, idx, window):
        B, Hkv, S, D = k.shape
        sel = idx[..., None].expand(-1, -1, -1, D)
        ko = torch.cat([torch.gather(k, 2, sel), k[:, :, S - window:]], dim=2)
        vo = torch.cat([torch.gather(v, 2, sel), v[:, :, S - window:]], dim=2)
        return ko.contiguous(), vo.contiguous()


class OracleTPOps:
    """Oracle-backed stand-in for fastkv_amd.dist.HipTPOps (CPU tests of tp_update_kv)."""

    def update_kv_local(self, q, k, v, window, kernel_size, pooling, capacity, order):
        ko, vo, idx, _, c, _ = O.update_kv(q, k, v, window, kernel_size, pooling, capacity, 0, order, return_scores=True)
        return ko, vo, idx, c

    def head_sum(self, c_all):
        return c_all.float().sum(dim=1).to(torch.float16)               # <= 2^20 addends of 11-bit values: fp32 exact for these sizes

    def select_tsp(self, t, k, window):
        n = t.shape[1]
        rows = [torch.cat([O.canonical_topk(t[b].contiguous(), k, "index"), torch.arange(n, n + window)]) for b in range(t.shape[0])]
        return torch.stack(rows)
