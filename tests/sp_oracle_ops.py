"""Oracle-backed rank-local stages for the CPU (gloo) tests of fastkv_amd/dist.py -- test infrastructure only.

Implements the `LocalOps` interface of fastkv_amd.dist with the sequence-sharded twins in oracle/fastkv_oracle.c, so the
distributed control flow (halo exchange, MAX / SUM all-reduces, candidate all-gather, ownership masks) can be checked on
machines without a GPU against the single-process oracle."""
import ctypes

import torch

from oracle import fastkv_oracle as O

SP_PAD_RECORD = (0xFC00 << 32) | 0xFFFFFFFF      # csrc/sp.hip: SP_PAD


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

    # ---- twins of csrc/sp.hip (index arithmetic only; the semantics the HIP stages are tested against on the GPU)
    def local_candidates(self, rows2d, k, pos0, records):
        rows, n_own = rows2d.shape
        kl = min(k, n_own)
        records.fill_(SP_PAD_RECORD)
        if kl > 0:
            li = self.select(rows2d, kl, "index")
            bits = torch.gather(rows2d, 1, li).contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
            records[:, :kl] = (bits << 32) | ((li + pos0) & 0xFFFFFFFF)

    def merge_candidates(self, allc, offset, rows, k, order, append=0, n_glob=0):
        P = allc.shape[0]
        rec = allc[:, offset:offset + rows * k].view(P, rows, k).permute(1, 0, 2).reshape(rows, P * k)     # rank-major candidate list
        bits = ((rec >> 32) & 0xFFFF).to(torch.int32)
        sc = torch.where(bits >= 0x8000, bits - 0x10000, bits).to(torch.int16).view(torch.float16)
        sel = self.select(sc, k, order)
        pos = torch.gather(rec & 0xFFFFFFFF, 1, sel)
        if append:
            pos = torch.cat([pos, torch.arange(n_glob, n_glob + append, dtype=torch.int64).expand(rows, -1)], dim=1)
        return pos

    def compact_owned(self, k, v, kv_idx, pos0, window, capacity, window_owner):
        B, Hkv, S_r, D = k.shape
        kk = capacity - window
        own = (kv_idx >= pos0) & (kv_idx < pos0 + S_r)
        li = torch.where(own, kv_idx - pos0, torch.zeros_like(kv_idx))[..., None].expand(-1, -1, -1, D)
        outs = []
        for t in (k, v):
            o = torch.zeros(B, Hkv, capacity, D, dtype=t.dtype)
            o[:, :, :kk] = torch.where(own[..., None], torch.gather(t, 2, li), torch.zeros((), dtype=t.dtype))
            if window_owner:
                o[:, :, kk:] = t[:, :, S_r - window:]
            outs.append(o)
        return outs[0], outs[1]


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
