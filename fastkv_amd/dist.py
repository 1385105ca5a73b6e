"""Sharded FastKV hot path.  sp_update_kv: ONE prompt split over P ranks on the sequence axis (BASELINE.json configs[2]).
tp_update_kv (end of file): the KV heads split over P ranks (tensor parallel, configs[4]).

The reference has no distributed code; the contract here is "same result as one GPU, bit for bit".
That is possible because the arithmetic contract makes every global quantity exact:

    position-local        logits, scaled/masked logits, probabilities, pooled scores (needs `kernel//2` halo positions)
    global per (b,h,row)  softmax max  -> all-reduce MAX            [B*H*W floats + NaN flags]
                          softmax sum  -> all-reduce SUM of int64   [2^-40 fixed point: exact, order-free]
    global per score row  top-k        -> local canonical top-k, ONE all-gather of (score, position) candidates,
                                          final canonical top-k over the P*k candidates (a global winner is always a
                                          local winner of its shard, ties included)

Collectives per layer (all tiny, latency bound on xGMI -- use direct all-gather / all-reduce, never rings of big
buffers): (1) all-gather of the window queries + K halo rows (a few KiB), (2) MAX, (3) SUM, (4) the candidate
all-gather named in the north star (k*8 bytes per score row per rank), and optionally (5) a SUM all-reduce that
replicates the compacted K/V rows (each row has exactly one non-zero contributor, so the fp16 sum is exact).

Rank r owns positions [pos0_r, pos0_r + S_r); the last rank owns the window (the prompt's last W positions).
The local compute goes through a small `LocalOps` interface: `HipLocalOps` (the product: C-ABI `fastkv_sp_*` stages and
the regular select / compact kernels) or, in the CPU tests, an oracle-backed stand-in injected by the test.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

NEG_INF_F16_BITS = 0xFC00


# ------------------------------------------------------------------------------------------------- local compute
class HipLocalOps:
    """Rank-local stages on the MI355X through the C ABI (include/fastkv_hip.h, `fastkv_sp_*`)."""

    def __init__(self):
        from . import ops
        from ._lib import Problem, SPWindow, check, load
        self.ops, self.Problem, self.SPWindow, self.check, self.lib = ops, Problem, SPWindow, check, load()

    def _problem(self, B, H, Hkv, S, D, window, kernel_size, pooling):
        return self.Problem(B=B, H=H, Hkv=Hkv, S=S, D=D, window=window, kernel=kernel_size, pooling=self.ops.POOLING[pooling],
                            capacity=S, tsp_len=0, order=0, reserved=self.ops._engine)

    def _ws(self, p, dev):
        return self.ops._workspace(self.lib.fastkv_sp_workspace_bytes(ctypes.byref(p)), dev, "scratch")

    def logits(self, q_win, k, logits_ext, col_off, window, kernel_size, pooling):
        B, H, W, D = q_win.shape
        Hkv, S = k.shape[1], k.shape[2]
        p = self._problem(B, H, Hkv, S, D, window, kernel_size, pooling)
        ws = self._ws(p, k.device)
        rc = self.lib.fastkv_sp_logits_f16(ctypes.byref(p), q_win.data_ptr(), self.ops._strides(q_win), k.data_ptr(),
                                           self.ops._strides(k), logits_ext.data_ptr(), logits_ext.shape[-1], col_off,
                                           ws.data_ptr(), ws.numel(), self.ops._stream())
        self.check(rc, "sp_logits")

    def _win(self, win):
        return self.SPWindow(*win)

    def rowmax(self, logits_ext, win, Hkv, D, window, kernel_size, pooling):
        B, H, W, _ = logits_ext.shape
        p = self._problem(B, H, Hkv, win[4], D, window, kernel_size, pooling)
        out = torch.empty(2 * B * H * W, dtype=torch.float32, device=logits_ext.device)
        w = self._win(win)
        self.check(self.lib.fastkv_sp_rowmax_f16(ctypes.byref(p), logits_ext.data_ptr(), ctypes.byref(w), out.data_ptr(),
                                                 self.ops._stream()), "sp_rowmax")
        return out

    def rowsum(self, logits_ext, win, gmax, Hkv, D, window, kernel_size, pooling):
        B, H, W, _ = logits_ext.shape
        p = self._problem(B, H, Hkv, win[4], D, window, kernel_size, pooling)
        out = torch.empty(B * H * W, dtype=torch.int64, device=logits_ext.device)
        w = self._win(win)
        self.check(self.lib.fastkv_sp_rowsum_f16(ctypes.byref(p), logits_ext.data_ptr(), ctypes.byref(w), gmax.data_ptr(),
                                                 out.data_ptr(), self.ops._stream()), "sp_rowsum")
        return out

    def scores(self, logits_ext, win, gmax, gsum, n_own, want_tsp, Hkv, D, window, kernel_size, pooling):
        B, H, W, _ = logits_ext.shape
        p = self._problem(B, H, Hkv, win[4], D, window, kernel_size, pooling)
        dev = logits_ext.device
        c = torch.empty(B, Hkv, max(n_own, 0), dtype=torch.float16, device=dev)
        t = torch.empty(B, max(n_own, 0), dtype=torch.float16, device=dev) if want_tsp else None
        ws = self._ws(p, dev)
        w = self._win(win)
        rc = self.lib.fastkv_sp_scores_f16(ctypes.byref(p), logits_ext.data_ptr(), ctypes.byref(w), gmax.data_ptr(), gsum.data_ptr(),
                                           c.data_ptr(), t.data_ptr() if t is not None else None, ws.data_ptr(), ws.numel(),
                                           self.ops._stream())
        self.check(rc, "sp_scores")
        return c, t

    def select(self, rows2d, k, order="index"):
        return self.ops.select(rows2d.contiguous(), k, order)

    def compact(self, k, v, idx, window):
        return self.ops.compact(k, v, idx, window)


# ------------------------------------------------------------------------------------------------- helpers
def _f16_bits(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF


def _pack(scores: torch.Tensor, gidx: torch.Tensor) -> torch.Tensor:
    """(fp16 score, global position) -> one int64 per candidate: score bits << 32 | position."""
    return (_f16_bits(scores) << 32) | (gidx & 0xFFFFFFFF)


def _unpack(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    bits = ((packed >> 32) & 0xFFFF).to(torch.int32)
    scores = torch.where(bits >= 0x8000, bits - 0x10000, bits).to(torch.int16).view(torch.float16)
    return scores, (packed & 0xFFFFFFFF)


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo has no device transport for ROCm tensors: stage through the host (used by the 2-process single-GPU test)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(t: torch.Tensor, group) -> List[torch.Tensor]:
    src = t.contiguous().cpu() if _host_staged(t, group) else t.contiguous()
    out = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, src, group=group)
    return [o.to(t.device) for o in out] if src is not t and src.device != t.device else out


def _all_reduce(t: torch.Tensor, op, group) -> None:
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


# ------------------------------------------------------------------------------------------------- the operator
def sp_update_kv(key_states: torch.Tensor, query_states: torch.Tensor, value_states: torch.Tensor, *, window_size: int,
                 kernel_size: int, pooling: str, capacity: int, tsp_len: int = 0, order: str = "score", group=None,
                 local_ops=None, shard_lengths: Optional[List[int]] = None, replicate: bool = True):
    """Sequence-sharded compress branch of FastKVCluster.update_kv (/root/reference/baselines/fastkv/utils.py:93-132).

    key/value_states [B,Hkv,S_r,D], query_states [B,H,S_r,D]: this rank's slice of the prompt (ranks in sequence order).
    `capacity` / `tsp_len` are GLOBAL (utils.py:86-87, :123-126 applied to the global length by the caller).
    Returns (k_out [B,Hkv,capacity,D], v_out, tsp_idx [B,tsp_len] int64 | None, kv_idx [B,Hkv,capacity-W] int64):
    identical on every rank and bit-identical to the single-GPU operator when `replicate`; with `replicate=False`
    k_out / v_out hold only the rows this rank owns (zeros elsewhere) and no K/V bytes cross the fabric.
    """
    lo = local_ops or HipLocalOps()
    P, r = dist.get_world_size(group), dist.get_rank(group)
    B, Hkv, S_r, D = key_states.shape
    H = query_states.shape[1]
    W, pad = window_size, kernel_size // 2
    dev = key_states.device
    if pooling not in ("avgpool", "maxpool"):
        raise ValueError("Pooling method not supported")
    if shard_lengths is None:
        lens = _all_gather(torch.tensor([S_r], dtype=torch.int64, device=dev), group)
        shard_lengths = [int(x.item()) for x in lens]
    assert shard_lengths[r] == S_r and min(shard_lengths) >= max(pad, 1) and shard_lengths[-1] >= W + pad, \
        "every shard needs >= kernel//2 positions and the last one the whole window"
    pos0 = sum(shard_lengths[:r])
    S = sum(shard_lengths)
    n = S - W
    kk = capacity - W
    assert W < capacity <= S and (tsp_len == 0 or W < tsp_len < S)
    common = dict(window=W, kernel_size=kernel_size, pooling=pooling)

    # (1) window queries (owned by the last rank) + K halo rows of every rank: one small all-gather
    nq, nk = B * H * W * D, B * Hkv * pad * D
    packet = torch.cat([query_states[:, :, S_r - W:, :].reshape(-1) if S_r >= W else torch.zeros(nq, dtype=torch.float16, device=dev),
                        key_states[:, :, :pad, :].reshape(-1), key_states[:, :, S_r - pad:, :].reshape(-1)])
    packets = _all_gather(packet, group)
    q_win = packets[P - 1][:nq].view(B, H, W, D)
    left = packets[r - 1][nq + nk:].view(B, Hkv, pad, D) if r > 0 and pad else None
    right = packets[r + 1][nq:nq + nk].view(B, Hkv, pad, D) if r < P - 1 and pad else None

    # (2) local logits with halo columns: column x <-> global position pos0 - pad + x
    ncols = S_r + 2 * pad
    Sp = (ncols + 7) // 8 * 8
    logits = torch.zeros(B, H, W, Sp, dtype=torch.float16, device=dev)
    lo.logits(q_win, key_states, logits, pad, **common)
    if left is not None:
        lo.logits(q_win, left, logits, 0, **common)
    if right is not None:
        lo.logits(q_win, right, logits, pad + S_r, **common)
    win = (ncols, pos0 - pad, pad, pad + S_r, S, Sp)

    # (3) + (4) global softmax statistics: MAX (with NaN flags), then the exact fixed-point SUM
    gmax = lo.rowmax(logits, win, Hkv, D, **common)
    _all_reduce(gmax, dist.ReduceOp.MAX, group)
    gsum = lo.rowsum(logits, win, gmax, Hkv, D, **common)
    _all_reduce(gsum, dist.ReduceOp.SUM, group)

    # (5) scores of the owned candidate positions
    n_own = max(0, min(S_r, n - pos0))
    c_loc, t_loc = lo.scores(logits, win, gmax, gsum, n_own, tsp_len > 0, Hkv, D, **common)

    # (6) local canonical top-k -> the candidate all-gather -> final canonical top-k
    def global_topk(rows2d: torch.Tensor, k: int, final_order: str) -> torch.Tensor:
        nrows = rows2d.shape[0]
        kl = min(k, n_own)
        cand = torch.full((nrows, k), (NEG_INF_F16_BITS << 32) | 0xFFFFFFFF, dtype=torch.int64, device=dev)
        if kl > 0:
            li = lo.select(rows2d, kl, "index")
            cand[:, :kl] = _pack(torch.gather(rows2d, 1, li), li + pos0)
        allc = torch.cat(_all_gather(cand, group), dim=1)            # [rows, P*k], ascending global position per rank block
        sc, gi = _unpack(allc)
        sel = lo.select(sc, k, final_order)
        return torch.gather(gi, 1, sel)

    kv_idx = global_topk(c_loc.reshape(B * Hkv, n_own), kk, order).view(B, Hkv, kk)
    tsp_idx = None
    if tsp_len:
        t_sel = global_topk(t_loc.reshape(B, n_own), tsp_len - W, "index")
        tsp_idx = torch.cat([t_sel, torch.arange(n, S, device=dev, dtype=torch.int64).expand(B, -1)], dim=1)   # utils.py:128-130

    # (7) compaction of the rows this rank owns (+ the window rows on the last rank), optional replication
    own = (kv_idx >= pos0) & (kv_idx < pos0 + S_r)
    li = torch.where(own, kv_idx - pos0, torch.zeros_like(kv_idx))
    wl = min(W, S_r)
    ko, vo = lo.compact(key_states, value_states, li.contiguous(), wl)     # [B,Hkv,kk+wl,D]: rows + this shard's last wl rows
    k_out = torch.zeros(B, Hkv, capacity, D, dtype=torch.float16, device=dev)
    v_out = torch.zeros_like(k_out)
    m = own[..., None]
    k_out[:, :, :kk] = torch.where(m, ko[:, :, :kk], torch.zeros((), dtype=torch.float16, device=dev))
    v_out[:, :, :kk] = torch.where(m, vo[:, :, :kk], torch.zeros((), dtype=torch.float16, device=dev))
    if r == P - 1:
        k_out[:, :, kk:] = ko[:, :, kk + wl - W:]
        v_out[:, :, kk:] = vo[:, :, kk + wl - W:]
    if replicate:
        # exactly one non-zero contributor per element; summed as int32 words (two fp16 each) so that even -0.0 keeps its bits
        both = torch.stack([k_out, v_out])
        _all_reduce(both.view(torch.int32), dist.ReduceOp.SUM, group)
        k_out, v_out = both[0], both[1]
    return k_out, v_out, tsp_idx, kv_idx


# ------------------------------------------------------------------------------------------------- tensor parallel
class HipTPOps:
    """Rank-local stages of tp_update_kv on the MI355X (the whole operator on the local heads + the head sum)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def update_kv_local(self, q, k, v, window, kernel_size, pooling, capacity, order):
        ko, vo, _, kv_idx, c = self.ops.update_kv(q, k, v, window, kernel_size, pooling, capacity, 0, order, return_indices=True,
                                                   return_scores=True)
        return ko, vo, kv_idx, c

    def head_sum(self, c_all):
        return self.ops.head_sum(c_all)

    def select_tsp(self, t, k, window):
        return self.ops.select(t, k, "index", append=window)


def tp_update_kv(key_states: torch.Tensor, query_states: torch.Tensor, value_states: torch.Tensor, *, window_size: int,
                 kernel_size: int, pooling: str, capacity: int, tsp_len: int = 0, order: str = "score", group=None,
                 local_ops=None):
    """Head-sharded (tensor-parallel) compress branch of FastKVCluster.update_kv (utils.py:93-132), e.g. Llama-3-70B over
    8 ranks: rank r holds KV heads [r*Hkv/P, (r+1)*Hkv/P) and their query heads, the whole sequence.

    Scoring, per-KV-head top-k and the K/V gather are local to the heads; only the TSP selection sums
    over ALL KV heads (utils.py:127).  ONE all-gather of the ranks' fp16 score rows [B,Hkv_local,n] (64 KiB per head at
    32k), then every rank adds the P*Hkv_local rows in head order (fp32 accumulate, one rounding: what the reference's
    `sum(dim=-2)` does) and runs the same canonical selection, so tsp_idx is identical on every rank and equal to the
    single-GPU result bit for bit.  Returns (k_out, v_out, tsp_idx | None, kv_idx) for the local heads."""
    lo = local_ops or HipTPOps()
    B, Hkv_l, S, D = key_states.shape
    W = window_size
    assert W < capacity <= S and (tsp_len == 0 or W < tsp_len < S)
    ko, vo, kv_idx, c = lo.update_kv_local(query_states, key_states, value_states, W, kernel_size, pooling, capacity, order)
    tsp = None
    if tsp_len:
        parts = _all_gather(c.contiguous(), group)                       # [B,Hkv_l,n] per rank, rank order = head order
        c_all = torch.cat(parts, dim=1).contiguous()                     # [B,Hkv,n]
        t = lo.head_sum(c_all)
        tsp = lo.select_tsp(t, tsp_len - W, W)
    return ko, vo, tsp, kv_idx
