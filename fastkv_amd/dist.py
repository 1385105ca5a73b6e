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

FOUR collectives per layer, all tiny and latency bound on xGMI (direct all-gather / all-reduce of a few KiB, never a ring
of big buffers), every layer the same four (the TSP layer adds none, unknown shard lengths add none):
    (1) all-gather   shard length + window queries (held by the last rank) + the K halo rows of every rank
    (2) all-reduce   MAX of the row maxima with their NaN flags
    (3) all-reduce   SUM of the fixed-point row sums
    (4) all-gather   candidate records of the per-head rows AND of the TSP row (the index all-gather of the north star)
Each needs the global result of the one before (queries -> max -> sum -> scores -> selection), and the exact sum
sum_j fix(exp(x_j - max)) cannot be rebuilt from partial sums taken against local maxima, so (2) and (3) do not fold into one
message the way a rescaling online softmax would (SURVEY.md 8(e) budgeted 3 under that assumption).
No K/V bytes cross the fabric: every rank keeps the selected rows it owns (`replicate=False`, the layout a sequence-parallel
decode consumes); `replicate=True` (tests, single-GPU consumers) adds (5), an exact all-reduce of the compacted rows.

Rank r owns positions [pos0_r, pos0_r + S_r); the last rank owns the window (the prompt's last W positions).
The local compute goes through a small `LocalOps` interface: `HipLocalOps` (the product: C-ABI `fastkv_sp_*` stages and
the regular select kernels) or, in the CPU tests, an oracle-backed stand-in injected by the test.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

NEG_INF_F16_BITS = 0xFC00
SP_PAD_RECORD = (NEG_INF_F16_BITS << 32) | 0xFFFFFFFF          # csrc/sp.hip: SP_PAD

# collectives issued by this module since import (tests assert the per-layer count)
COLLECTIVES = {"all_gather": 0, "all_reduce": 0}


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

    def local_candidates(self, rows2d, k, pos0, records):
        """records [rows, k] int64 (a view of the send buffer) <- the shard's canonical top-min(k, n_own) of every fp16 row as
        {score bits << 32 | global position}, ascending position, padded with {-inf, no position}."""
        rows, n_own = rows2d.shape
        kl = min(k, n_own)
        li = self.ops.select(rows2d, kl, "index") if kl > 0 else None
        self.check(self.lib.fastkv_sp_pack_f16(rows2d.data_ptr() if kl else None, rows, rows2d.stride(0) if kl else 0,
                                               li.data_ptr() if kl else None, kl, k, pos0, records.data_ptr(), self.ops._stream()),
                   "sp_pack")

    def merge_candidates(self, allc, offset, rows, k, order, append=0, n_glob=0):
        """allc [P, L] int64: every rank's records; this row set starts at `offset` of each block.  Final canonical top-k over
        the P*k candidates of each row -> global positions [rows, k + append] (`append` window positions n_glob.. after them)."""
        P = allc.shape[0]
        dev = allc.device
        sc = torch.empty(rows, P * k, dtype=torch.float16, device=dev)
        self.check(self.lib.fastkv_sp_unpack_f16(allc.data_ptr(), allc.stride(0), offset, P, rows, k, sc.data_ptr(), sc.stride(0),
                                                 self.ops._stream()), "sp_unpack")
        sel = self.ops.select(sc, k, order)
        out = torch.empty(rows, k + append, dtype=torch.int64, device=dev)
        self.check(self.lib.fastkv_sp_pick(allc.data_ptr(), allc.stride(0), offset, rows, k, sel.data_ptr(), k, append, n_glob,
                                           out.data_ptr(), self.ops._stream()), "sp_pick")
        return out

    def compact_owned(self, k, v, kv_idx, pos0, window, capacity, window_owner):
        """[B,Hkv,capacity,D] x2: the rows of kv_idx (global positions) inside this shard (+ the window rows when
        `window_owner`), zeros in the slots other ranks own."""
        B, Hkv, S_r, D = k.shape
        ko = torch.empty(B, Hkv, capacity, D, dtype=torch.float16, device=k.device)
        vo = torch.empty_like(ko)
        rc = self.lib.fastkv_sp_compact_f16(B, Hkv, S_r, D, window, capacity, k.data_ptr(), self.ops._strides(k), v.data_ptr(),
                                            self.ops._strides(v), kv_idx.data_ptr(), pos0, 1 if window_owner else 0, ko.data_ptr(),
                                            vo.data_ptr(), self.ops._stream())
        self.check(rc, "sp_compact")
        return ko, vo


# ------------------------------------------------------------------------------------------------- helpers
def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo has no device transport for ROCm tensors: stage through the host (used by the shared-GPU tests)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(t: torch.Tensor, group) -> torch.Tensor:
    """[L] -> [P, L] (rank order), one collective."""
    COLLECTIVES["all_gather"] += 1
    P = dist.get_world_size(group)
    src = t.contiguous()
    if _host_staged(t, group):
        src = src.cpu()
    out = torch.empty(P * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.view(P, -1).to(t.device)


def _all_reduce(t: torch.Tensor, op, group) -> None:
    COLLECTIVES["all_reduce"] += 1
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def check_shards(shard_lengths: List[int], window: int, kernel_size: int) -> None:
    """What the sharded operator needs from the split -- evaluated on the same list by every rank, so that all ranks fail
    (or go on) together and nobody is left waiting in a collective."""
    pad = kernel_size // 2
    if min(shard_lengths) < max(pad, 1) or shard_lengths[-1] < window + pad:
        raise ValueError(f"sp_update_kv: every shard needs >= kernel//2 = {pad} positions and the last one the whole window "
                         f"plus the halo (>= {window + pad}); got {shard_lengths}")


# ------------------------------------------------------------------------------------------------- the operator
def sp_update_kv(key_states: torch.Tensor, query_states: torch.Tensor, value_states: torch.Tensor, *, window_size: int,
                 kernel_size: int, pooling: str, capacity: int, tsp_len: int = 0, order: str = "score", group=None,
                 local_ops=None, shard_lengths: Optional[List[int]] = None, replicate: bool = False):
    """Sequence-sharded compress branch of FastKVCluster.update_kv (/root/reference/baselines/fastkv/utils.py:93-132).

    key/value_states [B,Hkv,S_r,D], query_states [B,H,S_r,D]: this rank's slice of the prompt (ranks in sequence order).
    `capacity` / `tsp_len` are GLOBAL (utils.py:86-87, :123-126 applied to the global length by the caller).
    Returns (k_out [B,Hkv,capacity,D], v_out, tsp_idx [B,tsp_len] int64 | None, kv_idx [B,Hkv,capacity-W] int64).
    kv_idx / tsp_idx are identical on every rank and equal to the single-GPU operator's.  k_out / v_out hold the selected
    rows THIS rank owns (zeros in the other slots; the ranks' tensors add up to the single-GPU result) -- no K/V bytes cross
    the fabric; `replicate=True` adds one all-reduce that makes them identical on every rank.
    Four collectives per call (module docstring), five with `replicate`.
    """
    lo = local_ops or HipLocalOps()
    P, r = dist.get_world_size(group), dist.get_rank(group)
    B, Hkv, S_r, D = key_states.shape
    H = query_states.shape[1]
    W, pad = window_size, kernel_size // 2
    dev = key_states.device
    if pooling not in ("avgpool", "maxpool"):
        raise ValueError("Pooling method not supported")
    if shard_lengths is not None:
        check_shards(shard_lengths, W, kernel_size)                  # before the first collective, same verdict on every rank
        assert shard_lengths[r] == S_r
    common = dict(window=W, kernel_size=kernel_size, pooling=pooling)

    # (1) shard length + window queries (owned by the last rank) + K halo rows of every rank: one small all-gather
    nq, nk = B * H * W * D, B * Hkv * pad * D
    HDR = 8                                                          # 16 bytes: the pieces behind it stay 16-B aligned
    packet = torch.zeros(HDR + nq + 2 * nk, dtype=torch.float16, device=dev)
    packet[:4].view(torch.int64).fill_(S_r)
    o_q, o_l, o_r = HDR, HDR + nq, HDR + nq + nk                     # window queries | first `pad` K rows | last `pad` K rows
    if S_r >= W:
        packet[o_q:o_l].view(B, H, W, D).copy_(query_states[:, :, S_r - W:, :])
    hl = min(pad, S_r)                                               # a shard shorter than the halo fails check_shards below
    if hl:
        packet[o_l:o_r].view(B, Hkv, pad, D)[:, :, :hl].copy_(key_states[:, :, :hl, :])
        packet[o_r:].view(B, Hkv, pad, D)[:, :, pad - hl:].copy_(key_states[:, :, S_r - hl:, :])
    packets = _all_gather(packet, group)
    if shard_lengths is None:
        shard_lengths = [int(x) for x in packets[:, :4].contiguous().view(torch.int64).view(-1).tolist()]
        check_shards(shard_lengths, W, kernel_size)                  # every rank sees the same list
    pos0 = sum(shard_lengths[:r])
    S = sum(shard_lengths)
    n = S - W
    kk = capacity - W
    assert W < capacity <= S and (tsp_len == 0 or W < tsp_len < S)
    q_win = packets[P - 1, o_q:o_l].view(B, H, W, D)
    left = packets[r - 1, o_r:].view(B, Hkv, pad, D) if r > 0 and pad else None               # the previous rank's last rows
    right = packets[r + 1, o_l:o_r].view(B, Hkv, pad, D) if r < P - 1 and pad else None       # the next rank's first rows

    # local logits with halo columns: column x <-> global position pos0 - pad + x
    ncols = S_r + 2 * pad
    Sp = (ncols + 7) // 8 * 8
    logits = torch.zeros(B, H, W, Sp, dtype=torch.float16, device=dev)
    lo.logits(q_win, key_states, logits, pad, **common)
    if left is not None:
        lo.logits(q_win, left, logits, 0, **common)
    if right is not None:
        lo.logits(q_win, right, logits, pad + S_r, **common)
    win = (ncols, pos0 - pad, pad, pad + S_r, S, Sp)

    # (2) + (3) global softmax statistics: MAX (with NaN flags), then the exact fixed-point SUM
    gmax = lo.rowmax(logits, win, Hkv, D, **common)
    _all_reduce(gmax, dist.ReduceOp.MAX, group)
    gsum = lo.rowsum(logits, win, gmax, Hkv, D, **common)
    _all_reduce(gsum, dist.ReduceOp.SUM, group)

    # scores of the owned candidate positions
    n_own = max(0, min(S_r, n - pos0))
    c_loc, t_loc = lo.scores(logits, win, gmax, gsum, n_own, tsp_len > 0, Hkv, D, **common)

    # (4) local canonical top-k of every row -> ONE all-gather of the candidate records -> final canonical top-k
    kt = tsp_len - W if tsp_len else 0
    n_kv = B * Hkv * kk
    cand = torch.empty(n_kv + B * kt, dtype=torch.int64, device=dev)
    lo.local_candidates(c_loc.view(B * Hkv, n_own), kk, pos0, cand[:n_kv].view(B * Hkv, kk))
    if tsp_len:
        lo.local_candidates(t_loc.view(B, n_own), kt, pos0, cand[n_kv:].view(B, kt))
    allc = _all_gather(cand, group)                                  # [P, n_kv + B*kt]
    kv_idx = lo.merge_candidates(allc, 0, B * Hkv, kk, order).view(B, Hkv, kk)
    tsp_idx = lo.merge_candidates(allc, n_kv, B, kt, "index", append=W, n_glob=n) if tsp_len else None     # utils.py:127-130

    # compaction of the rows this rank owns (+ the window rows on the last rank); (5) optional replication
    k_out, v_out = lo.compact_owned(key_states, value_states, kv_idx, pos0, W, capacity, r == P - 1)
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

    def update_kv_local(self, q, k, v, window, kernel_size, pooling, capacity, order, want_scores=True):
        out = self.ops.update_kv(q, k, v, window, kernel_size, pooling, capacity, 0, order, return_indices=True,
                                 return_scores=want_scores)                # (the score rows only leave the workspace on the TSP layer)
        return out[0], out[1], out[3], (out[4] if want_scores else None)

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
    try:
        ko, vo, kv_idx, c = lo.update_kv_local(query_states, key_states, value_states, W, kernel_size, pooling, capacity, order,
                                               want_scores=bool(tsp_len))
    except TypeError:                                             # (a LocalOps stand-in without the keyword: tests)
        ko, vo, kv_idx, c = lo.update_kv_local(query_states, key_states, value_states, W, kernel_size, pooling, capacity, order)
    tsp = None
    if tsp_len:
        P = dist.get_world_size(group)
        n = c.shape[2]
        parts = _all_gather(c.contiguous().view(-1), group).view(P, B, Hkv_l, n)   # rank order = head order
        c_all = parts.permute(1, 0, 2, 3).reshape(B, P * Hkv_l, n)       # [B,Hkv,n] (a view when B == 1)
        c_all = c_all.contiguous()
        t = lo.head_sum(c_all)
        tsp = lo.select_tsp(t, tsp_len - W, W)
    return ko, vo, tsp, kv_idx
