"""torch-tensor front-ends of the C ABI.  torch is used for device memory and the current HIP
stream only; every byte of the hot path is moved/computed by the kernels in csrc/.

All functions are stream-ordered on torch's current stream and never synchronise
(same contract as the reference operator, /root/reference/baselines/fastkv/utils.py:80-134)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from ._lib import FASTKV_EUNSUPPORTED, FastKVNativeError, Problem, check, load, raise_if_aborted  # noqa: F401

POOLING = {"avgpool": 0, "maxpool": 1}
ORDER = {"index": 0, "score": 1}

_ws_cache: dict = {}


def _require_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fastkv_amd: the HIP path needs tensors on a ROCm device (no CPU fallback exists)")


def _strides(t: torch.Tensor):
    return (ctypes.c_int64 * 4)(*t.stride())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _workspace(nbytes: int, device: torch.device, kind: str = "op") -> torch.Tensor:
    """Grow-only per-(device, stream, kind) scratch; reuse is safe because all users are ordered on that stream.

    kind "op": workspace of fastkv_update_kv_f16 / fastkv_score_f16 -- its first 8 KiB are the control block that
    fastkv_workspace_init sets up once per allocation and that only those entry points may touch.
    kind "scratch": plain scratch of the stand-alone select and the sequence-sharded stages (written from byte 0)."""
    key = (device.index, _stream(), kind)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        if kind == "op":
            check(load().fastkv_workspace_init(ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                               ctypes.c_void_p(_stream())), "workspace_init")
        _ws_cache[key] = ws
        if kind == "op" and os.environ.get("FASTKV_SELFTEST", "0") == "1":
            from . import selftest                                  # (once per process; runs on this device with its own workspaces)
            selftest.maybe_run_at_load(device)
    return ws


PLACEMENT_POLICY = {"count": 0, "strict": 1, "failsafe": 2}


def set_placement_policy(name: str) -> None:
    """What a fused scoring launch that did not get its compute-unit pairing leads to (include/fastkv_hip.h,
    fastkv_set_placement_policy): "failsafe" (default) = FASTKV_EPLACEMENT once + the process switches to the no-wait kernels;
    "strict" = the error only; "count" = `fastkv_placement_violations()` only."""
    check(load().fastkv_set_placement_policy(PLACEMENT_POLICY[name]), "set_placement_policy")


def set_fused_rolling(on: bool) -> bool:
    """Turn the rolling launch of the fused scoring kernel on / off (include/fastkv_hip.h fastkv_set_fused_rolling: three or more
    32k-class entries scored by ONE launch, two entries on the chip at a time and out of step); returns the previous setting."""
    return bool(load().fastkv_set_fused_rolling(1 if on else 0))


def no_wait_mode() -> bool:
    """True when the library launches no kernel with an in-launch wait (FASTKV_FUSED=0, or the fail-safe switch)."""
    return bool(load().fastkv_no_wait_mode())


def set_no_wait_mode(on: bool) -> bool:
    """Force / leave the no-wait mode at run time; returns the previous setting of the switch."""
    return bool(load().fastkv_set_no_wait_mode(1 if on else 0))


# contraction engines.  Two arithmetic CONTRACTS (oracle/fastkv_oracle.c, "the contraction"): the fp32 fma chain ("valu", "mfma": the
# fp32 matrix instruction; bit-identical) and "mfma16" (the gfx950 fp16 matrix instruction on the fp16 operands themselves).  "auto" =
# the library's default contract (FASTKV_CONTRACTION=fmaf | mfma16; fmaf unless set)
ENGINE = {"auto": 0, "valu": 1, "mfma": 2, "mfma16": 3}
_engine = ENGINE[os.environ.get("FASTKV_SCORE_ENGINE", "auto")]


def set_score_engine(name: str) -> None:
    """Force the contraction engine of the scoring kernels ("auto" | "valu" | "mfma" | "mfma16"); "valu" and "mfma" are bit-identical
    (the fp32 fma chain), "mfma16" is the other contract."""
    global _engine
    _engine = ENGINE[name]


def _problem(q, k, window, kernel_size, pooling, capacity, tsp_len, order) -> Problem:
    B, H, S, D = q.shape
    if pooling not in POOLING:
        raise ValueError("Pooling method not supported")          # utils.py:110
    return Problem(B=B, H=H, Hkv=k.shape[1], S=S, D=D, window=window, kernel=kernel_size, pooling=POOLING[pooling],
                   capacity=capacity, tsp_len=tsp_len, order=ORDER[order], reserved=_engine)


def _check_qkv(q, k, v=None):
    _require_cuda(q, k, v)
    for t in (q, k, v):
        if t is None:
            continue
        if t.dtype != torch.float16:
            raise TypeError(f"fastkv_amd: fp16 tensors expected, got {t.dtype}")
        if t.dim() != 4 or t.stride(3) != 1:
            raise ValueError("fastkv_amd: expected [B,H,S,D] tensors with unit head_dim stride")
    if k.shape[0] != q.shape[0] or k.shape[2] != q.shape[2] or k.shape[3] != q.shape[3]:
        raise ValueError("fastkv_amd: q/k shape mismatch")


def update_kv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, window: int, kernel_size: int, pooling: str,
              capacity: int, tsp_len: int = 0, order: str = "score", return_indices: bool = False,
              return_scores: bool = False, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, q_window: bool = False):
    """Compress branch of FastKVCluster.update_kv (utils.py:93-132) in one stream-ordered call.

    Returns (k_out [B,Hkv,cap,D], v_out, tsp_idx [B,tsp_len] int64 | None[, kv_idx][, scores]).
    `out` = (k_buf, v_buf): write the compacted rows into these [B,Hkv,cap,D] fp16 views instead of fresh tensors (unit
    head_dim stride, other strides multiples of 8 -- e.g. `slab[:, :, :cap]` of a pre-sized cache slab).
    `q_window`: q is `window_rows(q, window)` ([B,H,window,D] contiguous, the only query rows the operator reads); the library is
    handed an address `S - window` rows in front of it."""
    q_ptr = q.data_ptr()
    if q_window:
        if q.dim() != 4 or q.shape[2] != window or not q.is_contiguous() or q.dtype != torch.float16:
            raise ValueError("fastkv_amd.update_kv(q_window=True): a [B,H,window,D] contiguous fp16 copy of the window rows expected")
        _check_qkv(k, k, v)
        _require_cuda(q)
        if pooling not in POOLING:
            raise ValueError("Pooling method not supported")
        p = Problem(B=q.shape[0], H=q.shape[1], Hkv=k.shape[1], S=k.shape[2], D=q.shape[3], window=window, kernel=kernel_size,
                    pooling=POOLING[pooling], capacity=capacity, tsp_len=tsp_len, order=ORDER[order], reserved=_engine)
        q_ptr -= (k.shape[2] - window) * q.stride(2) * 2          # row S - window + r of "the query tensor" = row r of the copy
    else:
        _check_qkv(q, k, v)
        p = _problem(q, k, window, kernel_size, pooling, capacity, tsp_len, order)
    L = load()
    B, Hkv, D = p.B, p.Hkv, p.D
    dev = q.device
    if out is None:
        ko = torch.empty(B, Hkv, capacity, D, dtype=torch.float16, device=dev)
        vo = torch.empty_like(ko)
    else:
        ko, vo = out
        _require_cuda(ko, vo)
        if tuple(ko.shape) != (B, Hkv, capacity, D) or ko.shape != vo.shape or ko.stride() != vo.stride() or ko.stride(3) != 1 \
                or ko.dtype != torch.float16 or vo.dtype != torch.float16:
            raise ValueError("fastkv_amd: out must be two [B,Hkv,capacity,D] fp16 views with equal strides and unit head_dim stride")
    ostr = (ctypes.c_int64 * 3)(*ko.stride()[:3])
    # zeros, not empty: the ONE output that torch itself consumes as indices (torch.gather on the position ids, llama_model.py:254)
    # -- an abandoned launch (FASTKV_EABORTED) leaves entries unwritten, and a stale index there would be a device-side assert
    # in torch instead of a reported error; position 0 is always valid.  16 KiB memset once per prefill.
    tsp = torch.zeros(B, tsp_len, dtype=torch.int64, device=dev) if tsp_len else None
    kv_idx = torch.empty(B, Hkv, capacity - window, dtype=torch.int64, device=dev) if return_indices else None
    sc = torch.empty(B, Hkv, p.S - window, dtype=torch.float16, device=dev) if return_scores else None
    nbytes = L.fastkv_workspace_bytes(ctypes.byref(p))
    if nbytes == 0:
        check(-1, "workspace_bytes")
    ws = _workspace(nbytes, dev)
    rc = L.fastkv_update_kv_strided_f16(ctypes.byref(p), q_ptr, _strides(q), k.data_ptr(), _strides(k), v.data_ptr(),
                                _strides(v), ko.data_ptr(), vo.data_ptr(), ostr,
                                kv_idx.data_ptr() if kv_idx is not None else None,
                                tsp.data_ptr() if tsp is not None else None,
                                sc.data_ptr() if sc is not None else None, ws.data_ptr(), ws.numel(), _stream())
    check(rc, "update_kv")
    out = [ko, vo, tsp]
    if return_indices:
        out.append(kv_idx)
    if return_scores:
        out.append(sc)
    return tuple(out)


def update_kv_per_query_head(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, window: int, kernel_size: int, pooling: str,
                             capacity: int, order: str = "score", return_indices: bool = False):
    """The selection rule of the SnapKV baseline (/root/reference/baselines/snapkv/utils.py:57-102): the same window scoring,
    pooling and top-k as `update_kv`, but per QUERY head (no sum over the heads of a KV group, utils.py:112 of fastkv) and
    without TSP; the cache then holds H rows sets: K/V [B,H,capacity,D].

    q [B,H,S,D]; k, v either [B,H,S,D] (already repeated, as the SnapKV attention module does before the call:
    snapkv/llama_model.py:161-170) or [B,Hkv,S,D]: the H/Hkv query heads of a group then read the group's K/V rows through a
    zero head stride -- no `repeat_kv` copy is made (the reference materialises it: 4x the K/V bytes at G = 4).
    No kernel of its own: each batch row is the ordinary operator on the problem (batch = KV head, kv heads = query heads of
    the group, G = 1)."""
    _check_qkv(q, k, v)
    B, H, S, D = q.shape
    Hkv = k.shape[1]
    assert H % Hkv == 0 and k.shape == v.shape and k.shape[0] == B and k.shape[2] == S and k.shape[3] == D
    G = H // Hkv
    ko = torch.empty(B, H, capacity, D, dtype=torch.float16, device=q.device)
    vo = torch.empty_like(ko)
    idx = torch.empty(B, H, capacity - window, dtype=torch.int64, device=q.device) if return_indices else None
    for b in range(B):
        qb = q[b].as_strided((Hkv, G, S, D), (G * q.stride(1), q.stride(1), q.stride(2), 1), q[b].storage_offset())
        kb = k[b].as_strided((Hkv, G, S, D), (k.stride(1), 0, k.stride(2), 1), k[b].storage_offset())
        vb = v[b].as_strided((Hkv, G, S, D), (v.stride(1), 0, v.stride(2), 1), v[b].storage_offset())
        out = update_kv(qb, kb, vb, window, kernel_size, pooling, capacity, 0, order, return_indices=return_indices,
                        out=(ko[b].view(Hkv, G, capacity, D), vo[b].view(Hkv, G, capacity, D)))
        if return_indices:
            idx[b] = out[3].view(H, capacity - window)
    return (ko, vo, idx) if return_indices else (ko, vo)


def scores(q: torch.Tensor, k: torch.Tensor, window: int, kernel_size: int, pooling: str, want_tsp: bool = True):
    """attn_cache c[B,Hkv,n] (utils.py:112) and optionally the TSP row t[B,n] (utils.py:127)."""
    _check_qkv(q, k)
    L = load()
    p = _problem(q, k, window, kernel_size, pooling, q.shape[2], 0, "index")
    n = p.S - window
    c = torch.empty(p.B, p.Hkv, n, dtype=torch.float16, device=q.device)
    t = torch.empty(p.B, n, dtype=torch.float16, device=q.device) if want_tsp else None
    ws = _workspace(L.fastkv_workspace_bytes(ctypes.byref(p)), q.device)
    rc = L.fastkv_score_f16(ctypes.byref(p), q.data_ptr(), _strides(q), k.data_ptr(), _strides(k), c.data_ptr(),
                            t.data_ptr() if t is not None else None, ws.data_ptr(), ws.numel(), _stream())
    check(rc, "score")
    return c, t


def select(scores2d: torch.Tensor, k: int, order: str = "index", append: int = 0) -> torch.Tensor:
    """Canonical top-k of every row of a [rows, n] fp16 tensor -> int64 [rows, k + append]."""
    _require_cuda(scores2d)
    assert scores2d.dim() == 2 and scores2d.dtype == torch.float16 and scores2d.stride(1) == 1
    L = load()
    rows, n = scores2d.shape
    out = torch.empty(rows, k + append, dtype=torch.int64, device=scores2d.device)
    ws = _workspace(L.fastkv_select_workspace_bytes(rows, n, k), scores2d.device, "scratch")
    rc = L.fastkv_select_f16(scores2d.data_ptr(), rows, scores2d.stride(0), n, k, ORDER[order], append, out.data_ptr(),
                             ws.data_ptr(), ws.numel(), _stream())
    check(rc, "select")
    return out


def head_sum(c: torch.Tensor) -> torch.Tensor:
    """t[b,j] = fp16(sum_r c[b,r,j]) for a contiguous fp16 [B,R,n] tensor (utils.py:127)."""
    _require_cuda(c)
    assert c.dim() == 3 and c.dtype == torch.float16 and c.is_contiguous()
    t = torch.empty(c.shape[0], c.shape[2], dtype=torch.float16, device=c.device)
    check(load().fastkv_head_sum_f16(c.data_ptr(), c.shape[0], c.shape[1], c.shape[2], t.data_ptr(), _stream()), "head_sum")
    return t


def compact(k: torch.Tensor, v: torch.Tensor, idx: torch.Tensor, window: int, scores: Optional[torch.Tensor] = None,
            return_sorted: bool = False):
    """K/V gather + window append (utils.py:114-121) for given per-head indices [B,Hkv,cap-W] int64.

    `scores` None: rows in the order of `idx`.  `scores` [B,Hkv,>=S-W] fp16 (the score rows `idx` was selected from, `idx`
    ascending): rows in the reference's order -- score descending, ties by position (utils.py:113) -- ranked inside the copy
    kernel; `return_sorted` adds the positions in that order."""
    _check_qkv(k, k, v)
    L = load()
    B, Hkv, S, D = k.shape
    cap = idx.shape[2] + window
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.is_cuda
    p = Problem(B=B, H=Hkv, Hkv=Hkv, S=S, D=D, window=window, kernel=1, pooling=0, capacity=cap, tsp_len=0, order=0, reserved=0)
    ko = torch.empty(B, Hkv, cap, D, dtype=torch.float16, device=k.device)
    vo = torch.empty_like(ko)
    if scores is None:
        rc = L.fastkv_compact_f16(ctypes.byref(p), k.data_ptr(), _strides(k), v.data_ptr(), _strides(v), idx.data_ptr(),
                                  ko.data_ptr(), vo.data_ptr(), _stream())
        check(rc, "compact")
        return ko, vo
    _require_cuda(scores)
    assert scores.dtype == torch.float16 and scores.dim() == 3 and scores.shape[:2] == (B, Hkv) and scores.stride(2) == 1 \
        and scores.stride(0) == Hkv * scores.stride(1)
    srt = torch.empty_like(idx) if return_sorted else None
    ws = _workspace(B * Hkv * ((cap - window + 7) // 8 * 8) * 2, k.device, "scratch")
    rc = L.fastkv_compact_ranked_f16(ctypes.byref(p), k.data_ptr(), _strides(k), v.data_ptr(), _strides(v), idx.data_ptr(),
                                     scores.data_ptr(), scores.stride(1), srt.data_ptr() if srt is not None else None,
                                     ko.data_ptr(), vo.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "compact_ranked")
    return (ko, vo, srt) if return_sorted else (ko, vo)


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[b, r, ...] = src[b, idx[b, r], ...]: the TSP propagation gather (llama_model.py:254-257).

    `src` is [B, S, ...] with contiguous trailing dims whose byte size is a multiple of 16
    (hidden states), or [B, S] int64 position ids (handled as 8-byte rows via a 16-byte staging trick:
    not supported -> use torch.gather for those 16 KiB)."""
    _require_cuda(src, idx)
    assert idx.dtype == torch.int64 and idx.dim() == 2 and idx.stride(1) == 1
    B, S = src.shape[0], src.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size()
    if not src[0, 0].is_contiguous() and src[0, 0].numel() > 1:
        raise ValueError("fastkv_amd.gather_rows: trailing dims must be contiguous")
    L = load()
    out = torch.empty((B, idx.shape[1]) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    rc = L.fastkv_gather_rows(src.data_ptr(), src.stride(0) * src.element_size(), src.stride(1) * src.element_size(),
                              idx.data_ptr(), idx.stride(0), B, idx.shape[1], S, row_bytes, out.data_ptr(), _stream())
    check(rc, "gather_rows")
    return out


def tsp_propagate(hidden: torch.Tensor, position_ids: torch.Tensor, tsp_idx: torch.Tensor):
    """The decoder layer's TSP propagation (llama_model.py:252-259) in ONE launch: returns (hidden[b, tsp_idx[b]], position_ids
    .gather(1, tsp_idx)).  `hidden` [B,S,...] with contiguous trailing dims (16-byte multiple), `position_ids` [B,S] or [1,S]
    (broadcast over the batch) int64.  Indices outside [0, S) read a clamped row: the index tensor of a call that was REPORTED
    (FASTKV_EABORTED) cannot fault here, where `torch.gather` would assert on the device."""
    _require_cuda(hidden, tsp_idx)
    assert tsp_idx.dtype == torch.int64 and tsp_idx.dim() == 2 and tsp_idx.stride(1) == 1
    assert position_ids.dtype == torch.int64 and position_ids.dim() == 2 and position_ids.stride(1) == 1 and position_ids.is_cuda
    B, S = hidden.shape[0], hidden.shape[1]
    assert position_ids.shape[1] == S and position_ids.shape[0] in (1, B)
    row_bytes = hidden[0, 0].numel() * hidden.element_size()
    if not hidden[0, 0].is_contiguous() and hidden[0, 0].numel() > 1:
        raise ValueError("fastkv_amd.tsp_propagate: trailing dims must be contiguous")
    out = torch.empty((B, tsp_idx.shape[1]) + tuple(hidden.shape[2:]), dtype=hidden.dtype, device=hidden.device)
    pos = torch.empty(B, tsp_idx.shape[1], dtype=torch.int64, device=hidden.device)
    rc = load().fastkv_tsp_propagate(hidden.data_ptr(), hidden.stride(0) * hidden.element_size(), hidden.stride(1) * hidden.element_size(),
                                     position_ids.data_ptr(), position_ids.stride(0) if position_ids.shape[0] == B and B > 1 else 0,
                                     tsp_idx.data_ptr(), tsp_idx.stride(0), B, tsp_idx.shape[1], S, row_bytes, out.data_ptr(), pos.data_ptr(),
                                     _stream())
    check(rc, "tsp_propagate")
    return out, pos


# ------------------------------------------------------------------------------------------------- decode over the slab cache
DECODE_NSPLIT = int(os.environ.get("FASTKV_DECODE_NSPLIT", "16"))


def decode_append(kslab: torch.Tensor, vslab: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, len_dev: torch.Tensor) -> None:
    """Slab row `len_dev[0]` <- the step's K/V row ([B,Hkv,1,D]); the length itself is advanced by decode_attention."""
    _require_cuda(kslab, vslab, k_new, v_new, len_dev)
    B, Hkv, rows, D = kslab.shape
    assert k_new.shape == (B, Hkv, 1, D) and v_new.shape == k_new.shape and k_new.stride(3) == 1 and v_new.stride(3) == 1
    assert kslab.stride() == vslab.stride() and kslab.stride(3) == 1 and len_dev.dtype == torch.int32
    I2, I3 = ctypes.c_int64 * 2, ctypes.c_int64 * 3
    rc = load().fastkv_decode_append_f16(B, Hkv, D, k_new.data_ptr(), I2(k_new.stride(0), k_new.stride(1)), v_new.data_ptr(),
                                         I2(v_new.stride(0), v_new.stride(1)), kslab.data_ptr(), vslab.data_ptr(),
                                         I3(*kslab.stride()[:3]), rows, len_dev.data_ptr(), _stream())
    check(rc, "decode_append")


def decode_attention(q: torch.Tensor, kslab: torch.Tensor, vslab: torch.Tensor, len_dev: torch.Tensor, scaling: float,
                     nsplit: int = 0, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GQA attention of q [B,H,1,D] over slab rows 0 .. len_dev[0] (inclusive: the row decode_append just wrote) -> fp16
    [B,1,H*D]; advances len_dev on the device.  Static shapes: capturable in a HIP graph."""
    _require_cuda(q, kslab, vslab, len_dev)
    B, H, one, D = q.shape
    Hkv, rows = kslab.shape[1], kslab.shape[2]
    assert one == 1 and q.stride(3) == 1 and q.dtype == torch.float16 and kslab.dtype == torch.float16
    nsplit = nsplit or DECODE_NSPLIT
    L = load()
    out = torch.empty(B, 1, H * D, dtype=torch.float16, device=q.device)
    need = L.fastkv_decode_workspace_bytes(B, H, D, nsplit)
    ws = workspace if workspace is not None and workspace.numel() >= need else _workspace(need, q.device, "decode")
    I2, I3 = ctypes.c_int64 * 2, ctypes.c_int64 * 3
    rc = L.fastkv_decode_attention_f16(B, H, Hkv, D, q.data_ptr(), I2(q.stride(0), q.stride(1)), kslab.data_ptr(), vslab.data_ptr(),
                                       I3(*kslab.stride()[:3]), rows, len_dev.data_ptr(), ctypes.c_float(scaling), nsplit,
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "decode_attention")
    return out


def decode_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm of the step's hidden states [..., hidden] (fp16) in one launch; arithmetic of LlamaRMSNorm."""
    _require_cuda(x, weight)
    hidden = x.shape[-1]
    assert x.dtype == torch.float16 and weight.dtype == torch.float16 and x.stride(-1) == 1 and weight.is_contiguous()
    x2 = x.reshape(-1, hidden)
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    check(load().fastkv_decode_rmsnorm_f16(x2.data_ptr(), x2.shape[0], x2.stride(0), hidden, weight.data_ptr(), ctypes.c_float(eps),
                                           out.data_ptr(), _stream()), "decode_rmsnorm")
    return out


def decode_rope_(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> None:
    """apply_rotary_pos_emb on the step's q [B,H,1,D] / k [B,Hkv,1,D], IN PLACE; cos / sin [B,1,D] fp16."""
    _require_cuda(q, k, cos, sin)
    B, H, one, D = q.shape
    assert one == 1 and k.shape[2] == 1 and q.stride(3) == 1 and k.stride(3) == 1 and cos.shape == (B, 1, D) and cos.stride(2) == 1 \
        and sin.stride() == cos.stride() and cos.dtype == torch.float16 and q.dtype == torch.float16
    I2 = ctypes.c_int64 * 2
    check(load().fastkv_decode_rope_f16(B, H, k.shape[1], D, q.data_ptr(), I2(q.stride(0), q.stride(1)), k.data_ptr(),
                                        I2(k.stride(0), k.stride(1)), cos.data_ptr(), sin.data_ptr(), cos.stride(0), _stream()),
          "decode_rope")


def decode_silu_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up (fp16, contiguous, same shape) in one launch."""
    _require_cuda(gate, up)
    assert gate.shape == up.shape and gate.is_contiguous() and up.is_contiguous() and gate.dtype == torch.float16
    out = torch.empty_like(gate)
    check(load().fastkv_decode_silu_mul_f16(gate.data_ptr(), up.data_ptr(), gate.numel(), out.data_ptr(), _stream()), "decode_silu_mul")
    return out


def decode_rotary(inv_freq: torch.Tensor, position_ids: torch.Tensor, scaling: float, D: int):
    """cos / sin [B,1,D] fp16 of a one-token step in ONE launch: LlamaRotaryEmbedding.forward(x, position_ids [B,1]) for fp16 `x`
    (csrc/decode.hip decode_rotary_kernel; the stock module runs ten small launches)."""
    _require_cuda(inv_freq, position_ids)
    B = position_ids.shape[0]
    assert position_ids.shape == (B, 1) and position_ids.dtype == torch.int64 and position_ids.is_contiguous()
    assert inv_freq.dtype == torch.float32 and inv_freq.is_contiguous() and inv_freq.numel() == D // 2
    cos = torch.empty(B, 1, D, dtype=torch.float16, device=position_ids.device)
    sin = torch.empty_like(cos)
    check(load().fastkv_decode_rotary_f16(B, D, inv_freq.data_ptr(), position_ids.data_ptr(), ctypes.c_float(scaling), cos.data_ptr(),
                                          sin.data_ptr(), _stream()), "decode_rotary")
    return cos, sin


def new_greedy_scratch(device: torch.device, B: int) -> torch.Tensor:
    """Scratch of `decode_greedy` (zero between launches; allocate it before a capture)."""
    return torch.zeros(B + 1, dtype=torch.int64, device=device)


def decode_greedy(logits: torch.Tensor, scratch: torch.Tensor, tok: torch.Tensor, pos: Optional[torch.Tensor] = None,
                  log: Optional[torch.Tensor] = None, log_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Greedy sampling of a step in ONE launch (csrc/decode.hip decode_greedy_kernel): tok[b, 0] = argmax of logits[b, -1, :]
    (torch.argmax's rule), optionally pos[b, 0] += 1, log[log_index[0]] = tok[0, 0], log_index[0] += 1 -- the tail of
    benchmark/e2e.py's captured step.  logits [B, T, V] fp16; tok / pos [B, 1] int64; log [n] int64; log_index [1] int64."""
    _require_cuda(logits, scratch, tok)
    B, T, V = logits.shape
    row = logits[:, -1, :]
    assert logits.dtype == torch.float16 and row.stride(1) == 1 and tok.dtype == torch.int64 and tok.shape == (B, 1) and tok.is_contiguous()
    assert scratch.dtype == torch.int64 and scratch.numel() >= B + 1 and (B == 1 or row.stride(0) % 8 == 0)
    assert pos is None or (pos.dtype == torch.int64 and pos.shape == (B, 1) and pos.is_contiguous())
    assert (log is None) == (log_index is None) and (log is None or (log.dtype == torch.int64 and log.is_contiguous() and log_index.dtype == torch.int64))
    check(load().fastkv_decode_greedy_f16(B, V, row.data_ptr(), row.stride(0) if B > 1 else V, scratch.data_ptr(), tok.data_ptr(),
                                          pos.data_ptr() if pos is not None else None, log.data_ptr() if log is not None else None,
                                          log_index.data_ptr() if log_index is not None else None, log.numel() if log is not None else 0,
                                          _stream()), "decode_greedy")
    return tok


def decode_gemv(x: torch.Tensor, weights, norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0, glu: bool = False,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The projections of a one-token step in ONE launch (csrc/gemv.hip): x [B,1,K] fp16 times up to three `nn.Linear` weights
    [N_i,K] that share it -> [B,1,sum N_i]; `norm_weight`: RMSNorm of x first; `glu`: silu(W0 x) * (W1 x); `residual`
    [B,1,N]: added to the result.  Rounding points of the stock fp16 modules."""
    _require_cuda(x, *weights)
    B, one, K = x.shape
    assert one == 1 and x.dtype == torch.float16 and x.stride(2) == 1 and 1 <= len(weights) <= 3
    for w in weights:
        assert w.dtype == torch.float16 and w.is_contiguous() and w.shape[1] == K
    rows = [int(w.shape[0]) for w in weights]
    n_out = rows[0] if glu else sum(rows)
    out = torch.empty(B, 1, n_out, dtype=torch.float16, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == torch.float16 and residual.stride(2) == 1
    if norm_weight is not None:
        assert norm_weight.dtype == torch.float16 and norm_weight.is_contiguous() and norm_weight.numel() == K
    n = len(weights)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in weights])
    rp = (ctypes.c_int32 * n)(*rows)
    rc = load().fastkv_decode_gemv_f16(B, K, x.data_ptr(), x.stride(0), norm_weight.data_ptr() if norm_weight is not None else None,
                                       ctypes.c_float(eps), n, wp, rp, 1 if glu else 0,
                                       residual.data_ptr() if residual is not None else None,
                                       residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), _stream())
    check(rc, "decode_gemv")
    return out


_step_counters = {}
_step_ws = {}
DECODE_STEP_NSPLIT = int(os.environ.get("FASTKV_DECODE_STEP_NSPLIT", "0"))     # 0: the library chooses (one 128-row tile per slice)


def new_step_counters(device: torch.device) -> torch.Tensor:
    """State of the fused step kernel between launches: word 0 counts the workgroups that have read the cache length (zero between
    launches), word 1 is the launch epoch that tags the slice records of `new_decode_workspace` (only ever advanced).  A cache slab owns
    its own set (allocated eagerly by SlabLayer.enable_static_decode -- never for the first time inside a graph capture, where the
    allocation would live in the graph's private pool and its zero-fill would be replayed with every step), so steps over different
    caches may run concurrently and a capture only ever records launches."""
    return torch.zeros(1024, dtype=torch.int32, device=device)


def new_decode_workspace(device: torch.device, B: int, H: int, D: int, nsplit: int = 0) -> torch.Tensor:
    """Slice-record scratch of decode_step_attention / decode_attention for up to H query heads.  ZERO-filled: the step kernel's records
    are {launch token, value} granules that a reader accepts by their token (never 0), so the memory must not hold anything that
    could pass for one -- it belongs to ONE set of step counters (whose epoch the tokens come from) for good."""
    n = load().fastkv_decode_workspace_bytes(B, H, D, nsplit)
    return torch.zeros(max(int(n), 256), dtype=torch.uint8, device=device)


def _default_step_state(device: torch.device, need: int):
    # callers without a slab object (tests, tools): one {counters, records} pair per (device, stream), allocated outside captures only
    key = (device.index, _stream())
    cnt, ws = _step_counters.get(key), _step_ws.get(key)
    if cnt is None or ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("fastkv_amd.decode_step_attention: pass `counters` and `workspace` (ops.new_step_counters / "
                               "ops.new_decode_workspace, allocated before the capture) when the step is captured in a graph")
        if cnt is None:
            cnt = _step_counters[key] = new_step_counters(device)
        if ws is None or ws.numel() < need:
            ws = _step_ws[key] = torch.zeros(max(need, 1 << 20), dtype=torch.uint8, device=device)
    return cnt, ws


def reset_decode_state(*counters: torch.Tensor) -> None:
    """After a reported error (FASTKV_EABORTED, a killed kernel) the arrival counter may be non-zero for good: zero it from the
    host before the next step (the given sets, and every default set).  The epoch word stays: tokens are never reused."""
    for c in list(counters) + list(_step_counters.values()):
        c[:1].zero_()


def decode_step_attention(q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                          kslab: torch.Tensor, vslab: torch.Tensor, len_dev: torch.Tensor, scaling: float, nsplit: int = 0,
                          counters: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RoPE + append + GQA attention + slice merge of a one-token step in ONE launch (csrc/decode_step.hip): q [B,H,1,D] / k_new,
    v_new [B,Hkv,1,D] are the RAW projections, cos / sin [B,1,D] fp16; returns fp16 [B,1,H*D], the slab gains the row, len_dev is
    advanced.  `counters` + `workspace` come as a pair (new_step_counters / new_decode_workspace); nsplit 0 = the library chooses."""
    _require_cuda(q, k_new, v_new, cos, sin, kslab, vslab, len_dev)
    B, H, one, D = q.shape
    Hkv, rows = kslab.shape[1], kslab.shape[2]
    assert one == 1 and q.stride(3) == 1 and k_new.shape == (B, Hkv, 1, D) and v_new.shape == k_new.shape and k_new.stride(3) == 1 \
        and v_new.stride(3) == 1 and cos.shape == (B, 1, D) and cos.stride(2) == 1 and sin.stride() == cos.stride() \
        and cos.dtype == torch.float16 and q.dtype == torch.float16 and kslab.stride() == vslab.stride() and len_dev.dtype == torch.int32
    nsplit = nsplit or DECODE_STEP_NSPLIT
    L = load()
    out = torch.empty(B, 1, H * D, dtype=torch.float16, device=q.device)
    need = L.fastkv_decode_workspace_bytes(B, H, D, nsplit)
    assert (counters is None) == (workspace is None), "counters and workspace of the step kernel come as a pair"
    if counters is None:
        cnt, ws = _default_step_state(q.device, need)
    else:
        cnt, ws = counters, workspace
    assert cnt.dtype == torch.int32 and cnt.numel() >= 2 and cnt.is_cuda and ws.numel() >= need
    I2, I3 = ctypes.c_int64 * 2, ctypes.c_int64 * 3
    rc = L.fastkv_decode_step_attention_f16(B, H, Hkv, D, q.data_ptr(), I2(q.stride(0), q.stride(1)), k_new.data_ptr(),
                                            I2(k_new.stride(0), k_new.stride(1)), v_new.data_ptr(), I2(v_new.stride(0), v_new.stride(1)),
                                            cos.data_ptr(), sin.data_ptr(), cos.stride(0), kslab.data_ptr(), vslab.data_ptr(),
                                            I3(*kslab.stride()[:3]), rows, len_dev.data_ptr(), ctypes.c_float(scaling), nsplit,
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), cnt.data_ptr(), _stream())
    check(rc, "decode_step_attention")
    return out


# ------------------------------------------------------------------------------------------------- separately allocated entries
_ptr_ring = {}


_ptr_cache = {}


def _device_ptr_table(rows, device: torch.device) -> torch.Tensor:
    """int64 [len(rows), n] on `device` from lists of addresses.  A table only holds addresses, so a table built once for a
    set of addresses is good for ever: tables are cached by their content (a caching allocator hands a model the same blocks
    prompt after prompt -- no copy at all in the steady state).  A miss is staged through a small ring of pinned host buffers
    (asynchronous copy on the current stream; a buffer is reused only after the copy that read it has completed)."""
    key = (device.index, _stream(), tuple(map(tuple, rows)))
    hit = _ptr_cache.get(key)
    if hit is not None:
        return hit
    tab = _stage_ptr_table(rows, device)
    if len(_ptr_cache) >= 256:
        _ptr_cache.clear()
    _ptr_cache[key] = tab
    return tab


def _stage_ptr_table(rows, device: torch.device) -> torch.Tensor:
    if torch.cuda.is_current_stream_capturing():
        # a replay would re-read the pinned staging buffer, which other calls have rewritten since
        raise RuntimeError("fastkv_amd.update_kv_entries cannot be captured in a HIP graph (its address tables are staged through "
                           "host memory); call update_kv per entry inside a capture")
    n = len(rows[0])
    key = device.index
    ring = _ptr_ring.get(key)
    if ring is None:
        ring = _ptr_ring[key] = {"bufs": [torch.empty(5 * 512, dtype=torch.int64).pin_memory() for _ in range(8)],
                                 "events": [None] * 8, "next": 0}
    assert len(rows) * n <= 5 * 512
    i = ring["next"]
    ring["next"] = (i + 1) % 8
    if ring["events"][i] is not None:
        ring["events"][i].synchronize()
    buf = ring["bufs"][i][:len(rows) * n].view(len(rows), n)
    buf.copy_(torch.tensor(rows, dtype=torch.int64))
    dev = buf.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ring["events"][i] = ev
    return dev


def fused_entries(H: int, Hkv: int, S: int, D: int, window: int, kernel_size: int) -> int:
    """Entries ONE fused scoring launch holds for this geometry (0: the geometry takes the staged path; `update_kv_entries` then
    refuses it).  Host only."""
    p = Problem(B=1, H=H, Hkv=Hkv, S=S, D=D, window=window, kernel=kernel_size, pooling=0, capacity=S, tsp_len=0, order=0, reserved=_engine)
    return int(load().fastkv_fused_entries_f16(ctypes.byref(p)))


def window_rows(q: torch.Tensor, window: int) -> torch.Tensor:
    """The only rows of the query tensor the operator reads (utils.py:93: `query_states[..., -window:, :]`), as a contiguous
    [B,H,window,D] copy: 64 KiB for Llama-3-8B where the tensor itself is 256 MiB at 32k -- what a deferred entry keeps alive
    (`update_kv_entries(..., q_window=True)`)."""
    return q[:, :, q.shape[2] - window:, :].contiguous()


def update_kv_entries(qs, ks, vs, window: int, kernel_size: int, pooling: str, capacity: int, tsp_len: int = 0, order: str = "score",
                      outs=None, return_indices: bool = False, q_window: bool = False):
    """`update_kv` over SEPARATELY ALLOCATED entries in one launch sequence (fastkv_update_kv_ptrs_f16): qs / ks / vs are lists
    of [Bq,H,S,D] / [Bq,Hkv,S,D] fp16 tensors of ONE geometry and ONE memory layout (e.g. the layers of a model whose compression
    was deferred to the end of the forward pass; every batch row of every tensor becomes one entry of the library call).
    Returns (k_outs, v_outs, tsp_idx [n*Bq,tsp_len] | None[, kv_idx [n*Bq,Hkv,cap-W]]), rows i*Bq .. (i+1)*Bq-1 belonging to tensor i;
    `outs` = (list of k buffers, list of v buffers), [Bq,Hkv,cap,D] views of one stride pattern, to write into.
    `q_window`: every qs[i] is `window_rows(q_i, window)` ([Bq,H,window,D] contiguous) instead of the whole query tensor; the library
    is handed an address `S - window` rows in front of it, so that row S - window + r of "the query tensor" is row r of the copy.
    A batch that does not fit one fused scoring launch is scored by several (the library splits it) and selected / copied by ONE
    launch each.  Raises FastKVNativeError(FASTKV_EUNSUPPORTED) for geometries off the fused scoring path: call `update_kv` per
    entry then."""
    n = len(qs)
    assert n >= 1 and len(ks) == n and len(vs) == n
    q0, k0, v0 = qs[0], ks[0], vs[0]
    if q_window:
        S_full = k0.shape[2]
        for q in qs:
            if q.dim() != 4 or q.shape[2] != window or not q.is_contiguous() or q.dtype != torch.float16:
                raise ValueError("fastkv_amd.update_kv_entries(q_window=True): [B,H,window,D] contiguous fp16 copies of the window rows expected")
        _check_qkv(k0, k0, v0)
        _require_cuda(q0)
    else:
        _check_qkv(q0, k0, v0)
    Bq = q0.shape[0]

    def same_layout(ok: bool, what: str):
        # entries that do not share one geometry / layout / alignment cannot go through one launch sequence: the caller runs them
        # one by one (the only condition besides the library's own FASTKV_EUNSUPPORTED that DeferredCompression falls back on)
        if not ok:
            e = FastKVNativeError(f"fastkv_amd.update_kv_entries: entries differ in {what}", code=FASTKV_EUNSUPPORTED)
            e.layout = True                                       # a property of THESE tensors, not of the geometry (DeferredCompression)
            raise e

    for q, k, v in zip(qs, ks, vs):
        same_layout(q.shape == q0.shape and k.shape == k0.shape and v.shape == v0.shape, "shape")
        same_layout(q.stride() == q0.stride() and k.stride() == k0.stride() and v.stride() == v0.stride(), "strides")
        same_layout(q.dtype == torch.float16 and k.dtype == torch.float16 and v.dtype == torch.float16, "dtype")
        same_layout((q.data_ptr() | k.data_ptr() | v.data_ptr()) % 16 == 0, "16-byte alignment")
    L = load()
    if q_window:
        if pooling not in POOLING:
            raise ValueError("Pooling method not supported")
        p = Problem(B=Bq, H=q0.shape[1], Hkv=k0.shape[1], S=S_full, D=q0.shape[3], window=window, kernel=kernel_size, pooling=POOLING[pooling],
                    capacity=capacity, tsp_len=tsp_len, order=ORDER[order], reserved=_engine)
    else:
        p = _problem(q0, k0, window, kernel_size, pooling, capacity, tsp_len, order)
    p.B = n * Bq
    Hkv, D, dev = p.Hkv, p.D, q0.device
    if outs is None:
        kb = torch.empty(n, Bq, Hkv, capacity, D, dtype=torch.float16, device=dev)
        vb = torch.empty_like(kb)
        k_outs, v_outs = [kb[i] for i in range(n)], [vb[i] for i in range(n)]
    else:
        k_outs, v_outs = outs
        for ko, vo in zip(k_outs, v_outs):
            same_layout(ko.shape == (Bq, Hkv, capacity, D) and vo.shape == ko.shape and ko.stride() == k_outs[0].stride() == vo.stride(),
                        "output shape / strides")
            same_layout(ko.stride(3) == 1 and (ko.data_ptr() | vo.data_ptr()) % 16 == 0, "output alignment")
    ostr = (ctypes.c_int64 * 3)(*k_outs[0].stride()[:3])
    if Bq > 1:
        for t in (q0, k0, v0, k_outs[0]):
            same_layout((t.stride(0) * 2) % 16 == 0, "16-byte alignment of the batch rows")

    def rows_of(lst, back=0):                                      # one address per batch row of every tensor
        return [t.data_ptr() + b * t.stride(0) * 2 - back for t in lst for b in range(Bq)]

    # (window copies: the kernel addresses query row S - window + r as base + (S - window + r) * row stride)
    q_back = (S_full - window) * q0.stride(2) * 2 if q_window else 0
    tab = _device_ptr_table([rows_of(qs, q_back)] + [rows_of(lst) for lst in (ks, vs, k_outs, v_outs)], dev)
    kv_idx = torch.empty(n * Bq, Hkv, capacity - window, dtype=torch.int64, device=dev) if return_indices else None
    # (empty, not zeros as in update_kv: the consumer of a grouped call's TSP index is this package's own wiring, whose propagation
    # -- ops.tsp_propagate -- clamps its indices; the 128 KiB fill in front of the TSP group's scoring launch cost 4.8 us per prefill)
    tsp = torch.empty(n * Bq, tsp_len, dtype=torch.int64, device=dev) if tsp_len else None
    ws = _workspace(L.fastkv_workspace_bytes(ctypes.byref(p)), dev)
    rc = L.fastkv_update_kv_ptrs_f16(ctypes.byref(p), tab[0].data_ptr(), _strides(q0), tab[1].data_ptr(), _strides(k0), tab[2].data_ptr(),
                                     _strides(v0), tab[3].data_ptr(), tab[4].data_ptr(), ostr,
                                     kv_idx.data_ptr() if kv_idx is not None else None, tsp.data_ptr() if tsp is not None else None,
                                     ws.data_ptr(), ws.numel(), _stream())
    check(rc, "update_kv_entries")
    out = [k_outs, v_outs, tsp]
    if return_indices:
        out.append(kv_idx)
    return tuple(out)
