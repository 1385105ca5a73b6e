"""ctypes front-end of oracle/fastkv_oracle.c -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's hot path
(/root/reference/baselines/fastkv/utils.py:80-134 and
/root/reference/baselines/fastkv/llama_model.py:252-259).  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
the product (fastkv_amd) never does.

Tensors cross this boundary as torch CPU tensors (fp16 viewed as int16 bit patterns);
strides are passed in elements, so the `[B,S,H,D]`-physical / `[B,H,S,D]`-logical views
the attention module produces (llama_model.py:117-122) are consumed as they are.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "fastkv_oracle.c")
_LIB = os.path.join(_HERE, "libfastkv_oracle.so")
CFLAGS = ["-O3", "-ffp-contract=off", "-mavx2", "-mfma", "-mf16c", "-fopenmp", "-shared", "-fPIC"]

_lib = None
CONTRACTION_FMAF, CONTRACTION_MFMA16 = 0, 1                      # (fastkv_oracle.c FK_CONTRACT_FMAF / FK_CONTRACT_MFMA16)


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Returns the path of the .so.
    FASTKV_ORACLE_SANITIZE=1 (tests/test_oracle_golden.py::test_oracle_under_address_and_ub_sanitizers): the same source with
    -fsanitize=address,undefined into its own file -- to be loaded by a process started with the sanitizer runtime preloaded."""
    if os.environ.get("FASTKV_ORACLE_SANITIZE", "0") == "1":
        san = os.path.join(_HERE, "libfastkv_oracle_asan.so")
        if force or not os.path.exists(san) or os.path.getmtime(san) < os.path.getmtime(_SRC):
            flags = [f for f in CFLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                                                         "-fno-sanitize-recover=undefined"]
            subprocess.check_call(["gcc", *flags, "-o", san, _SRC, "-lm"])
        return san
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", *CFLAGS, "-o", _LIB, _SRC, "-lm"])
    return _LIB


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        i64p, vp, ci = ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.c_int
        L.fastkv_oracle_scores_f16.argtypes = [vp, i64p, vp, i64p] + [ci] * 8 + [vp, vp, vp]
        L.fastkv_oracle_scores_f16.restype = ci
        L.fastkv_oracle_topk_f16.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ci, vp]
        L.fastkv_oracle_topk_f16.restype = ci
        L.fastkv_oracle_gather_rows.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64]
        L.fastkv_oracle_gather_rows.restype = ci
        L.fastkv_oracle_update_kv_f16.argtypes = [vp, i64p, vp, i64p, vp, i64p] + [ci] * 11 + [vp] * 6
        L.fastkv_oracle_update_kv_f16.restype = ci
        L.fastkv_oracle_det_expf.argtypes = [ctypes.c_float]
        L.fastkv_oracle_det_expf.restype = ctypes.c_float
        L.fastkv_oracle_fix_to_f32.argtypes = [ctypes.c_uint64]
        L.fastkv_oracle_fix_to_f32.restype = ctypes.c_float
        L.fastkv_oracle_exp_to_fix.argtypes = [ctypes.c_float]
        L.fastkv_oracle_exp_to_fix.restype = ctypes.c_uint64
        L.fastkv_oracle_f2h.argtypes = [ctypes.c_float]
        L.fastkv_oracle_f2h.restype = ctypes.c_uint16
        L.fastkv_oracle_h2f.argtypes = [ctypes.c_uint16]
        L.fastkv_oracle_h2f.restype = ctypes.c_float
        L.fastkv_oracle_scale_logit.argtypes = [ctypes.c_uint16, ci]
        L.fastkv_oracle_scale_logit.restype = ctypes.c_uint16
        L.fastkv_oracle_set_contraction.argtypes = [ci]
        L.fastkv_oracle_get_contraction.restype = ci
        L.fastkv_oracle_mfma16_tiles.argtypes = [vp, vp, vp, vp, ci, ci]
        L.fastkv_oracle_mfma16_tiles.restype = ci
        L.fastkv_oracle_stages_f16.argtypes = [vp, i64p, vp, i64p] + [ci] * 8 + [vp, vp, vp, vp]
        L.fastkv_oracle_stages_f16.restype = ci
        L.fastkv_oracle_set_softmax.argtypes = [ci]
        L.fastkv_oracle_get_softmax.restype = ci
        L.fastkv_oracle_sleef_expf.argtypes = [ctypes.c_float]
        L.fastkv_oracle_sleef_expf.restype = ctypes.c_float
        L.fastkv_oracle_set_threads.argtypes = [ci]
        L.fastkv_oracle_get_threads.restype = ci
        # the default contract follows FASTKV_CONTRACTION, like the HIP side's "auto" (capi.hip default_contract_f16): the fp32 fma chain
        # unless the process was started with FASTKV_CONTRACTION=mfma16 (the HIP side's opt-in fast mode) -- a suite run under either
        # compares like with like without every caller having to say so
        L.fastkv_oracle_set_contraction(CONTRACTION_MFMA16 if os.environ.get("FASTKV_CONTRACTION", "fmaf")[:1] in ("m", "M") else CONTRACTION_FMAF)
        _lib = L
    return _lib


POOLING = {"avgpool": 0, "maxpool": 1}
ORDER = {"index": 0, "score": 1}


def _strides(t: torch.Tensor):
    assert t.dim() == 4 and t.dtype == torch.float16 and t.device.type == "cpu"
    return (ctypes.c_int64 * 4)(*t.stride())


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"fastkv oracle: {what} failed with code {rc}")


CONTRACTION = {"fmaf": 0, "mfma16": 1}
assert CONTRACTION["fmaf"] == 0


def set_contraction(name: str) -> None:
    """The arithmetic contract of the contraction (utils.py:94; fastkv_oracle.c "the contraction"): "fmaf" (default) = the fp32 fma
    chain, "mfma16" = what the gfx950 fp16 matrix instruction computes.  The HIP side's twins: ops.set_score_engine("mfma" | "valu") /
    ("mfma16"); "auto" there follows FASTKV_CONTRACTION."""
    lib().fastkv_oracle_set_contraction(CONTRACTION[name])


def get_contraction() -> str:
    return {v: k for k, v in CONTRACTION.items()}[int(lib().fastkv_oracle_get_contraction())]


SOFTMAX = {"contract": 0, "torch_avx512": 1, "torch_avx2": 2}


def set_softmax(name: str) -> None:
    """TESTS ONLY (fastkv_oracle.c "the softmax"): "contract" (default) = the arithmetic the HIP kernels share -- det_expf, 2^-40
    fixed-point denominator; "torch_avx512" / "torch_avx2" = the reference's own torch CPU kernel restated (SLEEF exp, 16 / 8
    lane-wise sequential fp32 sums + tree): with the "fmaf" contraction the oracle then reproduces the reference bit for bit.  The
    HIP library has no twin of these modes."""
    lib().fastkv_oracle_set_softmax(SOFTMAX[name])


def get_softmax() -> str:
    return {v: k for k, v in SOFTMAX.items()}[int(lib().fastkv_oracle_get_softmax())]


def stages(q: torch.Tensor, k: torch.Tensor, window: int = 8, kernel_size: int = 7, pooling: str = "avgpool"):
    """(c [B,Hkv,n], t [B,n], logits [B,H,W,S], probabilities [B,H,W,S]), fp16: the scores and the two internal stages the
    stage-level goldens pin -- what enters the reference's softmax (utils.py:94-101) and what leaves it (utils.py:103)."""
    B, H, S, D = q.shape
    Hkv = k.shape[1]
    n = S - window
    c = torch.empty(B, Hkv, n, dtype=torch.float16)
    t = torch.empty(B, n, dtype=torch.float16)
    lg = torch.empty(B, H, window, S, dtype=torch.float16)
    pr = torch.empty(B, H, window, S, dtype=torch.float16)
    _check(lib().fastkv_oracle_stages_f16(q.data_ptr(), _strides(q), k.data_ptr(), _strides(k), B, H, Hkv, S, D, window, kernel_size,
                                          POOLING[pooling], c.data_ptr(), t.data_ptr(), lg.data_ptr(), pr.data_ptr()), "stages")
    return c, t, lg, pr


def softmax_row_f32(x: torch.Tensor) -> torch.Tensor:
    """One softmax row in fp32 under the selected softmax mode, for a 1-D fp16 row (the probe against torch's kernel)."""
    assert x.dim() == 1 and x.dtype == torch.float16 and x.is_contiguous()
    out = torch.empty(x.numel(), dtype=torch.float32)
    L = lib()
    L.fastkv_oracle_softmax_row_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    _check(L.fastkv_oracle_softmax_row_f32(x.data_ptr(), x.numel(), out.data_ptr()), "softmax_row")
    return out


def mfma16_tiles(a: torch.Tensor, bt: torch.Tensor, c: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The restated v_mfma_f32_32x32x16_f16 chain on tiles: a, bt [T,32,dd] fp16 (CPU, contiguous), c [T,32,32] fp32 or None (+0)
    -> [T,32,32] fp32."""
    assert a.dtype == torch.float16 and bt.dtype == torch.float16 and a.is_contiguous() and bt.is_contiguous() and a.shape == bt.shape
    T, _, dd = a.shape
    out = torch.empty(T, 32, 32, dtype=torch.float32)
    if c is not None:
        assert c.dtype == torch.float32 and c.is_contiguous() and c.shape == (T, 32, 32)
    _check(lib().fastkv_oracle_mfma16_tiles(a.data_ptr(), bt.data_ptr(), c.data_ptr() if c is not None else None, out.data_ptr(), T, dd),
           "mfma16_tiles")
    return out


def set_threads(n: int) -> None:
    lib().fastkv_oracle_set_threads(int(n))


def get_threads() -> int:
    return int(lib().fastkv_oracle_get_threads())


def scores(q: torch.Tensor, k: torch.Tensor, window: int = 8, kernel_size: int = 7, pooling: str = "avgpool",
           want_tsp: bool = True, want_logits: bool = False):
    """attn_cache `c[B,Hkv,n]` (utils.py:112) and the TSP row `t[B,n]` (utils.py:127), fp16."""
    B, H, S, D = q.shape
    Hkv = k.shape[1]
    n = S - window
    c = torch.empty(B, Hkv, n, dtype=torch.float16)
    t = torch.empty(B, n, dtype=torch.float16) if want_tsp else None
    lg = torch.empty(B, H, window, S, dtype=torch.float16) if want_logits else None
    rc = lib().fastkv_oracle_scores_f16(q.data_ptr(), _strides(q), k.data_ptr(), _strides(k), B, H, Hkv, S, D, window,
                                        kernel_size, POOLING[pooling], c.data_ptr(),
                                        t.data_ptr() if t is not None else None,
                                        lg.data_ptr() if lg is not None else None)
    _check(rc, "scores")
    return c, t, lg


def canonical_topk(row: torch.Tensor, k: int, order: str = "index") -> torch.Tensor:
    """Canonical top-k (value desc, index asc on ties) of a 1-D fp16 score row -> int64 indices."""
    assert row.dim() == 1 and row.dtype == torch.float16 and row.is_contiguous()
    out = torch.empty(k, dtype=torch.int64)
    _check(lib().fastkv_oracle_topk_f16(row.data_ptr(), row.numel(), k, ORDER[order], out.data_ptr()), "topk")
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[r] = src[idx[r]] for a 2-D contiguous src (TSP hidden gather, llama_model.py:255-257)."""
    assert src.dim() == 2 and src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    dst = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype)
    rb = src.shape[1] * src.element_size()
    _check(lib().fastkv_oracle_gather_rows(src.data_ptr(), rb, idx.data_ptr(), idx.numel(), rb, dst.data_ptr(), rb), "gather")
    return dst


def update_kv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, window: int, kernel_size: int, pooling: str,
              capacity: int, tsp_len: int = 0, order: str = "score", return_scores: bool = False):
    """Compress branch of FastKVCluster.update_kv (utils.py:93-132).

    Returns (k_out, v_out, kv_idx[B,Hkv,cap-W] int64, tsp_idx[B,tsp_len] int64 | None[, c, t]).
    `tsp_len == 0` means "not a TSP layer" (the caller applies the guard of utils.py:126).
    """
    B, H, S, D = q.shape
    Hkv = k.shape[1]
    n = S - window
    ko = torch.empty(B, Hkv, capacity, D, dtype=torch.float16)
    vo = torch.empty_like(ko)
    kv_idx = torch.empty(B, Hkv, capacity - window, dtype=torch.int64)
    tsp_idx = torch.empty(B, tsp_len, dtype=torch.int64) if tsp_len else None
    c = torch.empty(B, Hkv, n, dtype=torch.float16) if return_scores else None
    t = torch.empty(B, n, dtype=torch.float16) if (return_scores and tsp_len) else None
    rc = lib().fastkv_oracle_update_kv_f16(
        q.data_ptr(), _strides(q), k.data_ptr(), _strides(k), v.data_ptr(), _strides(v),
        B, H, Hkv, S, D, window, kernel_size, POOLING[pooling], capacity, tsp_len, ORDER[order],
        ko.data_ptr(), vo.data_ptr(), kv_idx.data_ptr(), tsp_idx.data_ptr() if tsp_idx is not None else None,
        c.data_ptr() if c is not None else None, t.data_ptr() if t is not None else None)
    _check(rc, "update_kv")
    if return_scores:
        return ko, vo, kv_idx, tsp_idx, c, t
    return ko, vo, kv_idx, tsp_idx


class OracleFastKVCluster:
    """Host logic of FastKVCluster (utils.py:48-134) on top of the C restatement; used by the
    parity tests as the stand-in for the reference object on machines without /root/reference."""

    def __init__(self, window_size=8, max_capacity_prompt=512, kernel_size=7, pooling="avgpool", tsp_layer=False,
                 tsp_length=2048, tsp_rate=0.25, retain_rate=0.25, eviction_mode="constant", order="score"):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0          # utils.py:54
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.tsp_layer = tsp_layer
        self.tsp_length = tsp_length
        self.retain_rate = retain_rate
        self.eviction_mode = eviction_mode
        self.tsp_rate = tsp_rate
        self.order = order

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups, layer_idx):
        assert key_states.shape[-2] == query_states.shape[-2]            # utils.py:82
        q_len = query_states.shape[2]
        if self.eviction_mode == "proportional":                         # utils.py:86-87
            self.max_capacity_prompt = int(q_len * self.retain_rate)
        if q_len < self.max_capacity_prompt:                             # utils.py:89-91
            return key_states, value_states, None
        if self.pooling not in POOLING:
            raise ValueError("Pooling method not supported")             # utils.py:110
        if self.tsp_layer and self.eviction_mode == "proportional":      # utils.py:123-124
            self.tsp_length = int(q_len * self.tsp_rate)
        tsp_len = self.tsp_length if (self.tsp_layer and q_len > self.tsp_length) else 0   # utils.py:126
        ko, vo, _, tsp_idx = update_kv(query_states, key_states, value_states, self.window_size, self.kernel_size,
                                       self.pooling, self.max_capacity_prompt, tsp_len, self.order)
        return ko, vo, tsp_idx


def standard_dis_index(data: torch.Tensor, queries: torch.Tensor, k: int, norm=1, pool: bool = False, kernel_size: int = 5,
                       sum_over_heads: bool = False, return_scores: bool = False):
    """CPU restatement of /root/reference/baselines/gemfilter/utils.py:25-38 (`standard_dis_index`): (distances, indices) with
    the canonical tie rule (value descending, lowest position first)."""
    B, H, _, D = queries.shape
    Hd, n = data.shape[1], data.shape[2]
    q0 = queries[:, :, :1].contiguous()
    data = data if data.stride(3) == 1 else data.contiguous()
    R = 1 if sum_over_heads else H
    out = torch.empty(B, R, n, dtype=torch.float16)
    L = lib()
    L.fastkv_oracle_last_query_scores.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                                  ctypes.POINTER(ctypes.c_int64)] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
    _check(L.fastkv_oracle_last_query_scores(q0.data_ptr(), _strides(q0), data.data_ptr(), _strides(data), B, H, Hd, n, D,
                                             int(sum_over_heads), int(pool), kernel_size, out.data_ptr()), "last_query_scores")
    idx = torch.stack([torch.stack([canonical_topk(out[b, r], k, "score") for r in range(R)]) for b in range(B)])
    dist = torch.gather(out, 2, idx)
    if norm != 1:
        dist = dist / norm
    return (dist, idx, out) if return_scores else (dist, idx)
