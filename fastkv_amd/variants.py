"""Selection rules of the reference's OTHER baselines that fall out of this path's stages.

* SnapKV (per-query-head selection, /root/reference/baselines/snapkv/utils.py:57-102): `ops.update_kv_per_query_head`.
* GemFilter (/root/reference/baselines/gemfilter/utils.py:25-38, `standard_dis_index`): the LAST query row's raw inner products
  with every key, optionally summed over the heads, optionally average-pooled, then top-k -- a 1-row window without softmax.
  Here: `fastkv_sp_logits_f16` (window 1) -> `fastkv_head_sum_f16` -> `fastkv_pool_f16` -> `fastkv_select_f16`, all on the
  GPU through the C ABI; same name, arguments and return value as the reference function.
Only the selection rules: the two-pass generation loop around GemFilter's rule is that baseline's model code, out of scope."""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import Problem, check, load


def standard_dis_index(data: torch.Tensor, queries: torch.Tensor, k: int, norm=1, pool: bool = False, kernel_size: int = 5,
                       sum_over_heads: bool = False):
    """(distances, indices) of the k largest `queries[:, :, 0] . data` per head ([B,H,k]) or of their head sum ([B,1,k]),
    after `avg_pool1d(kernel_size)` when `pool`: gemfilter/utils.py:25-38.  `data` [B,H or Hkv,n,D] fp16 keys (the reference
    passes them repeated to H heads; unrepeated keys give the same result), `queries` [B,H,>=1,D] fp16 (row 0 is used: the
    caller slices the last query, utils.py:50).  Ties at the k-th value: lowest position first (torch.topk leaves it open)."""
    ops._require_cuda(data, queries)
    B, H, _, D = queries.shape
    Hd, n = data.shape[1], data.shape[2]
    assert data.dtype == torch.float16 and queries.dtype == torch.float16 and H % Hd == 0 and 1 <= k <= n
    L = load()
    q0 = queries[:, :, :1]
    Sp = (n + 7) // 8 * 8
    p = Problem(B=B, H=H, Hkv=Hd, S=n, D=D, window=1, kernel=1, pooling=0, capacity=n, tsp_len=0, order=0, reserved=ops._engine)
    logits = torch.zeros(B, H, 1, Sp, dtype=torch.float16, device=data.device)
    ws = ops._workspace(L.fastkv_sp_workspace_bytes(ctypes.byref(p)), data.device, "scratch")
    check(L.fastkv_sp_logits_f16(ctypes.byref(p), q0.data_ptr(), ops._strides(q0), data.data_ptr(), ops._strides(data),
                                 logits.data_ptr(), Sp, 0, ws.data_ptr(), ws.numel(), ops._stream()), "sp_logits")
    rows = logits.view(B, H, Sp)                                    # inner_product[:, :, 0, :]
    if sum_over_heads:
        rows = ops.head_sum(rows).view(B, 1, Sp)                    # torch.sum(dim=1, keepdim=True): fp32 accumulation -> fp16
    R = rows.shape[1]
    if pool:
        pooled = torch.empty_like(rows)
        check(L.fastkv_pool_f16(rows.data_ptr(), B * R, Sp, n, kernel_size, ops.POOLING["avgpool"], pooled.data_ptr(), Sp,
                                ops._stream()), "pool")
        rows = pooled
    idx = ops.select(rows.view(B * R, Sp)[:, :n], k, order="score").view(B, R, k)
    dist = torch.gather(rows, 2, idx)
    if norm != 1:
        dist = dist / norm
    return dist, idx
