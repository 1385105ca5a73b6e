"""Sequence-parallel prefill of the whole model: ONE prompt split over P ranks on the sequence axis.

The reference has no multi-GPU path; what it fixes is the single-device behaviour this has to reproduce:
attention is causal over the whole prompt, the cache receives `FastKVCluster.update_kv`'s rows
(/root/reference/baselines/fastkv/llama_model.py:136-145), and from the TSP layer on only the `tsp_len` selected tokens
propagate (`llama_model.py:252-259`, `:368-371`).

Layout per layer, rank r holding positions [pos0_r, pos0_r + S_r):
  * embeddings, norms, projections, RoPE, MLP: token-local -> run on the shard as they are;
  * attention: ONE all-gather of the ranks' K/V shards (context parallelism by all-gather: every peer is one xGMI hop away;
    a 16k-token shard of Llama-3-8B is 64 MiB of K+V per layer), then the local queries attend over keys [0, pos0_r + S_r)
    with a LOWER-RIGHT aligned causal mask (`torch.nn.attention.bias.causal_lower_right`: the flash kernel's own alignment);
  * KV compression: `fastkv_amd.dist.sp_update_kv` (four small collectives; every rank keeps the rows it owns);
  * the TSP layer: every rank contributes the surviving hidden rows it owns, ONE exact all-reduce assembles the
    [B, tsp_len, hidden] tensor everywhere -- the re-shard point.  The remaining layers see tsp_len (2048) tokens: too few
    to shard, they run replicated on every rank through the ordinary single-device path (fused HIP kernels).
Collectives before the TSP layer: 1 (K/V) + 4 (sp_update_kv) per layer; at the TSP layer + 1; afterwards none.

The model is the patched one (baselines.monkeypatch.replace_llama / replace_mistral + set_model); `sp_prefill` switches its
forward into this mode through a context object and returns what `model(...)` returns.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F


@dataclass
class SPContext:
    shard_lengths: List[int]
    group: Optional[object] = None
    local_ops: Optional[object] = None          # fastkv_amd.dist LocalOps (None: HipLocalOps); tests inject an oracle-backed one
    replicate: bool = False                     # also replicate the compressed K/V rows of the sharded layers (tests)
    reduced: bool = field(default=False, init=False)     # set once the TSP layer has re-assembled the surviving tokens

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group)

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group)

    @property
    def pos0(self) -> int:
        return sum(self.shard_lengths[:self.rank])

    @property
    def total(self) -> int:
        return sum(self.shard_lengths)


def plan_for(cluster, S: int):
    """(early_out, capacity, tsp_len) of `FastKVCluster.update_kv` for a prompt of S tokens (utils.py:86-91, :123-126), from the
    cluster's attributes; proportional mode mutates them as the reference does."""
    if cluster.eviction_mode == "proportional":
        cluster.max_capacity_prompt = int(S * cluster.retain_rate)
    cap = cluster.max_capacity_prompt
    if S < cap:
        return True, cap, 0
    if cluster.tsp_layer and cluster.eviction_mode == "proportional":
        cluster.tsp_length = int(S * cluster.tsp_rate)
    tsp = cluster.tsp_length if (cluster.tsp_layer and S > cluster.tsp_length) else 0
    return False, cap, tsp


_GQA_NATIVE = None          # does SDPA take grouped K/V heads with the lower-right causal bias on this build? (probed once)


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_kv(key_states: torch.Tensor, value_states: torch.Tensor, ctx: SPContext):
    """K/V of positions [0, pos0 + S_r) as [B,Hkv,pos0+S_r,D] tensors: one all-gather of the ranks' (padded) shards."""
    P, r, lens = ctx.world, ctx.rank, ctx.shard_lengths
    B, Hkv, S_r, D = key_states.shape
    smax = max(lens)
    mine = torch.zeros(2, B, Hkv, smax, D, dtype=key_states.dtype, device=key_states.device)
    mine[0, :, :, :S_r] = key_states
    mine[1, :, :, :S_r] = value_states
    src = mine.cpu() if _staged(mine, ctx.group) else mine
    out = torch.empty((P,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out.view(-1), src.view(-1), group=ctx.group)
    out = out.to(key_states.device)
    ks = [out[p, 0, :, :, :lens[p]] for p in range(r + 1)]
    vs = [out[p, 1, :, :, :lens[p]] for p in range(r + 1)]
    return torch.cat(ks, dim=2), torch.cat(vs, dim=2)


def sp_attention(query_states, k_cat, v_cat, scaling: float):
    """Local queries [B,H,S_r,D] over keys [0, pos0 + S_r): causal, lower-right aligned (query i sees keys 0 .. pos0 + i)."""
    from torch.nn.attention.bias import causal_lower_right
    G = query_states.shape[1] // k_cat.shape[1]
    bias = causal_lower_right(query_states.shape[2], k_cat.shape[2])
    global _GQA_NATIVE
    if G > 1 and _GQA_NATIVE is not False and query_states.is_cuda:
        # grouped-query attention inside the kernel: the gathered K/V (the largest tensors of the sharded prefill) are not
        # expanded to H heads -- 4x fewer bytes at G = 4.  Probed once: a backend without it raises, and the expanded form runs.
        try:
            out = F.scaled_dot_product_attention(query_states, k_cat, v_cat, attn_mask=bias, scale=scaling, enable_gqa=True)
            _GQA_NATIVE = True
            return out.transpose(1, 2)
        except (RuntimeError, TypeError):
            if _GQA_NATIVE:
                raise
            _GQA_NATIVE = False
    if G > 1:                                                      # repeat_kv (utils.py:13-22)
        B, Hkv, L, D = k_cat.shape
        k_cat = k_cat[:, :, None].expand(B, Hkv, G, L, D).reshape(B, Hkv * G, L, D)
        v_cat = v_cat[:, :, None].expand(B, Hkv, G, L, D).reshape(B, Hkv * G, L, D)
    out = F.scaled_dot_product_attention(query_states, k_cat, v_cat, attn_mask=bias, scale=scaling)
    return out.transpose(1, 2)                                     # [B,S_r,H,D]


def tsp_assemble(hidden_states: torch.Tensor, tsp_idx: torch.Tensor, ctx: SPContext) -> torch.Tensor:
    """[B,S_r,hidden] shard + global tsp_idx [B,tsp_len] -> the surviving rows [B,tsp_len,hidden] on every rank: each rank
    fills the rows it owns, one all-reduce adds the (otherwise zero) contributions -- exact, one contributor per row."""
    pos0, S_r = ctx.pos0, hidden_states.shape[1]
    own = (tsp_idx >= pos0) & (tsp_idx < pos0 + S_r)
    li = torch.where(own, tsp_idx - pos0, torch.zeros_like(tsp_idx))
    rows = torch.gather(hidden_states, 1, li.unsqueeze(-1).expand(-1, -1, hidden_states.shape[2]))
    rows = torch.where(own.unsqueeze(-1), rows, torch.zeros((), dtype=rows.dtype, device=rows.device)).contiguous()
    # summed as integer words so that even -0.0 keeps its bits
    words = rows.view(torch.int32 if rows.element_size() * rows.shape[-1] % 4 == 0 else torch.int16)
    if _staged(words, ctx.group):
        h = words.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=ctx.group)
        words.copy_(h)
    else:
        dist.all_reduce(words, op=dist.ReduceOp.SUM, group=ctx.group)
    return rows


def sp_prefill(model, input_ids_shard: torch.Tensor, ctx: SPContext, **kwargs):
    """`model(input_ids)` for ONE prompt of which this rank holds `input_ids_shard` (ranks in sequence order).  Returns the
    model's output (last-token logits, identical on every rank; the cache: owned rows of the sharded layers -- all rows with
    `ctx.replicate` -- and the full rows of the layers behind the TSP layer)."""
    from fastkv_amd.dist import check_shards
    cl0 = model.model.layers[0].self_attn.kv_cluster
    check_shards(ctx.shard_lengths, cl0.window_size, cl0.kernel_size)
    assert input_ids_shard.shape[1] == ctx.shard_lengths[ctx.rank]
    ctx.reduced = False
    model.model._fastkv_sp = ctx
    try:
        return model(input_ids_shard, **kwargs)
    finally:
        model.model._fastkv_sp = None
