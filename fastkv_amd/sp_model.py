"""Sequence-parallel prefill of the whole model: ONE prompt split over P ranks on the sequence axis.

The reference has no multi-GPU path; what it fixes is the single-device behaviour this has to reproduce:
attention is causal over the whole prompt, the cache receives `FastKVCluster.update_kv`'s rows
(/root/reference/baselines/fastkv/llama_model.py:136-145), and from the TSP layer on only the `tsp_len` selected tokens
propagate (`llama_model.py:252-259`, `:368-371`).

Rank r holds positions [pos0_r, pos0_r + S_r).  Embeddings, norms, projections, RoPE and the MLP are token-local and run on
the shard as they are.  Attention + KV compression come in two layouts (`SPContext.mode`):

  "heads"  (default whenever the KV heads split over the ranks: Hkv % P == 0) -- head-parallel attention:
    * ONE all-to-all turns the sequence shards of q / k / v into head shards: rank r receives ALL positions of its H/P query
      heads and Hkv/P KV heads (Llama-3-8B over 8 ranks: 4 query heads + 1 KV head, the whole prompt);
    * plain causal attention over the whole prompt for those heads: every rank does exactly 1/P of the S^2/2 work;
    * KV compression is head-local: `fastkv_amd.dist.tp_update_kv` = the ordinary FUSED operator on the local heads (one
      launch sequence, no collective) -- only the TSP layer adds ONE all-gather of the fp16 score rows for the head sum;
      the cache of a sharded layer holds the local KV heads' rows (the layout a head-parallel decode consumes);
    * ONE all-to-all brings the attention output back to sequence shards for o_proj and the MLP.
    Collectives per layer: 2 (3 on the TSP layer).  Bytes per rank and layer at S = 131072, P = 8 (Llama-3-8B): 7/8 of
    S * (H/P + 2 Hkv/P) * D * 2 B = 176 MB in, the same out, + 7/8 of S * (H/P) * D * 2 B = 117 MB each way for the
    output: 280 MiB each direction, spread over the 7 xGMI links of the rank.
  "gather" (any P; the fallback) -- context parallelism by all-gather:
    * ONE all-gather of the ranks' K/V shards (a 16k-token shard of Llama-3-8B is 64 MiB of K+V: 448 MiB received per rank
      and layer at P = 8), then the local queries attend over keys [0, pos0_r + S_r) with a LOWER-RIGHT aligned causal mask;
      rank r does (r + 1/2) / (P^2 / 2) of the attention work: the last rank 15/64 at P = 8 -- attention scales 4.3x at best;
    * KV compression: `fastkv_amd.dist.sp_update_kv` (four small collectives; every rank keeps the rows it owns).
    Collectives per layer: 5.

  Per-rank model of one pre-TSP layer, S = 131072, P = 8, Llama-3-8B (attention = 4 * D * H * S^2 / 2 = 1.41e14 flop):
                    attention flop (max rank)   bytes received      collectives   update_kv
    one GPU         1.41e14                     -                   -             fused, 8 heads
    "gather"        3.30e13  (15/64: 4.3x)      448 MiB             5             staged kernels (logits round trip) + 4 collectives
    "heads"         1.76e13  (1/8:   8.0x)      280 MiB             2             fused, 1 head, no collective
  (280 MiB over 7 links at ~45 GB/s each is ~0.9 ms against ~44 ms of attention at 400 TFLOP/s: the exchange is 2 %.)

The TSP layer: every rank contributes the surviving hidden rows it owns, ONE exact all-reduce assembles the
[B, tsp_len, hidden] tensor everywhere -- the re-shard point.  The remaining layers see tsp_len (2048) tokens: too few to
shard, they run replicated on every rank through the ordinary single-device path (fused HIP kernels).

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
    mode: str = "auto"                          # "heads" | "gather" | "auto" (heads when the KV heads split over the ranks)
    tp_ops: Optional[object] = None             # fastkv_amd.dist HipTPOps for the "heads" layout (tests: an oracle-backed one)
    reduced: bool = field(default=False, init=False)     # set once the TSP layer has re-assembled the surviving tokens

    def layout(self, H: int, Hkv: int) -> str:
        """The layout of a layer with H query / Hkv KV heads: the same answer on every rank (it depends on P and the model only)."""
        P = self.world
        ok = Hkv % P == 0 and H % P == 0
        if self.mode == "heads" and not ok:
            raise ValueError(f"sp_prefill(mode='heads'): {Hkv} KV heads / {H} query heads do not split over {P} ranks")
        return "heads" if (self.mode in ("heads", "auto") and ok) else "gather"

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


# ------------------------------------------------------------------------------------------------- head-parallel layout
COLLECTIVES = {"all_to_all": 0}


def _all_to_all(send: torch.Tensor, send_rows: List[int], recv_rows: List[int], ctx: SPContext) -> torch.Tensor:
    """send [sum(send_rows), ...] (chunk p goes to rank p) -> [sum(recv_rows), ...] (chunk p came from rank p); one collective."""
    COLLECTIVES["all_to_all"] += 1
    inner = send[0].numel() if send.shape[0] else 1
    src = send.contiguous()
    staged = _staged(src, ctx.group)
    if staged:
        src = src.cpu()
    out = torch.empty((sum(recv_rows),) + tuple(send.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_to_all_single(out.view(-1), src.view(-1), output_split_sizes=[r * inner for r in recv_rows],
                           input_split_sizes=[r * inner for r in send_rows], group=ctx.group)
    return out.to(send.device) if staged else out


def heads_exchange(query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor, ctx: SPContext):
    """Sequence shards -> head shards, ONE all-to-all: [B,H,S_r,D] / [B,Hkv,S_r,D] x2 (this rank's positions, all heads) ->
    q [B,H/P,S,D], k, v [B,Hkv/P,S,D] (ALL positions, this rank's heads: query heads [r*H/P, (r+1)*H/P) and their KV heads).
    The results are views of the receive buffer in the attention module's own memory layout ([B,S,heads,D] physical: what the
    HIP operator and SDPA take as they are) -- nothing is copied on arrival when B == 1."""
    P, r, lens = ctx.world, ctx.rank, ctx.shard_lengths
    B, H, S_r, D = query_states.shape
    Hkv = key_states.shape[1]
    hq, hk = H // P, Hkv // P
    C = hq + 2 * hk                                                   # head slots per token and destination
    # chunk for rank p: [S_r, B, C, D] = its query heads, K heads, V heads of my tokens (token-major: chunks of different
    # sources concatenate along the sequence on arrival)
    send = torch.empty(P, S_r, B, C, D, dtype=query_states.dtype, device=query_states.device)
    send[:, :, :, :hq] = query_states.reshape(B, P, hq, S_r, D).permute(1, 3, 0, 2, 4)
    send[:, :, :, hq:hq + hk] = key_states.reshape(B, P, hk, S_r, D).permute(1, 3, 0, 2, 4)
    send[:, :, :, hq + hk:] = value_states.reshape(B, P, hk, S_r, D).permute(1, 3, 0, 2, 4)
    recv = _all_to_all(send.view(P * S_r, B, C, D), [S_r] * P, list(lens), ctx)      # [S, B, C, D]
    full = recv.permute(1, 2, 0, 3)                                    # [B, C, S, D] view: strides (C*D, D, B*C*D, 1)
    return full[:, :hq], full[:, hq:hq + hk], full[:, hq + hk:]


def heads_attention(q_full, k_full, v_full, scaling: float):
    """Causal attention of the local heads over the WHOLE prompt: [B,H/P,S,D] -> [B,H/P,S,D] (every rank: 1/P of the work)."""
    G = q_full.shape[1] // k_full.shape[1]
    global _GQA_NATIVE
    if G > 1 and _GQA_NATIVE is not False and q_full.is_cuda:
        try:
            out = F.scaled_dot_product_attention(q_full, k_full, v_full, is_causal=True, scale=scaling, enable_gqa=True)
            _GQA_NATIVE = True
            return out
        except (RuntimeError, TypeError):
            if _GQA_NATIVE:
                raise
            _GQA_NATIVE = False
    if G > 1:                                                         # repeat_kv (utils.py:13-22)
        B, Hk, L, D = k_full.shape
        k_full = k_full[:, :, None].expand(B, Hk, G, L, D).reshape(B, Hk * G, L, D)
        v_full = v_full[:, :, None].expand(B, Hk, G, L, D).reshape(B, Hk * G, L, D)
    return F.scaled_dot_product_attention(q_full, k_full, v_full, is_causal=True, scale=scaling)


def heads_return(attn: torch.Tensor, ctx: SPContext) -> torch.Tensor:
    """Head shards of the attention output [B,H/P,S,D] -> this rank's sequence shard with all heads [B,S_r,H,D]: ONE all-to-all."""
    P, r, lens = ctx.world, ctx.rank, ctx.shard_lengths
    B, hq, S, D = attn.shape
    S_r = lens[r]
    send = attn.permute(2, 0, 1, 3).contiguous()                       # [S, B, hq, D]: rank p's tokens are rows [pos0_p, pos0_p + S_p)
    recv = _all_to_all(send, list(lens), [S_r] * P, ctx)               # [P * S_r, B, hq, D]: source p holds query heads [p*hq, (p+1)*hq)
    return recv.view(P, S_r, B, hq, D).permute(2, 1, 0, 3, 4).reshape(B, S_r, P * hq, D)


def gather_heads(t: torch.Tensor, ctx: SPContext) -> torch.Tensor:
    """[B,Hkv/P,...] on every rank -> [B,Hkv,...] everywhere (rank order = head order); `replicate` (tests) only."""
    P = ctx.world
    src = t.contiguous()
    staged = _staged(src, ctx.group)
    if staged:
        src = src.cpu()
    out = torch.empty((P,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out.view(-1), src.view(-1), group=ctx.group)
    out = out.to(t.device) if staged else out
    return out.transpose(0, 1).reshape((t.shape[0], P * t.shape[1]) + tuple(t.shape[2:]))


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
