"""FastKV model wiring shared by the Llama and Mistral patches, written against the INSTALLED transformers (5.x:
`Attention.forward(hidden_states, position_embeddings, attention_mask, past_key_values, **kw)`, decoder layers return a
bare tensor, `DynamicCache(config=...)`).  What is FastKV-specific mirrors the reference:

  * prefill (q_len > 1): `kv_cluster.update_kv(K, Q, V, mask, groups, layer_idx)`; the cache receives the COMPRESSED
    K/V while attention runs over the full current K/V          (/root/reference/baselines/fastkv/llama_model.py:136-145)
  * TSP layer: hidden states and position ids are gathered by `tsp_idx` AFTER the MLP residual    (llama_model.py:252-259)
  * the model loop adopts `new_position_ids` and refreshes the rotary tables                      (llama_model.py:368-371)
  * only the last token's hidden state reaches `lm_head`                                          (llama_model.py:392)
"""
from __future__ import annotations

import torch
from transformers.cache_utils import DynamicCache
from transformers.modeling_outputs import BaseModelOutputWithPast

import os

from fastkv_amd import ops
from fastkv_amd.cache import FastKVSlabCache, SlabLayer
from fastkv_amd.cluster import init_fastkv


def _static_step(past_key_values, hidden_states) -> bool:
    """One-token step over a slab cache in static-decode mode on the GPU: the small operators of the step take their
    single-launch HIP versions (csrc/decode.hip; the stock modules issue 7 / 8 / 2 elementwise launches for them)."""
    return getattr(past_key_values, "static_decode", False) and hidden_states.is_cuda and hidden_states.shape[1] == 1 \
        and hidden_states.dtype == torch.float16


def _window_allows_static(config, past_key_values) -> bool:
    """The HIP decode attention attends over ALL rows of the slab (no mask: one new token sees the whole cache).  A model with a
    sliding window (Mistral v0.1: `config.sliding_window`) limits attention to the last `window` positions in the eager path
    (mistral_model.py passes the window to the mask builder and the attention call): the two agree only while the slabs cannot
    hold more rows than the window.  Otherwise static decode is refused up front instead of diverging silently."""
    win = getattr(config, "sliding_window", None)
    if win is None:
        return True
    layers = getattr(past_key_values, "layers", None) or []
    return all(getattr(l, "kslab", None) is None or l.kslab.shape[2] <= win for l in layers)


def _norm(module, hidden_states, static):
    if static and hasattr(module, "variance_epsilon") and module.weight.dtype == torch.float16:
        return ops.decode_rmsnorm(hidden_states, module.weight, module.variance_epsilon)
    return module(hidden_states)


def _mlp(module, hidden_states, static):
    if static and getattr(getattr(module, "config", None), "hidden_act", None) == "silu" and hasattr(module, "gate_proj") \
            and getattr(module.gate_proj, "bias", None) is None:
        return module.down_proj(ops.decode_silu_mul(module.gate_proj(hidden_states), module.up_proj(hidden_states)))
    return module(hidden_states)


def _step_rotary(rotary_emb, hidden_states, position_ids, past_key_values):
    """The rotary tables: the stock module, or -- for the one-token step over the static slab cache, default rope, fp16, on the GPU --
    ONE launch with the module's arithmetic (`ops.decode_rotary`; the stock module runs ten small launches per step).  FASTKV_DECODE_ROTARY=0
    keeps the module."""
    if (hidden_states.shape[1] == 1 and hidden_states.is_cuda and hidden_states.dtype == torch.float16
            and getattr(past_key_values, "static_decode", False) and os.environ.get("FASTKV_DECODE_ROTARY", "1") != "0"
            and getattr(rotary_emb, "rope_type", "default") in ("default", "llama3") and position_ids.dtype == torch.int64
            and position_ids.shape == (hidden_states.shape[0], 1) and position_ids.is_contiguous()
            and rotary_emb.inv_freq.dtype == torch.float32 and rotary_emb.inv_freq.is_cuda and rotary_emb.inv_freq.is_contiguous()
            and rotary_emb.inv_freq.numel() * 2 <= 256):
        from fastkv_amd import ops
        return ops.decode_rotary(rotary_emb.inv_freq, position_ids, float(rotary_emb.attention_scaling), 2 * rotary_emb.inv_freq.numel())
    return rotary_emb(hidden_states, position_ids=position_ids)


def _gemv_ok(*linears) -> bool:
    # plain fp16 `nn.Linear` without bias -- and without hooks: the fused step reads the weights directly, a hook on the module
    # (an adapter, a profiler) would silently not run
    return all(type(m) is torch.nn.Linear and m.bias is None and m.weight.dtype == torch.float16 and m.weight.is_contiguous()
               and not m._forward_hooks and not m._forward_pre_hooks
               and m.in_features % 512 == 0 and m.in_features * 2 <= 65536 - 256 for m in linears)


def _static_layer_ok(layer, hidden_states, position_embeddings, past_key_values) -> bool:
    """Whole-layer fast path of a one-token step (see `_static_layer`): stock Llama / Mistral module shapes, fp16, no biases,
    batch 1 or 2 (the input rows of the widest projection live in LDS), the layer's cache a slab in static-decode mode.
    Batch 2 shares ONE device-side length per layer (SlabLayer.len_dev): both rows hold the same number of cached tokens -- true by
    construction for the unpadded, equal-length prompts this wiring supports (as the reference: SURVEY 3.2) after a compression to
    one budget."""
    if os.environ.get("FASTKV_DECODE_GEMV", "1") == "0" or hidden_states.shape[0] not in (1, 2):
        return False
    attn, mlp = layer.self_attn, layer.mlp
    layers = getattr(past_key_values, "layers", None)
    slab = layers[attn.layer_idx] if layers is not None and attn.layer_idx < len(layers) else None
    if not (isinstance(slab, SlabLayer) and slab.static_decode) or position_embeddings[0].dtype != torch.float16:
        return False
    if not all(hasattr(m, "variance_epsilon") and m.weight.dtype == torch.float16
               for m in (layer.input_layernorm, layer.post_attention_layernorm)):
        return False
    if getattr(getattr(mlp, "config", None), "hidden_act", None) != "silu" or not hasattr(mlp, "gate_proj"):
        return False
    if max(mlp.down_proj.in_features, attn.q_proj.in_features) * 2 * hidden_states.shape[0] > 65536 - 256:
        return False
    return _gemv_ok(attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj)


def _static_layer(layer, x, position_embeddings, past_key_values):
    """One decoder layer of a one-token step over the slab cache in 5 launches (the stock modules: ~17 + 7 hipBLASLt GEMVs):
    [RMSNorm + q/k/v projections] -> [RoPE + append + attention + merge] -> [o_proj + residual] -> [RMSNorm + gate/up +
    SiLU*up] -> [down_proj + residual]: `ops.decode_gemv` (csrc/gemv.hip) and `ops.decode_step_attention` (csrc/decode.hip).  Same data flow as
    /root/reference/baselines/fastkv/llama_model.py:100-190 (q_len == 1 branch) + the stock MLP."""
    attn, mlp = layer.self_attn, layer.mlp
    slab = past_key_values.layers[attn.layer_idx]
    B = x.shape[0]
    D = attn.head_dim
    nq, nk = attn.q_proj.out_features, attn.k_proj.out_features
    ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
    qkv = ops.decode_gemv(x, [attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight], norm_weight=ln1.weight, eps=ln1.variance_epsilon)
    q = qkv[..., :nq].view(B, 1, nq // D, D).transpose(1, 2)
    k = qkv[..., nq:nq + nk].view(B, 1, nk // D, D).transpose(1, 2)
    v = qkv[..., nq + nk:].view(B, 1, nk // D, D).transpose(1, 2)
    cos, sin = position_embeddings
    a = ops.decode_step_attention(q, k, v, cos, sin, slab.kslab, slab.vslab, slab.len_dev, attn.scaling, counters=slab.step_counters,
                                  workspace=slab.decode_ws)
    slab.host_step()
    attn.tsp_idx = None
    h1 = ops.decode_gemv(a, [attn.o_proj.weight], residual=x)
    m = ops.decode_gemv(h1, [mlp.gate_proj.weight, mlp.up_proj.weight], norm_weight=ln2.weight, eps=ln2.variance_epsilon, glu=True)
    layer.new_position_ids = None
    return ops.decode_gemv(m, [mlp.down_proj.weight], residual=h1)


def make_cache(config):
    """DynamicCache as in the reference, or (FASTKV_SLAB_CACHE=1) pre-sized per-layer slabs that the compaction writes
    into directly and decode steps append to in place (fastkv_amd/cache.py)."""
    if os.environ.get("FASTKV_SLAB_CACHE", "0") == "1":
        return FastKVSlabCache(config.num_hidden_layers, reserve=int(os.environ.get("FASTKV_SLAB_RESERVE", "256")))
    return DynamicCache(config=config)


def make_attention_class(base_cls, modeling, extra_attn_kwargs):
    """Subclass of the stock attention whose prefill routes K/V through FastKVCluster.update_kv."""

    class FastKVAttention(base_cls):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            init_fastkv(self)                                     # utils.py:137-138
            self.tsp_idx = None

        def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, fastkv_sp=None,
                    fastkv_defer=None, **kwargs):
            input_shape = hidden_states.shape[:-1]
            hidden_shape = (*input_shape, -1, self.head_dim)
            q_len = input_shape[1]
            query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
            key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
            value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
            cos, sin = position_embeddings
            if _static_step(past_key_values, hidden_states) and cos.dtype == torch.float16:
                ops.decode_rope_(query_states, key_states, cos, sin)      # in place on the fresh projections, one launch
            else:
                query_states, key_states = modeling.apply_rotary_pos_emb(query_states, key_states, cos, sin)

            if fastkv_sp is not None and q_len > 1:
                return self._forward_sequence_parallel(query_states, key_states, value_states, past_key_values, fastkv_sp, input_shape)

            if past_key_values is not None:
                if q_len > 1:                                     # prefill: compress what goes into the cache
                    layers = getattr(past_key_values, "layers", None)
                    slab = layers[self.layer_idx] if layers is not None and self.layer_idx < len(layers) else None
                    if fastkv_defer is not None and fastkv_defer.eligible(self.kv_cluster, key_states, query_states):
                        # a layer whose compressed cache nobody needs before decode: compressed together with its peers when
                        # the forward pass is over (fastkv_amd.cluster.DeferredCompression); attention below runs over the
                        # full current K/V either way
                        self.tsp_idx = None
                        k_c = v_c = None
                        of = slab.prefill_views if isinstance(slab, SlabLayer) else None
                        if self.kv_cluster.tsp_layer:              # needed at once; takes a waiting peer along
                            res = fastkv_defer.add_tsp_layer(self.layer_idx, self.kv_cluster, key_states, query_states, value_states,
                                                             out_factory=of)
                            if res is None:
                                k_c, v_c = key_states, value_states
                            else:
                                k_c, v_c, self.tsp_idx, ready = res
                                for idx, kr, vr in ready:
                                    past_key_values.update(kr, vr, idx)
                        else:
                            ready = fastkv_defer.add(self.layer_idx, self.kv_cluster, key_states, query_states, value_states, out_factory=of)
                            if ready is None:
                                k_c, v_c = key_states, value_states
                            else:
                                for idx, kr, vr in ready:          # (a launch sequence became full: this layer and its peer)
                                    past_key_values.update(kr, vr, idx)
                    elif isinstance(slab, SlabLayer) and key_states.is_cuda and getattr(self.kv_cluster, "supports_out_factory", False):
                        # the compaction writes straight into the layer's cache slab; `update` then only adopts the rows
                        k_c, v_c, self.tsp_idx = self.kv_cluster.update_kv(key_states, query_states, value_states, attention_mask,
                                                                           self.num_key_value_groups, self.layer_idx,
                                                                           out_factory=slab.prefill_views)
                    else:
                        k_c, v_c, self.tsp_idx = self.kv_cluster.update_kv(key_states, query_states, value_states, attention_mask,
                                                                           self.num_key_value_groups, self.layer_idx)
                    if k_c is not None:
                        past_key_values.update(k_c, v_c, self.layer_idx)
                else:                                             # decode: plain append (llama_model.py:143-145)
                    self.tsp_idx = None
                    layers = getattr(past_key_values, "layers", None)
                    slab = layers[self.layer_idx] if layers is not None and self.layer_idx < len(layers) else None
                    if isinstance(slab, SlabLayer) and slab.static_decode and key_states.is_cuda and q_len == 1:
                        # static decode over the slab: append + GQA attention through the HIP decode kernels, the length is a
                        # device-side counter -> no shape changes, the step is graph-capturable (fastkv_amd/cache.py)
                        ops.decode_append(slab.kslab, slab.vslab, key_states, value_states, slab.len_dev)
                        attn_output = ops.decode_attention(query_states, slab.kslab, slab.vslab, slab.len_dev, self.scaling, workspace=slab.attn_ws)
                        slab.host_step()
                        return self.o_proj(attn_output.view(*input_shape, -1)), None
                    key_states, value_states = past_key_values.update(key_states, value_states, self.layer_idx)

            attention_interface = modeling.ALL_ATTENTION_FUNCTIONS.get_interface(self.config._attn_implementation,
                                                                                modeling.eager_attention_forward)
            attn_output, attn_weights = attention_interface(
                self, query_states, key_states, value_states, attention_mask,
                dropout=0.0 if not self.training else self.attention_dropout, scaling=self.scaling,
                **extra_attn_kwargs(self), **kwargs)
            attn_output = attn_output.reshape(*input_shape, -1).contiguous()
            return self.o_proj(attn_output), attn_weights

        def _forward_sequence_parallel(self, query_states, key_states, value_states, past_key_values, sp, input_shape):
            """Prefill of ONE prompt whose sequence is split over ranks (fastkv_amd/sp_model.py): this rank's queries / keys /
            values are positions [pos0, pos0 + S_r).  Same two things as the single-device branch -- the cache gets
            update_kv's rows (llama_model.py:139-142), attention runs causally over the full current K/V (:181-183) -- with
            the sequence-sharded operator and an all-gather of the K/V shards."""
            from fastkv_amd import sp_model
            from fastkv_amd.dist import sp_update_kv
            cl = self.kv_cluster
            self.tsp_idx = None
            if sp.layout(query_states.shape[1], key_states.shape[1]) == "heads":
                return self._forward_head_parallel(query_states, key_states, value_states, past_key_values, sp, input_shape)
            if past_key_values is not None:
                early, cap, tsp = sp_model.plan_for(cl, sp.total)
                if early:                                         # utils.py:89-91: nothing is dropped, the shard's rows are cached
                    k_c, v_c = key_states, value_states
                else:
                    dt = key_states.dtype
                    k16, q16, v16 = (t if t.dtype == torch.float16 else t.half() for t in (key_states, query_states, value_states))
                    k_c, v_c, self.tsp_idx, _ = sp_update_kv(k16, q16, v16, window_size=cl.window_size, kernel_size=cl.kernel_size,
                                                             pooling=cl.pooling, capacity=cap, tsp_len=tsp,
                                                             order=getattr(cl, "kv_order", None) or getattr(cl, "order", "score"),
                                                             group=sp.group, local_ops=sp.local_ops,
                                                             shard_lengths=sp.shard_lengths, replicate=sp.replicate)
                    k_c, v_c = k_c.to(dt), v_c.to(dt)
                past_key_values.update(k_c, v_c, self.layer_idx)
            k_cat, v_cat = sp_model.gather_kv(key_states, value_states, sp)
            attn_output = sp_model.sp_attention(query_states, k_cat, v_cat, self.scaling)
            return self.o_proj(attn_output.reshape(*input_shape, -1).contiguous()), None

        def _forward_head_parallel(self, query_states, key_states, value_states, past_key_values, sp, input_shape):
            """The same with head-parallel attention (fastkv_amd/sp_model.py, layout "heads"): one all-to-all hands this rank ALL
            positions of its H/P query heads and Hkv/P KV heads; attention is plain causal attention over the whole prompt for
            those heads (1/P of the work on every rank); the cache gets update_kv's rows for the local KV heads through the
            ordinary fused operator (`tp_update_kv`: no collective, the TSP layer adds one all-gather of score rows); a second
            all-to-all returns the attention output to sequence shards."""
            from fastkv_amd import sp_model
            from fastkv_amd.dist import tp_update_kv
            cl = self.kv_cluster
            q_f, k_f, v_f = sp_model.heads_exchange(query_states, key_states, value_states, sp)
            if past_key_values is not None:
                early, cap, tsp = sp_model.plan_for(cl, sp.total)
                if early:                                         # utils.py:89-91: nothing is dropped
                    k_c, v_c = k_f, v_f
                else:
                    dt = k_f.dtype
                    k16, q16, v16 = (t if t.dtype == torch.float16 else t.half() for t in (k_f, q_f, v_f))
                    k_c, v_c, self.tsp_idx, _ = tp_update_kv(k16, q16, v16, window_size=cl.window_size, kernel_size=cl.kernel_size,
                                                             pooling=cl.pooling, capacity=cap, tsp_len=tsp,
                                                             order=getattr(cl, "kv_order", None) or getattr(cl, "order", "score"),
                                                             group=sp.group, local_ops=sp.tp_ops)
                    k_c, v_c = k_c.to(dt), v_c.to(dt)
                if sp.replicate:                                  # (tests: every rank holds all KV heads' rows)
                    k_c, v_c = sp_model.gather_heads(k_c, sp), sp_model.gather_heads(v_c, sp)
                past_key_values.update(k_c, v_c, self.layer_idx)
            attn = sp_model.heads_attention(q_f, k_f, v_f, self.scaling)
            attn_output = sp_model.heads_return(attn, sp)
            return self.o_proj(attn_output.reshape(*input_shape, -1).contiguous()), None

    return FastKVAttention


def decoderlayer_forward_fastkv(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None,
                                use_cache=False, position_embeddings=None, fastkv_sp=None, fastkv_defer=None, **kwargs):
    static = _static_step(past_key_values, hidden_states)
    if static and fastkv_sp is None and _static_layer_ok(self, hidden_states, position_embeddings, past_key_values):
        return _static_layer(self, hidden_states, position_embeddings, past_key_values)
    residual = hidden_states
    hidden_states = _norm(self.input_layernorm, hidden_states, static)
    hidden_states, _ = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                      past_key_values=past_key_values, use_cache=use_cache,
                                      position_embeddings=position_embeddings, fastkv_sp=fastkv_sp, fastkv_defer=fastkv_defer, **kwargs)
    hidden_states = residual + hidden_states
    residual = hidden_states
    hidden_states = _norm(self.post_attention_layernorm, hidden_states, static)
    hidden_states = _mlp(self.mlp, hidden_states, static)
    hidden_states = residual + hidden_states
    # [FastKV] token-selective propagation: keep only the selected tokens from this layer on
    tsp_idx = getattr(self.self_attn, "tsp_idx", None)
    if self.self_attn.kv_cluster.tsp_layer and tsp_idx is not None and fastkv_sp is not None:
        # sequence-parallel prompt: tsp_idx holds GLOBAL positions (= the position ids of a prompt that starts at 0); every rank
        # contributes the surviving rows it owns, from here on all ranks hold the reduced sequence (fastkv_amd/sp_model.py)
        from fastkv_amd import sp_model
        self.new_position_ids = tsp_idx
        hidden_states = sp_model.tsp_assemble(hidden_states, tsp_idx, fastkv_sp)
        fastkv_sp.reduced = True
    elif self.self_attn.kv_cluster.tsp_layer and tsp_idx is not None:
        if hidden_states.is_cuda and position_ids.is_cuda and position_ids.dtype == torch.int64 and position_ids.dim() == 2 \
                and position_ids.stride(1) == 1 and position_ids.shape[0] in (1, hidden_states.shape[0]):
            # llama_model.py:254 and :255-257 in one HIP launch (clamped indices: the index tensor of a reported call cannot fault)
            hidden_states, self.new_position_ids = ops.tsp_propagate(hidden_states.contiguous(), position_ids, tsp_idx)
        elif hidden_states.is_cuda:
            self.new_position_ids = torch.gather(position_ids, dim=1, index=tsp_idx)
            hidden_states = ops.gather_rows(hidden_states.contiguous(), tsp_idx)          # HIP row gather
        else:
            self.new_position_ids = torch.gather(position_ids, dim=1, index=tsp_idx)
            hidden_states = torch.gather(hidden_states, 1, tsp_idx.unsqueeze(-1).expand(-1, -1, hidden_states.size(2)))
    else:
        self.new_position_ids = None
    return hidden_states


def defer_hold_for(config, batch: int, seq_len: int, itemsize: int = 2) -> int:
    """Layers per group of the deferred compression: FASTKV_DEFER_HOLD (default 8), capped by what the waiting layers hold alive
    (ADVICE r04): a waiting layer keeps its q / k / v -- batch x seq_len x (H + 2 Hkv) x D elements, 0.4 GB per batch row at 32k, 1.6 GB
    at 128k -- and hold - 1 layers wait at a time; FASTKV_DEFER_HOLD_GIB (default 4) bounds that sum.  32k, batch 1: 8 (2.8 GB held);
    128k or 32k x batch 4: 3 (3.2 GB); a prompt whose single layer exceeds the bound: 1 = layer by layer."""
    hold = int(os.environ.get("FASTKV_DEFER_HOLD", "8"))
    budget = float(os.environ.get("FASTKV_DEFER_HOLD_GIB", "4")) * 2 ** 30
    heads = getattr(config, "num_attention_heads", 32)
    kvh = getattr(config, "num_key_value_heads", None) or heads
    hd = getattr(config, "head_dim", None) or config.hidden_size // heads
    per_layer = batch * seq_len * (heads + 2 * kvh) * hd * itemsize
    return max(1, min(hold, int(budget // max(1, per_layer)) + 1))


def defer_max_len_for(config, batch: int, itemsize: int = 2) -> int:
    """Layers of at most this many tokens wait for the END of the forward pass and are compressed in one launch sequence
    (DeferredCompression `max_len`): FASTKV_DEFER_MAX_LEN, default 8192 since round 5 -- the 6553-token layers behind the TSP layer of
    the published recipe at 32k then run as ONE sequence of 16 instead of two groups of eight (0.776 -> 0.750 ms per step) -- capped so
    that what all of a model's layers would hold alive at that length stays within FASTKV_DEFER_HOLD_GIB (4): 3.2 GB at batch 1."""
    want = int(os.environ.get("FASTKV_DEFER_MAX_LEN", "8192"))
    budget = float(os.environ.get("FASTKV_DEFER_HOLD_GIB", "4")) * 2 ** 30
    heads = getattr(config, "num_attention_heads", 32)
    kvh = getattr(config, "num_key_value_heads", None) or heads
    hd = getattr(config, "head_dim", None) or config.hidden_size // heads
    per_token = max(1, getattr(config, "num_hidden_layers", 32) * batch * (heads + 2 * kvh) * hd * itemsize)
    return max(0, min(want, int(budget // per_token)))


def make_model_forward(modeling, mask_fn_for):
    def model_forward_fastkv(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                             inputs_embeds=None, use_cache=None, **kwargs):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        use_cache = use_cache if use_cache is not None else self.config.use_cache
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        if use_cache and past_key_values is None:
            past_key_values = make_cache(self.config)
        sp = getattr(self, "_fastkv_sp", None)                    # sequence-parallel prefill (fastkv_amd/sp_model.py)
        if sp is not None and position_ids is None:
            position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + sp.pos0).unsqueeze(0)
            position_ids = position_ids.expand(inputs_embeds.shape[0], -1)
        if getattr(past_key_values, "static_decode", False) and not _window_allows_static(self.config, past_key_values):
            raise ValueError(f"static decode attends over the whole cache slab, but this model has sliding_window = "
                             f"{self.config.sliding_window} and the slabs hold more rows than that: decode eagerly "
                             "(cache.finish_static_decode()) or size the slabs within the window")
        if position_ids is None:
            if getattr(past_key_values, "static_decode", False):
                raise ValueError("static decode (graph-capturable) needs position_ids as a device tensor: a host-side position "
                                 "would be frozen into the captured graph")
            past_seen = past_key_values.get_seq_length() if past_key_values is not None else 0
            position_ids = (torch.arange(inputs_embeds.shape[1], device=inputs_embeds.device) + past_seen).unsqueeze(0)
            position_ids = position_ids.expand(inputs_embeds.shape[0], -1)
        if sp is not None:
            causal_mask = None                                    # the sharded attention builds its own (lower-right causal) mask
        elif getattr(past_key_values, "static_decode", False) and inputs_embeds.shape[1] == 1:
            causal_mask = None                                    # one new token attends to the whole cache: nothing to mask
        else:
            causal_mask = mask_fn_for(self.config)(config=self.config, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                                   past_key_values=past_key_values, position_ids=position_ids)
        hidden_states = inputs_embeds
        position_embeddings = _step_rotary(self.rotary_emb, hidden_states, position_ids, past_key_values)
        # Prefill: layers whose compressed cache is not needed while the prompt is in flight (all but the TSP layer) are compressed
        # together with their peers -- the <= 4096-token layers behind the TSP layer after the last layer, the long ones in front of
        # it in groups of up to FASTKV_DEFER_HOLD (default 8: up to seven more layers' q / k / v alive, 0.4 GB each per batch row at
        # 32k; size it down for long prompts x big batches) (fastkv_amd.cluster.DeferredCompression; FASTKV_DEFER=0: layer by layer
        # as the reference)
        defer = None
        if sp is None and (type(past_key_values) is DynamicCache or isinstance(past_key_values, FastKVSlabCache)) \
                and inputs_embeds.shape[1] > 1 and inputs_embeds.is_cuda \
                and os.environ.get("FASTKV_DEFER", "1") != "0":
            from fastkv_amd.cluster import DeferredCompression
            defer = DeferredCompression(max_len=defer_max_len_for(self.config, inputs_embeds.shape[0], inputs_embeds.element_size()),
                                        hold_long=defer_hold_for(self.config, inputs_embeds.shape[0], inputs_embeds.shape[1],
                                                                 inputs_embeds.element_size()))
        for decoder_layer in self.layers[: self.config.num_hidden_layers]:
            hidden_states = decoder_layer(hidden_states, attention_mask=causal_mask, position_embeddings=position_embeddings,
                                          position_ids=position_ids, past_key_values=past_key_values, use_cache=use_cache,
                                          fastkv_sp=sp if (sp is not None and not sp.reduced) else None, fastkv_defer=defer, **kwargs)
            new_position_ids = getattr(decoder_layer, "new_position_ids", None)
            if new_position_ids is not None:                      # after the TSP layer: fewer tokens, new rotary tables
                position_ids = new_position_ids
                position_embeddings = self.rotary_emb(hidden_states, position_ids=position_ids)
                # The mask of the reduced sequence: causal over the surviving TOKENS (the reference's flash-attention call is
                # causal=True on whatever sequence it gets, llama_model.py:181-183).  Rebuilt rather than dropped: eager
                # attention masks only when it is handed a mask, so `None` would make every later layer bidirectional.
                # No 2-D padding mask (unpadded prompts only, as in the reference: SURVEY 3.2), no cache (prefill attention runs
                # over the current K/V), and no position ids: the gaps between surviving positions are not sequence boundaries.
                causal_mask = mask_fn_for(self.config)(config=self.config, inputs_embeds=hidden_states, attention_mask=None,
                                                       past_key_values=None, position_ids=None)
        if defer is not None:
            for idx, k_c, v_c in defer.flush():                   # (layer order: the cache grows by appending)
                past_key_values.update(k_c, v_c, idx)
        if inputs_embeds.is_cuda and inputs_embeds.shape[1] > 1:
            # Once per prefill: has a launch of this process given up a bounded in-kernel wait (include/fastkv_hip.h,
            # "Residency")?  A host-only read of pinned memory, no synchronisation: it reports what has run by now -- the
            # harnesses ask again behind their synchronisation point (benchmark/prefill.py, e2e.py), and any later operator call
            # reports the rest.  The caches of this forward pass are invalid then: raise, never hand them to decode.
            if os.environ.get("FASTKV_CHECK_SYNC", "0") == "1":   # debugging aid: wait for this stream, then the report is complete
                torch.cuda.current_stream().synchronize()
            ops.raise_if_aborted("prefill")
        hidden_states = _norm(self.norm, hidden_states, _static_step(past_key_values, hidden_states))
        hidden_states = hidden_states[:, -1:, :]                  # only the last token feeds lm_head
        if sp is not None and not sp.reduced:
            # no TSP reduction happened (short prompt / no TSP layer): the prompt's last token lives on the last rank
            import torch.distributed as dist
            src = sp.world - 1 if sp.group is None else dist.get_global_rank(sp.group, sp.world - 1)
            hidden_states = hidden_states.contiguous()
            if hidden_states.is_cuda and dist.get_backend(sp.group) == "gloo":
                h = hidden_states.cpu()
                dist.broadcast(h, src=src, group=sp.group)
                hidden_states = h.to(hidden_states.device)
            else:
                dist.broadcast(hidden_states, src=src, group=sp.group)
        return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=past_key_values)

    return model_forward_fastkv
