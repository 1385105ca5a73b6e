"""Drop-in `FastKVCluster` (same names, attributes, argument meaning and error behaviour as
/root/reference/baselines/fastkv/utils.py:48-138) whose compress branch runs on the HIP kernels.

Host logic kept exactly: the 9 plain attributes `compress_fastkv` pushes (utils.py:29-46), the
`cap - window > 0` assertion (utils.py:54), the proportional-mode mutation of
`max_capacity_prompt` / `tsp_length` (utils.py:86-87, :123-124), the `q_len < cap` early-out that
returns the SAME tensor objects (utils.py:89-91), the strict `q_len > tsp_length` TSP guard
(utils.py:126) and `ValueError('Pooling method not supported')` (utils.py:110).
"""
from __future__ import annotations

import os
from typing import NamedTuple, Optional

import torch

from . import ops
from ._lib import FASTKV_EUNSUPPORTED, FastKVNativeError


class Plan(NamedTuple):
    """What one `update_kv` call will do for a prompt of `q_len` tokens (pure host logic)."""
    early_out: bool
    capacity: int
    tsp_len: int          # 0 = no TSP index on this call


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """[B,Hkv,S,D] -> [B,Hkv*n_rep,S,D] (utils.py:13-22).  Kept for API parity; the HIP path never
    materialises this tensor."""
    if n_rep == 1:
        return hidden_states
    b, h, s, d = hidden_states.shape
    return hidden_states.unsqueeze(2).expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


class FastKVCluster:
    def __init__(self, window_size=8, max_capacity_prompt=512, kernel_size=7, pooling="avgpool", tsp_layer=False,
                 tsp_length=2048, tsp_rate=0.25, retain_rate=0.25, eviction_mode="constant"):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.tsp_layer = tsp_layer
        self.tsp_length = tsp_length
        self.retain_rate = retain_rate
        self.eviction_mode = eviction_mode
        self.tsp_rate = tsp_rate
        # row order of the compressed K/V: "score" = the reference's topk(sorted=True) order with the
        # canonical tie rule; "index" = ascending position (no sort pass).  Not a reference attribute.
        self.kv_order = os.environ.get("FASTKV_KV_ORDER", "score")

    def reset(self, window_size=8, max_capacity_prompt=512, kernel_size=7, pooling="avgpool", tsp_layer=False,
              tsp_length=2048, tsp_rate=0.25, retain_rate=0.25, eviction_mode="constant"):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.tsp_layer = tsp_layer
        self.tsp_length = tsp_length          # (the reference stores a 1-tuple here, utils.py:73: a bug, not reproduced)
        self.retain_rate = retain_rate
        self.eviction_mode = eviction_mode
        self.tsp_rate = tsp_rate

    # ------------------------------------------------------------------ host logic (no device work)
    def plan(self, q_len: int) -> Plan:
        """Applies and returns the length rules of utils.py:86-91 and :123-126 (mutating state as the
        reference does)."""
        if self.eviction_mode == "proportional":
            self.max_capacity_prompt = int(q_len * self.retain_rate)
        if q_len < self.max_capacity_prompt:
            return Plan(True, self.max_capacity_prompt, 0)
        if self.pooling not in ("avgpool", "maxpool"):
            raise ValueError("Pooling method not supported")
        if self.tsp_layer and self.eviction_mode == "proportional":
            self.tsp_length = int(q_len * self.tsp_rate)
        tsp = self.tsp_length if (self.tsp_layer and q_len > self.tsp_length) else 0
        return Plan(False, self.max_capacity_prompt, tsp)

    # ------------------------------------------------------------------ the operator
    supports_out_factory = True

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups, layer_idx, *,
                  out_factory=None):
        # `attention_mask` and `layer_idx` are accepted and ignored, as in the reference (utils.py:99).
        # out_factory (keyword only, not in the reference): callable (B, Hkv, capacity, D, dtype, device) -> (k_view, v_view)
        # asked for the output buffers once the capacity is known, e.g. fastkv_amd.cache.SlabLayer.prefill_views
        assert key_states.shape[-2] == query_states.shape[-2]
        q_len = query_states.shape[2]
        plan = self.plan(q_len)
        if plan.early_out:
            return key_states, value_states, None
        assert query_states.shape[1] == key_states.shape[1] * num_key_value_groups
        out = None
        if out_factory is not None:
            B, Hkv, _, D = key_states.shape
            out = out_factory(B, Hkv, plan.capacity, D, key_states.dtype, key_states.device)
        k_out, v_out, tsp_indices = ops.update_kv(query_states, key_states, value_states, self.window_size,
                                                  self.kernel_size, self.pooling, plan.capacity, plan.tsp_len,
                                                  self.kv_order, out=out)
        return k_out, v_out, tsp_indices


class DeferredCompression:
    """Compression of several layers in ONE launch sequence instead of one sequence per layer.

    The reference compresses layer by layer inside the attention forward (llama_model.py:136-142), but nothing reads a
    layer's compressed cache before decode -- only the TSP layer's index is needed while the prompt is still in flight.  An
    attention module hands a layer's q / k / v over with `add`; layers of one geometry are then compressed together through
    `ops.update_kv_entries` (device-side pointer tables: no stacking copies; the library scores three or more 32k layers with ONE
    rolling launch -- the layers follow each other over the chip two at a time and out of step, csrc/fused.hip launch_score_fused --, fewer or
    smaller ones in as many launches as the chip's residency asks for, and selects / copies ALL of them with one launch each).
    A waiting layer keeps its q / k / v alive (400 MiB at 32k); with `q_window` (FASTKV_DEFER_QWINDOW=1) only K, V and a 64 KiB copy of
    the query WINDOW rows -- the only query rows the operator reads (utils.py:93) -- at the price of one small copy launch per layer.
    Two regimes:
      * short layers (<= `max_len` tokens: the layers behind the TSP layer, whose launches are all latency) wait for `flush` at the
        end of the forward pass: 16 post-TSP layers in ~107 us instead of 407;
      * long layers wait for `hold_long` - 1 peers (default 8: up to seven more layers' q / k / v held, 2.8 GB at 32k -- the TSP layer 15 of the reference's recipe then closes the second group of eight; FASTKV_DEFER_HOLD): the group runs as
        soon as it is full.  `hold_long = 0` with `max_len` at the prompt length defers every layer to the end.
    `add` returns None when the layer keeps everything (utils.py:89-91: the caller caches K/V as they are), else the list of
    (layer_idx, k_compressed, v_compressed) that became ready with this call (usually empty); `flush` returns the rest.
    Same rows, same order as the per-layer calls (tests/test_wiring_gpu.py)."""

    _max_entries = {}                                              # geometry -> entries per launch sequence (process-wide)
    _fused_ok = {}                                                 # (H, Hkv, S, D, window, kernel) -> entries one fused launch holds

    def __init__(self, max_len: int = 4096, hold_long: int = 8, q_window: Optional[bool] = None):
        self.max_len = max_len
        self.hold_long = hold_long
        # keep only the window rows of q per waiting layer (a 64 KiB copy instead of the 256 MiB tensor at 32k).  Off by default: the copy
        # is one more small launch per layer -- measured 0.15 ms per 32-layer step, more than grouping four layers saves -- so it is
        # for callers who are short of memory, not of time (FASTKV_DEFER_QWINDOW=1)
        self.q_window = (os.environ.get("FASTKV_DEFER_QWINDOW", "0") == "1") if q_window is None else bool(q_window)
        self.groups = {}

    def _q_of(self, q, window):
        return ops.window_rows(q, window) if self.q_window else q

    @classmethod
    def _on_fused_path(cls, cluster, key_states, query_states) -> bool:
        """Batched entries exist on the fused scoring path only: asked from the library once per geometry (host only)."""
        if ops.no_wait_mode():                                    # FASTKV_FUSED=0 or the fail-safe switch after a placement report
            return False
        g = (query_states.shape[1], key_states.shape[1], key_states.shape[2], key_states.shape[3], cluster.window_size, cluster.kernel_size)
        ok = cls._fused_ok.get(g)
        if ok is None:
            try:
                ok = cls._fused_ok[g] = ops.fused_entries(*g)
            except Exception:   # noqa: BLE001 -- an odd kernel size etc.: the per-layer call reports it
                ok = cls._fused_ok[g] = 0
        return ok > 0

    def eligible(self, cluster, key_states, query_states) -> bool:
        # (an instance whose update_kv was wrapped -- a spy, an adapter -- expects to be called: not deferred)
        return (type(cluster) is FastKVCluster and "update_kv" not in vars(cluster)
                and key_states.is_cuda and key_states.dtype == torch.float16
                and query_states.dtype == torch.float16
                and (key_states.shape[2] <= self.max_len or self.hold_long >= 2)
                and not torch.cuda.is_current_stream_capturing()          # (the address tables are staged through the host)
                and self._on_fused_path(cluster, key_states, query_states))

    @staticmethod
    def _key(cluster, plan, q, k, v, outs):
        return ((cluster.window_size, cluster.kernel_size, cluster.pooling, plan.capacity, cluster.kv_order), tuple(q.shape), q.stride(),
                tuple(k.shape), k.stride(), v.stride(), None if outs is None else outs[0].stride())

    def add_tsp_layer(self, layer_idx, cluster, key_states, query_states, value_states, out_factory=None):
        """The TSP layer cannot wait (its index is needed at once), but it takes the WAITING peers of its geometry along: the
        group runs now, with the TSP selection computed for every entry and the peers' discarded.  Returns
        (k_compressed, v_compressed, tsp_idx, ready) -- `ready` = the peers' (layer_idx, k, v) -- or None when the layer keeps
        everything (utils.py:89-91)."""
        plan = cluster.plan(query_states.shape[2])
        if plan.early_out:
            return None
        outs = None
        if out_factory is not None:
            B, Hkv, _, D = key_states.shape
            outs = out_factory(B, Hkv, plan.capacity, D, key_states.dtype, key_states.device)
        q, k, v = query_states, key_states, value_states
        key = self._key(cluster, plan, q, k, v, outs)
        peers = self.groups.get(key, [])
        if plan.tsp_len and peers and self._max_entries.get(key, len(peers) + 1) >= len(peers) + 1 \
                and not torch.cuda.is_current_stream_capturing():
            try:
                qw = [p_[1] for p_ in peers] + [self._q_of(q, cluster.window_size)]
                o = None if outs is None else ([p_[4][0] for p_ in peers] + [outs[0]], [p_[4][1] for p_ in peers] + [outs[1]])
                k_outs, v_outs, tsp = ops.update_kv_entries(qw, [p_[2] for p_ in peers] + [k], [p_[3] for p_ in peers] + [v],
                                                            cluster.window_size, cluster.kernel_size, cluster.pooling, plan.capacity,
                                                            plan.tsp_len, cluster.kv_order, outs=o, q_window=self.q_window)
                self.groups.pop(key)
                Bq, n = q.shape[0], len(peers)                    # (rows i*Bq .. of the index tensor belong to entry i)
                return k_outs[n], v_outs[n], tsp[n * Bq:(n + 1) * Bq], [(p_[0], k_outs[i], v_outs[i]) for i, p_ in enumerate(peers)]
            except FastKVNativeError as e:
                # Only "this group cannot go through one launch sequence" (nothing was launched) sends the TSP layer on alone; the
                # peers stay in their group for `flush`.  Anything else -- FASTKV_EABORTED from an EARLIER launch, FASTKV_ELAUNCH --
                # is not this group's to absorb: it must reach the caller.
                if e.code != FASTKV_EUNSUPPORTED:
                    raise
        ko, vo, tsp = ops.update_kv(q, k, v, cluster.window_size, cluster.kernel_size, cluster.pooling, plan.capacity, plan.tsp_len,
                                    cluster.kv_order, out=outs)
        return ko, vo, tsp, []

    def add(self, layer_idx, cluster, key_states, query_states, value_states, out_factory=None):
        assert not cluster.tsp_layer, "the TSP layer goes through add_tsp_layer"
        plan = cluster.plan(query_states.shape[2])
        if plan.early_out:
            return None
        outs = None
        if out_factory is not None:
            B, Hkv, _, D = key_states.shape
            outs = out_factory(B, Hkv, plan.capacity, D, key_states.dtype, key_states.device)
        q, k, v = query_states, key_states, value_states
        key = self._key(cluster, plan, q, k, v, outs)
        pending = self.groups.setdefault(key, [])
        pending.append((layer_idx, self._q_of(q, cluster.window_size), k, v, outs))
        if k.shape[2] > self.max_len:
            # a long layer: run as soon as a launch sequence is full (or at once if this geometry only ever runs alone)
            if len(pending) >= max(1, min(self.hold_long, self._max_entries.get(key, self.hold_long))):
                return self._run(key)
        return []

    def flush(self):
        done = []
        for key in list(self.groups):
            done += self._run(key)
        return sorted(done, key=lambda t: t[0])

    def _run(self, key):
        its = self.groups.pop(key, [])
        window, ksize, pooling, cap, order = key[0]
        done, pos = [], 0
        # as many entries per launch sequence as the library takes for this geometry (all of them, normally: it splits the scoring
        # into resident launches itself; a refusal is remembered per geometry and the sequence shrinks)
        while pos < len(its):
            n = min(self._max_entries.get(key, len(its)), len(its) - pos)
            if torch.cuda.is_current_stream_capturing():
                n = 1                                            # (the entries call stages its address tables through the host)
            chunk = its[pos:pos + n]
            qs, ks, vs = [i[1] for i in chunk], [i[2] for i in chunk], [i[3] for i in chunk]
            outs = None if chunk[0][4] is None else ([i[4][0] for i in chunk], [i[4][1] for i in chunk])
            try:
                k_outs, v_outs, _ = ops.update_kv_entries(qs, ks, vs, window, ksize, pooling, cap, 0, order, outs=outs, q_window=self.q_window)
            except FastKVNativeError as e:
                # FASTKV_EUNSUPPORTED = more entries than the library takes for this geometry or entries of different layouts:
                # nothing was launched, retry with fewer (remembered per geometry).  Every other code (FASTKV_EABORTED: an earlier
                # launch of this process gave up a wait and ITS outputs are invalid; FASTKV_ELAUNCH) is an error of the run, not a
                # property of the geometry: the pending entries go back so that a caller who handles the error can flush again,
                # `_max_entries` is left alone, and the error is raised.
                if e.code != FASTKV_EUNSUPPORTED:
                    self.groups.setdefault(key, [])[:0] = its[pos:]
                    raise
                if n == 1 or getattr(e, "layout", False):
                    # the pointer-table path does not take this entry / these tensors (a view whose address or batch stride is not
                    # 16-byte aligned, outputs of another stride pattern, the no-wait mode since the group was queued): nothing was
                    # launched -- one by one through the strided entry point, and nothing is remembered about the GEOMETRY
                    for i in chunk:
                        o = ops.update_kv(i[1], i[2], i[3], window, ksize, pooling, cap, 0, order, out=i[4], q_window=self.q_window)
                        done.append((i[0], o[0], o[1]))
                    pos += n
                    continue
                self._max_entries[key] = n // 2 if n > 3 else n - 1
                continue
            done += [(i[0], ko, vo) for i, ko, vo in zip(chunk, k_outs, v_outs)]
            pos += n
        return done


def init_fastkv(self):
    """Attach a cluster to an attention module (utils.py:137-138)."""
    self.kv_cluster = FastKVCluster()


def compress_fastkv(model, args):
    """Per-layer configuration push with the reference's rules (utils.py:25-46): layer `tsp_idx` is the
    TSP layer; layers after it keep `retain_rate / tsp_rate` of their (already reduced) input."""
    for i, layer in enumerate(model.model.layers):
        c = layer.self_attn.kv_cluster
        c.window_size = args.window_size[i]
        c.kernel_size = args.kernel_size[i]
        c.pooling = args.pooling
        c.max_capacity_prompt = args.max_capacity_prompts
        c.tsp_length = args.tsp_len
        c.tsp_rate = args.tsp_rate
        c.eviction_mode = args.eviction_mode
        c.tsp_layer = (i == args.tsp_idx)
        c.retain_rate = args.retain_rate if i <= args.tsp_idx else args.retain_rate / args.tsp_rate
