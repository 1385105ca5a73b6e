"""Pre-sized per-layer K/V cache slabs.

The reference hands the compressed K/V to `past_key_value.update(...)` (/root/reference/baselines/fastkv/llama_model.py:142),
which in transformers' DynamicLayer is a `torch.cat` -- one more copy of every kept row at prefill, and a re-allocation +
copy of the WHOLE layer cache at every decode step.  `SlabLayer` owns one `[B,Hkv,capacity + reserve,D]` buffer per layer:
the HIP compaction writes its rows straight into it (`FastKVCluster.update_kv(..., out_factory=layer.prefill_views)` ->
`fastkv_update_kv_strided_f16`), decode steps append one row in place, and attention reads the `[:, :, :len]` view.
Drop-in for `DynamicCache` in greedy decoding (`update`, `get_seq_length`, `get_mask_sizes`); pure torch, device-agnostic.

Static decode: `enable_static_decode(extra_rows)` makes room for the tokens to come and puts every layer's
length into device memory (`len_dev`).  From then on the attention module appends and attends through the HIP decode
kernels (`fastkv_decode_append_f16` / `fastkv_decode_attention_f16`), which read and advance that device-side length: no shape
changes from step to step, so the whole decode step can be captured once in a HIP graph and replayed (benchmark/e2e.py).
"""
from __future__ import annotations

import torch
from transformers.cache_utils import Cache, DynamicLayer


class SlabLayer(DynamicLayer):
    def __init__(self, reserve: int = 256):
        super().__init__()
        self.reserve = reserve
        self.kslab = self.vslab = None
        self.len = 0
        self.static_decode = False
        self.len_dev = None
        self.step_counters = None
        self.decode_ws = None       # slice records of the fused step kernel: tagged granules, paired with step_counters
        self.attn_ws = None         # slice records of the separate decode_attention launches (plain fp32)

    def _alloc(self, B, H, rows, D, dtype, device):
        self.kslab = torch.empty(B, H, rows, D, dtype=dtype, device=device)
        self.vslab = torch.empty_like(self.kslab)
        self.dtype, self.device = dtype, device
        self.is_initialized = True

    def _views(self):
        self.keys = self.kslab[:, :, :self.len]
        self.values = self.vslab[:, :, :self.len]
        return self.keys, self.values

    def prefill_views(self, B, H, rows, D, dtype, device):
        """[B,H,rows,D] views of the slab at the current end, for a producer that writes the rows itself; they become part
        of the cache when passed to `update`."""
        if self.kslab is None:
            self._alloc(B, H, rows + self.reserve, D, dtype, device)
        elif self.len + rows > self.kslab.shape[2]:
            self._grow(self.len + rows + self.reserve)
        return self.kslab[:, :, self.len:self.len + rows], self.vslab[:, :, self.len:self.len + rows]

    def _grow(self, rows):
        k, v = self.kslab, self.vslab
        self._alloc(k.shape[0], k.shape[1], rows, k.shape[3], k.dtype, k.device)
        self.kslab[:, :, :self.len].copy_(k[:, :, :self.len])
        self.vslab[:, :, :self.len].copy_(v[:, :, :self.len])

    def update(self, key_states, value_states, *args, **kwargs):
        n = key_states.shape[-2]
        if self.kslab is None:
            B, H, _, D = key_states.shape
            self._alloc(B, H, n + self.reserve, D, key_states.dtype, key_states.device)
        elif self.len > 0 and (self.keys is None or self.keys.shape[-2] != self.len or
                               self.keys.data_ptr() != self.kslab.data_ptr()):
            # someone replaced keys / values through the base-class API (crop, reorder, ...): rebuild the slab from them
            k, v = self.keys, self.values
            self.len = k.shape[-2]
            self._alloc(k.shape[0], k.shape[1], self.len + n + self.reserve, k.shape[3], k.dtype, k.device)
            self.kslab[:, :, :self.len].copy_(k)
            self.vslab[:, :, :self.len].copy_(v)
        if self.len + n > self.kslab.shape[2]:
            self._grow(2 * (self.len + n))
        dst_k = self.kslab[:, :, self.len:self.len + n]
        in_place = key_states.data_ptr() == dst_k.data_ptr() and key_states.stride() == dst_k.stride() and \
            value_states.data_ptr() == self.vslab[:, :, self.len:self.len + n].data_ptr()
        if not in_place:
            dst_k.copy_(key_states)
            self.vslab[:, :, self.len:self.len + n].copy_(value_states)
        self.len += n
        return self._views()

    def get_seq_length(self) -> int:
        return self.len

    # ---- static decode: the length lives on the device, the kernels advance it
    def enable_static_decode(self, extra_rows: int, shared=None):
        """`shared` = (step counters, step records, decode_attention records) of the cache this layer belongs to (the layers of
        one cache run one after the other, so they share them); a stand-alone layer gets its own."""
        assert self.kslab is not None and self.kslab.is_cuda, "static decode needs a prefilled slab on the GPU"
        if self.len + extra_rows > self.kslab.shape[2]:
            self._grow(self.len + extra_rows)
        self.len_dev = torch.tensor([self.len], dtype=torch.int32, device=self.kslab.device)
        self.static_decode = True
        # What the step kernels keep between launches is allocated (and zeroed) NOW, eagerly, and owned by the cache -- never
        # for the first time inside a graph capture, where the allocation would live in that graph's private pool, its zero-fill
        # would be replayed with every step, and a later capture would find a tensor of a pool that may be gone.
        from . import ops
        if shared is None:
            B, Hkv, _, D = self.kslab.shape
            shared = _new_shared(self.kslab.device, B, Hkv, D)
        self.step_counters, self.decode_ws, self.attn_ws = shared

    def rows_left(self) -> int:
        """Steps the slab still has room for (host mirror of the length; graph replays do not advance it -- see reserve_steps)."""
        return self.kslab.shape[2] - self.len

    def host_step(self):
        """Host mirror of one decode step (the device-side length is advanced by the attention kernel)."""
        if self.len >= self.kslab.shape[2]:
            # the kernel clamps (the step's row overwrites the last cached row) and reports FASTKV_EOVERFLOW through
            # fastkv_last_status; an eager caller learns it here, before the launch's results are used
            raise RuntimeError(f"fastkv_amd.cache: static decode ran out of slab rows ({self.kslab.shape[2]}): "
                               "enable_static_decode(extra_rows) reserved too few")
        self.len += 1

    def finish_static_decode(self, true_len=None):
        """Leave static mode: the host mirror takes the device's length (one small copy + sync) unless it is given."""
        if self.static_decode:
            self.len = int(self.len_dev.item()) if true_len is None else int(true_len)
            self.static_decode = False
            self._views()


def _new_shared(device, B, Hkv, D):
    from . import ops
    return (ops.new_step_counters(device), ops.new_decode_workspace(device, B, Hkv * 8, D),
            ops.new_decode_workspace(device, B, Hkv * 8, D, ops.DECODE_NSPLIT))


class FastKVSlabCache(Cache):
    """`DynamicCache` stand-in made of `SlabLayer`s (one per decoder layer)."""

    def __init__(self, num_layers: int, reserve: int = 256):
        super().__init__(layers=[SlabLayer(reserve) for _ in range(num_layers)])

    @property
    def static_decode(self) -> bool:
        return bool(self.layers) and all(getattr(l, "static_decode", False) for l in self.layers)

    def enable_static_decode(self, extra_rows: int):
        from . import ops
        first = self.layers[0]
        B, Hkv, _, D = first.kslab.shape
        shared = _new_shared(first.kslab.device, B, Hkv, D)
        for l in self.layers:
            l.enable_static_decode(extra_rows, shared if l.kslab.device == first.kslab.device and l.kslab.shape[:2] == first.kslab.shape[:2]
                                   and l.kslab.shape[3] == D else None)

    def rows_left(self) -> int:
        return min(l.rows_left() for l in self.layers)

    def reserve_steps(self, steps: int) -> None:
        """Before replaying a captured step `steps` times: the replays advance only the DEVICE-side lengths, nothing on the host
        sees a slab fill up -- so the room is checked here, once, for all of them."""
        left = self.rows_left()
        if steps > left:
            raise RuntimeError(f"fastkv_amd.cache: {steps} decode steps planned, the slabs have rows for {left}: "
                               "call enable_static_decode with more extra_rows")

    def finish_static_decode(self):
        for l in self.layers:
            l.finish_static_decode()
