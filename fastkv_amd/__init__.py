"""fastkv_amd -- MI355X (gfx950) implementation of FastKV's hot path: window-attention scoring,
pooled canonical top-k selection (per KV head and TSP) and K/V gather/compact.

    csrc/      hand-written HIP kernels + the C ABI (include/fastkv_hip.h)
    _lib.py    ctypes binding (fails loudly when the extension is missing; there is no fallback)
    ops.py     torch-tensor front-ends (device pointers + current stream only)
    cluster.py drop-in FastKVCluster / compress_fastkv / init_fastkv / repeat_kv
"""
from .cluster import FastKVCluster, Plan, compress_fastkv, init_fastkv, repeat_kv  # noqa: F401

__version__ = "0.1.0"
