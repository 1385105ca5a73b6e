"""The KV gather/compact kernel alone (HIP-event brackets through the library's profiler): roofline shape (32 layers stacked, both row
orders, min / median / max over buffer rotations) and the two-layer launch of the deferred schedule.  FASTKV_COMPACT_NT=1: non-temporal."""
import os, sys, statistics as st
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from fastkv_amd import ops
from fastkv_amd._lib import load
lib = load(); dev = torch.device("cuda:0")
r = bench.compact_roofline_shape(lib, dev, 10)
print("roofline shape:", {k: r[k] for k in ("index", "score")})
gen = torch.Generator(device=dev); gen.manual_seed(1)
sets = [[bench.make_layer_inputs(32768, gen, dev) for _ in range(2)] for _ in range(4)]
for order in ("score", "index"):
    for _ in range(3):
        for s_ in sets:
            ops.update_kv_entries([t[0] for t in s_], [t[1] for t in s_], [t[2] for t in s_], 8, 7, "maxpool", 2048, 0, order)
    torch.cuda.synchronize(); bench.profile_read(lib); lib.fastkv_profile_enable(1)
    for _ in range(10):
        for s_ in sets:
            ops.update_kv_entries([t[0] for t in s_], [t[1] for t in s_], [t[2] for t in s_], 8, 7, "maxpool", 2048, 0, order)
    torch.cuda.synchronize(); lib.fastkv_profile_enable(0)
    print("pair,", order, {k: round(ms / c * 1e3, 2) for k, (c, ms) in bench.profile_read(lib).items() if c})
