"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage times of
the grouping pass in front of the score-order copy (csrc/compact.hip rank_group_kernel), eight 32k layers per call as the default schedule
runs it."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W, S, n = 32, 8, 128, 8, 32768, 8
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(5)
qs = [torch.randn(1, S, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2) for _ in range(n)]
ks = [torch.randn(1, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2) for _ in range(n)]
vs = [torch.randn(1, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2) for _ in range(n)]
for _ in range(5):
    ops.update_kv_entries(qs, ks, vs, W, 7, "maxpool", 2048, 0, "score")
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.fastkv_debug_read_rank_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
st = buf.reshape(4096, 8).astype(np.int64)
st = st[st[:, 6] > 0]
t0 = st[:, 0].min()
rel = (st - t0) * 10 / 1000.0
print(f"waves {len(st)}")
for i, nm in enumerate(["start", "keys loaded, min/max per wave", "barrier 1", "histogram (barrier 2)", "suffix sums (barriers 3, 4)", "scatter (barrier 5)", "counted, slots stored"]):
    col = rel[:, i]
    print(f"  {nm:32s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
