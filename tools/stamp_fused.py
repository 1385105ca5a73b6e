"""Measurement build only (FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage times of score_fused."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W = 32, 8, 128, 8
lib = _lib.load()
for S in (32768, 2048):
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(1, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)
    for _ in range(50): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    nw = 2048 if S == 32768 else 16 * 8 * 4
    st = buf.reshape(4096, 8)[:nw].astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) * 10 / 1000.0
    names = ["start", "MFMA + epilogue done", "max known (hand-off 1)", "(unordered stamp)", "sum known (hand-off 2)", "row sums in LDS", "halo received", "end"]
    print(f"S={S}: waves={nw}")
    for i, nm in enumerate(names):
        col = rel[:, i]
        print(f"  {nm:22s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
