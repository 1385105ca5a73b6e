"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage times of score_fused."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W = 32, 8, 128, 8
lib = _lib.load()
names = ["start", "A0 done (+publish)", "A1 done (+publish)", "max0 known", "B0 done (+publish)", "max1 known", "B1 done (+publish)",
         "sum0 known", "C0 done (+halo out)", "sum1 known", "C1 done (+halo out)", "halo0 in", "D0 done", "halo1 in", "end"]
for S in (32768, 2048):
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(1, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)
    for _ in range(50): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    st = buf.reshape(4096, 16).astype(np.int64)
    st = st[st[:, 14] > 0]                     # waves of the last launch (smaller launches leave old rows behind: take the newest start)
    st = st[st[:, 0] >= st[:, 0].max() - 100000]
    t0 = st[:, 0].min()
    rel = (st - t0) * 10 / 1000.0
    print(f"S={S}: waves={len(st)}")
    for i, nm in enumerate(names):
        col = rel[:, i]
        if (st[:, i] == 0).all():
            continue
        print(f"  {nm:22s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
