"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage cycles of score_logits_mfma."""
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
    for _ in range(100): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.fastkv_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    nw = 2048 if S == 32768 else 64 * 4
    st = buf.reshape(4096, 8)[:nw].astype(np.int64)
    print(f"S={S}: waves={nw}; A ready after {np.median(st[:,1]-st[:,0])*10/1000:.2f} us")
    names = ["ph0 commit+fetch", "ph0 64 MFMAs", "gap", "ph1 commit+fetch", "ph1 64 MFMAs"]
    for i, nm in enumerate(names):
        d = st[:, 3 + i] - st[:, 2 + i]
        print(f"  {nm:18s} cycles: min {d.min():7d}  median {int(np.median(d)):7d}  max {d.max():7d}")
