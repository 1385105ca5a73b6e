"""Measurement build only: phase-A completion of score_fused by kv head (= XCD under round-robin placement), by wave slot and by workgroup index."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W, S = 32, 8, 128, 8, 32768
lib = _lib.load()
q = torch.randn(1, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2)
k = torch.randn(1, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)
for _ in range(50): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
st = buf.reshape(4096, 8)[:2048].astype(np.int64)
t0 = st[:, 0].min()
a = (st[:, 1] - t0) * 10 / 1000.0          # phase A done, us; wave index = (blockIdx.x * 4 + w)
start = (st[:, 0] - t0) * 10 / 1000.0
wg = np.arange(2048) // 4; w = np.arange(2048) % 4; head = wg % 8; blk = wg // 8
print("by head/XCD :", " ".join(f"{a[head == h].mean():5.1f}" for h in range(8)))
print("by wave slot:", " ".join(f"{a[w == i].mean():5.1f}" for i in range(4)))
print("by blk octile:", " ".join(f"{a[(blk // 8) == i].mean():5.1f}" for i in range(8)))
print("start by blk octile:", " ".join(f"{start[(blk // 8) == i].mean():5.2f}" for i in range(8)))
print("within-WG spread (max-min) mean:", np.mean([a[i*4:(i+1)*4].max() - a[i*4:(i+1)*4].min() for i in range(512)]))
print("percentiles 5/25/50/75/95:", np.percentile(a, [5, 25, 50, 75, 95]).round(1))
