"""Measurement build only (FASTKV_CXXFLAGS=-DFK_STAMP): fine per-wave timeline of score_fused at the PAIR shape (two 32k layers per
launch: score_fused_kernel<128,4,2,1>) and at the one-layer shape, slots as in tools/stamp_fused_fine.py."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W = 32, 8, 128, 8
lib = _lib.load()
order = [(0, "start"), (16, "Q in LDS (A operand ready)"), (17, "tile even MFMA done (last)"), (18, "tile even epilogue done (last)"),
         (19, "tile odd MFMA done (last)"), (20, "tile odd epilogue done (last)"), (21, "max reduced in wave"), (22, "max published"),
         (1, "phase A end"), (23, "max: first granules seen"), (24, "max: records read"), (3, "max known"), (25, "B: exp loop done"),
         (26, "B: sums reduced in wave"), (4, "B end (published)"), (27, "sum: first granules seen"), (28, "sum: records read"),
         (7, "sum known"), (29, "C: ri loaded"), (30, "C: loop done"), (8, "C end (halo out)"), (11, "halo in"),
         (31, "D: pool+stores done"), (12, "D end (hist flushed)")]
for B, S in ((2, 32768), (1, 32768)):
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)
    for _ in range(30): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 48, dtype=np.uint64)
    lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    st = buf.reshape(4096, 48).astype(np.int64)
    st = st[st[:, 12] > 0]
    st = st[st[:, 0] >= st[:, 0].max() - 100000]
    t0 = st[:, 0].min()
    rel = (st - t0) * 10 / 1000.0
    print(f"B={B} S={S}: waves={len(st)}")
    prev = None
    for slot, nm in order:
        if (st[:, slot] == 0).all():
            continue
        col = rel[:, slot]
        med = np.median(col)
        d = "" if prev is None else f"  (+{med - prev:5.2f})"
        print(f"  {nm:32s} min {col.min():6.2f}  p10 {np.percentile(col, 10):6.2f}  median {med:6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us{d}")
        prev = med
