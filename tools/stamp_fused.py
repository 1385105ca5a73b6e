"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage times of
score_fused (csrc/fused.hip FKF_STAMP slots).  Usage: stamp_fused.py [S B] ...  (default: 2048 1, 2048 16, 32768 1)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W = 32, 8, 128, 8
lib = _lib.load()
NSLOT = 48
names = {0: "start", 16: "query block staged", 17: "tile 0 contracted", 18: "tile 0 epilogue", 19: "tile 1 contracted", 20: "tile 1 epilogue",
         21: "row maxima reduced", 22: "A done, maxima published", 23: "max granules seen", 24: "max records read", 3: "max known",
         25: "B: exponentials", 26: "B: sums reduced", 4: "B done, sums published", 27: "sum granules seen", 28: "sum records read", 7: "sum known",
         30: "C: window-row sums", 8: "C done, halo out", 11: "halo in", 31: "D: scores stored", 12: "D done", 14: "end"}
order = [0, 16, 17, 18, 19, 20, 21, 22, 23, 24, 3, 25, 26, 4, 27, 28, 7, 30, 8, 11, 31, 12, 14]
args = [int(a) for a in sys.argv[1:]] or [2048, 1, 2048, 16, 32768, 1]
for S, B in zip(args[0::2], args[1::2]):
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)
    for _ in range(20): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.scores(q, k, W, 7, 'maxpool', want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(4096 * NSLOT, dtype=np.uint64)
    lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    st = buf.reshape(4096, NSLOT).astype(np.int64)
    st = st[st[:, 14] > 0]                     # waves of the last launch (smaller launches leave old rows behind: take the newest start)
    st = st[st[:, 0] >= st[:, 0].max() - 100000]
    t0 = st[:, 0].min()
    rel = (st - t0) * 10 / 1000.0
    print(f"S={S} B={B}: waves seen={len(st)} (of min(4096, launched)); {e0.elapsed_time(e1) * 50:.1f} us per call (events, incl. the stamps' stores)")
    if os.environ.get("STAMP_BY_ENTRY") and B >= 3:
        # a rolling launch: the buffer holds the last 4096 waves = the last 4096 / (waves per entry) entries; one block per entry, times
        # relative to the launch's first stamp in the buffer
        raw = buf.reshape(4096, NSLOT).astype(np.int64)
        wpe = 4 * ((S + 1023) // 1024) * Hkv                      # waves per entry (four tiles of 64 keys per wave)
        for e0_ in range(0, 4096, wpe):
            blk = raw[e0_:e0_ + wpe]
            blk = blk[blk[:, 14] > 0]
            if not len(blk):
                continue
            r = (blk - t0) * 10 / 1000.0
            print(f"  entry block at wave {e0_}: " + "  ".join(f"{names[i].split(',')[0][:14]} {np.median(r[:, i]):6.1f}" for i in (0, 22, 3, 4, 7, 8, 11, 14)))
        continue
    for i in order:
        col = rel[:, i]
        if (st[:, i] == 0).all():
            continue
        print(f"  {names[i]:26s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
