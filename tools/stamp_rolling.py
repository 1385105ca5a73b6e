"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave timeline of the ROLLING
launch of score_fused (eight 32k layers in one grid, csrc/fused.hip launch_score_fused), per entry: when its waves start, end phase A, know the
maxima / sums, end phases B, C, D -- relative to the launch's first wave.  The stamp table holds 4096 waves: entries 0-3 of the launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W, S, B = 32, 8, 128, 8, 32768, int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _lib.load()
slots = [(0, "start"), (15, "phase A begins (paced)"), (1, "phase A end"), (3, "max known"), (4, "B end (published)"), (7, "sum known"), (8, "C end (halo out)"), (11, "halo in"), (12, "D end")]
sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(3)]
for i in range(9):
    ops.scores(*sets[i % 3], W, 7, 'maxpool', want_tsp=False)
torch.cuda.synchronize()
buf = np.zeros(4096 * 48, dtype=np.uint64)
lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
st = buf.reshape(4096, 48).astype(np.int64)
# wave index = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) % 4096: with 256 workgroups per entry, entry e (mod 4) owns rows e*1024 .. e*1024+1023
t0 = st[:1024, 0].min()
print(f"rolling launch of {B} entries (the table keeps the LAST entry written to each quarter: entries {[e for e in range(B)][-4:]} or 0-3)")
for e in range(min(B, 4)):
    blk = st[e * 1024:(e + 1) * 1024]
    print(f" table quarter {e}:")
    prev = None
    for slot, nm in slots:
        col = (blk[:, slot] - t0) / 100.0
        med = float(np.median(col))
        d = "" if prev is None else f"  (+{med - prev:5.2f})"
        print(f"   {nm:20s} min {col.min():7.2f}  median {med:7.2f}  max {col.max():7.2f} us{d}")
        prev = med
