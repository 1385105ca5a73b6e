"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): per-wave stage times of
decode_step_kernel (csrc/decode_step.hip) at the shape of benchmark/e2e.py, 32 dependent launches replayed from a graph."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device("cuda:0")
B, H, Hkv, D = 1, 32, 8, 128
L0, rows, layers = 2048, 2304, 32
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator(device=dev).manual_seed(1)
slabs = [(torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16), torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16))
         for _ in range(layers)]
lens = [torch.tensor([L0], dtype=torch.int32, device=dev) for _ in range(layers)]
qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, generator=g, device=dev, dtype=torch.float16)
q = qkv[..., :H * D].view(B, 1, H, D).transpose(1, 2)
k = qkv[..., H * D:(H + Hkv) * D].view(B, 1, Hkv, D).transpose(1, 2)
v = qkv[..., (H + Hkv) * D:].view(B, 1, Hkv, D).transpose(1, 2)
ang = torch.rand(B, 1, D // 2, generator=g, device=dev) * 6.28
cos, sin = torch.cat([ang.cos(), ang.cos()], -1).half(), torch.cat([ang.sin(), ang.sin()], -1).half()
cnt, ws = ops.new_step_counters(dev), ops.new_decode_workspace(dev, B, H, D)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    def step():
        for (ks, vs), ld in zip(slabs, lens):
            ops.decode_step_attention(q, k, v, cos, sin, ks, vs, ld, D ** -0.5, nsplit=nsplit, counters=cnt, workspace=ws)
    step(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        step()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(2048 * 16, dtype=np.uint64)
lib.fastkv_debug_read_step_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
st = buf.reshape(2048, 16).astype(np.int64)
st = st[st[:, 0] > 0]
st = st[st[:, 0] >= st[:, 0].max() - 3000]          # waves of the last launch (30 us window)
t0 = st[:, 0].min()
names = ["start", "len read, arrival atomic issued", "q rotated (barrier passed)", "tile loads arrived", "tile done", "record stored",
         "merger: batch 0 valid", "merger: batch 1 valid", "merger: batch 2 valid", "merger: batch 3+ valid", "merger: output written", "before arrival check", "end (thread 0)"]
print(f"waves of the last launch: {len(st)}")
for i, nm in enumerate(names):
    col = st[:, i]
    sel = col > 0
    if not sel.any():
        continue
    rel = (col[sel] - t0) / 100.0
    print(f"  {nm:34s} n={sel.sum():4d}  min {rel.min():6.2f}  median {np.median(rel):6.2f}  max {rel.max():6.2f} us")

# where the workgroups ran: XCC_ID per (slice, head) of the last launch (grid x = head)
full = buf.reshape(2048, 16).astype(np.int64)
nh = B * Hkv
ns = len(st) // 4 // nh
tab = np.array([[int(full[(sl * nh + h) * 4, 13]) & 0xf for h in range(nh)] for sl in range(ns)])
print("XCC_ID [slice][head]:")
for sl in range(ns):
    print("  ", sl, tab[sl].tolist(), "merge done %.2f us" % ((full[(sl * nh) * 4, 10] - t0) / 100.0) if full[(sl * nh) * 4, 10] else "")
for h in range(nh):
    m = full[((ns - 1) * nh + h) * 4]
    print(f"  head {h}: merger XCC {int(m[13]) & 0xf}, producers on {sorted(set(tab[:-1, h].tolist()))}, output at {(m[10] - t0) / 100.0:.2f} us")

# several more replays: which mergers gave up on their L2, and what was stale then
for rep in range(12):
    with torch.cuda.stream(s):
        gr.replay()
        torch.cuda.synchronize()
    lib.fastkv_debug_read_step_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
    full = buf.reshape(2048, 16).astype(np.int64)
    out = []
    for h in range(nh):
        for wv in range(4):
            m = full[((ns - 1) * nh + h) * 4 + wv]
            if m[15]:
                out.append(f"head {h} wave {wv}: {int(m[15])} lanes gave up, stale mask {int(m[14]) & 0xffffffff:08x}")
    print(f"replay {rep}: " + ("; ".join(out) if out else "all mergers fast"))

# which workgroups store their record last? (slice index of the ten latest "record stored" stamps of the last launch read above)
rs = [(float(full[(sl * nh + h) * 4 + wv, 5] - t0) / 100.0, sl, h, wv) for sl in range(ns - 1) for h in range(nh) for wv in range(4) if full[(sl * nh + h) * 4 + wv, 5] > 0]
rs.sort(reverse=True)
print("latest records (us, slice, head, wave):", [(round(a, 2), b, c, d) for a, b, c, d in rs[:10]], " owner slice =", (L0 + 20 + 12 + 1) // 128)
