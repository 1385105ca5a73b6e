"""ops.scores of one / two layers at 64k - 256k tokens (us per call) and which scoring kernels ran (the library's launch counters)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
lib = ops.load()
H, Hkv, D, W = 32, 8, 128, 8
def counters():
    n = lib.fastkv_profile_kernels()
    c, ms = (ctypes.c_int64 * n)(), (ctypes.c_double * n)()
    assert lib.fastkv_profile_read(c, ms) == 0
    return {lib.fastkv_profile_kernel_name(i).decode(): (int(c[i]), round(float(ms[i]) * 1e3 / max(1, int(c[i])), 1)) for i in range(n) if c[i]}
for B, S in ((1, 65536), (1, 98304), (1, 131072), (2, 131072), (1, 262144)):
    sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(2)]
    for i in range(4):
        ops.scores(*sets[i % 2], W, 7, "avgpool", want_tsp=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for i in range(n):
        ops.scores(*sets[i % 2], W, 7, "avgpool", want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    counters(); lib.fastkv_profile_enable(1)
    ops.scores(*sets[0], W, 7, "avgpool", want_tsp=False); torch.cuda.synchronize()
    lib.fastkv_profile_enable(0)
    print(f"B={B} S={S}: {e0.elapsed_time(e1) * 1000 / n:7.1f} us per call;  K = {B * Hkv * S * D * 2 / 1e6:.0f} MB; kernels (launches, us): {counters()}", flush=True)
    del sets
