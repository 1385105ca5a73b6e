"""A/B of the interleaved schedule of launch_score_fused (FASTKV_FUSED_ROLLING=0/1, set per process): ops.scores of B 32k layers
(rotating over K sets larger than the Infinity Cache), us per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
H, Hkv, D, W, S = 32, 8, 128, 8, int(os.environ.get("EXP_S", "32768"))
for B in (2, 3, 4, 8, 16):
    nset = max(2, 16 // B)
    sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(nset)]
    outs = []
    for i in range(6):
        outs.append(ops.scores(*sets[i % nset], W, 7, "maxpool", want_tsp=False))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        ops.scores(*sets[i % nset], W, 7, "maxpool", want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"ROLLING={os.environ.get('FASTKV_FUSED_ROLLING', '1')} B={B}: {e0.elapsed_time(e1) * 1000 / n:7.1f} us per call, {e0.elapsed_time(e1) * 1000 / n / B:6.1f} us per layer", flush=True)
    del sets
from fastkv_amd._lib import raise_if_aborted
raise_if_aborted()
