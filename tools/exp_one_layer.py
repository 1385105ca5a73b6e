"""ops.scores of ONE layer per call at 32k and at 2048 tokens, and of a pair (us per call; K sets rotate beyond the Infinity Cache at 32k)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
H, Hkv, D, W = 32, 8, 128, 8
for B, S, nset in ((1, 32768, 6), (2, 32768, 3), (1, 2048, 6), (16, 2048, 2)):
    sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(nset)]
    for i in range(12):
        ops.scores(*sets[i % nset], W, 7, "avgpool", want_tsp=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for i in range(n):
        ops.scores(*sets[i % nset], W, 7, "avgpool", want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} S={S}: {e0.elapsed_time(e1) * 1000 / n:6.1f} us per call", flush=True)
from fastkv_amd._lib import raise_if_aborted
raise_if_aborted()
