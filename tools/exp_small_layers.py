"""Experiment: ops.scores of the 16 S=2048 layers behind the TSP layer in one launch, under FASTKV_FUSED_MAX_WGS (workgroups per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
H, Hkv, D, W, S, B = 32, 8, 128, 8, 2048, 16
sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(4)]
for i in range(8):
    ops.scores(*sets[i % 4], W, 7, "avgpool", want_tsp=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
e0.record()
for i in range(n):
    ops.scores(*sets[i % 4], W, 7, "avgpool", want_tsp=False)
e1.record()
torch.cuda.synchronize()
print(f"FASTKV_FUSED_MAX_WGS={os.environ.get('FASTKV_FUSED_MAX_WGS', '-')}: {e0.elapsed_time(e1) * 1000 / n:6.1f} us per call (16 layers of 2048 tokens)")
