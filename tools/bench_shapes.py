import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.'))
import torch
from fastkv_amd import ops
dev = torch.device('cuda:0')
def run(H, Hkv, S, cap, label):
    q = torch.randn(1, S, H, 128, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(1, S, Hkv, 128, device=dev, dtype=torch.float16).transpose(1, 2)
    v = torch.randn(1, S, Hkv, 128, device=dev, dtype=torch.float16).transpose(1, 2)
    for _ in range(5): ops.update_kv(q, k, v, 8, 7, 'maxpool', cap, 0, 'score')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): ops.update_kv(q, k, v, 8, 7, 'maxpool', cap, 0, 'score')
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter()-t0)/n*1e6:.1f} us per update_kv")
run(8, 1, 32768, 2048, "cfg5 rank (H=8,Hkv=1,G=8) 32k")
run(8, 1, 2048, 2048, "cfg5 rank post-TSP 2k")
run(32, 8, 32768, 2048, "cfg2 (H=32,Hkv=8) 32k")
run(64, 8, 32768, 2048, "70B unsharded (H=64,Hkv=8,G=8) 32k")
# 1-3 query heads per KV head (round 3: ONE zero-padded 32-row block per KV head on the fused kernel; FASTKV_FUSED=0 in the environment
# gives the staged three-kernel path for comparison)
run(16, 8, 32768, 2048, "G=2 (H=16,Hkv=8) 32k")
run(8, 8, 32768, 2048, "MHA G=1 (H=8,Hkv=8) 32k")
run(16, 8, 2048, 2048, "G=2 (H=16,Hkv=8) 2k keep-all")
run(24, 8, 8192, 1024, "G=3 (H=24,Hkv=8) 8k")
