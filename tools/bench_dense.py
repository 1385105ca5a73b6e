"""Dense back-to-back launches of single stages (one event pair around N launches)."""
import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import torch
from fastkv_amd import ops
dev=torch.device('cuda:0')
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
for S in (2048, 32768):
    H,Hkv,D,W=32,8,128,8
    q=torch.randn(1,S,H,D,device=dev,dtype=torch.float16).transpose(1,2)
    k=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    v=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    cap=2048
    c,t=ops.scores(q,k,W,7,"maxpool")
    idx=ops.select(c[0],cap-W,"score")[None].contiguous()
    print(f"S={S}: scores(4 kernels) {timeit(lambda: ops.scores(q,k,W,7,'maxpool',want_tsp=False)):.1f}us  "
          f"select(score) {timeit(lambda: ops.select(c[0],cap-W,'score')):.1f}us  select(index) {timeit(lambda: ops.select(c[0],cap-W,'index')):.1f}us  "
          f"compact {timeit(lambda: ops.compact(k,v,idx,W)):.1f}us  update_kv {timeit(lambda: ops.update_kv(q,k,v,W,7,'maxpool',cap,0,'score')):.1f}us")
x=torch.empty(1<<20,device=dev)
print("torch tiny add_ kernel", f"{timeit(lambda: x.add_(1.0)):.1f}us", " empty alloc only", f"{timeit(lambda: torch.empty(1,8,2048,128,device=dev,dtype=torch.float16)):.1f}us")
