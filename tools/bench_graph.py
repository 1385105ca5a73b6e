"""GPU-side stage timing with hipGraph replay (no host overhead between launches)."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import torch
from fastkv_amd import ops
dev=torch.device('cuda:0')
def graph_time(fn, n=50, reps=5):
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/(n*reps)*1e3
for S in (2048, 32768):
    H,Hkv,D,W=32,8,128,8
    q=torch.randn(1,S,H,D,device=dev,dtype=torch.float16).transpose(1,2)
    k=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    v=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    cap=2048
    c,t=ops.scores(q,k,W,7,"maxpool")
    idx=ops.select(c[0],cap-W,"score")[None].contiguous()
    print(f"S={S}: scores(4 kernels) {graph_time(lambda: ops.scores(q,k,W,7,'maxpool',want_tsp=False)):.1f}us  "
          f"select(score) {graph_time(lambda: ops.select(c[0],cap-W,'score')):.1f}us  select(index) {graph_time(lambda: ops.select(c[0],cap-W,'index')):.1f}us  "
          f"compact {graph_time(lambda: ops.compact(k,v,idx,W)):.1f}us  update_kv {graph_time(lambda: ops.update_kv(q,k,v,W,7,'maxpool',cap,0,'score')):.1f}us")
x=torch.empty(1<<20,device=dev)
print("torch tiny add_ kernel", f"{graph_time(lambda: x.add_(1.0)):.1f}us")
