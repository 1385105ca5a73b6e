"""Dense update_kv calls only (fused operator) at S=32768 (TSP layer on) and S=2048, for rocprofv3 per-kernel stats."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import torch
from fastkv_amd import ops
dev=torch.device('cuda:0')
order=os.environ.get("ORDER","score")
H,Hkv,D,W=32,8,128,8
for S in (32768, 2048):
    ins=[]
    for i in range(4):
        q=torch.randn(1,S,H,D,device=dev,dtype=torch.float16).transpose(1,2)
        k=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
        v=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
        ins.append((q,k,v))
    for it in range(40):
        q,k,v=ins[it%4]
        ops.update_kv(q,k,v,W,7,'maxpool',2048,0,order)
    torch.cuda.synchronize()
