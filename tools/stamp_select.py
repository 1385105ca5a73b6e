import os, sys, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import torch
from fastkv_amd import ops
dev=torch.device('cuda:0')
H,Hkv,D,W,S=32,8,128,8,32768
q=torch.randn(1,S,H,D,device=dev,dtype=torch.float16).transpose(1,2)
k=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
v=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
for _ in range(3): ops.update_kv(q,k,v,W,7,'maxpool',2048,0,'score')
torch.cuda.synchronize()
ws=list(ops._ws_cache.values())[0]
# find stamps: g_cnt region is inside workspace; scan for plausible: easier - recompute offsets
from fastkv_amd._lib import Problem, load
# layout mirror
def al(x,a=256): return (x+a-1)//a*a
B=1;R_alloc=32;Sp=S;n=S-W;n_pad=(n+7)//8*8;kk=2040
o=0
o+=al(B*Hkv*R_alloc*D*4); o+=al(B*H*W*Sp*2); o+=al(B*H*W*4)*2; o+=al(B*Hkv*n_pad*2); o+=al(B*n_pad*2); o+=al(B*Hkv*4096*4); o+=al(B*4096*4); o+=al(B*Hkv*kk*8)
off_sel=o; rows=8; kal=2040
g_cnt=off_sel+al(rows*2*kal*4)+al(rows*2*kal*2)
st=ws[off_sel:off_sel+8*16*8].view(torch.int64).view(8,16).cpu()
for r in range(8):
    t=st[r,:8]; print("row",r," ".join(f"{(int(t[i+1])-int(t[i]))*10/1000:.1f}us" for i in range(7)))
