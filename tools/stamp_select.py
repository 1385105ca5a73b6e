import os, sys
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
def al(x,a=256): return (x+a-1)//a*a
B=1;Sp=S;n=S-W;n_pad=(n+7)//8*8
o=0
o+=al(B*Hkv*32*D*4); o+=al(B*H*W*Sp*2); o+=al(B*H*W*4)*2; o+=al(B*Hkv*n_pad*2); o+=al(B*n_pad*2)
off_hist=o
st=ws[off_hist+8*4096*4:off_hist+8*4096*4+8*16*8].view(torch.int64).view(8,16).cpu()
names=["load+histcopy","find12","pass2","find4","pass3a(count+scan)","pass3b(emit)"]
for r in range(2):
    t=st[r,:7]; print("row",r," ".join(f"{names[i]}={(int(t[i+1])-int(t[i]))*10/1000:.1f}us" for i in range(6)))
