import sys, os, ctypes, math
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch, numpy as np
from fastkv_amd._lib import load
from oracle import fastkv_oracle as O
L=load(); Lo=O.lib()
dev=torch.device('cuda:0')
def run(op,a,b=None):
    ad=a.to(dev); bd=b.to(dev) if b is not None else None
    out=torch.zeros_like(ad); o64=torch.zeros(a.numel(),dtype=torch.int64,device=dev)
    rc=L.fastkv_debug_contract(op, ad.data_ptr(), bd.data_ptr() if bd is not None else None, out.data_ptr(), o64.data_ptr(), a.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc==0; torch.cuda.synchronize(); return out.cpu(), o64.cpu()
# exp over all fp16 differences grid
d = -torch.rand(1<<20)*90
g,_=run(0,d)
c=torch.tensor([Lo.fastkv_oracle_det_expf(float(x)) for x in d[:200000].tolist()])
print("exp mismatches", int((g[:200000].view(torch.int32)!=c.view(torch.int32)).sum()))
# division by sqrt(128) of all fp16 values
x=torch.arange(0,65536,dtype=torch.int32).to(torch.int16).view(torch.float16).float()
x=x[torch.isfinite(x)]
s=torch.full_like(x, float(np.float32(math.sqrt(128))))
g,_=run(1,x,s)
c=(x/s)
print("div sqrtD mismatches", int((g.view(torch.int32)!=c.view(torch.int32)).sum()), "of", x.numel())
# 1/x random
y=torch.rand(1<<20)*4000+1
g,_=run(1,torch.ones_like(y),y); c=1.0/y
print("rcp mismatches", int((g.view(torch.int32)!=c.view(torch.int32)).sum()))
y=torch.rand(1<<20); s7=torch.full_like(y,7.0)
g,_=run(1,y,s7); c=y/s7
print("div7 mismatches", int((g.view(torch.int32)!=c.view(torch.int32)).sum()))
# fp16 round trip incl subnormal range
z=(torch.rand(1<<20)*2e-4).float()
g,_=run(2,z); c=z.half().float()
print("f2h mismatches", int((g.view(torch.int32)!=c.view(torch.int32)).sum()))
# fix round trip
e=torch.rand(1<<18)
g,g64=run(3,e)
c64=torch.tensor([Lo.fastkv_oracle_exp_to_fix(float(v)) for v in e[:100000].tolist()],dtype=torch.int64)
print("exp_to_fix mismatches", int((g64[:100000]!=c64).sum()))
cf=torch.tensor([Lo.fastkv_oracle_fix_to_f32(int(v)) for v in c64.tolist()])
print("fix_to_f32 mismatches", int((g[:100000].view(torch.int32)!=cf.view(torch.int32)).sum()))
# mul small e * rinv
a=torch.rand(1<<20)*1e-3; b=torch.rand(1<<20)
g,_=run(6,a,b); print("mul mismatches", int((g.view(torch.int32)!=(a*b).view(torch.int32)).sum()))
