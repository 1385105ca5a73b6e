import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
from gen_inputs import make_qkv
from golden_cases import CASES
from fastkv_amd import ops
from oracle import fastkv_oracle as O
name = sys.argv[1] if len(sys.argv)>1 else 'cfg1'
case=CASES[name]
q,k,v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"], peaked=case.get("peaked",0))
dev=torch.device('cuda:0')
qd,kd,vd=(t.transpose(1,2).contiguous().to(dev).transpose(1,2) for t in (q,k,v))
c,t = ops.scores(qd,kd,case["W"],case["ks"],case["pooling"])
co,to,lg = O.scores(q,k,case["W"],case["ks"],case["pooling"],want_logits=True)
c=c.cpu()
d=(c.view(torch.int16).int()-co.view(torch.int16).int())
nz=torch.nonzero(d)
print("mismatches", nz.shape[0], "of", d.numel(), "max", d.abs().max().item())
print(nz[:20].tolist())
for (b,g,j) in nz[:10].tolist():
    print((b,g,j), "gpu", c[b,g,j].item(), "cpu", co[b,g,j].item(), "d", d[b,g,j].item())
# ---- intermediate comparison via the workspace
def al(x,a=256): return (x+a-1)//a*a
B,H,Hkv,S,D,W=case["B"],case["H"],case["Hkv"],case["S"],case["D"],case["W"]
G=H//Hkv; R=G*W; r8=(R+7)//8*8
if r8<=64:
    RB = 8 if r8<=8 else 16 if r8<=16 else 32 if r8<=32 else 64; passes=1
else: RB=64; passes=(r8+63)//64
R_alloc=RB*passes; Sp=(S+7)//8*8; ntA=(S+255)//256; nchB=(S+2047)//2048
off_qf=0; off_logits=al(B*Hkv*R_alloc*D*4); off_pm=off_logits+al(B*H*W*Sp*2); off_ps=off_pm+al(B*H*W*ntA*4)
ws=list(ops._ws_cache.values())[0].cpu()
lg_g=ws[off_logits:off_logits+B*H*W*Sp*2].view(torch.float16).view(B,H,W,Sp)[...,:S]
dl=(lg_g.view(torch.int16).int()-lg.view(torch.int16).int())
print("logit mismatches", int((dl!=0).sum()), "of", dl.numel())
nzl=torch.nonzero(dl)
for (b,h,r,j) in nzl[:10].tolist():
    print("  logit",(b,h,r,j),"gpu",lg_g[b,h,r,j].item(),"cpu",lg[b,h,r,j].item())
pm=ws[off_pm:off_pm+B*H*W*ntA*4].view(torch.float32).view(B,H,W,ntA)
print("max mismatch rows", int((pm.max(-1).values!=lg.float().max(-1).values).sum()))
ps=ws[off_ps:off_ps+B*H*W*nchB*8].view(torch.int64).view(B,H,W,nchB).sum(-1)
# oracle sums
Lo=O.lib()
import numpy as np
bad=0
for b in range(B):
  for h in range(H):
    for r in range(W):
      row=lg[b,h,r].float(); m=row.max()
      # vectorised replica impossible; sample: compute using oracle scalar fns (slow) only for first few rows
      if h<2 and r<2:
        tot=0
        for x in (row-m).tolist(): tot+=Lo.fastkv_oracle_exp_to_fix(Lo.fastkv_oracle_det_expf(x))
        if tot!=int(ps[b,h,r]): bad+=1; print("sum mismatch",(b,h,r),tot,int(ps[b,h,r]))
print("sum mismatches (sampled rows)", bad)
# ---- full replica of the tail using verified GPU primitives
import ctypes
from fastkv_amd._lib import load
Lh=load()
def run(op,a,b=None):
    ad=a.contiguous().to(dev); bd=b.contiguous().to(dev) if b is not None else None
    out=torch.zeros_like(ad); o64=torch.zeros(a.numel(),dtype=torch.int64,device=dev)
    rc=Lh.fastkv_debug_contract(op, ad.data_ptr(), bd.data_ptr() if bd is not None else None, out.data_ptr(), o64.data_ptr(), a.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc==0; torch.cuda.synchronize(); return out.cpu(), o64.cpu()
x=lg.float(); m=x.max(-1,keepdim=True).values
dd=(x-m).reshape(-1)
e,_=run(0,dd); _,f64=run(3,e)
tot=f64.view(B,H,W,S).sum(-1)
print("row-sum mismatches vs gpu ps:", int((tot!=ps).sum()), "of", tot.numel())
sumf=torch.tensor([Lo.fastkv_oracle_fix_to_f32(int(v)) for v in tot.reshape(-1).tolist()]).view(B,H,W,1)
rinv=1.0/sumf
p16=(e.view(B,H,W,S)*rinv).half()
n=S-W
a=torch.zeros(B,H,n)
for r in range(W): a=a+p16[:,:,r,:n].float()
s16=a.half()
import torch.nn.functional as F
if case["pooling"]=="maxpool": pooled=F.max_pool1d(s16.float(),case["ks"],1,case["ks"]//2).half()
else:
    pad=case["ks"]//2; sp=F.pad(s16.float(),(pad,pad)); acc=torch.zeros(B,H,n)
    for u in range(case["ks"]): acc=acc+sp[...,u:u+n]
    pooled=(acc/float(case["ks"])).half()
pg=pooled.view(B,Hkv,G,n); acc=torch.zeros(B,Hkv,n)
for i in range(G): acc=acc+pg[:,:,i].float()
crep=acc.half()
print("replica vs gpu:", int((crep.view(torch.int16)!=c.view(torch.int16)).sum()), " replica vs oracle:", int((crep.view(torch.int16)!=co.view(torch.int16)).sum()))
