"""Per-kernel timing at the 32k config (HIP events inside the library, dense launch stream)."""
import ctypes, os, sys, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
from fastkv_amd import ops
from fastkv_amd._lib import load
lib=load()
dev=torch.device('cuda:0')
S=int(sys.argv[1]) if len(sys.argv)>1 else 32768
H,Hkv,D,W=32,8,128,8
NL=6
ins=[]
for i in range(NL):
    q=torch.randn(1,S,H,D,device=dev,dtype=torch.float16).transpose(1,2)
    k=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    v=torch.randn(1,S,Hkv,D,device=dev,dtype=torch.float16).transpose(1,2)
    ins.append((q,k,v))
def read():
    n=lib.fastkv_profile_kernels(); c=(ctypes.c_int64*n)(); ms=(ctypes.c_double*n)()
    lib.fastkv_profile_read(c,ms)
    return {lib.fastkv_profile_kernel_name(i).decode():(c[i],ms[i]) for i in range(n)}
cap=min(2048,S); tsp=2048 if S>2048 else 0
for order in ("score","index"):
    for _ in range(2):
        for (q,k,v) in ins: ops.update_kv(q,k,v,W,7,"maxpool",cap,tsp,order)
    torch.cuda.synchronize(); read()
    lib.fastkv_profile_enable(1)
    for _ in range(5):
        for (q,k,v) in ins: ops.update_kv(q,k,v,W,7,"maxpool",cap,tsp,order)
    torch.cuda.synchronize(); lib.fastkv_profile_enable(0)
    r=read()
    print(f"S={S} order={order}: "+"  ".join(f"{n}={ms/c*1e3:.1f}us" for n,(c,ms) in r.items() if c))
