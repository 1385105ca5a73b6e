import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
from oracle import fastkv_oracle as O
S=32768
q=torch.randn(1,S,32,128).half().transpose(1,2); k=torch.randn(1,S,8,128).half().transpose(1,2); v=torch.randn(1,S,8,128).half().transpose(1,2)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ")
for nt in (1, 8, 16, 32, 64, 128, 256):
    O.set_threads(nt)
    O.update_kv(q,k,v,8,7,'maxpool',2048,2048,'score')
    t=time.perf_counter(); O.update_kv(q,k,v,8,7,'maxpool',2048,2048,'score'); dt=time.perf_counter()-t
    print(f"threads={nt:4d}: {dt*1e3:8.1f} ms")
