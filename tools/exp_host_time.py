"""Is the layer-by-layer schedule (32 sequential update_kv calls, the reference's call pattern) bound by the GPU or by the host that
enqueues it?  Enqueue time of a step (no synchronisation inside) against its end-to-end time, bench.py's workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
for defer in (False, True):
    work = bench.HotPathPrefill(dev, seed=1)
    work.defer = defer
    for _ in range(3):
        work.step()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        work.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    enq.sort(); tot.sort()
    print(f"{'deferred (default)' if defer else 'layer by layer'}: enqueue {enq[len(enq)//2]*1e3:.3f} ms, end to end {tot[len(tot)//2]*1e3:.3f} ms per step (medians of 20)", flush=True)
    del work
