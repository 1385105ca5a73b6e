"""Round 5: the hot path of one prefill (bench.py HotPathPrefill.step) with the selection / grouping / copy of every launch sequence on
the library's finish stream (FASTKV_FINISH_STREAM=1: they run beside the NEXT sequence's scoring launch) against everything on one
stream.  Prints ms per step (HIP events around 30 steps, each step joined) and a digest of every cache row, the TSP rows and the gathered
hidden states: the two modes must agree bit for bit.  Usage: [FASTKV_FINISH_STREAM=1] exp_finish_stream.py [recipe]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from fastkv_amd._lib import raise_if_aborted
dev = torch.device("cuda:0")
recipe = len(sys.argv) > 1 and sys.argv[1] == "recipe"
work = bench.HotPathPrefill(dev, seed=7000, recipe=recipe)
dig = hashlib.sha256()
for _ in range(2):
    cache, hidden = work.step()
    torch.cuda.synchronize()
    for ko, vo in cache:
        dig.update(ko.cpu().numpy().tobytes()); dig.update(vo.cpu().numpy().tobytes())
    dig.update(hidden.cpu().numpy().tobytes())
    del cache, hidden
res = []
for rep in range(3):
    for _ in range(3):
        work.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        work.step()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 30)
raise_if_aborted("exp_finish_stream")
print(f"FINISH_STREAM={os.environ.get('FASTKV_FINISH_STREAM', '0')} recipe={int(recipe)}: " + " / ".join(f"{r:.4f}" for r in res) + f" ms per step, digest {dig.hexdigest()[:16]}", flush=True)
